# tools/envab.sh <workload> <ENVVAR> <val1> <val2> ...: same library, one env knob, two repetitions each
w=$1; v=$2; shift 2
for rep in 1 2; do for x in "$@"; do
  echo -n "$w $v=$x: "; env $v=$x python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
