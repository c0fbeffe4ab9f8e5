# A/B two builds of the library inside one gpurun call: tools/ab.sh <libA> <libB> [workloads...]
A=$1; B=$2; shift 2
for w in "$@"; do for rep in 1 2; do for L in $A $B; do
  echo -n "$w $L: "; DEX_AMD_LIB=$PWD/dex_tts_amd/lib/$L python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done; done
