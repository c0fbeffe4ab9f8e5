# End-to-end same-box sweep of one launcher knob (no rocprof):  bash tools/ab_e2e.sh KNOB "v0 v1 ..." workload [workload ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; K=$1; VALS=$2; shift 2; O=$R/gpurun_out/e2e_$K.txt; : > $O
B="--no-cpu-baseline --no-profile --no-configs"
for w in "$@"; do
  p=bf16; [ $w = gedex_long ] && p=fp16
  for rep in 1 2; do for v in $VALS; do
    env $K=$v python $R/bench.py --workload $w --precision $p --steps 6 --warmup 2 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   end to end $w $p $K=$v: %.1f frames/s, %.3f ms per call' % (d['value'], d['ms_per_step']))" >> $O
  done; done
done
cat $O
