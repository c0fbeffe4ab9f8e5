#!/bin/bash
# one Euler step launch by launch in fp16 and fp16x2 (rocprofv3 kernel trace): tools/trace_fp16x2.sh [workload]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/fp16x2; mkdir -p $O
w=${1:-gedex_b1}
for prec in fp16 fp16x2; do
  rm -rf /tmp/px_$prec
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px_$prec -o t -- python $R/bench.py --workload $w --precision $prec --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-profile > $O/${w}_${prec}_bench.json 2>/dev/null
  f=$(find /tmp/px_$prec -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python $R/tools/trace_step.py "$f" > $O/${w}_${prec}_step_trace.txt
done
paste <(cut -c1-44,68-82 $O/${w}_fp16_step_trace.txt) <(cut -c1-30,68-82 $O/${w}_fp16x2_step_trace.txt) | head -60
