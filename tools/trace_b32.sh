# one B=32 Euler step, per launch, for each value of DEX_CONV_STREAM given as arguments (default: current default)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/b32t; mkdir -p $O
for m in ${@:-1}; do
  rm -rf /tmp/p2
  DEX_CONV_STREAM=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o b32 -- python $R/bench.py --workload ${WL:-gedex_b32} --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
  python $R/tools/trace_step.py /tmp/p2/b32_kernel_trace.csv > $O/b32_one_step_$m.txt
  echo "== DEX_CONV_STREAM=$m"; grep -E "conv3x3|^void  |step:" $O/b32_one_step_$m.txt | head -14
done
