set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/b32t; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o b32 -- python $R/bench.py --workload gedex_b32 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
python $R/tools/trace_step.py /tmp/p2/b32_kernel_trace.csv > $O/b32_one_step.txt
tail -60 $O/b32_one_step.txt
