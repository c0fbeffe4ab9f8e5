set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r1g; mkdir -p $O
python $R/bench.py > $O/bench_default.json 2>$O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o b1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/bench_under_rocprof.json 2>/dev/null
cp /tmp/p1/b1_kernel_stats.csv $O/b1_kernel_stats.csv
python $R/tools/trace_step.py /tmp/p1/b1_kernel_trace.csv > $O/b1_one_euler_step_trace.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o b32 -- python $R/bench.py --workload gedex_b32 --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
cp /tmp/p2/b32_kernel_stats.csv $O/b32_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/p3_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/p3_$c -name "*counter_collection.csv" | head -1) $c > $O/b1_pmc_$c.txt 2>&1
done
ls -la $O
