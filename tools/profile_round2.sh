# Round-2 evidence in one gpurun call: bench lines, rocprofv3 kernel stats + one-step traces and PMC traffic for the
# BASELINE configs.  Usage (on the GPU box, from the repo root): bash tools/profile_round2.sh [tag]
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-round2}; O=$R/gpurun_out/$TAG; mkdir -p $O
B="--no-cpu-baseline --no-profile"
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
for w in gedex_b1 gedex_b32 dex_b32 gedex_long; do
  rm -rf /tmp/p_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$w -o t -- python $R/bench.py --workload $w --steps 2 --warmup 1 --graph off $B > $O/${w}_bench_under_rocprof.json 2>/dev/null
  cp $(find /tmp/p_$w -name "*kernel_stats.csv" | head -1) $O/${w}_kernel_stats.csv
  python $R/tools/trace_step.py $(find /tmp/p_$w -name "*kernel_trace.csv" | head -1) > $O/${w}_one_euler_step_trace.txt
done
# the DiT attention as its own launch (profiler evidence for roofline_attention)
for w in gedex_b1 gedex_b32 dex_b32 gedex_long; do
  rm -rf /tmp/pa_$w
  DEX_ATTN_SEPARATE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_$w -o t -- python $R/bench.py --workload $w --steps 1 --warmup 1 --graph off $B > /dev/null 2>&1
  grep -E "Name|attn_direct" $(find /tmp/pa_$w -name "*kernel_stats.csv" | head -1) > $O/${w}_attention_separate_kernel_stats.csv
done
# HBM traffic: one --pmc pass per counter, no tracing
for w in gedex_b1 dex_b32 gedex_long gedex_b32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_${w}_$c
    rocprofv3 --pmc $c --output-format csv -d /tmp/pm_${w}_$c -o pmc -- python $R/bench.py --workload $w --steps 1 --warmup 0 --graph off $B > /dev/null 2>&1
  done
  python $R/tools/pmc_json.py $w $(find /tmp/pm_${w}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pm_${w}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/${w}_pmc_top.txt 2>&1
done
ls -la $O
