#!/bin/bash
# kernel durations of the DiT attention and its consumer with the attention's tail split off / on (rocprofv3 --stats, DEX B=32)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/tail; mkdir -p $O
for w in ${1:-dex_b32}; do
for t in 0 1; do
  rm -rf /tmp/pt_$t
  DEX_ATTN_Q64_TAIL=$t timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$t -o t -- python $R/bench.py --workload $w --precision bf16 --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-profile > $O/prof_${w}_$t.log 2>&1
  f=$(find /tmp/pt_$t -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then echo "== $w tail=$t"; grep -h "attn_q64\|rowchain64" "$f" | cut -c1-160; fi
done
done > $O/stats_tail.txt
cat $O/stats_tail.txt
