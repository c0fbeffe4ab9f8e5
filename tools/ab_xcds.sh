#!/bin/bash
# The XCD-local cluster form of the DiT block dealt to 3 / 4 / 6 / 8 XCDs (DEX_DIT_XCDS): end to end at B = 1, the kernel's duration
# (rocprofv3 --stats) and its HBM traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled).   tools/ab_xcds.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/xcds; mkdir -p $O
B="--no-cpu-baseline --no-profile"
for rep in 1 2; do
for x in 8 3 4 6; do
  DEX_DIT_XCDS=$x python $R/bench.py --workload gedex_b1 --precision bf16 --steps 20 --warmup 5 --graph on $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcds=$x', d['value'], 'frames/s', d['ms_per_euler_step'], 'ms/step')"
done
done
for x in 8 3 4; do
  rm -rf /tmp/px_$x
  DEX_DIT_XCDS=$x rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px_$x -o t -- python $R/bench.py --workload gedex_b1 --precision bf16 --steps 2 --warmup 1 --graph off $B > /dev/null 2>&1
  f=$(find /tmp/px_$x -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "== xcds=$x kernel stats"; grep -h "cluster_kernel" "$f" | cut -c1-150; }
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_${x}_$c
    DEX_DIT_XCDS=$x rocprofv3 --pmc $c --output-format csv -d /tmp/pm_${x}_$c -o pmc -- python $R/bench.py --workload gedex_b1 --precision bf16 --steps 1 --warmup 0 --graph off $B > /dev/null 2>&1
  done
  python $R/tools/pmc_json.py gedex_b1 $(find /tmp/pm_${x}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pm_${x}_WRITE_SIZE -name "*counter_collection.csv" | head -1) /tmp/pmc_$x.json 2>&1 | grep -i "cluster" | head -3
done
