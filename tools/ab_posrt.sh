#!/bin/bash
# positional convolution with one / two output rows per workgroup (DEX_POS_RT=1 / 2): end to end + the kernel's duration.  tools/ab_posrt.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B="--no-cpu-baseline --no-profile"
for rep in 1 2; do
for w in dex_b32 gedex_b32; do
for rt in 1 2 4; do
  DEX_POS_RT=$rt python $R/bench.py --workload $w --precision bf16 --steps 4 --warmup 2 --graph on $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w rt=$rt', d['value'], 'frames/s', d['ms_per_euler_step'], 'ms/step')"
done; done; done
for w in dex_b32 gedex_b32; do for rt in 1 2 4; do
  rm -rf /tmp/pr_$rt
  DEX_POS_RT=$rt rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$rt -o t -- python $R/bench.py --workload $w --precision bf16 --steps 2 --warmup 1 --graph off $B > /dev/null 2>&1
  f=$(find /tmp/pr_$rt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo "== $w rt=$rt"; grep -h "pos_conv" "$f" | cut -c1-150; }
done; done
