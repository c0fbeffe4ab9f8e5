cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/gedex_sep; mkdir -p $O
B="--no-cpu-baseline --no-profile"
run() { tag=$1; shift
  rm -rf /tmp/p_$tag
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o t -- python $R/bench.py --workload gedex_b32 --precision bf16 --steps 2 --warmup 1 --graph off $B > $O/${tag}_bench.json 2>/dev/null
  grep -E "Name|attn_|dit_rowchain" $(find /tmp/p_$tag -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 > $O/${tag}_stats.csv
  env "$@" python $R/bench.py --workload gedex_b32 --precision bf16 --steps 10 --warmup 3 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$tag', d['value'], d['ms_per_step'])" >> $O/e2e.txt
}
run base DEX_X=0
run q64rc64 DEX_ATTN_Q64=1 DEX_ROWCHAIN64=2
run q64rc64_old DEX_ATTN_Q64=1 DEX_ROWCHAIN64=2 DEX_ROWCHAIN64A=0
run sep_rc64 DEX_ATTN_SEPARATE=1 DEX_ROWCHAIN64=2
cat $O/e2e.txt; for t in base q64rc64 q64rc64_old sep_rc64; do echo "== $t"; cat $O/${t}_stats.csv; done
