# Same-box end-to-end A/B of this tree against the tree at the start of the second round-6 session (a git worktree of 5d9150c built under .s1tree/): every workload x mode
# the bench line reports.  bash tools/ab_vs_round5.sh > gpurun_out/ab_vs_round5.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
B="--no-cpu-baseline --no-profile --no-configs"
one() { (cd $1 && python bench.py --workload $2 --precision $3 --steps $4 --warmup 2 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.1f' % d['value'])"); }
for spec in "gedex_b1 bf16 10" "gedex_b1 fp16 10" "gedex_b1 fp16x2 10" "gedex_b1 fp32 5" "gedex_b32 bf16 5" "gedex_b32 fp16x2 4" "dex_b32 bf16 5" "dex_b32 fp16x2 4" "dex_b32 fp32 2" \
            "dex_esd_b32_n100 bf16 3" "dex_esd_b32_n100 fp16x2 2" "gedex_long fp16 8" "gedex_long_x2 fp16x2 5" "dex_b32_t512 bf16 3" "gedex_b1_t800 bf16 8" "dex_b1 bf16 8"; do
  set -- $spec
  a1=$(one $R/.s1tree $1 $2 $3); b1=$(one $R $1 $2 $3); a2=$(one $R/.s1tree $1 $2 $3); b2=$(one $R $1 $2 $3)
  python -c "a=($a1+$a2)/2; b=($b1+$b2)/2; print('%-18s %-7s before %9.1f %9.1f   after  %9.1f %9.1f   %+.1f %%' % ('$1','$2',$a1,$a2,$b1,$b2,(b/a-1)*100))"
done
