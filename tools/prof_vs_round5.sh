# Per-kernel rocprofv3 diff of the round-5 tree (.r5tree, a git worktree of d511422 built in place) against this tree for one workload:
#   bash tools/prof_vs_round5.sh <workload> <precision>      (GPU box, repo root; result in gpurun_out/prof_vs_round5.txt)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; W=${1:-dex_b1}; P=${2:-bf16}; O=$R/gpurun_out/prof_vs_round5.txt
B="--no-cpu-baseline --no-profile --no-configs"
for which in r5 r6; do
  T=$R; [ $which = r5 ] && T=$R/.r5tree
  rm -rf /tmp/p_$which
  (cd $T && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$which -o t -- python bench.py --workload $W --precision $P --steps 3 --warmup 1 --graph off $B > /dev/null 2>&1)
done
python - "$(find /tmp/p_r5 -name '*kernel_stats.csv' | head -1)" "$(find /tmp/p_r6 -name '*kernel_stats.csv' | head -1)" $W $P > $O <<'PY'
import csv, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
print(f"{sys.argv[3]} {sys.argv[4]}: kernel time round5 {sum(v[1] for v in a.values()):.0f} us, round6 {sum(v[1] for v in b.values()):.0f} us")
rows = []
for n in set(a) | set(b):
    ca, ta = a.get(n, (0, 0.0)); cb, tb = b.get(n, (0, 0.0))
    rows.append((tb - ta, n, ca, ta, cb, tb))
for d, n, ca, ta, cb, tb in sorted(rows, key=lambda r: -abs(r[0]))[:25]:
    print(f"{d:+9.1f} us  r5 {ca:5d} x {ta / max(ca, 1):8.2f}  r6 {cb:5d} x {tb / max(cb, 1):8.2f}  {n[:110]}")
PY
cat $O
