"""MFMA utilisation per kernel symbol from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES pass of a bench.py command.

    python tools/pmc_mfma.py <workload> <attention separate 0/1> <counter_collection.csv> [out_json]

SQ_VALU_MFMA_BUSY_CYCLES sums the cycles the matrix pipe of each of the 1024 SIMDs is busy (= 32 per v_mfma_f32_32x32x16_*, checked
against SQ_INSTS_MFMA in round 2); SQ_BUSY_CYCLES sums the busy cycles of the 32 shader engines, i.e. kernel cycles x 32.  So
    utilisation = MFMA_BUSY / (1024 SIMDs x kernel cycles) = SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES).
This is utilisation at the clock the kernel actually ran at (dense bf16 MFMA throttles the chip to ~1.7 GHz: profiles/
round3_mfma_ceiling_random_operands.txt), NOT a fraction of the nominal 2.5 PF."""
import collections
import csv
import json
import os
import sys


def norm(name):
    n = name.replace("dex::", "").replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    return n.split("(")[0].replace(" ", "")


def main():
    workload, sep, path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    out = sys.argv[4] if len(sys.argv) > 4 else None
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = norm(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            cnt[k] += 1
    rows = []
    for k, c in agg.items():
        busy, mf = c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy > 0:
            rows.append((busy, k, cnt[k], mf / (32.0 * busy)))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    table = {k: {"launches": n, "share_of_busy_cycles": round(b / tot, 4), "mfma_util": round(u, 4)} for b, k, n, u in rows}
    for b, k, n, u in rows[:14]:
        print(f"{k:78s} x{n:5d}  share {b / tot:6.3f}  MFMA util {u:6.3f}")
    if out:
        data = json.load(open(out)) if os.path.exists(out) else {}
        data[f"{workload}{'_attention_separate' if sep else ''}"] = table
        json.dump(data, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
