"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE counter_collection CSVs of the same bench.py command) into
profiles/pmc_traffic.json: HBM bytes per launch per kernel symbol.

    python tools/pmc_json.py <workload> <fetch_csv> <write_csv> [out_json]

Corrections (MI355X_MICROARCH.md §HBM): the counters are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced streaming reads, so it is doubled.  WRITE_SIZE is taken as reported (uncalibrated)."""
import collections
import csv
import json
import os
import sys


def norm(name):
    n = name.replace("dex::", "").replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    n = n.split("(")[0]
    return n.replace(" ", "").replace("true", "1").replace("false", "0")


def avg_per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = norm(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def main():
    workload, fcsv, wcsv = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    f = avg_per_kernel(fcsv, "FETCH_SIZE")
    w = avg_per_kernel(wcsv, "WRITE_SIZE")
    table = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[0] * f.get(k, (0, 0))[1])):
        nf, fk = f.get(k, (0, 0.0))
        nw, wk = w.get(k, (0, 0.0))
        table[k] = {"launches": nf or nw, "fetch_kb_reported": round(fk, 1), "write_kb_reported": round(wk, 1),
                    "hbm_bytes_per_launch": int(round((2.0 * fk + wk) * 1024)),
                    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --workload {workload}; FETCH x2 (gfx950)"}
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[workload] = table
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    for k in list(table)[:8]:
        print(f"{k:70s} {table[k]['hbm_bytes_per_launch'] / 1e6:9.3f} MB/launch  x{table[k]['launches']}")


if __name__ == "__main__":
    main()
