"""The driver reads bench.py's result out of an 8 KB tail of stdout (BENCH_r04.json came back ``parsed: null`` for a 30.6 KB line):
the ONE stdout line must stay a compact object that round-trips through json.loads and keeps the contract's keys."""
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


FULL = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_bench_default.json")))


@pytest.mark.parametrize("path", FULL[-3:], ids=os.path.basename)
def test_compact_line_fits_the_driver_tail(path):
    b = _bench()
    full = json.load(open(path))
    full["_precision"] = "bf16"
    line = json.dumps(b.compact(full))
    assert len(line) < b.COMPACT_LIMIT <= 8192, len(line)
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert k in back, k
    assert back["value"] == full["value"] and back["config"]["workload"] == full["config"]["workload"]
    if "roofline" in full:
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in back["roofline"], k
    if "cpu_baseline" in full:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], k


def test_compact_line_survives_a_bloated_record():
    """whatever later rounds add to the full record, the line sheds optional blocks instead of growing"""
    b = _bench()
    full = json.load(open(FULL[-1]))
    full["_precision"] = "bf16"
    full["configs"].update({f"extra {i}": dict(full["configs"]["configs[2]"]) for i in range(40)})
    line = json.dumps(b.compact(full))
    assert len(line) < b.COMPACT_LIMIT
    back = json.loads(line)
    assert "roofline" in back and "cpu_baseline" in back and back["value"] == full["value"]


def test_bench_has_no_unresolved_global_names():
    """bench.py's GPU-only branches never run in the CPU suite: a name that resolves nowhere (round 6: `np` in job_abs_err) would only
    surface in the driver's run.  Every implicitly-global name referenced inside any function of bench.py must be defined at module
    level or be a builtin."""
    import builtins
    import os
    import symtable
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    top = symtable.symtable(src, "bench.py", "exec")
    defined = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()} | set(dir(builtins))
    missing = []

    def walk(tab):
        for ch in tab.get_children():
            for s in ch.get_symbols():
                if s.is_referenced() and s.is_global() and s.get_name() not in defined:
                    missing.append((ch.get_name(), s.get_name()))
            walk(ch)

    walk(top)
    assert not missing, missing
