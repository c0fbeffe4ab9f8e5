"""Development check of the split-weight mode (--precision fp16x2): distance of fp16 / fp16x2 from the library's exact-fp32 mode on the
same job (the fp32 mode sits 4e-6 from the CPU oracle, tests/test_gpu_baseline_shapes.py) and the speed of each, per workload.
    python tools/fp16x2_check.py [workload ...]      (default: gedex_b1 small)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from dex_tts_amd import synth, config as C
from dex_tts_amd.engine import ScoreNetEngine

dev = torch.device("cuda:0")
names = sys.argv[1:] or ["small", "gedex_b1"]
for name in names:
    if name == "small":
        preset, B, T, n, TrTs = "gedex_lj", 3, 132, 10, 0
    else:
        preset, B, T, n, TrTs, _ = Bn.WORKLOADS[name]
    cfg = C.PRESETS[preset]()
    eng = ScoreNetEngine(cfg, dev)
    eng.load_weights({k: torch.from_numpy(v) for k, v in synth.make_weights(C.param_shapes(cfg)).items()})
    lengths = Bn.lengths_for(B, T)
    mu, mask, z, kw = Bn.make_inputs(cfg, lengths, T, TrTs, dev, 1234)
    ys = {}
    for prec in ("fp32", "fp16", "fp16x2"):
        eng.set_precision(prec)
        y = eng.sample(z, mask, mu, n, **kw)           # eager
        torch.cuda.synchronize()
        use_graph = os.environ.get("CHECK_GRAPH", "1") != "0"
        call = lambda: eng.sample(z, mask, mu, n, use_graph=use_graph, **kw)
        k = 3
        dt, ev, yg = Bn.timed_calls(call, k, 2, dev)
        ys[prec] = y
        eq = bool((yg == y).all())
        fin = bool(torch.isfinite(y).all())
        line = f"{name:10s} {prec:7s} {sum(lengths) * k / dt:10.1f} frames/s  {dt / k / n * 1e3:8.4f} ms/Euler step  graph==eager {eq} finite {fin}"
        if prec != "fp32":
            d = (y - ys["fp32"]).abs()
            line += f"   vs fp32 mode: max {float(d.max()):.3e} mean {float(d.mean()):.3e}"
        print(line, flush=True)
    del eng
