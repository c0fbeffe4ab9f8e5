import os, sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gpu_util as U
cfg, eng, w = U.engine_for("dex_vctk")
eng.set_precision("bf16")
for B, T in ((1, 128), (1, 256), (1, 512), (2, 256), (3, 132), (4, 256), (8, 256), (12, 256), (4, 512), (8, 512)):
    case = U.make_case(cfg, B=B, T=T, lengths=[T - 3 * i for i in range(B)], Tr=348, Ts=348, sty_lengths=[348 - 5 * i for i in range(B)])
    mu, mask, z = (torch.from_numpy(case[k]).cuda() for k in ("mu", "mask", "z"))
    kw = U.engine_kwargs(case)
    res = {}
    for flag in ("0", "2"):
        os.environ["DEX_TV_CHAIN"] = flag
        for _ in range(2): eng.sample(z, mask, mu, 10, use_graph=True, **kw)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): eng.sample(z, mask, mu, 10, use_graph=True, **kw)
        torch.cuda.synchronize(); res[flag] = (time.time() - t0) / 5
    print(f"B={B} T={T}: three launches {res['0']*1e3:.2f} ms, one launch {res['2']*1e3:.2f} ms per 10-step call  ({res['0']/res['2']:.3f}x)", flush=True)
