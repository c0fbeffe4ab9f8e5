# Soak of the cluster form of the fused DiT block (XCD-local hand-offs): thousands of sampler calls, eager and replayed, under a side
# stream that streams memory; every 25th result is compared bitwise with the first, and no hand-off may report an error.
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests import gpu_util as U
cfg, eng, w = U.engine_for("gedex_lj")
eng.set_precision(os.environ.get("SOAK_PREC", "bf16"))
print("xcd_local", eng.xcd_local())
bad = 0; t0 = time.time()
for name_kw in [dict(B=1, T=512), dict(B=3, T=512, lengths=[512, 300, 77]), dict(B=2, T=100, lengths=[100, 61])]:
    case = U.make_case(cfg, **name_kw)
    mu, mask, z = (torch.from_numpy(case[k]).cuda() for k in ("mu", "mask", "z"))
    ref = eng.sample(z, mask, mu, 4).cpu().numpy()
    big = torch.empty(32 << 20, dtype=torch.float32, device="cuda"); side = torch.cuda.Stream()
    n = 0
    for it in range(int(os.environ.get("SOAK_CALLS", "3000"))):
        if it % 50 == 0:
            with torch.cuda.stream(side):
                big.mul_(1.0001)
        y = eng.sample(z, mask, mu, 4, use_graph=(it % 2 == 0))
        if it % 25 == 0:
            yh = y.cpu().numpy()
            if not np.array_equal(yh, ref): bad += 1
            if eng.handoff_timeouts() != 0: bad += 1000
        n += 1
    torch.cuda.synchronize()
    print(name_kw, "calls", n, "bad", bad, "elapsed %.1fs" % (time.time() - t0), flush=True)
print("SOAK", "OK" if bad == 0 else "FAILED", bad)
