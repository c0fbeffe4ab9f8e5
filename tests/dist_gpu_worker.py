"""Worker of tests/test_gpu_dist.py: one rank of a world_size-2 job whose ranks share cuda:0 (a 1-GPU box), gloo
rendezvous.  Every rank builds the real engine, holds only its shard of the batch, samples it and joins the one
all-gather of dex_tts_amd.dist.sample_sharded; rank 0 also samples the whole batch alone and compares BITWISE."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dex_tts_amd import config as C, dist as D, synth  # noqa: E402
from dex_tts_amd.engine import ScoreNetEngine  # noqa: E402


def main():
    preset, prec, n_steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    lengths = [int(v) for v in sys.argv[4].split(",")]
    out_path = sys.argv[5]
    backend = os.environ.get("DEX_DIST_BACKEND", "gloo")
    if backend == "nccl":                     # one process per GPU, RCCL (tests/test_gpu_dist.py::test_nccl_backend_two_gpus)
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = C.PRESETS[preset]()
    eng = ScoreNetEngine(cfg, dev)
    eng.load_weights({k: torch.from_numpy(v) for k, v in synth.make_weights(C.param_shapes(cfg)).items()})
    eng.set_precision(prec)
    T = D.padded_length(lengths)
    mu, mask, z, _ = synth.make_inputs(len(lengths), T, lengths)
    full_in = [torch.from_numpy(a).to(dev) for a in (mu, mask, z)]
    mu_l, mask_l, z_l = (D.take_shard(t, lengths) for t in full_in)          # what a rank would hold in production
    extras_full, extras_l = {}, {}
    if cfg.variant == "dex":                                                 # the style inputs shard with their utterances
        B = len(lengths)
        ref, _, sty, sl = synth.make_dex_style(B, 44, 52, cfg.mid_dim, sty_lengths=[52 - 7 * (i % 5) for i in range(B)])
        extras_full = {"ref": [torch.from_numpy(r).to(dev) for r in ref], "sty": torch.from_numpy(sty).to(dev),
                       "sty_lengths": torch.from_numpy(np.asarray(sl)).to(dev)}
        extras_l = {k: D.take_shard(v, lengths) for k, v in extras_full.items()}
    fn = lambda zz, mm, uu, **kw: eng.sample(zz, mm, uu, n_steps, **kw)
    got = D.sample_sharded(fn, mu_l, mask_l, z_l, lengths, local=True, extras=extras_l)
    res = {"rank": rank, "world": world, "shape": list(got.shape)}
    if rank == 0:
        ref = eng.sample(full_in[2], full_in[1], full_in[0], n_steps, **extras_full)
        res["bitwise_equal"] = bool(torch.equal(got, ref))
        res["max_abs_diff"] = float((got - ref).abs().max())
        res["finite"] = bool(torch.isfinite(got).all())
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
