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
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    cfg = C.PRESETS[preset]()
    eng = ScoreNetEngine(cfg, dev)
    eng.load_weights({k: torch.from_numpy(v) for k, v in synth.make_weights(C.param_shapes(cfg)).items()})
    eng.set_precision(prec)
    T = D.padded_length(lengths)
    mu, mask, z, _ = synth.make_inputs(len(lengths), T, lengths)
    full_in = [torch.from_numpy(a).to(dev) for a in (mu, mask, z)]
    mu_l, mask_l, z_l = (D.take_shard(t, lengths) for t in full_in)          # what a rank would hold in production
    fn = lambda zz, mm, uu: eng.sample(zz, mm, uu, n_steps)
    got = D.sample_sharded(fn, mu_l, mask_l, z_l, lengths, local=True)
    res = {"rank": rank, "world": world, "shape": list(got.shape)}
    if rank == 0:
        ref = eng.sample(full_in[2], full_in[1], full_in[0], n_steps)
        res["bitwise_equal"] = bool(torch.equal(got, ref))
        res["max_abs_diff"] = float((got - ref).abs().max())
        res["finite"] = bool(torch.isfinite(got).all())
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
