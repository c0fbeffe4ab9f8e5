"""The multi-GPU PRODUCT path with the real engine on one MI355X: two ranks (gloo rendezvous on 127.0.0.1, both on
cuda:0) run dex_tts_amd.dist.sample_sharded — each rank holds only its shard — and the gathered result equals the
single-process batched run BITWISE (utterances never interact; the norm statistics are order-independent integers)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("preset,prec,n,lengths,bitwise", [
    ("gedex_lj", "fp32", 4, [64, 40, 52, 30], True),
    ("gedex_lj", "bf16", 4, [64, 40, 52, 30], True),
    # uneven deal (3 + 2).  The library picks kernel variants by grid size (here: B=5 splits the linear-attention context
    # pass in two sub-tiles, B=3 / B=2 do not), so a shard and the full batch may sum in different orders: equal to bf16
    # rounding, not bitwise.  Identical variants (the two cases above) give identical bits.
    ("gedex_lj", "bf16", 4, [64, 40, 52, 30, 64], False),
    # the same uneven deal in the parity mode: within the fp32 sampler tolerance of tests/tolerances.py
    ("gedex_lj", "fp32", 4, [64, 40, 52, 30, 64], False),
    # DEX: the style inputs (six TIV skips, sty, sty_lengths) travel through sample_sharded's `extras` with their utterances
    ("dex_vctk", "fp32", 3, [64, 40, 52, 30], True),
    ("dex_vctk", "bf16", 3, [64, 40, 52, 30, 48], False),
])
def test_two_ranks_sharded_equals_single_process(tmp_path, preset, prec, n, lengths, bitwise):
    out = tmp_path / "r0.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_gpu_worker.py"), preset, prec, str(n),
           ",".join(map(str, lengths)), str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(out.read_text())
    assert res["finite"] and res["shape"][0] == len(lengths)
    if bitwise:
        assert res["bitwise_equal"], res
    else:
        from tests.tolerances import LOWP, FP32_SAMPLER_MAX
        assert res["max_abs_diff"] <= (FP32_SAMPLER_MAX if prec == "fp32" else LOWP[prec]["sampler"][0]), res
