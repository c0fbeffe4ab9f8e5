"""The cluster form of the fused DiT block (dit_rowchain_cluster_kernel): four workgroups per 32-row tile exchange split-K partial
sums inside ONE launch (write-through stores + flags).  What can go wrong is visibility — a stale slab or flag — so the tests
hammer it: many back-to-back calls must be bitwise identical, eager == graph replay, results equal the one-workgroup form to
reduced-precision rounding and the oracle to the stated tolerance, and no hand-off wait ever times out."""
import os

import numpy as np
import pytest
import torch

from tests import gpu_util as U
from tests.tolerances import LOWP

pytestmark = pytest.mark.gpu


def _run(eng, case, n, graph=False):
    mu, mask, z = (torch.from_numpy(case[k]).cuda() for k in ("mu", "mask", "z"))
    return eng.sample(z, mask, mu, n, use_graph=graph, **U.engine_kwargs(case)).cpu().numpy()


@pytest.mark.parametrize("name,kw,n", [
    ("gedex_lj", dict(B=1, T=512), 6),                                    # the headline shape: 21 clusters of 4
    ("gedex_lj", dict(B=3, T=512, lengths=[512, 300, 77]), 3),            # 63 clusters: the largest grid that takes the cluster form
    ("gedex_lj", dict(B=2, T=100, lengths=[100, 61]), 4),                 # N not a multiple of 32, ragged
    ("dex_vctk", dict(B=1, T=64, lengths=[57], Tr=40, Ts=40, sty_lengths=[33]), 4),
])
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_cluster_form_repeatable_and_equal_to_single_workgroup_form(name, kw, n, prec):
    cfg, eng, w = U.engine_for(name)
    case = U.make_case(cfg, **kw)
    eng.set_precision(prec)
    try:
        a = _run(eng, case, n)
        assert eng.handoff_timeouts() == 0
        for _ in range(12):                                               # stale slabs / flags would show up as a changed bit
            assert np.array_equal(a, _run(eng, case, n))
        assert np.array_equal(a, _run(eng, case, n, graph=True))
        assert np.array_equal(a, _run(eng, case, n, graph=True))          # second replay: flags re-zeroed by the replayed memset node
        assert eng.handoff_timeouts() == 0
        os.environ["DEX_DIT_CLUSTER"] = "0"
        try:
            single = _run(eng, case, n)
        finally:
            del os.environ["DEX_DIT_CLUSTER"]
        assert not np.array_equal(single, a)                              # another kernel really ran
        got, ref = U.run_sampler(name, case, n)
        for tag, y in (("cluster", a), ("single", single)):
            e = np.abs(y - ref)
            U.record(f"clusterAB_{name}_B{kw['B']}_T{kw['T']}_{tag}:{prec}:sampler", max=e.max(), mean=e.mean())
            assert e.max() <= LOWP[prec]["sampler"][0] and e.mean() <= LOWP[prec]["sampler"][1], (tag, float(e.max()), float(e.mean()))
    finally:
        eng.set_precision("fp32")


def test_cluster_form_under_concurrent_load():
    """Hand-offs under UNEVEN load (the guide's advice): a second stream streams memory while the sampler runs; bits unchanged."""
    cfg, eng, w = U.engine_for("gedex_lj")
    case = U.make_case(cfg, B=1, T=512)
    eng.set_precision("bf16")
    try:
        a = _run(eng, case, 4)
        big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        side = torch.cuda.Stream()
        for _ in range(6):
            with torch.cuda.stream(side):
                for _ in range(8):
                    big.mul_(1.0001)
            assert np.array_equal(a, _run(eng, case, 4))
        torch.cuda.synchronize()
        assert eng.handoff_timeouts() == 0
    finally:
        eng.set_precision("fp32")


def test_lost_handoff_is_loud():
    """A hand-off that never arrives (DEX_DEBUG_DROP_HANDOFF=1: member 3 of every cluster keeps its flags down) must neither hang
    the GPU nor pass as a mel: the bounded waits time out, the device word is set, and every output of the call is NaN."""
    cfg, eng, w = U.engine_for("gedex_lj")
    case = U.make_case(cfg, B=1, T=64)
    eng.set_precision("bf16")
    try:
        good = _run(eng, case, 2)
        assert np.isfinite(good).all() and eng.handoff_timeouts() == 0
        os.environ["DEX_DEBUG_DROP_HANDOFF"] = "1"
        try:
            eng.check_handoffs = True
            try:
                with pytest.raises(RuntimeError, match="hand-off timed out"):   # the synchronous check: raises before the mel is handed out
                    _run(eng, case, 2)
                eng.check_handoffs = False                                       # ... unless told not to: the poisoned result itself
                bad = _run(eng, case, 2)
            finally:
                eng.check_handoffs = "deferred"
            assert eng.handoff_timeouts() == 1
            # the default, asynchronous check: the call returns (poisoned), the verdict surfaces at status() - or at the next call
            bad2 = _run(eng, case, 2)
            assert np.isnan(bad2).all()
            with pytest.raises(RuntimeError, match="hand-off timed out"):
                eng.status()
            assert eng.status() is True                                          # (one verdict per call)
            _run(eng, case, 2)
            with pytest.raises(RuntimeError, match="hand-off timed out"):
                _run(eng, case, 2)                                               # the previous call's verdict, before anything new is enqueued
            assert eng.status() is True
        finally:
            del os.environ["DEX_DEBUG_DROP_HANDOFF"]
        assert np.isnan(bad).all()
        assert np.array_equal(good, _run(eng, case, 2)) and eng.handoff_timeouts() == 0      # the next call is clean again
    finally:
        eng.set_precision("fp32")


@pytest.mark.parametrize("name,kw,n", [
    ("gedex_lj", dict(B=1, T=512), 6),                                    # 21 clusters in 3 rounds of 8 (3 idle cluster slots)
    ("gedex_lj", dict(B=3, T=512, lengths=[512, 300, 77]), 3),            # 63 clusters -> 64 slots = 256 workgroups: the largest XCD-local grid
    ("gedex_lj", dict(B=2, T=100, lengths=[100, 61]), 4),
    ("dex_vctk", dict(B=1, T=64, lengths=[57], Tr=40, Ts=40, sty_lengths=[33]), 4),
])
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_xcd_local_clusters_equal_cross_xcd_clusters_bitwise(name, kw, n, prec):
    """The members of a cluster on ONE XCD (hand-offs through its L2: plain stores, sc0 loads) against the members dealt across
    XCDs (write-through stores, memory-scope loads; DEX_DIT_CLUSTER_LOCAL=0): the arithmetic and its order are the same, so the
    bits are; repeated and replayed calls stay identical (a stale L2 line would show), and no hand-off reports an error."""
    cfg, eng, w = U.engine_for(name)
    if eng.xcd_local() != 1:
        pytest.skip("this device does not deal workgroup b to XCD b % 8: the XCD-local form is off")
    case = U.make_case(cfg, **kw)
    eng.set_precision(prec)
    try:
        a = _run(eng, case, n)
        assert eng.handoff_timeouts() == 0
        for _ in range(12):
            assert np.array_equal(a, _run(eng, case, n)), eng.handoff_timeouts()
        for _ in range(2):
            g = _run(eng, case, n, graph=True)
            assert np.array_equal(a, g), (eng.handoff_timeouts(), int(np.isnan(g).sum()))
        assert eng.handoff_timeouts() == 0
        os.environ["DEX_DIT_CLUSTER_LOCAL"] = "0"
        try:
            b = _run(eng, case, n)
            assert eng.handoff_timeouts() == 0
        finally:
            del os.environ["DEX_DIT_CLUSTER_LOCAL"]
        assert np.isfinite(a).all() and np.array_equal(a, b), float(np.abs(a - b).max())
    finally:
        eng.set_precision("fp32")


def test_lost_handoff_is_loud_on_graph_replays_too():
    """a hipGraph REPLAY enqueues nothing on the host, so the status check must read the word the captured launches write
    (kept in the graph entry).  With the drop knob set, the first graph call captures (and fails), the second is a cache hit under an
    unchanged key - both must raise, and handoff_timeouts() must see the word after each."""
    cfg, eng, w = U.engine_for("gedex_lj")
    case = U.make_case(cfg, B=1, T=64)
    eng.set_precision("bf16")
    try:
        good = _run(eng, case, 2, graph=True)
        assert np.array_equal(good, _run(eng, case, 2, graph=True)) and eng.handoff_timeouts() == 0     # clean capture, clean replay
        os.environ["DEX_DEBUG_DROP_HANDOFF"] = "1"
        try:
            for mode in (True, "deferred"):                                      # the synchronous and the asynchronous check
                eng.check_handoffs = mode
                for _ in range(3):                                               # capture, then two replay hits
                    with pytest.raises(RuntimeError, match="hand-off timed out"):
                        _run(eng, case, 2, graph=True)
                        eng.status()
                    assert eng.handoff_timeouts() == 1
        finally:
            eng.check_handoffs = "deferred"
            del os.environ["DEX_DEBUG_DROP_HANDOFF"]
        assert np.array_equal(good, _run(eng, case, 2, graph=True)) and eng.handoff_timeouts() == 0     # the clean graph is still cached and clean
    finally:
        eng.set_precision("fp32")


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_clusters_packed_onto_three_xcds_are_bit_identical(prec):
    """DEX_DIT_XCDS=3: the 21 clusters of the headline shape dealt to 3 XCDs (7 x 4 workgroups each) instead of 8 - same arithmetic, same
    bits; keeps the packed blockIdx -> (cluster, member) mapping covered while the default stays 8."""
    cfg, eng, w = U.engine_for("gedex_lj")
    if eng.xcd_local() != 1:
        pytest.skip("XCD-local form off on this device")
    case = U.make_case(cfg, B=1, T=512)
    eng.set_precision(prec)
    try:
        a = _run(eng, case, 3)
        os.environ["DEX_DIT_XCDS"] = "3"
        try:
            b = _run(eng, case, 3)
            g = _run(eng, case, 3, graph=True)
            assert eng.handoff_timeouts() == 0
        finally:
            del os.environ["DEX_DIT_XCDS"]
        assert np.isfinite(a).all() and np.array_equal(a, b) and np.array_equal(a, g)
    finally:
        eng.set_precision("fp32")


def test_l2_scope_handoff_across_xcds_is_loud():
    """DEX_DEBUG_DROP_HANDOFF=2 runs the XCD-local protocol on clusters whose members sit on DIFFERENT XCDs (what a device with
    another workgroup placement rule would do to it): the peers' plain stores never reach this XCD's L2 in time (time-out, 1) or
    arrive with a foreign XCC id in the flag (2) - either way the device word is set and the call's outputs are NaN."""
    cfg, eng, w = U.engine_for("gedex_lj")
    if eng.xcd_local() != 1:
        pytest.skip("XCD-local form off on this device")
    case = U.make_case(cfg, B=1, T=64)
    eng.set_precision("bf16")
    try:
        good = _run(eng, case, 2)
        assert np.isfinite(good).all() and eng.handoff_timeouts() == 0
        os.environ["DEX_DEBUG_DROP_HANDOFF"] = "2"
        try:
            with pytest.raises(RuntimeError, match="hand-off"):
                _run(eng, case, 2)
                eng.status()
            eng.check_handoffs = False
            try:
                bad = _run(eng, case, 2)
            finally:
                eng.check_handoffs = "deferred"
            assert eng.handoff_timeouts() in (1, 2)
        finally:
            del os.environ["DEX_DEBUG_DROP_HANDOFF"]
        assert np.isnan(bad).all()
        assert eng.xcd_local() == 1                                        # (the debug run does not switch the form off)
        assert np.array_equal(good, _run(eng, case, 2)) and eng.handoff_timeouts() == 0
    finally:
        eng.set_precision("fp32")


@pytest.mark.parametrize("name,kw", [
    ("gedex_lj", dict(B=1, T=512)),                                        # 7 x 7 / stride 4 patches
    ("gedex_lj", dict(B=2, T=100, lengths=[100, 61])),                     # width not a multiple of the patch: right zero padding
    ("dex_vctk", dict(B=1, T=64, lengths=[57], Tr=40, Ts=40, sty_lengths=[33])),      # 3 x 3 / stride 2 patches
])
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_fused_patch_embed_is_bit_identical(name, kw, prec):
    """PatchEmbed2D as one launch at small grids (patch_embed.hip) against the two-kernel form (DEX_PATCH_FUSED=0): same bits."""
    cfg, eng, w = U.engine_for(name)
    case = U.make_case(cfg, **kw)
    eng.set_precision(prec)
    try:
        a = _run(eng, case, 3)
        os.environ["DEX_PATCH_FUSED"] = "0"
        try:
            b = _run(eng, case, 3)
        finally:
            del os.environ["DEX_PATCH_FUSED"]
        assert np.isfinite(a).all() and np.array_equal(a, b), float(np.abs(a - b).max())
    finally:
        eng.set_precision("fp32")
