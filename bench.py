#!/usr/bin/env python
"""bench.py — mel-frames/s of the reverse-diffusion hot path (Diffusion.forward(infer=True) == EDM Euler
sampler) on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE full sampler call (n_timesteps Euler steps) over one batch of synthetic utterances.
Default workload = BASELINE.json configs[1]: GeDEX-LJ, B=1, T=512 mel frames, n_timesteps=50, bf16.
For N>1 the batch is B utterances PER GPU (weak scaling): every rank holds only its shard, samples it with the product
sharder (dex_tts_amd.dist.sample_sharded) and the finished mels are all-gathered over RCCL inside the timed region
(the path's one exchange step).  Prints ONE JSON line on rank 0.

The default N=1 run also carries, in the same line: `roofline` for the dominant kernel SYMBOL of the workload (HIP
events on the launch stream; HBM traffic from the committed PMC summary), `roofline_attention` (the DiT attention
kernel on its own), `fp32_mode`, `roofline_batch32`, driver-timed blocks for BASELINE.json configs[2], [3] (per-GPU
share) and [4] under `configs`, and `cpu_baseline` (the CPU oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import statistics
import sys
import time

# kernel arguments in device memory: shaves ~0.7 us off every launch of this launch-bound workload (must be set
# before the HIP runtime initialises)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dex_tts_amd import config as C, synth  # noqa: E402

WORKLOADS = {
    # name: (preset, B per GPU, T, n_timesteps, Tr/Ts, precision or None = --precision)
    "gedex_b1": ("gedex_lj", 1, 512, 50, 0, None),               # BASELINE.json configs[1]
    "gedex_b1_t800": ("gedex_lj", 1, 800, 50, 0, None),
    "gedex_long_x2": ("gedex_lj", 1, 4000, 50, 0, "fp16x2"),     # configs[4]'s job in the split-weight mode (its parity leg)
    "gedex_b2": ("gedex_lj", 2, 512, 50, 0, None),               # small batches: the cluster form of the DiT block covers B x 21 <= 64 row tiles
    "gedex_b3": ("gedex_lj", 3, 512, 50, 0, None),
    "gedex_b32": ("gedex_lj", 32, 512, 50, 0, None),
    "dex_b1": ("dex_vctk", 1, 512, 50, 348, None),
    "dex_b32": ("dex_vctk", 32, 256, 50, 348, None),             # configs[2]
    "dex_b32_t512": ("dex_vctk", 32, 512, 50, 348, None),        # SURVEY 8(d) C3, the longer utterances
    "dex_esd_b32_n100": ("dex_esd", 32, 256, 100, 348, None),    # per-GPU share of configs[3] (256 utterances / 8 GPUs)
    "dex_esd_b256_n100": ("dex_esd", 256, 256, 100, 348, None),  # configs[3] undivided: --scaling strong deals the 256 utterances over the ranks
    "gedex_long": ("gedex_lj", 1, 4000, 50, 0, None),            # configs[4] shape
    "dex_libritts_b8": ("dex_libritts", 8, 256, 50, 348, None),  # not a BASELINE config: the dim-128 / hidden-384 geometry (per-operation reduced precision)
}
CONFIG_TAG = {"gedex_b1": "BASELINE.json configs[1]", "dex_b32": "BASELINE.json configs[2]", "dex_b32_t512": "SURVEY 8(d) C3 at T=512",
              "gedex_b1_t800": "SURVEY 8(d) C2 at T=800", "gedex_b32": "BASELINE.json metric: batch = 32",
              "dex_esd_b32_n100": "BASELINE.json configs[3], per-GPU share (256 utterances / 8 GPUs)",
              "dex_esd_b256_n100": "BASELINE.json configs[3], all 256 utterances (strong scaling: / N ranks)",
              "gedex_long": "BASELINE.json configs[4] shape", "gedex_long_x2": "BASELINE.json configs[4] shape, fp16x2"}
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
# what the chip SUSTAINS on dense 16-bit MFMA with random operands on all CUs (tools/mfmaceil, profiles/round3_mfma_ceiling_random_operands.txt):
# it clocks down to ~1.7 GHz under that load.  Reported next to the nominal-peak fraction, never instead of it.
MEASURED_MFMA_CEILING_TFLOPS = {"bf16": 1772.0, "f16": 1642.0}
DTYPE_KEY = {"fp32": "f32", "bf16": "bf16", "fp16": "f16", "fp16x2": "f16"}      # (fp16x2: fp16 operands, weights as hi + lo - two MFMAs per weight product)
# library profile row -> rocprofv3 kernel symbol (rows of the conv kernels already carry their symbol)
SYMBOL_OF = {"dit_block": "dit_rowchain_kernel<true>", "dit_qkv": "dit_rowchain_kernel<false>", "dit_rowchain": "dit_rowchain_kernel<false>",
             "dit_attention": ("attn_q64_kernel", "attn_direct"),      # (whichever form the launch took: 64-query form / round-3 forms) "linattn_kvctx": "linattn_kvctx_kernel", "linattn_out": "linattn_out2",
             "linattn_merge": "linattn_merge_kernel", "first_conv": "first_conv_kernel", "final_conv_euler": "final_kernel",
             "pos_conv": "pos_conv_direct_kernel", "upsample_convT": "igemm_lp_ss_kernel", "downsample": "igemm_lp_kernel",
             "dit_final_unpatchify": "igemm_lp_ss_kernel", "tv_attention": "attn_lp", "patch_dwconv_silu": "dwconv_silu_kernel"}


# Whole-job errors against the CPU oracle are MEASURED IN THE RUN (they were constants copied from a test log): the pinned
# job of a preset (dex_tts_amd.synth.pinned_job_case: the same inputs tests/test_gpu_full_jobs.py uses) runs once on the engine and is
# compared with the oracle's committed OUTPUT of that job (tests/golden/oracle_jobs/*.npy - data written by oracle/make_oracle_jobs.py on
# the CPU; its file name hashes inputs, weights, config and oracle version, so a stale file is simply not found -> abs_err null).
PINNED_PRESET = {"dex_b32": "dex_vctk", "dex_esd_b32_n100": "dex_esd", "gedex_long": "gedex_lj", "gedex_long_x2": "gedex_lj"}


def job_abs_err(workload, precision, device, stream):
    """{"abs_err": [max, mean], "abs_err_source": ...} of the workload's pinned whole job in this mode, measured now"""
    from dex_tts_amd.engine import ScoreNetEngine
    preset = PINNED_PRESET[workload]
    case, n_steps = synth.pinned_job_case(preset)
    path = synth.stored_job_path(ROOT, preset, case, n_steps)
    if not os.path.exists(path):
        return {"abs_err": None, "abs_err_source": f"no stored oracle job for these inputs ({os.path.basename(path)})"}
    cfg = C.PRESETS[preset]()
    eng = ScoreNetEngine(cfg, device)
    eng.load_weights({k: torch.from_numpy(v) for k, v in synth.make_weights(C.param_shapes(cfg)).items()})
    eng.set_precision(precision)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    kw = {}
    if "ref" in case:
        kw = dict(ref=[t(r) for r in case["ref"]], sty=t(case["sty"]), sty_lengths=t(np.asarray(case["sty_lengths"])))
    if "spk" in case:
        kw["spk"] = t(case["spk"])
    with torch.cuda.stream(stream):
        got = eng.sample(t(case["z"]), t(case["mask"]), t(case["mu"]), n_steps, use_graph=True, **kw)
        torch.cuda.synchronize(device)
    e = np.abs(got.float().cpu().numpy() - np.load(path))
    del eng
    torch.cuda.empty_cache()
    return {"abs_err": [float(f"{e.max():.3e}"), float(f"{e.mean():.3e}")],
            "abs_err_source": f"measured in this run: the pinned {preset} job ({n_steps} steps, whole-call graph) vs {os.path.relpath(path, ROOT)}"}


def dtype_name(precision):
    return precision


def lengths_for(B, T, salt=0):
    return [T] * B if B == 1 else [int(T * (0.6 + 0.4 * ((7 * i + 3 * salt) % 11) / 10.0)) for i in range(B)]


def make_inputs(cfg, lengths, T, TrTs, device, seed):
    B = len(lengths)
    mu, mask, z, _ = synth.make_inputs(B, T, list(lengths), seed=seed)
    kw = {}
    if cfg.variant == "dex":
        ref, rl, sty, sl = synth.make_dex_style(B, TrTs, TrTs, cfg.mid_dim)
        kw = dict(ref=[torch.from_numpy(r).to(device) for r in ref], sty=torch.from_numpy(sty).to(device),
                  sty_lengths=torch.from_numpy(sl).to(device))
    t = lambda a: torch.from_numpy(a).to(device)
    return t(mu), t(mask), t(z), kw


def host_cpu():
    """What the CPU baseline ran on: logical CPUs visible to this process, physical cores and the model string (north_star: "core count stated")."""
    info = {"logical_cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "physical_cores": None, "model": None}
    try:
        cores, model = set(), None
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
        info["physical_cores"], info["model"] = (len(cores) or None), model
    except OSError:
        pass
    return info


def cpu_baseline(cfg, weights, B, T, n_timesteps, TrTs, full=False):
    """Oracle (CPU restatement of the reference, torch ops on the host cores) on a bounded sample of the same workload: a 3-step
    sampler scaled to n_timesteps, or (full=True, the B=1 headline job: ~5-10 s) the WHOLE n_timesteps job - its result is then
    returned too, as the checker the modes' ``abs_err`` fields are measured against."""
    from oracle import dex_oracle as O
    W = O.as_torch(weights)
    mu, mask, z, _ = synth.make_inputs(B, T, None, seed=1234)
    kw = {}
    if cfg.variant == "dex":
        ref, rl, sty, sl = synth.make_dex_style(B, TrTs, TrTs, cfg.mid_dim)
        kw = dict(ref=[torch.from_numpy(r) for r in ref], sty=torch.from_numpy(sty), sty_lengths=torch.from_numpy(sl))
    mu, mask, z = map(torch.from_numpy, (mu, mask, z))
    nsub = 3
    all_cores = torch.get_num_threads()
    best = None
    with torch.no_grad():
        # oversubscribed hosts run this small-tensor workload slower on all cores: take the better of two settings
        for nt in sorted({all_cores, min(all_cores, 16)}):
            torch.set_num_threads(nt)
            O.diffusion_infer(W, cfg, mask, mu, 2, z, **kw)        # warm-up (oneDNN primitive caches)
            t0 = time.perf_counter()
            O.diffusion_infer(W, cfg, mask, mu, nsub, z, **kw)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, nt)
        dt, nt = best
        torch.set_num_threads(nt)
        y = None
        if full:
            t0 = time.perf_counter()
            y = O.diffusion_infer(W, cfg, mask, mu, n_timesteps, z, **kw)
            dt, nsub = time.perf_counter() - t0, n_timesteps
    per_step = dt / nsub
    frames_s = B * T / (per_step * n_timesteps)
    sample = (f"the whole job: {n_timesteps} Euler steps (B={B}, T={T})" if full else
              f"{nsub} of {n_timesteps} Euler steps of the same workload (B={B}, T={T}), scaled x{n_timesteps}/{nsub}")
    return {"value": round(frames_s, 2), "unit": "mel-frames/s", "cores": torch.get_num_threads(), "kind": "port", "host": host_cpu(),
            "sample": f"{sample}; {per_step * 1e3:.1f} ms/Euler-step on {torch.get_num_threads()} threads"}, y


_PMC = None


_PHEAD = None


def profiles_head():
    """{"commit", "kernel_sources_sha", "stale"}: the tree the committed rocprof / PMC summaries (spliced into roofline entries as
    rocprof_avg_launch_us / traffic) were measured on, and whether this run's kernel sources differ from it"""
    global _PHEAD
    if _PHEAD is None:
        path = os.path.join(ROOT, "profiles", "profiles_head.json")
        cur = synth.kernel_sources_sha(ROOT)
        if os.path.exists(path):
            d = json.load(open(path))
            _PHEAD = {"commit": d.get("commit_at_collection"), "kernel_sources_sha": d.get("kernel_sources_sha"), "this_tree_sha": cur,
                      "stale": d.get("kernel_sources_sha") != cur}
        else:
            _PHEAD = {"commit": None, "kernel_sources_sha": None, "this_tree_sha": cur, "stale": True}
        if _PHEAD["stale"]:
            sys.stderr.write("bench.py: WARNING the committed profiles/ summaries were measured on other kernel sources than this tree "
                             "(profiles_head.stale): the spliced rocprof_avg_launch_us / traffic fields describe the OLD kernels\n")
    return _PHEAD


def symbols_of(row_name):
    """rocprofv3 kernel symbol(s) a library profile row may appear under, most specific first"""
    v = SYMBOL_OF.get(row_name, row_name)
    return list(v) if isinstance(v, tuple) else [v]


def pmc_traffic(workload, row_name):
    """HBM bytes per launch of one kernel symbol from the committed PMC summary (profiles/pmc_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, FETCH doubled as MI355X_MICROARCH.md §HBM says)."""
    global _PMC
    if _PMC is None:
        path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        _PMC = json.load(open(path)) if os.path.exists(path) else {}
    # (the ESD workload launches the same kernels on the same shapes as dex_b32, twice as many steps)
    table = _PMC.get({"dex_esd_b32_n100": "dex_b32"}.get(workload, workload), {})
    norm = lambda n: n.replace(" ", "").replace("true", "1").replace("false", "0")
    for sym in symbols_of(row_name):
        for k, v in table.items():              # keys are normalised kernel names (tools/pmc_json.py), most-fetched first
            if norm(sym) in norm(k):
                return v
    return None


_STATS = {}
PROFILE_DTYPE = {"gedex_b1": "bf16", "gedex_b32": "bf16", "dex_b32": "bf16", "dex_esd_b32_n100": "bf16", "gedex_long": "f16"}


def rocprof_avg_us(workload, row_name, attention=False):
    """Average launch duration of one kernel symbol in the COMMITTED rocprofv3 --kernel-trace --stats summary of this workload
    (profiles/round<N>_<workload>[_attention_separate]_kernel_stats.csv, newest round first) — kernel-only time, without the
    ~2-3 us of dispatch a HIP-event bracket carries; the live event number must agree with it."""
    import csv
    key = (workload, attention)
    if key not in _STATS:
        _STATS[key] = (None, {})
        for rnd in (6, 5, 4, 3, 2):
            path = os.path.join(ROOT, "profiles", f"round{rnd}_{workload}{'_attention_separate' if attention else ''}_kernel_stats.csv")
            if os.path.exists(path):
                with open(path) as f:
                    _STATS[key] = (os.path.relpath(path, ROOT), {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(f)})
                break
    src, table = _STATS[key]
    norm = lambda n: n.replace(" ", "").replace("true", "1").replace("false", "0")
    hits = []
    for sym in symbols_of(row_name):
        hits = [(k, v) for k, v in table.items() if norm(sym) in norm(k)]
        if hits:
            break
    if not hits:
        return None, src
    return max(v for _, v in hits), src        # several instantiations of one symbol: the slowest (the row the event profile ranks first)


def roof(r, dtype_key, workload=None, force_mfma=False):
    """Roofline entry of one kernel symbol from its event-timed profile row.  The binding roof is the one that takes
    longer for the kernel's ALGORITHMIC work: flops / MFMA peak  vs  bytes / HBM peak."""
    sec = r["ms"] * 1e-3
    t_mfma = r["flops"] / (PEAK_TFLOPS[dtype_key] * 1e12)
    t_hbm = r["bytes"] / (PEAK_HBM_GBS * 1e9)
    ent = {"kernel": " | ".join(symbols_of(r["name"])), "profile_row": r["name"], "launches": r["calls"],
           "avg_launch_us": round(r["ms"] / r["calls"] * 1e3, 2),
           "algorithmic_GFLOP_per_launch": round(r["flops"] / r["calls"] / 1e9, 4),
           "algorithmic_MB_per_launch": round(r["bytes"] / r["calls"] / 1e6, 4),
           "mfma_TFLOP/s": round(r["flops"] / sec / 1e12, 2), "hbm_GB/s": round(r["bytes"] / sec / 1e9, 1), "traffic": None}
    if t_mfma >= t_hbm or force_mfma:
        ach = r["flops"] / sec / 1e12
        ent.update({"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[dtype_key], "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_TFLOPS[dtype_key], 4)})
        if dtype_key in MEASURED_MFMA_CEILING_TFLOPS:
            ent["frac_of_measured_mfma_ceiling"] = round(ach / MEASURED_MFMA_CEILING_TFLOPS[dtype_key], 4)
    else:
        ach = r["bytes"] / sec / 1e9
        ent.update({"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4)})
    if workload:
        t = pmc_traffic(workload, r["name"])
        if t:
            ent["traffic"] = t.get("hbm_bytes_per_launch")
            ent["traffic_source"] = t.get("source")
            ent["profiles_stale"] = profiles_head()["stale"]
        # (the committed profiles were taken in the mode each config names: bf16, long-form fp16 - other modes launch other kernels)
        us, src = (rocprof_avg_us(workload, r["name"], attention=force_mfma and r["name"] == "dit_attention")
                   if dtype_key == PROFILE_DTYPE.get(workload) else (None, None))
        if us:
            ent["rocprof_avg_launch_us"] = round(us, 2)
            ent["rocprof_source"] = src
            ent["profiles_stale"] = profiles_head()["stale"]
            ent["frac_at_rocprof_duration"] = round(ent["frac"] * ent["avg_launch_us"] / us, 4)
    return ent


def timed_calls(call, steps, warmup, device, dist=None):
    """W untimed calls, then EXACTLY K calls between barrier + synchronize on both sides.  Returns (wall seconds,
    per-call HIP-event milliseconds measured on the launch stream)."""
    for _ in range(warmup):
        call()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    out = None
    for a, b in evs:
        a.record()
        out = call()
        b.record()
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    return dt, [a.elapsed_time(b) for a, b in evs], out


def profile_rows(eng, call, device):
    eng.profile(True)
    call()
    torch.cuda.synchronize(device)
    rows = sorted(eng.profile_rows(), key=lambda r: -r["ms"])
    eng.profile(False)
    return rows


def attention_row(eng, call, device, rows):
    att = [r for r in rows if r["name"] == "dit_attention"]
    if att:
        return att[0]
    # reduced-precision modes run the attention core inside the fused DiT-block launch; one extra profiling pass with
    # the separate attention kernel (same wave body, attention_direct.hip) times it on its own
    os.environ["DEX_ATTN_SEPARATE"] = "1"
    try:
        att = [r for r in profile_rows(eng, call, device) if r["name"] == "dit_attention"]
    finally:
        del os.environ["DEX_ATTN_SEPARATE"]
    return att[0] if att else None


def pick_graph(eng_call, device, mode):
    """--graph auto: two calls each way, keep the faster (reported in config.hipgraph)."""
    if mode != "auto":
        return mode == "on"
    best = {}
    for g in (False, True):
        eng_call(g); eng_call(g)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(3):
            eng_call(g)
        torch.cuda.synchronize(device)
        best[g] = time.perf_counter() - t0
    return best[True] < best[False]


def side_workload(name, precision, device, stream, graph_mode, steps=3, warmup=1, profile=True):
    """Driver-timed block for another BASELINE.json config on this GPU (same timing discipline, fewer calls)."""
    from dex_tts_amd.engine import ScoreNetEngine
    preset, B, T, n_steps, TrTs, prec_o = WORKLOADS[name]
    prec = prec_o or precision
    cfg = C.PRESETS[preset]()
    eng = ScoreNetEngine(cfg, device)
    eng.load_weights({k: torch.from_numpy(v) for k, v in synth.make_weights(C.param_shapes(cfg)).items()})
    eng.set_precision(prec)
    lengths = lengths_for(B, T)
    mu, mask, z, kw = make_inputs(cfg, lengths, T, TrTs, device, 1234)
    with torch.cuda.stream(stream):
        g = pick_graph(lambda gg: eng.sample(z, mask, mu, n_steps, use_graph=gg, **kw), device, graph_mode)
        call = lambda: eng.sample(z, mask, mu, n_steps, use_graph=g, **kw)
        dt, ev, out = timed_calls(call, steps, warmup, device)
        assert torch.isfinite(out).all()
        rows = profile_rows(eng, lambda: eng.sample(z, mask, mu, n_steps, use_graph=False, **kw), device) if profile else None
        att = attention_row(eng, lambda: eng.sample(z, mask, mu, n_steps, use_graph=False, **kw), device, rows) if profile else None
    valid = sum(lengths)
    key = DTYPE_KEY[prec]
    ent = {"workload": f"{name}: {preset} B={B} T={T} n_timesteps={n_steps}" + (f" Tr=Ts={TrTs}" if TrTs else "") + f" ({CONFIG_TAG.get(name, '')})",
           "value": round(valid * steps / dt, 1), "unit": "mel-frames/s", "dtype": key, "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 3), "hip_event_median_ms": round(statistics.median(ev), 3),
           "ms_per_euler_step": round(dt / steps * 1e3 / n_steps, 4), "hipgraph": g,
           "rtf": round((dt / steps) / (valid * 256 / 22050.0), 6)}
    if rows:
        ent["roofline"] = roof(rows[0], key, name)
    if att:
        ent["roofline_attention"] = roof(att, key, name, force_mfma=True)
    if rows:
        ent["kernels"] = [{"kernel": r["name"], "calls": r["calls"], "avg_us": round(r["ms"] / r["calls"] * 1e3, 2),
                       "share": round(r["ms"] / sum(q["ms"] for q in rows), 3)} for r in rows[:6]]
    del eng
    torch.cuda.empty_cache()
    return ent


def vocoder_block(device, stream, B=1, T=512, steps=10, warmup=3, big=False, precision="fp32"):
    """HiFi-GAN V1 generator (the step right after the sampler) on the mel of the headline workload; big: BigVGAN-base."""
    from dex_tts_amd import vocoder as V
    h = V.BIGVGAN_BASE if big else V.HIFIGAN_V1
    gen = V.Generator(V.AttrDict(h))
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_vocoder_weights(V.param_shapes(h)).items()})
    gen = gen.to(device).eval()
    gen.precision = precision
    mel = torch.from_numpy(synth.make_inputs(B, T, None, seed=1234)[0]).to(device)
    # algorithmic work: every Conv1d / ConvTranspose1d as 2 * L_out * Cin * Cout * taps-per-output
    fl, L, c = 2.0 * T * 80 * h["upsample_initial_channel"] * 7, T, h["upsample_initial_channel"]
    for u, k in zip(h["upsample_rates"], h["upsample_kernel_sizes"]):
        fl += 2.0 * L * c * (c // 2) * k
        L, c = L * u, c // 2
        fl += sum(2.0 * L * c * c * kk * 6 for kk in h["resblock_kernel_sizes"])
    fl += 2.0 * L * c * 7
    with torch.cuda.stream(stream):
        dt, ev, wav = timed_calls(lambda: gen(mel), steps, warmup, device)
    assert torch.isfinite(wav).all()
    sec = dt / steps
    name = "BigVGAN-base generator (anti-aliased snakebeta activations)" if big else "HiFi-GAN V1 generator (hifigan/config.json)"
    mode = "exact-fp32 MFMA" if precision == "fp32" else f"{precision} MFMA operands, fp32 accumulation"
    peak = PEAK_TFLOPS[DTYPE_KEY[precision]]
    return {"workload": f"{name}, B={B}, T={T} mel frames -> {wav.shape[-1]} samples, {mode}",
            "value": round(B * T / sec, 1), "unit": "mel-frames/s", "ms_per_call": round(sec * 1e3, 3), "hip_event_median_ms": round(statistics.median(ev), 3),
            "rtf": round(sec / (B * T * 256 / 22050.0), 6), "algorithmic_GFLOP": round(B * fl / 1e9, 1),
            "mfma_TFLOP/s": round(B * fl / sec / 1e12, 1), f"frac_of_{DTYPE_KEY[precision]}_mfma_peak": round(B * fl / sec / 1e12 / peak, 3)}


def frontend_block(device, stream, steps=10, warmup=3):
    """the stages in front of the sampler, once per utterance: DEX style encoders (348 reference frames) and the
    text encoder + durations + alignment (100 tokens), exact-fp32 MFMA, synthetic weights of the shipped geometries."""
    from dex_tts_amd import style as S, text as TX
    out = {}
    st = S.StyleEncoders()
    st.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_style_weights(st.shapes).items()})
    st = st.to(device).eval()
    mel, lf0, L = synth.make_style_inputs(1, 348, [348])
    mel, lf0, L = torch.from_numpy(mel).to(device), torch.from_numpy(lf0).to(device), torch.from_numpy(L).to(device)
    with torch.cuda.stream(stream):
        dt, ev, _ = timed_calls(lambda: st(mel, L, mel, L, lf0, L)[1], steps, warmup, device)
    out["style_encoders"] = {"workload": "TIV + TV + LF0 encoders + conv_sty, B=1, 348 frames", "ms_per_call": round(dt / steps * 1e3, 3),
                             "hip_event_median_ms": round(statistics.median(ev), 3)}
    kw = dict(n_vocab=149, n_feats=80, n_channels=192, filter_channels=1024, filter_channels_dp=256, n_heads=2, n_layers=8, kernel_size=3)
    te = TX.TextEncoder(**kw)
    te.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_text_weights(te.shapes).items()})
    te = te.to(device).eval()
    tok, tl = synth.make_text_inputs(1, 100, [100])
    tok, tl = torch.from_numpy(tok).to(device), torch.from_numpy(tl).to(device)

    def run():
        te(tok, tl)
        return te.align(return_attn=False)[0]
    with torch.cuda.stream(stream):
        dt, ev, mu_y = timed_calls(run, steps, warmup, device)
    out["text_encoder"] = {"workload": f"TextEncoder (8 RetNet layers, width 192) + durations + alignment, B=1, 100 tokens -> {mu_y.shape[-1]} frames",
                           "ms_per_call": round(dt / steps * 1e3, 3), "hip_event_median_ms": round(statistics.median(ev), 3)}
    return out


_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_us", "rocprof_avg_launch_us",
              "frac_at_rocprof_duration", "algorithmic_MB_per_launch", "algorithmic_GFLOP_per_launch")
COMPACT_LIMIT = 8192        # bytes: the driver reads the bench line out of an 8 KB tail of stdout (BENCH_r04.json: a 30.6 KB line came back parsed = null)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _err2(e):
    return None if not e else {"max": float(f"{e['max']:.3g}"), "mean": float(f"{e['mean']:.3g}")}


def compact(res):
    """The ONE stdout line: the contract's keys + roofline / roofline_attention / cpu_baseline / parity_mode / batch32 / per-config
    values, <= COMPACT_LIMIT bytes whatever the full record holds (tests/test_bench_line.py).  The full record goes to
    gpurun_out/bench_full.json and to stderr."""
    out = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                      "dtype", "data", "config", "rtf", "ms_per_euler_step", "hip_event_median_ms", "gpu_over_cpu"))
    if res.get("abs_err"):
        out["abs_err"] = _err2(res["abs_err"])
    for k in ("roofline", "roofline_attention"):
        if k in res:
            out[k] = _pick(res[k], _ROOF_KEYS)
    if "cpu_baseline" in res:
        cb = res["cpu_baseline"]
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample"))
        h = cb.get("host") or {}
        out["cpu_baseline"]["host"] = f"{h.get('model')}, {h.get('physical_cores')} physical cores, {h.get('logical_cpus')} logical CPUs"
    if "parity_mode" in res:
        pm = res["parity_mode"]
        out["parity_mode"] = _pick(pm, ("dtype", "value", "unit", "ms_per_step", "value_batch32", "ms_per_step_batch32"))
        out["parity_mode"]["abs_err"] = _err2(pm.get("abs_err"))
        if "exact_fp32_mode" in pm:
            e = pm["exact_fp32_mode"]
            out["exact_fp32_mode"] = dict(_pick(e, ("value", "ms_per_step", "value_batch32")), abs_err=_err2(e.get("abs_err")))
    if "fp16_mode" in res:
        out["fp16_mode"] = dict(_pick(res["fp16_mode"], ("value", "ms_per_step")), abs_err=_err2(res["fp16_mode"].get("abs_err")))
    if "batch32" in res:
        out["batch32"] = _pick(res["batch32"], ("value", "unit", "ms_per_step", "dtype", "workload", "hipgraph"))
        r32 = res.get("roofline_batch32", {}).get(res.get("_precision"), {})
        if "dominant" in r32:
            out["batch32"]["roofline"] = _pick(r32["dominant"], _ROOF_KEYS)
        if "dit_attention" in r32:
            out["batch32"]["roofline_attention"] = _pick(r32["dit_attention"], _ROOF_KEYS)
    if "batch32_bucketed" in res:
        out["batch32_bucketed"] = _pick(res["batch32_bucketed"], ("value", "ms_per_step", "bucket_width", "hipgraph"))
    if "configs" in res:
        out["configs"] = {}
        for name, c in res["configs"].items():
            ent = _pick(c, ("value", "dtype", "ms_per_step", "hipgraph"))
            ent["workload"] = c.get("workload", "").split(":")[0]
            for k, short in (("roofline", "dominant"), ("roofline_attention", "attention")):
                if k in c:
                    ent[short] = _pick(c[k], ("kernel", "bound", "frac", "avg_launch_us", "rocprof_avg_launch_us", "frac_at_rocprof_duration", "traffic"))
            if "abs_err" in c:
                ent["abs_err"] = c["abs_err"]
            out["configs"][name] = ent
    for k in ("vocoder", "vocoder_bf16", "vocoder_bigvgan"):
        if k in res:
            out[k] = _pick(res[k], ("value", "ms_per_call"))
    if "frontend" in res:
        out["frontend_ms"] = {k: v.get("ms_per_call") for k, v in res["frontend"].items()}
    out["full_record"] = res.get("_full_path")
    ph = profiles_head()
    out["profiles_head"] = {"commit": ph["commit"], "stale": ph["stale"]}       # the tree the spliced rocprof_avg_launch_us / traffic fields were measured on
    out["abs_err_measured"] = "in this run"                                     # configs[*].abs_err: pinned jobs vs the oracle's stored outputs (job_abs_err)
    # belt and braces: drop the optional blocks, least important first, until the line fits
    for k in ("frontend_ms", "vocoder_bigvgan", "vocoder_bf16", "vocoder", "batch32_bucketed", "fp16_mode", "exact_fp32_mode", "configs"):
        if len(json.dumps(out)) < COMPACT_LIMIT - 256:
            break
        out.pop(k, None)
    return out


def emit(res):
    """Full record -> gpurun_out/bench_full.json (+ stderr); compact record -> the last (and only) stdout line."""
    full_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(full_dir, exist_ok=True)
        path = os.path.join(full_dir, f"bench_full_{res.get('_workload', 'run')}_n{res.get('n_gpus', 1)}.json")
        res["_full_path"] = os.path.relpath(path, ROOT)
        with open(path, "w") as f:
            json.dump({k: v for k, v in res.items() if not k.startswith("_")}, f)
    except OSError:
        res["_full_path"] = None
    res["profiles_head"] = profiles_head()
    print("[bench full record] " + json.dumps({k: v for k, v in res.items() if not k.startswith("_")}), file=sys.stderr, flush=True)
    line = json.dumps(compact(res))
    assert len(line) < COMPACT_LIMIT, len(line)
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="gedex_b1", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="bf16", choices=sorted(__import__("dex_tts_amd._lib", fromlist=["x"]).PRECISION))
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="whole-sampler hipGraph replay: auto = measure both at start-up and keep the faster")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): the workload's B utterances PER GPU; strong: its B utterances IN TOTAL, dealt over the ranks "
                         "(BASELINE configs[3]: --workload dex_esd_b256_n100 --scaling strong)")
    ap.add_argument("--bucket-width", type=int, default=0,
                    help="opt-in length bucketing of the timed job (dex_tts_amd.dist.sample_bucketed): utterances whose length rounds up to the "
                         "same multiple of this many frames are padded to their own maximum and sampled as a batch of their own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs[2]/[3]/[4] blocks of the default run")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (nccl == RCCL over xGMI; gloo only for single-GPU smoke tests)")
    ap.add_argument("--all-on-device0", action="store_true",
                    help="testing aid: every rank uses cuda:0 (exercises the N>1 code path on a 1-GPU box; use with --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback in the product path)"
    if args.all_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    from dex_tts_amd import dist as D
    from dex_tts_amd.engine import ScoreNetEngine

    preset, B, T, n_steps, TrTs, prec_o = WORKLOADS[args.workload]
    precision = prec_o or args.precision
    cfg = C.PRESETS[preset]()
    weights = synth.make_weights(C.param_shapes(cfg))
    eng = ScoreNetEngine(cfg, device)
    eng.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
    eng.set_precision(precision)
    # the job: B utterances per GPU.  Every rank derives the same global length list and deal; it materialises ONLY the
    # utterances of its own shard (dist.partition is a pure function of the lengths)
    if args.scaling == "strong":
        # the workload's B utterances in total (the same list whatever N: groups of 32 with the salted length law of the weak mode)
        lengths = [l for r in range((B + 31) // 32) for l in lengths_for(min(32, B), T, r)][:B] if B > 1 else lengths_for(B, T)
    else:
        lengths = [l for r in range(world) for l in lengths_for(B, T, r)] if world > 1 else lengths_for(B, T)
    mine = D.partition(lengths, world)[rank]
    if args.bucket_width:
        # bucketing re-deals every bucket over the ranks: each rank materialises the full job (inputs are 164 KB per utterance)
        mu, mask, z, kw = make_inputs(cfg, lengths, T, TrTs, device, 1234)
    else:
        mu, mask, z, kw = make_inputs(cfg, [lengths[i] for i in mine], T, TrTs, device, 1234 + rank)
    stream = torch.cuda.Stream(device)
    with torch.cuda.stream(stream):
        use_graph = pick_graph(lambda g: eng.sample(z, mask, mu, n_steps, use_graph=g, **kw), device, args.graph)
        if args.bucket_width:
            sample_fn = lambda zz, mm, uu, **k2: eng.sample(zz, mm, uu, n_steps, use_graph=use_graph, **k2)
            one_call = lambda: D.sample_bucketed(sample_fn, mu, mask, z, lengths, args.bucket_width, extras=kw)
        else:
            sample_fn = lambda zz, mm, uu: eng.sample(zz, mm, uu, n_steps, use_graph=use_graph, **kw)
            # N == 1: sample_sharded degenerates to one sampler call; N > 1: shard sampler + the ONE all-gather + index_copy_
            one_call = lambda: D.sample_sharded(sample_fn, mu, mask, z, lengths, local=True)
        dt, ev_ms, out = timed_calls(one_call, args.steps, args.warmup, device, dist)
    gdev = device if (world == 1 or args.backend == "nccl") else torch.device("cpu")
    if world > 1:
        t = torch.tensor([dt], device=gdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        assert out.shape[0] == len(lengths)
    valid_total = sum(lengths)
    assert torch.isfinite(out).all()

    if rank == 0:
        dtype = DTYPE_KEY[precision]
        ms_per_step = dt / args.steps * 1e3
        frames_s = valid_total * args.steps / dt
        audio_s = valid_total * 256 / 22050.0
        res = {
            "metric": f"mel-frames/s at n_timesteps={n_steps}, 80-ch mel (sampler only); RTF",
            "value": round(frames_s, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": dtype, "data": "synthetic (portable random weights with zero-inits overridden; mel-like mu; mask from lengths)",
            "config": {"workload": f"{args.workload}: {preset} B={B}{'/GPU' if args.scaling == 'weak' else ' in total'} T={T} n_timesteps={n_steps}"
                                   + (f" Tr=Ts={TrTs}" if TrTs else "") + (f" ({CONFIG_TAG[args.workload]})" if args.workload in CONFIG_TAG else ""),
                       "global_batch": len(lengths), "bucket_width": args.bucket_width or None, "frames": T, "n_timesteps": n_steps, "hipgraph": use_graph,
                       "parallelism": f"{world} replica(s), utterances dealt by dex_tts_amd.dist.partition, each rank holds its shard only; "
                                      "one RCCL all-gather of the finished mels per call" if world > 1 else "1 GPU"},
            "rtf": round((dt / args.steps) / audio_s, 6),
            "ms_per_euler_step": round(ms_per_step / n_steps, 4),
            "hip_event_median_ms": round(statistics.median(ev_ms), 3),
            "hip_event_min_ms": round(min(ev_ms), 3),
            "_precision": precision, "_workload": args.workload,
        }
        prof = world == 1 and not args.no_profile
        call_eager = lambda: eng.sample(z, mask, mu, n_steps, use_graph=False, **kw)
        y_oracle = None
        err = lambda y: None if y_oracle is None else {"max": float((y.float().cpu() - y_oracle).abs().max()), "mean": float((y.float().cpu() - y_oracle).abs().mean()),
                                                       "against": "CPU oracle (fp32 restatement of the reference, pinned to it bit for bit), the whole job"}
        if world == 1 and not args.no_cpu_baseline:
            # (timed on the host cores while the GPU idles; for the B=1 headline job the oracle runs the WHOLE job and its result is the
            # checker every mode's abs_err below is measured against)
            res["cpu_baseline"], y_oracle = cpu_baseline(cfg, weights, B, T, n_steps, TrTs, full=(args.workload == "gedex_b1" and not args.bucket_width))
            res["gpu_over_cpu"] = round(frames_s / res["cpu_baseline"]["value"], 1)
            if y_oracle is not None:
                res["abs_err"] = err(out)
        if prof:
            with torch.cuda.stream(stream):
                # per-kernel HIP-event timing on the launch stream (eager launches, same work)
                rows = profile_rows(eng, call_eager, device)
                tot = sum(r["ms"] for r in rows)
                res["roofline"] = roof(rows[0], dtype, args.workload)          # dominant SYMBOL by time
                att = attention_row(eng, call_eager, device, rows)
            if att:
                res["roofline_attention"] = roof(att, dtype, args.workload, force_mfma=True)   # judged against the MFMA peak
                res["roofline_attention"]["note"] = "timed as a separate launch; the default path fuses it into dit_block"
            res["kernels"] = [{"kernel": r["name"], "calls": r["calls"], "avg_us": round(r["ms"] / r["calls"] * 1e3, 2),
                               "share": round(r["ms"] / tot, 3), "TFLOP/s": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2),
                               "GB/s": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1)} for r in rows[:int(os.environ.get("DEX_BENCH_TOPK", "8"))]]
            res["eager_event_total_ms"] = round(tot, 2)
        if prof and precision != "fp32":
            # companion number in the exact-fp32 MFMA mode (parity mode of the tests)
            eng.set_precision("fp32")
            n32c = max(5, args.steps // 4)
            with torch.cuda.stream(stream):
                g32 = lambda: eng.sample(z, mask, mu, n_steps, use_graph=use_graph, **kw)
                dt32, ev32, y_fp32 = timed_calls(g32, n32c, 2, device)
                r32 = profile_rows(eng, call_eager, device)
            eng.set_precision(precision)
            res["fp32_mode"] = {"value": round(valid_total * n32c / dt32, 1), "unit": "mel-frames/s", "ms_per_step": round(dt32 / n32c * 1e3, 3),
                                "steps": n32c, "warmup": 2, "hipgraph": use_graph, "hip_event_median_ms": round(statistics.median(ev32), 3),
                                "abs_err": err(y_fp32), "roofline": roof(r32[0], "f32")}
            del y_fp32
        if prof and precision == "bf16":
            # the fp16 operand mode of the same kernels: the mode that sits INSIDE the fp32 tolerance against the oracle at this shape
            # (tests/test_gpu_baseline_shapes.py: 50-step sampler max|d| 8.3e-4 vs 7.2e-3 in bf16); here its speed, and how far each
            # reduced-precision mode lands from the library's own exact-fp32 mode on this very job
            with torch.cuda.stream(stream):
                eng.set_precision("fp32")
                y32 = call_eager()
                ybf = None
                eng.set_precision("bf16")
                ybf = call_eager()
                eng.set_precision("fp16")
                g16 = lambda: eng.sample(z, mask, mu, n_steps, use_graph=use_graph, **kw)
                dt16, ev16, y16 = timed_calls(g16, max(3, args.steps // 4), 2, device)
                n16 = max(3, args.steps // 4)
            eng.set_precision(precision)
            dd = lambda a: {"max": float((a - y32).abs().max()), "mean": float((a - y32).abs().mean())}
            res["fp16_mode"] = {"value": round(valid_total * n16 / dt16, 1), "unit": "mel-frames/s", "ms_per_step": round(dt16 / n16 * 1e3, 3),
                                "steps": n16, "hip_event_median_ms": round(statistics.median(ev16), 3), "hipgraph": use_graph,
                                "abs_err": err(y16), "abs_diff_vs_fp32_mode": dd(y16), "bf16_abs_diff_vs_fp32_mode": dd(ybf),
                                "note": "same kernels compiled for fp16 MFMA operands (--precision fp16); fp32 sampler tolerance of tests/tolerances.py: "
                                        "max 5e-5 / mean 1e-5 (fp32 mode), reduced-precision bounds ibid."}
            del y32, ybf, y16
            if "fp16x2" in __import__("dex_tts_amd._lib", fromlist=["x"]).PRECISION:
                # the split-weight mode (fp16 operands, every weight as hi + lo, two MFMAs per product): the fast mode INSIDE the fp32-grade
                # sampler bound (max <= 1e-3, mean <= 1e-4 against the oracle) - timed exactly like the headline (same calls, same warm-ups,
                # same graph mode), at batch 1 here and at batch 32 below
                eng.set_precision("fp16x2")
                with torch.cuda.stream(stream):
                    gx2 = lambda: eng.sample(z, mask, mu, n_steps, use_graph=use_graph, **kw)
                    dtx, evx, yx = timed_calls(gx2, args.steps, args.warmup, device)
                eng.set_precision(precision)
                res["parity_mode"] = {"dtype": "f16x2", "precision": "fp16x2 (fp16 MFMA operands, weights split hi + lo, activations rounded once)",
                                      "value": round(valid_total * args.steps / dtx, 1), "unit": "mel-frames/s", "ms_per_step": round(dtx / args.steps * 1e3, 3),
                                      "ms_per_euler_step": round(dtx / args.steps * 1e3 / n_steps, 4), "steps": args.steps, "warmup": args.warmup,
                                      "hipgraph": use_graph, "hip_event_median_ms": round(statistics.median(evx), 3), "abs_err": err(yx),
                                      "bound": "50-step sampler against the oracle: max <= 1e-3 and mean <= 1e-4 (tests/test_gpu_fp16x2.py holds 8e-4 / 1e-4)"}
                del yx
        if prof and args.workload == "gedex_b1":
            # The B=1 headline workload is latency-bound (35 dependent launches of ~12 us per Euler step), so its roofline
            # fractions say little about the kernels.  Same kernels in the bandwidth/MFMA regime: one B=32 sampler call per
            # precision, event-timed per kernel symbol.
            p32, B32, T32, n32, _, _ = WORKLOADS["gedex_b32"]
            l32 = lengths_for(B32, T32)
            mu2, mask2, z2, kw2 = make_inputs(cfg, l32, T32, 0, device, 1234)
            scale = {}
            x2 = ["fp16x2"] if (precision != "fp32" and "parity_mode" in res) else []
            for prec in ([precision, "fp32"] + x2 if precision != "fp32" else ["fp32"]):
                key = DTYPE_KEY[prec]
                eng.set_precision(prec)
                nb = 5 if prec != "fp32" else 3
                with torch.cuda.stream(stream):
                    c2 = lambda: eng.sample(z2, mask2, mu2, n32, use_graph=use_graph, **kw2)
                    dtb, evb, _ = timed_calls(c2, nb, 2 if prec != "fp32" else 1, device)
                    dtb /= nb
                    c4 = lambda: eng.sample(z2, mask2, mu2, 4, **kw2)
                    rb = profile_rows(eng, c4, device)
                    attb = attention_row(eng, c4, device, rb)
                ent = {"value": round(sum(l32) / dtb, 1), "unit": "mel-frames/s", "workload": f"gedex_lj B={B32} T={T32} n_timesteps={n32}",
                       "steps": nb, "warmup": 2 if prec != "fp32" else 1, "hipgraph": use_graph, "ms_per_step": round(dtb * 1e3, 3),
                       "hip_event_median_ms": round(statistics.median(evb), 3), "ms_per_euler_step": round(dtb * 1e3 / n32, 4),
                       "dominant": roof(rb[0], key, "gedex_b32")}
                if attb:
                    ent["dit_attention"] = roof(attb, key, "gedex_b32", force_mfma=True)
                for r in rb:
                    if r["name"] in ("dit_block", "pos_conv"):
                        ent[r["name"]] = roof(r, key, "gedex_b32")
                ent["convs"] = [roof(r, key, "gedex_b32") for r in rb if r["name"].startswith("conv3x3")][:4]
                scale[prec] = ent
            eng.set_precision(precision)
            res["roofline_batch32"] = scale
            # BASELINE's metric is "batch = 1 & 32": the B = 32 value of the headline model next to the B = 1 one, same mode, same graph setting
            b32 = scale[precision]
            res["batch32"] = {k: b32[k] for k in ("value", "unit", "workload", "steps", "warmup", "hipgraph", "ms_per_step", "ms_per_euler_step", "hip_event_median_ms")}
            res["batch32"]["dtype"] = dtype
            # opt-in length bucketing on the same 32 utterances (lengths 0.6 T .. T): valid frames / s with every bucket padded to its own maximum
            eng.set_precision(precision)
            with torch.cuda.stream(stream):
                # (one cached hipGraph per bucket shape: the four buckets are four graph-cache entries of the engine, replayed back to back)
                # (the buckets are small grids - per-call host work matters - so they always run as graph replays, whatever
                # the padded batch picked)
                fnb = lambda zz, mm, uu, **k2: eng.sample(zz, mm, uu, n32, use_graph=True, **k2)
                cb = lambda: D.sample_bucketed(fnb, mu2, mask2, z2, l32, 64)
                dtk, evk, yk = timed_calls(cb, 3, 3, device)
            res["batch32_bucketed"] = {"value": round(sum(l32) * 3 / dtk, 1), "unit": "mel-frames/s", "bucket_width": 64,
                                       "buckets": [[Tb, len(ix)] for Tb, ix in D.buckets_of(l32, 64)], "steps": 3, "warmup": 3, "hipgraph": True,
                                       "ms_per_step": round(dtk / 3 * 1e3, 3),
                                       "note": "dex_tts_amd.dist.sample_bucketed: per bucket the result of the reference run on that bucket, NOT of the globally padded batch (opt-in)"}
            del mu2, mask2, z2, yk
            if "fp32_mode" in res and "fp32" in scale:
                # the exact-fp32 mode (every operation of the reference in fp32): its speed at batch 1 and 32 and its distance from the oracle
                fp32_leg = {"dtype": "f32", "value": res["fp32_mode"]["value"], "unit": "mel-frames/s", "ms_per_step": res["fp32_mode"]["ms_per_step"],
                            "steps": res["fp32_mode"]["steps"], "hipgraph": use_graph, "abs_err": res["fp32_mode"]["abs_err"],
                            "value_batch32": scale["fp32"]["value"], "ms_per_step_batch32": scale["fp32"]["ms_per_step"]}
                if "parity_mode" in res and "fp16x2" in scale:
                    res["parity_mode"].update({"value_batch32": scale["fp16x2"]["value"], "ms_per_step_batch32": scale["fp16x2"]["ms_per_step"],
                                               "steps_batch32": scale["fp16x2"]["steps"], "warmup_batch32": scale["fp16x2"]["warmup"],
                                               "exact_fp32_mode": fp32_leg,
                                               "note": "abs_err of every mode against the same oracle run: parity_mode.abs_err (fp16x2), fp16_mode.abs_err, abs_err (bf16 headline), fp32_mode.abs_err"})
                else:
                    res["parity_mode"] = fp32_leg
        if prof and args.workload == "gedex_b1" and not args.no_configs:
            del eng
            torch.cuda.empty_cache()
            from dex_tts_amd import _lib
            res["configs"] = {
                "configs[2]": side_workload("dex_b32", precision, device, stream, args.graph),
                "configs[3]": side_workload("dex_esd_b32_n100", precision, device, stream, args.graph, steps=2),
                "configs[4]": side_workload("gedex_long", "fp16" if "fp16" in _lib.PRECISION else precision, device, stream, "on"),
                "C3 (SURVEY 8d) T=512": side_workload("dex_b32_t512", precision, device, stream, "on", steps=3),
                "C2 (SURVEY 8d) T=800": side_workload("gedex_b1_t800", precision, device, stream, "on", steps=5, warmup=2),
            }
            # configs[2] / [3] name no reduced precision: their exact-fp32 leg and their split-weight (fp16x2) legs, driver-timed like the blocks
            # above.  abs_err = the pinned WHOLE job against the CPU oracle's stored output of it, measured in this run (job_abs_err): at
            # configs[2] the split mode's mean sits just outside the 1e-4 it holds at configs[1], so these legs are reported as a mode with
            # its measured error, not as "parity mode".
            res["configs"]["configs[2] exact fp32 mode"] = side_workload("dex_b32", "fp32", device, stream, "on", steps=2, profile=False)
            res["configs"]["configs[2] exact fp32 mode"].update(job_abs_err("dex_b32", "fp32", device, stream))
            res["configs"]["configs[2]"].update(job_abs_err("dex_b32", precision, device, stream))
            res["configs"]["configs[3]"].update(job_abs_err("dex_esd_b32_n100", precision, device, stream))
            res["configs"]["configs[4]"].update(job_abs_err("gedex_long", "fp16" if "fp16" in _lib.PRECISION else precision, device, stream))
            if "fp16x2" in _lib.PRECISION:
                res["configs"]["configs[2] fp16x2"] = side_workload("dex_b32", "fp16x2", device, stream, "on", steps=3, profile=False)
                res["configs"]["configs[2] fp16x2"].update(job_abs_err("dex_b32", "fp16x2", device, stream))
                res["configs"]["configs[3] fp16x2"] = side_workload("dex_esd_b32_n100", "fp16x2", device, stream, "on", steps=2, profile=False)
                res["configs"]["configs[3] fp16x2"].update(job_abs_err("dex_esd_b32_n100", "fp16x2", device, stream))
                res["configs"]["configs[4] fp16x2"] = side_workload("gedex_long_x2", "fp16x2", device, stream, "on", steps=3, profile=False)
                res["configs"]["configs[4] fp16x2"].update(job_abs_err("gedex_long_x2", "fp16x2", device, stream))
        if prof and args.workload == "gedex_b1" and not args.no_configs:
            res["vocoder"] = vocoder_block(device, stream)
            res["vocoder_bf16"] = vocoder_block(device, stream, precision="bf16")
            res["vocoder_fp16"] = vocoder_block(device, stream, precision="fp16")
            res["vocoder_bigvgan"] = vocoder_block(device, stream, big=True)
            res["frontend"] = frontend_block(device, stream)
        emit(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
