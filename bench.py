#!/usr/bin/env python
"""bench.py — mel-frames/s of the reverse-diffusion hot path (Diffusion.forward(infer=True) == EDM Euler
sampler) on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE full sampler call (n_timesteps Euler steps) over one batch of synthetic utterances.
Default workload = BASELINE.json configs[1]: GeDEX-LJ, B=1, T=512 mel frames, n_timesteps=50.
For N>1 every rank samples its own shard of independent utterances (weak scaling, no data-path collective)
and the finished mels are all-gathered over RCCL inside the timed region (the path's one exchange step).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# kernel arguments in device memory: shaves ~0.7 us off every launch of this launch-bound workload (must be set
# before the HIP runtime initialises)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dex_tts_amd import config as C, synth  # noqa: E402

WORKLOADS = {
    # name: (preset, B, T, n_timesteps, Tr/Ts)
    "gedex_b1": ("gedex_lj", 1, 512, 50, 0),
    "gedex_b1_t800": ("gedex_lj", 1, 800, 50, 0),
    "gedex_b32": ("gedex_lj", 32, 512, 50, 0),
    "dex_b1": ("dex_vctk", 1, 512, 50, 348),
    "dex_b32": ("dex_vctk", 32, 256, 50, 348),
    "gedex_long": ("gedex_lj", 1, 4000, 50, 0),
    "dex_esd_b32_n100": ("dex_esd", 32, 256, 100, 348),     # per-GPU share of BASELINE.json configs[3] (256 utterances / 8 GPUs)
}
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def make_inputs(cfg, B, T, TrTs, device, rank):
    lengths = None if B == 1 else [int(T * (0.6 + 0.4 * ((7 * i + 3 * rank) % 11) / 10.0)) for i in range(B)]
    mu, mask, z, _ = synth.make_inputs(B, T, lengths, seed=1234 + rank)
    kw = {}
    if cfg.variant == "dex":
        ref, rl, sty, sl = synth.make_dex_style(B, TrTs, TrTs, cfg.mid_dim)
        kw = dict(ref=[torch.from_numpy(r).to(device) for r in ref], sty=torch.from_numpy(sty).to(device),
                  sty_lengths=torch.from_numpy(sl).to(device))
    t = lambda a: torch.from_numpy(a).to(device)
    valid = int(mask.sum())
    return t(mu), t(mask), t(z), kw, valid


def cpu_baseline(cfg, weights, B, T, n_timesteps, TrTs):
    """Oracle (CPU restatement of the reference, torch ops on the host cores) on a bounded sample:
    a 3-step sampler of the same workload, scaled to n_timesteps."""
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
    per_step = dt / nsub
    frames_s = B * T / (per_step * n_timesteps)
    return {"value": round(frames_s, 2), "unit": "mel-frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{nsub} of {n_timesteps} Euler steps of the same workload (B={B}, T={T}), scaled x{n_timesteps}/{nsub}; "
                      f"{per_step * 1e3:.1f} ms/Euler-step on {torch.get_num_threads()} threads"}


def roof(r, dtype_key):
    """Roofline entry of one kernel class from its event-timed profile row.  The binding roof is the one that takes
    longer for the kernel's ALGORITHMIC work: flops / MFMA peak  vs  bytes / HBM peak."""
    sec = r["ms"] * 1e-3
    t_mfma = r["flops"] / (PEAK_TFLOPS[dtype_key] * 1e12)
    t_hbm = r["bytes"] / (PEAK_HBM_GBS * 1e9)
    ent = {"kernel": r["name"], "avg_launch_us": round(r["ms"] / r["calls"] * 1e3, 2),
           "mfma_TFLOP/s": round(r["flops"] / sec / 1e12, 2), "hbm_GB/s": round(r["bytes"] / sec / 1e9, 1), "traffic": None}
    if t_mfma >= t_hbm or r["name"] == "dit_attention":       # the DiT attention is judged against the MFMA peak
        ach = r["flops"] / sec / 1e12
        ent.update({"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[dtype_key], "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_TFLOPS[dtype_key], 4)})
    else:
        ach = r["bytes"] / sec / 1e9
        ent.update({"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4)})
    return ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="gedex_b1", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--graph", action="store_true", help="replay one captured hipGraph per Euler step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N>1 (nccl == RCCL over xGMI; gloo only for single-GPU smoke tests)")
    ap.add_argument("--all-on-device0", action="store_true",
                    help="testing aid: every rank uses cuda:0 (exercises the N>1 code path on a 1-GPU box; use with --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
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

    preset, B, T, n_steps, TrTs = WORKLOADS[args.workload]
    cfg = C.PRESETS[preset]()
    from dex_tts_amd.engine import ScoreNetEngine
    weights = synth.make_weights(C.param_shapes(cfg))
    eng = ScoreNetEngine(cfg, device)
    eng.load_weights({k: torch.from_numpy(v) for k, v in weights.items()})
    eng.set_precision(args.precision)
    mu, mask, z, kw, valid = make_inputs(cfg, B, T, TrTs, device, rank)
    use_graph = args.graph
    stream = torch.cuda.Stream(device)
    gdev = device if args.backend == "nccl" else torch.device("cpu")
    gathered = torch.empty(world * B, 80, T, device=gdev) if world > 1 else None

    def one_call():
        out = eng.sample(z, mask, mu, n_steps, use_graph=use_graph, **kw)
        if world > 1:       # the path's one exchange step: finished mels of every shard
            dist.all_gather_into_tensor(gathered, out if args.backend == "nccl" else out.cpu())
        return out

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            one_call()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = one_call()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=gdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        v = torch.tensor([float(valid)], device=gdev, dtype=torch.float64)
        dist.all_reduce(v)
        valid_total = int(v.item())
    else:
        valid_total = valid
    assert torch.isfinite(out).all()

    if rank == 0:
        dtype = "f32" if args.precision == "fp32" else "bf16"
        ms_per_step = dt / args.steps * 1e3
        frames_s = valid_total * args.steps / dt
        audio_s = valid_total * 256 / 22050.0
        res = {
            "metric": f"mel-frames/s at n_timesteps={n_steps}, 80-ch mel (sampler only); RTF",
            "value": round(frames_s, 1), "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic (portable random weights with zero-inits overridden; mel-like mu; mask from lengths)",
            "config": {"workload": f"{args.workload}: {preset} B={B}/GPU T={T} n_timesteps={n_steps}"
                                   + (f" Tr=Ts={TrTs}" if TrTs else "") + " (BASELINE.json configs[1])" * (args.workload == "gedex_b1"),
                       "global_batch": B * world, "frames": T, "n_timesteps": n_steps, "hipgraph": use_graph,
                       "parallelism": f"{world} independent replicas, utterance-sharded; RCCL all-gather of finished mels"},
            "rtf": round((dt / args.steps) / audio_s, 6),
            "ms_per_euler_step": round(ms_per_step / n_steps, 4),
        }
        if world == 1 and not args.no_profile:
            # per-kernel HIP-event timing on the launch stream (eager launches, same work)
            eng.profile(True)
            with torch.cuda.stream(stream):
                eng.sample(z, mask, mu, n_steps, use_graph=False, **kw)
                torch.cuda.synchronize(device)
            rows = sorted(eng.profile_rows(), key=lambda r: -r["ms"])
            eng.profile(False)
            tot = sum(r["ms"] for r in rows)
            kern = []
            for r in rows[:int(os.environ.get("DEX_BENCH_TOPK", "8"))]:
                tf = r["flops"] / (r["ms"] * 1e-3) / 1e12
                gb = r["bytes"] / (r["ms"] * 1e-3) / 1e9
                kern.append({"kernel": r["name"], "calls": r["calls"], "avg_us": round(r["ms"] / r["calls"] * 1e3, 2),
                             "share": round(r["ms"] / tot, 3), "TFLOP/s": round(tf, 2), "GB/s": round(gb, 1)})
            res["roofline"] = roof(rows[0], dtype)
            res["roofline"]["launches"] = rows[0]["calls"]
            att = [r for r in rows if r["name"] == "dit_attention"]
            if not att and args.precision == "bf16":
                # bf16 mode runs the attention core inside the fused DiT-block launch ("dit_block"); one extra profiling
                # pass with the separate attention kernel (same wave body, attention_direct.hip) times it on its own
                os.environ["DEX_ATTN_SEPARATE"] = "1"
                eng.profile(True)
                with torch.cuda.stream(stream):
                    eng.sample(z, mask, mu, n_steps, use_graph=False, **kw)
                    torch.cuda.synchronize(device)
                att = [r for r in eng.profile_rows() if r["name"] == "dit_attention"]
                eng.profile(False)
                del os.environ["DEX_ATTN_SEPARATE"]
            if att:
                res["roofline_attention"] = roof(att[0], dtype)
                res["roofline_attention"]["note"] = "timed as a separate launch; the default path fuses it into dit_block"
            res["kernels"] = kern
            res["eager_event_total_ms"] = round(tot, 2)
        if world == 1 and not args.no_profile and args.precision == "bf16":
            # companion number in the exact-fp32 MFMA mode (parity mode of the tests)
            eng.set_precision("fp32")
            with torch.cuda.stream(stream):
                eng.sample(z, mask, mu, n_steps, **kw)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(2):
                    eng.sample(z, mask, mu, n_steps, **kw)
                torch.cuda.synchronize(device)
                dt32 = (time.perf_counter() - t0) / 2
            eng.profile(True)
            with torch.cuda.stream(stream):
                eng.sample(z, mask, mu, n_steps, **kw)
                torch.cuda.synchronize(device)
            r32 = sorted(eng.profile_rows(), key=lambda r: -r["ms"])
            eng.profile(False)
            eng.set_precision("bf16")
            res["fp32_mode"] = {"value": round(valid * 1.0 / dt32, 1), "unit": "mel-frames/s", "ms_per_step": round(dt32 * 1e3, 3),
                                "roofline": roof(r32[0], "f32")}
        if world == 1 and not args.no_profile and args.workload == "gedex_b1":
            # The B=1 headline workload is launch/latency-bound (~47 dependent launches of a few us per Euler step), so
            # its roofline fractions say little about the kernels.  Same kernels in the bandwidth/MFMA regime:
            # one B=32 sampler call per precision, event-timed per kernel class.
            p32, B32, T32, n32, _ = WORKLOADS["gedex_b32"]
            mu2, mask2, z2, kw2, valid2 = make_inputs(cfg, B32, T32, 0, device, 0)
            scale = {}
            for prec, key in (("bf16", "bf16"), ("fp32", "f32")):
                eng.set_precision(prec)
                with torch.cuda.stream(stream):
                    eng.sample(z2, mask2, mu2, n32, **kw2)      # full warm-up call (plan, conditioning tables, clocks)
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    eng.sample(z2, mask2, mu2, n32, **kw2)
                    torch.cuda.synchronize(device)
                    dtb = time.perf_counter() - t0
                    eng.profile(True)
                    eng.sample(z2, mask2, mu2, 4, **kw2)
                    torch.cuda.synchronize(device)
                rb = {r["name"]: r for r in eng.profile_rows()}
                eng.profile(False)
                if "dit_attention" not in rb:
                    os.environ["DEX_ATTN_SEPARATE"] = "1"
                    eng.profile(True)
                    with torch.cuda.stream(stream):
                        eng.sample(z2, mask2, mu2, 4, **kw2)
                        torch.cuda.synchronize(device)
                    rb.update({r["name"]: r for r in eng.profile_rows() if r["name"] == "dit_attention"})
                    eng.profile(False)
                    del os.environ["DEX_ATTN_SEPARATE"]
                ent = {"value": round(valid2 / dtb, 1), "unit": "mel-frames/s", "workload": f"gedex_lj B={B32} T={T32} n_timesteps={n32}"}
                for kname in ("conv3x3", "dit_attention", "dit_block", "pos_conv"):
                    if kname in rb:
                        ent[kname] = roof(rb[kname], key)
                scale[prec] = ent
            eng.set_precision(args.precision)
            res["roofline_batch32"] = scale
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, weights, B, T, n_steps, TrTs)
            res["gpu_over_cpu"] = round(frames_s / res["cpu_baseline"]["value"], 1)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
