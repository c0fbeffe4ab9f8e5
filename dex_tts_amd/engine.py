"""Host-side owner of one libdexamd context: device memory and streams come from PyTorch-ROCm
(plumbing), all arithmetic of the sampler runs in the HIP library."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib
from .config import ScoreNetConfig, param_shapes


def edm_sigmas(n_steps: int) -> torch.Tensor:
    """t_0..t_{N-1}, t_N = 0 with the reference's exact fp32 expression (edm.py:141,157,184-185)."""
    if n_steps < 2:
        raise ValueError("n_timesteps must be >= 2 (the reference divides by num_steps - 1, edm.py:157)")
    sigma_min, sigma_max, rho = 0.002, 80, 7
    idx = torch.arange(n_steps)
    s = (sigma_max ** (1 / rho) + idx / (n_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([s.to(torch.float32), torch.zeros(1)])


def heun_eval_sigmas(n_steps: int) -> torch.Tensor:
    """Sigma of each of the 2n-1 network evaluations of ablation_sampler(solver='heun', alpha=1), plus a trailing
    0: t_i for step i's predictor and fl(t_i + fl(t_{i+1} - t_i)) for its corrector (edm.py:199-207).  The library
    builds the same table on the device; this host copy exists for tests and tooling."""
    ts = edm_sigmas(n_steps)
    out = []
    for i in range(n_steps):
        out.append(ts[i])
        if i < n_steps - 1:
            out.append(ts[i] + (ts[i + 1] - ts[i]))
    return torch.stack(out + [torch.zeros(())]).to(torch.float32)


class ScoreNetEngine:
    def __init__(self, cfg: ScoreNetConfig, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("dex_tts_amd runs on an AMD GPU (torch device 'cuda' on ROCm); no CPU path exists")
        self.lib = _lib.load()
        self.cfg, self.device = cfg, device
        self.shapes = param_shapes(cfg)
        ccfg = _lib.make_config(cfg)
        h = C.c_void_p()
        rc = self.lib.dex_ctx_create(C.byref(ccfg), C.byref(h))
        self.h = h
        self._check(rc)
        self._ws: Optional[torch.Tensor] = None
        self._keep: list = []
        self._side = None
        self._stage: dict = {}
        self._stage_out: dict = {}
        self.loaded_version = None

    # ---------------------------------------------------------------------------------------
    # sample(): how the in-launch hand-offs of a call (small grids: the cluster form of the DiT block) are verified.
    #   "deferred" (default): asynchronous - a copy of the call's hand-off word into pinned host memory + an event are enqueued behind the
    #       call (dex_call_status_begin); the verdict is read at the NEXT call into this engine or at .status() and raises there.  sample()
    #       returns at once, so a caller can put the vocoder of this utterance behind it and go on to the next one.  A lost
    #       hand-off also poisons every output of its call with NaN, so nothing wrong can look right in the meantime.
    #   True / "sync": the host waits for the call and raises before the mel is handed out (rounds 3-5; repeats the call by itself when
    #       the XCD-local form had to be switched off).   False: no check.
    check_handoffs = "deferred"

    def status(self, wait: bool = True) -> bool:
        """Verdict of the deferred hand-off check of the last call (check_handoffs = "deferred"): raises RuntimeError if a hand-off was
        lost; returns False while the stream has not reached the check yet (wait=False), True once it is known to be clean."""
        with torch.cuda.device(self.device):
            rc = self.lib.dex_call_status_poll(self.h, 1 if wait else 0)
        if rc == _lib.DEX_PENDING:
            return False
        self._check(rc)
        return True

    def _check(self, rc: int):
        if rc != 0:
            msg = self.lib.dex_last_error(self.h)
            raise RuntimeError(f"libdexamd error {rc}: {msg.decode() if msg else '?'}")

    def __del__(self):
        try:
            if getattr(self, "h", None) and self.h.value:
                self.lib.dex_ctx_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_precision(self, name: str):
        self._check(self.lib.dex_ctx_set_precision(self.h, _lib.PRECISION[name]))

    def load_weights(self, weights: Dict[str, torch.Tensor]):
        """weights: state-dict entries relative to ``denoise_fn.`` in the reference layout."""
        with torch.cuda.device(self.device):
            # every copy is enqueued on torch's CURRENT stream, i.e. behind the .to()/.contiguous() kernels that may have
            # produced the staging tensor on it; dex_ctx_finalize packs on the same stream and waits for it on the host,
            # so the packed weights are complete for sampler calls on any stream afterwards
            st = self._stream()
            staged = []
            for key, shape in self.shapes.items():
                if key not in weights:
                    raise KeyError(f"missing weight {key}")
                w = weights[key].detach().to(device=self.device, dtype=torch.float32).contiguous()
                if tuple(w.shape) != tuple(shape):
                    raise ValueError(f"{key}: shape {tuple(w.shape)} != {tuple(shape)}")
                shp = (C.c_int64 * 4)(*([int(s) for s in shape] + [0] * (4 - len(shape))))
                self._check(self.lib.dex_ctx_load_weight_async(self.h, key.encode(), C.c_void_p(w.data_ptr()), shp, len(shape), st))
                staged.append(w)                  # alive until the stream has consumed them
            self._check(self.lib.dex_ctx_finalize(self.h, st))
            del staged

    def workspace(self, B, T, Tr, Ts, n_steps) -> torch.Tensor:
        need = int(self.lib.dex_workspace_bytes(self.h, B, T, Tr, Ts, n_steps))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---------------------------------------------------------------------------------------
    def _fill_args(self, a: _lib.DexSampleArgs, mu, mask, sigmas, out, n_steps, spk, ref, sty, sty_lengths, use_graph,
                   solver="euler", noise=None, churn=None):
        if solver not in _lib.SOLVER:
            raise ValueError(f"solver must be 'euler' or 'heun' (edm.py:107), got {solver!r}")
        B, F, T = mu.shape
        if F != 80:
            raise ValueError("mel dimension must be 80")
        if T % 4 != 0:
            raise ValueError(f"T={T} must be a multiple of 4 (fix_len_compatibility)")
        Tr = Ts = 0
        keep = [mu, mask, sigmas, out]
        a.B, a.T, a.n_steps = B, T, n_steps
        a.mu_dev, a.mask_dev, a.sigmas_dev, a.out_dev = mu.data_ptr(), mask.data_ptr(), sigmas.data_ptr(), out.data_ptr()
        a.spk_dev = None
        if spk is not None and self.cfg.n_spks > 1:
            spk = spk.to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(spk); a.spk_dev = spk.data_ptr()
        a.ref_skips_dev, a.n_ref, a.Tr, a.sty_dev, a.sty_lengths_dev, a.Ts = None, 0, 0, None, None, 0
        if self.cfg.variant == "dex":
            ref = [r.to(device=self.device, dtype=torch.float32).contiguous() for r in ref]
            sty = sty.to(device=self.device, dtype=torch.float32).contiguous()
            sl = sty_lengths.to(device=self.device, dtype=torch.int32).contiguous()
            if not 1 <= len(ref) <= 7:
                raise ValueError(f"DEX needs 1..7 reference skips, got {len(ref)}")
            mid = self.cfg.mid_dim
            Tr, Ts = ref[0].shape[-1], sty.shape[-1]
            for j, r in enumerate(ref):           # the library reads B*mid*Tr floats per skip: shapes are checked here
                if tuple(r.shape) != (B, mid, Tr):
                    raise ValueError(f"ref[{j}] has shape {tuple(r.shape)}, expected {(B, mid, Tr)}")
            if sty.dim() != 3 or sty.shape[0] != B or sty.shape[1] != mid:
                raise ValueError(f"sty has shape {tuple(sty.shape)}, expected ({B}, {mid}, Ts)")
            if tuple(sl.shape) != (B,):
                raise ValueError(f"sty_lengths has shape {tuple(sl.shape)}, expected ({B},)")
            if Tr < 2 or Ts < 1:
                raise ValueError("DEX needs Tr >= 2 reference frames and Ts >= 1 style tokens")
            arr = (C.c_void_p * len(ref))(*[r.data_ptr() for r in ref])
            keep += ref + [sty, sl, arr]
            a.ref_skips_dev, a.n_ref, a.Tr = C.cast(arr, C.POINTER(C.c_void_p)), len(ref), Tr
            a.sty_dev, a.sty_lengths_dev, a.Ts = sty.data_ptr(), sl.data_ptr(), Ts
        a.solver = _lib.SOLVER[solver]
        ws = self.workspace(B, T, Tr, Ts, max(int(self.lib.dex_num_evals(n_steps, a.solver)), 1))
        base = (ws.data_ptr() + 255) // 256 * 256
        a.workspace_dev, a.workspace_bytes = base, ws.numel() - (base - ws.data_ptr())
        a.use_graph = 1 if use_graph else 0
        a.noise_dev, a.S_churn, a.S_min, a.S_max, a.S_noise = None, 0.0, 0.0, 0.0, 1.0
        if churn is not None and churn[0] > 0:
            S_churn, S_min, S_max, S_noise = churn
            if noise is None or tuple(noise.shape) != (n_steps, B, F, T):
                raise ValueError(f"S_churn > 0 needs noise of shape {(n_steps, B, F, T)} (one randn_like draw per step, edm.py:196)")
            keep.append(noise)
            a.noise_dev = noise.data_ptr()
            a.S_churn, a.S_min, a.S_noise = float(S_churn), float(S_min), float(S_noise)
            a.S_max = 0.0 if S_max == float("inf") else float(S_max)          # <= 0 is the library's +inf
        return keep

    @staticmethod
    def _prep_mask(mask, B, T, device):
        m = mask.to(device=device, dtype=torch.float32).reshape(B, T).contiguous()
        return m

    def _staged(self, key, tensors):
        """Persistent device buffers for graph replays: a captured graph dereferences fixed addresses, so the inputs of a
        call are copied into buffers that live as long as the engine (one set per shape), which makes every call of a
        shape a hit in the library's graph cache."""
        bufs = self._stage.get(key)
        if bufs is None:
            if len(self._stage) >= 8:                       # least recently used shape goes, together with its output buffer
                old = next(iter(self._stage))
                self._stage.pop(old)
                self._stage_out.pop(old, None)
            bufs = [torch.empty_like(t) for t in tensors]
            self._stage[key] = bufs
        else:
            self._stage[key] = self._stage.pop(key)          # dicts keep insertion order: re-insert = most recently used
        cur = torch.cuda.current_stream(self.device)
        for b, t in zip(bufs, tensors):
            b.copy_(t, non_blocking=True)
            if t.is_cuda:
                t.record_stream(cur)       # the caller's tensor may be freed (and its block reused on ITS stream) before this copy has run
        return bufs

    def sample(self, z, mask, mu, n_steps, spk=None, ref=None, sty=None, sty_lengths=None, use_graph=False,
               solver="euler", noise=None, S_churn=0.0, S_min=0.0, S_max=float("inf"), S_noise=1.0):
        """ablation_sampler(solver, edm, linear, none) for latent z — edm.py:109-216.  ``solver`` is 'euler' (what
        Diffusion wires, diffusion.py:216) or 'heun' (edm.py:207-214; 2n-1 network evaluations).  Asynchronous.
        ``use_graph``: the whole call (conditioning tables + every network evaluation) is one cached hipGraph.
        Small grids (B x row tiles <= 64: the cluster form of the DiT block) have their in-launch hand-offs checked: asynchronously by
        default (``check_handoffs = "deferred"``: the verdict surfaces at the next call or at ``status()``), or before the mel is handed out
        (``check_handoffs = True``: blocks the host until the call has finished).
        ``S_churn > 0`` turns on the stochastic sampler (edm.py:194-196); ``noise`` [n_steps,B,80,T] then holds step i's
        ``randn_like(x_cur)`` draw (the caller owns the RNG, as with ablation_sampler's ``randn_like`` argument)."""
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            run_on = cur
            if use_graph and cur.cuda_stream == 0:
                # the legacy default stream cannot be captured: replay on a private stream, ordered
                # after / before the caller's stream
                if self._side is None:
                    self._side = torch.cuda.Stream(self.device)
                self._side.wait_stream(cur)
                run_on = self._side
            with torch.cuda.stream(run_on):
                mu = mu.to(device=self.device, dtype=torch.float32).contiguous()
                z = z.to(device=self.device, dtype=torch.float32).contiguous()
                B, _, T = mu.shape
                mask = self._prep_mask(mask, B, T, self.device)
                sig = edm_sigmas(n_steps).to(self.device)
                if self.cfg.variant == "dex" and ref is not None and sty is not None and sty_lengths is not None:
                    ref = [r.to(device=self.device, dtype=torch.float32).contiguous() for r in ref]
                    sty = sty.to(device=self.device, dtype=torch.float32).contiguous()
                    sty_lengths = sty_lengths.to(device=self.device, dtype=torch.int32).contiguous()
                if spk is not None and self.cfg.n_spks > 1:
                    spk = spk.to(device=self.device, dtype=torch.float32).contiguous()
                else:
                    spk = None
                churn = (S_churn, S_min, S_max, S_noise) if S_churn and S_churn > 0 else None
                if churn is not None:
                    if noise is None:
                        raise ValueError("S_churn > 0 needs the per-step noise draws (noise=[n_steps,B,80,T])")
                    noise = noise.to(device=self.device, dtype=torch.float32).contiguous()
                else:
                    noise = None
                if use_graph:
                    dexin = ([sty, sty_lengths] + list(ref)) if (self.cfg.variant == "dex" and ref is not None) else []
                    flat = [z, mu, mask, sig] + ([spk] if spk is not None else []) + ([noise] if noise is not None else []) + dexin
                    key = (n_steps, solver, spk is not None, noise is not None) + tuple((tuple(t.shape), t.dtype) for t in flat)
                    st = self._staged(key, flat)
                    z, mu, mask, sig = st[:4]
                    k = 4
                    if spk is not None:
                        spk = st[k]; k += 1
                    if noise is not None:
                        noise = st[k]; k += 1
                    if dexin:
                        sty, sty_lengths, ref = st[k], st[k + 1], st[k + 2:]
                    out = self._stage_out.setdefault(key, torch.empty_like(mu))
                else:
                    out = torch.empty_like(mu)
                a = _lib.DexSampleArgs()
                keep = self._fill_args(a, mu, mask, sig, out, n_steps, spk, ref, sty, sty_lengths, use_graph, solver, noise, churn)
                a.z_dev = z.data_ptr()
                deferred = self.check_handoffs == "deferred"
                if deferred:
                    self.status(wait=True)         # the previous call's verdict (its event has long passed; raises if that call lost a hand-off)
                for attempt in (0, 1):
                    self._check(self.lib.dex_sample(self.h, C.byref(a), self._stream()))
                    # calls that used in-launch hand-offs (small grids: the cluster form of the DiT block) are checked: a lost hand-off
                    # poisons the outputs with NaN, and a NaN mel must not leave silently.  Deferred: the check rides the
                    # stream behind the call; sync: dex_call_status waits (it returns at once for calls that used no hand-offs).
                    if deferred:
                        self._check(self.lib.dex_call_status_begin(self.h, self._stream()))
                        break
                    rc = self.lib.dex_call_status(self.h, self._stream()) if self.check_handoffs else 0
                    if rc == 0:
                        break
                    if attempt == 0 and rc == _lib.DEX_ERR_HANDOFF_XCD:
                        continue                   # the XCD-local form was just switched off for this device: the repeat is placement-independent
                    msg = self.lib.dex_last_error(self.h) or b""
                    raise RuntimeError(f"libdexamd error {rc}: {msg.decode()}")
                self._keep = keep + [z]           # keep inputs alive until the stream work is enqueued & consumed
                if use_graph:
                    out = out.clone()             # the staging output is overwritten by the next replay
            if run_on is not cur:
                cur.wait_stream(run_on)
            return out

    def denoise_once(self, x, sigma: float, mask, mu, spk=None, ref=None, sty=None, sty_lengths=None):
        """One EDMPrecond.forward (edm.py:88-98); also records debug taps."""
        with torch.cuda.device(self.device):
            mu = mu.to(device=self.device, dtype=torch.float32).contiguous()
            x = x.to(device=self.device, dtype=torch.float32).contiguous()
            B, _, T = mu.shape
            mask = self._prep_mask(mask, B, T, self.device)
            sig = torch.tensor([float(sigma), 0.0], dtype=torch.float32).to(self.device)
            out = torch.empty_like(mu)
            d = _lib.DexDenoiseArgs()
            keep = self._fill_args(d.s, mu, mask, sig, out, 1, spk, ref, sty, sty_lengths, False)
            d.x_dev = x.data_ptr()
            self._check(self.lib.dex_denoise_once(self.h, C.byref(d), self._stream()))
            self._keep = keep + [x]
            return out

    def taps(self) -> Dict[str, torch.Tensor]:
        """Named intermediates of the last denoise_once call, each [rows, C] (channels-last)."""
        out = {}
        with torch.cuda.device(self.device):
            for i in range(self.lib.dex_num_taps(self.h)):
                name = self.lib.dex_tap_name(self.h, i)
                shp = (C.c_int64 * 4)()
                nd = C.c_int()
                self._check(self.lib.dex_tap_info(self.h, name, shp, C.byref(nd)))
                t = torch.empty(int(shp[0]), int(shp[1]), dtype=torch.float32, device=self.device)
                self._check(self.lib.dex_tap_copy(self.h, name, C.c_void_p(t.data_ptr()), t.numel() * 4, self._stream()))
                out[name.decode()] = t
            torch.cuda.synchronize(self.device)
        return out

    # profiling --------------------------------------------------------------------------------
    def profile(self, on: bool):
        self._check(self.lib.dex_profile_enable(self.h, 1 if on else 0))

    def profile_rows(self) -> List[dict]:
        rows = []
        for i in range(self.lib.dex_profile_num(self.h)):
            name = C.c_char_p(); calls = C.c_int(); ms = C.c_double(); fl = C.c_double(); by = C.c_double()
            self._check(self.lib.dex_profile_get(self.h, i, C.byref(name), C.byref(calls), C.byref(ms), C.byref(fl), C.byref(by)))
            rows.append({"name": name.value.decode(), "calls": calls.value, "ms": ms.value, "flops": fl.value, "bytes": by.value})
        return rows

    # mel front-end ----------------------------------------------------------------------------
    def handoff_timeouts(self) -> int:
        """Debug: did a workgroup hand-off of the last sample / denoise_once call time out (cluster form of the DiT row chain)?"""
        with torch.cuda.device(self.device):
            return int(self.lib.dex_debug_handoff_timeouts(self.h, self._stream()))

    def xcd_local(self) -> int:
        """Debug: 1 if this device deals workgroup b to XCD b % 8 (the cluster row chain then keeps its hand-offs inside one L2)."""
        with torch.cuda.device(self.device):
            return int(self.lib.dex_debug_xcd_local())

    def mel_from_wav(self, wav: torch.Tensor):
        with torch.cuda.device(self.device):
            wav = wav.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1)
            n = wav.numel()
            frames = self.lib.dex_mel_frames(n)
            mel = torch.empty(80, frames, dtype=torch.float32, device=self.device)
            energy = torch.empty(frames, dtype=torch.float32, device=self.device)
            self._check(self.lib.dex_mel_from_wav(self.h, wav.data_ptr(), n, mel.data_ptr(), energy.data_ptr(), self._stream()))
            self._keep = [wav]
            return mel, energy
