"""ctypes binding of libmpv_hip.so -- the C ABI declared in include/mpv_hip.h.

There is deliberately no fallback: if the library is missing, cannot be loaded, or no
gfx950 device is visible, every compute entry point raises.  PyTorch is used only for
device memory (``data_ptr()``) and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional, Sequence

import torch

PKG = Path(__file__).resolve().parent
import os
LIB_PATH = Path(os.environ.get("MPV_LIB", PKG / "libmpv_hip.so"))   # MPV_LIB: kernel-variant experiments only
ABI_VERSION = 4
HIDDEN = 128
MAX_NF = 4
COEF_COLS = 20
GEMM_MODES = {"f16x3": 0, "f32": 1, "f16x2": 2, "f16": 3}

EXPORTS = (
    "mpv_abi_version", "mpv_ctx_create", "mpv_ctx_destroy", "mpv_ctx_set_stream", "mpv_last_error",
    "mpv_nsdiff_load", "mpv_nsdiff_sample", "mpv_chain_stats", "mpv_nsdiff_philox_normals", "mpv_sigma_load", "mpv_sigma_head",
    "mpv_reduce", "mpv_chain_timing", "mpv_chain_timing_enable", "mpv_tmdm_load", "mpv_tmdm_sample",
    "mpv_stg_load", "mpv_stg_set_graph", "mpv_stg_forward", "mpv_stg_sample",
    "mpv_nss_load", "mpv_nss_set_graph", "mpv_nss_forward", "mpv_nss_sample",
    "mpv_biomass_simulate", "mpv_network_sde_simulate", "mpv_dts_load", "mpv_dts_forward", "mpv_dts_vjp", "mpv_dts_debug_tensor", "mpv_dts_sample",
    "mpv_nst_load", "mpv_nst_set_graph", "mpv_nst_forward", "mpv_nst_debug_tensor", "mpv_slbp_simulate",
)


class MpvError(RuntimeError):
    pass


_F = C.POINTER(C.c_float)


class NsdiffDesc(C.Structure):
    _fields_ = [("steps", C.c_int32), ("nf", C.c_int32)]


class NsdiffWeights(C.Structure):
    _fields_ = [("lin_w", _F * 3), ("lin_b", _F * 3), ("embed", _F * 3),
                ("eps_w", _F), ("eps_b", _F), ("sig_w", _F), ("sig_b", _F)]


class SampleShape(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("nodes", C.c_int32), ("pred_len", C.c_int32), ("n_samples", C.c_int32),
                ("chunk", C.c_int32), ("gemm", C.c_int32), ("seed", C.c_uint64), ("offset", C.c_uint64)]


class SampleOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("samples", "mpv", "pred_mean", "mpv_feat", "std", "mean")]


class DtsDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("seq_len", "nf", "d_model", "n_heads", "n_enc", "n_dec", "mlp_times", "timesteps", "combine_taps", "gemm")]


class DtsSampleShape(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("nodes", C.c_int32), ("windows", C.c_int32), ("n_samples", C.c_int32), ("chunk", C.c_int32),
                ("n_pairs", C.c_int32), ("infill_coef", C.c_float), ("rows_per_pass", C.c_int32), ("seed", C.c_uint64)]


DTS_PLAN_COLS = 11


class StgDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("T_h", "T_p", "nf", "d_h", "Td_h", "timesteps", "gemm")]


class StgStep(C.Structure):
    _fields_ = [("kind", C.c_int32), ("c_in", C.c_int32), ("c_out", C.c_int32), ("T", C.c_int32), ("pop", C.c_int32), ("push", C.c_int32),
                ("w", _F * 24)]


class StgSampleShape(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("parallel", C.c_int32), ("sequential", C.c_int32), ("n_steps", C.c_int32), ("seed", C.c_uint64)]


STG_PLAN_COLS = 6


class NssDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("pred_len", "nf", "d_h", "Td_h", "steps", "gemm")]


class NssSampleShape(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("nodes", C.c_int32), ("n_samples", C.c_int32), ("chunk", C.c_int32), ("seed", C.c_uint64),
                ("rows_per_pass", C.c_int64)]


class NstDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("enc_in", "c_out", "seq_len", "label_len", "pred_len", "d_model", "n_heads", "d_ff", "e_layers", "d_layers",
                                          "activation", "p_layers")] + [("p_hidden", C.c_int32 * 4), ("latent", C.c_int32), ("pe_rows", C.c_int32),
                                                                   ("fT_h", C.c_int32), ("spatial_layers", C.c_int32),
                                                                   ("spatial_gnn", C.c_int32), ("spatial_heads", C.c_int32), ("spatial_slope", C.c_float)]


class SigmaDesc(C.Structure):
    _fields_ = [("seq_len", C.c_int32), ("rolling", C.c_int32), ("pred_len", C.c_int32), ("nf", C.c_int32),
                ("hidden", C.c_int32)]


class SigmaWeights(C.Structure):
    _fields_ = [(n, _F) for n in ("w0", "b0", "ln0_w", "ln0_b", "w1", "b1", "ln1_w", "ln1_b", "w2", "b2")]


class BiomassDesc(C.Structure):
    _fields_ = [("n_trajectories", C.c_int32), ("nodes", C.c_int32), ("ticks", C.c_int32), ("burn_ticks", C.c_int32),
                ("stride", C.c_int32), ("r", C.c_float), ("k", C.c_float), ("d", C.c_float), ("eta", C.c_float),
                ("dt", C.c_float), ("eps_start", C.c_float), ("eps_end", C.c_float), ("seed", C.c_uint64)]


class SdeDesc(C.Structure):
    _fields_ = [("model", C.c_int32), ("n_trajectories", C.c_int32), ("nodes", C.c_int32), ("ticks", C.c_int32), ("burn_ticks", C.c_int32),
                ("stride", C.c_int32), ("p0", C.c_float), ("p1", C.c_float), ("p2", C.c_float), ("eta", C.c_float), ("dt", C.c_float),
                ("eps_start", C.c_float), ("eps_end", C.c_float), ("seed", C.c_uint64)]


class SlbpDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_trajectories", "ticks", "burn_ticks", "stride")] + \
               [(n, C.c_float) for n in ("i_b", "i_p", "r", "H1", "H2", "H3", "H4", "c_b", "c_p", "p_r", "c_e", "m_p", "K", "D", "dt", "N_start", "N_end")] + \
               [("seed", C.c_uint64)]


# parameter table of the reference generator (dataset/SLBP_dynamic_gene.py:10-24), D = its first value
SLBP_PARAMS = dict(i_b=3e-4, i_p=3e-4, r=7.5e-3, H1=0.5, H2=0.1, H3=20.0, H4=15.0, c_b=7.5e-5, c_p=2.75e-4, p_r=5e-2, c_e=0.1, m_p=2.25e-3, K=1.0, D=1e-5)
SDE_MODELS = {"biomass": 0, "SIS": 1, "neuronal": 2}
_lib = None


def load_library():
    """dlopen the in-tree library and declare the prototypes.  Raises MpvError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise MpvError("{} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950); "
                       "there is no CPU fallback for this path".format(LIB_PATH))
    try:
        lib = C.CDLL(str(LIB_PATH))
    except OSError as exc:
        raise MpvError("cannot load {}: {}".format(LIB_PATH, exc)) from exc
    vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
    protos = {
        "mpv_abi_version": (C.c_int, []),
        "mpv_ctx_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
        "mpv_ctx_destroy": (None, [vp]),
        "mpv_ctx_set_stream": (C.c_int, [vp, vp]),
        "mpv_last_error": (C.c_char_p, [vp]),
        "mpv_nsdiff_load": (C.c_int, [vp, C.POINTER(NsdiffDesc), C.POINTER(NsdiffWeights), _F]),
        "mpv_nsdiff_sample": (C.c_int, [vp, C.POINTER(SampleShape), vp, vp, vp, C.POINTER(SampleOut)]),
        "mpv_tmdm_load": (C.c_int, [vp, C.POINTER(NsdiffDesc), C.POINTER(NsdiffWeights), _F]),
        "mpv_tmdm_sample": (C.c_int, [vp, C.POINTER(SampleShape), vp, vp, C.POINTER(SampleOut)]),
        "mpv_biomass_simulate": (C.c_int, [vp, C.POINTER(BiomassDesc), vp, vp, vp, vp]),
        "mpv_network_sde_simulate": (C.c_int, [vp, C.POINTER(SdeDesc), vp, vp, vp, vp]),
        "mpv_slbp_simulate": (C.c_int, [vp, C.POINTER(SlbpDesc), vp, vp]),
        "mpv_nsdiff_philox_normals": (C.c_int, [vp, u64, u64, i64, vp]),
        "mpv_chain_stats": (C.c_int, [vp, C.POINTER(u64), C.POINTER(u64), C.c_int]),
        "mpv_sigma_load": (C.c_int, [vp, C.POINTER(SigmaDesc), C.POINTER(SigmaWeights)]),
        "mpv_sigma_head": (C.c_int, [vp, vp, i32, i64, i32, i64, i64, vp, vp, f32, vp]),
        "mpv_reduce": (C.c_int, [vp, vp, i32, i64, i32, i32, vp, vp, vp, vp, vp]),
        "mpv_stg_load": (C.c_int, [vp, C.POINTER(StgDesc), C.POINTER(StgStep), i32, C.POINTER(_F), _F]),
        "mpv_stg_set_graph": (C.c_int, [vp, i32, i64, C.POINTER(i64), C.POINTER(i64)]),
        "mpv_stg_forward": (C.c_int, [vp, vp, vp, i64, i32, vp]),
        "mpv_stg_sample": (C.c_int, [vp, C.POINTER(StgSampleShape), vp, C.POINTER(C.c_double), vp, vp]),
        "mpv_nss_load": (C.c_int, [vp, C.POINTER(NssDesc), C.POINTER(StgStep), i32, C.POINTER(_F), _F]),
        "mpv_nss_set_graph": (C.c_int, [vp, i32, i64, C.POINTER(i64), C.POINTER(i64)]),
        "mpv_nss_forward": (C.c_int, [vp, vp, vp, vp, i64, i32, vp, vp]),
        "mpv_nss_sample": (C.c_int, [vp, C.POINTER(NssSampleShape), vp, vp, vp, vp]),
        "mpv_dts_load": (C.c_int, [vp, C.POINTER(DtsDesc), C.POINTER(_F), i32, _F, _F, _F]),
        "mpv_dts_forward": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "mpv_dts_vjp": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "mpv_dts_debug_tensor": (C.c_int, [vp, C.c_char_p, i32, vp, i64]),
        "mpv_dts_sample": (C.c_int, [vp, C.POINTER(DtsSampleShape), vp, _F, vp, vp]),
        "mpv_nst_load": (C.c_int, [vp, C.POINTER(NstDesc), C.POINTER(_F), i32]),
        "mpv_nst_set_graph": (C.c_int, [vp, i32, i64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "mpv_nst_forward": (C.c_int, [vp, vp, i32, i64, i32, i64, i64, vp, vp, i32, i32, i64, vp]),
        "mpv_nst_debug_tensor": (C.c_int, [vp, i32, vp, i64]),
        "mpv_chain_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(i64)]),
        "mpv_chain_timing_enable": (C.c_int, [vp, C.c_int]),
    }
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)          # AttributeError here = symbol missing from the .so
        fn.restype, fn.argtypes = res, args
    if lib.mpv_abi_version() != ABI_VERSION:
        raise MpvError("libmpv_hip.so ABI {} != binding ABI {}".format(lib.mpv_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _host_f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(device="cpu", dtype=torch.float32).contiguous()


def _fptr(t: torch.Tensor):
    return C.cast(t.data_ptr(), _F)


def _dev_ptr(t: Optional[torch.Tensor], device: torch.device, name: str):
    if t is None:
        return None
    if t.device != device:
        raise MpvError("{} is on {}, the context is on {}".format(name, t.device, device))
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise MpvError("{} must be a contiguous float32 tensor".format(name))
    return C.c_void_p(t.data_ptr())


class Context:
    """One mpv_ctx: (process, device, stream).  Owns the device copies of the weights."""

    def __init__(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise MpvError("the MPV sampling path runs only on a ROCm device (got {!r}); no CPU fallback".format(str(device)))
        if not torch.cuda.is_available():
            raise MpvError("no ROCm device is visible to PyTorch")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.lib = load_library()
        self.device = device
        self._h = C.c_void_p()
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = self.lib.mpv_ctx_create(device.index, C.c_void_p(stream), C.byref(self._h))
        if rc != 0:
            raise MpvError("mpv_ctx_create: {} ({})".format(rc, (self.lib.mpv_last_error(None) or b"").decode()))
        self.nsdiff = None     # (steps, nf)
        self.tmdm = None       # (steps, nf)
        self.sigma = None      # SigmaDesc fields
        self.nst = None        # NstDesc of the loaded f(x) backbone
        self.dts = None        # DtsDesc fields
        self.stg = None        # (T_h, T_p, F, graph nodes | None)
        self._stg_graph = None
        self.nss = None        # (pred_len, F, steps) of the loaded NsDiff_spatial network
        self._nss_graph = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.mpv_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise MpvError("{} failed ({}): {}".format(what, rc, (self.lib.mpv_last_error(self._h) or b"").decode()))

    def _bind_stream(self):
        self._check(self.lib.mpv_ctx_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                    "mpv_ctx_set_stream")

    # ---- weights -------------------------------------------------------------------
    def load_nsdiff(self, steps: int, nf: int, lin_w: Sequence[torch.Tensor], lin_b: Sequence[torch.Tensor],
                    embed: Sequence[torch.Tensor], eps_w, eps_b, sig_w, sig_b, coef: torch.Tensor):
        keep = [_host_f32(t) for t in (*lin_w, *lin_b, *embed, eps_w, eps_b, sig_w, sig_b, coef)]
        want = [(HIDDEN, 3 * nf), (HIDDEN, HIDDEN), (HIDDEN, HIDDEN), (HIDDEN,), (HIDDEN,), (HIDDEN,),
                (steps, HIDDEN), (steps, HIDDEN), (steps, HIDDEN), (nf, HIDDEN), (nf,), (nf, HIDDEN), (nf,),
                (steps, COEF_COLS)]
        for t, shp in zip(keep, want):
            if tuple(t.shape) != shp:
                raise MpvError("NsDiff weight has shape {}, expected {}".format(tuple(t.shape), shp))
        w = NsdiffWeights()
        for i in range(3):
            w.lin_w[i], w.lin_b[i], w.embed[i] = _fptr(keep[i]), _fptr(keep[3 + i]), _fptr(keep[6 + i])
        w.eps_w, w.eps_b, w.sig_w, w.sig_b = (_fptr(t) for t in keep[9:13])
        self._bind_stream()
        self._check(self.lib.mpv_nsdiff_load(self._h, C.byref(NsdiffDesc(steps, nf)), C.byref(w), _fptr(keep[13])),
                    "mpv_nsdiff_load")
        self.nsdiff = (steps, nf)

    def load_sigma(self, seq_len: int, rolling: int, pred_len: int, nf: int, hidden: int, tensors: Sequence[torch.Tensor]):
        keep = [_host_f32(t) for t in tensors]
        k0 = seq_len - rolling
        want = [(hidden, k0), (hidden,), (nf, hidden), (nf, hidden), (hidden, hidden), (hidden,), (nf, hidden),
                (nf, hidden), (pred_len, hidden), (pred_len,)]
        for t, shp in zip(keep, want):
            if tuple(t.shape) != shp:
                raise MpvError("sigma-head weight has shape {}, expected {}".format(tuple(t.shape), shp))
        w = SigmaWeights(*[_fptr(t) for t in keep])
        self._bind_stream()
        self._check(self.lib.mpv_sigma_load(self._h, C.byref(SigmaDesc(seq_len, rolling, pred_len, nf, hidden)), C.byref(w)),
                    "mpv_sigma_load")
        self.sigma = (seq_len, rolling, pred_len, nf, hidden)

    # ---- compute (all stream-ordered, asynchronous) -------------------------------------
    def sigma_head(self, series: torch.Tensor, n_windows: int, first: int, step: int, mean=None, std=None,
                   add_eps: float = 0.0) -> torch.Tensor:
        """series [V, L, F] on device -> gx [n_windows, V, O, F]."""
        if self.sigma is None:
            raise MpvError("sigma head weights not loaded")
        V, L, F = series.shape
        if F != self.sigma[3]:
            raise MpvError("series has F={}, sigma head was loaded with F={}".format(F, self.sigma[3]))
        gx = torch.empty((n_windows, V, self.sigma[2], F), device=self.device, dtype=torch.float32)
        self._bind_stream()
        self._check(self.lib.mpv_sigma_head(self._h, _dev_ptr(series, self.device, "series"), V, L, n_windows, first, step,
                                            _dev_ptr(mean, self.device, "mean"), _dev_ptr(std, self.device, "std"),
                                            float(add_eps), _dev_ptr(gx, self.device, "gx")), "mpv_sigma_head")
        return gx

    def _sample_out(self, shape5, want_samples: bool, summary: bool, per_feature: bool, std, mean):
        """Allocate the outputs of a sampling call and fill the mpv_sample_out struct (tensors kept alive by the caller)."""
        W, F = shape5[0], shape5[3]
        samples = torch.empty(shape5, device=self.device, dtype=torch.float32) if want_samples else None
        mpv = torch.empty(W, device=self.device, dtype=torch.float32) if summary else None
        pm = torch.empty(W, device=self.device, dtype=torch.float32) if summary else None
        feat = torch.empty((W, F), device=self.device, dtype=torch.float32) if (summary and per_feature) else None
        out = SampleOut(_dev_ptr(samples, self.device, "samples"), _dev_ptr(mpv, self.device, "mpv"),
                        _dev_ptr(pm, self.device, "pred_mean"), _dev_ptr(feat, self.device, "mpv_feat"),
                        _dev_ptr(std, self.device, "std"), _dev_ptr(mean, self.device, "mean"))
        return out, samples, mpv, pm, feat

    def sample(self, gx: torch.Tensor, y0hat: Optional[torch.Tensor], n_samples: int, chunk: int,
               noise: Optional[torch.Tensor] = None, seed: int = 0, offset: int = 0, gemm: str = "f16x3",
               want_samples: bool = True, summary: bool = False, per_feature: bool = False, std=None, mean=None):
        """gx / y0hat [W, B, O, F] on device -> samples [W, B, O, F, S]  (summary=False, the default), or
        (samples | None, mpv [W], pred_mean [W], mpv_feat [W, F] | None) with summary=True: the MPV reduction fused into
        the sampler's epilogue; want_samples=False then skips the sample tensor altogether."""
        if self.nsdiff is None:
            raise MpvError("NsDiff weights not loaded")
        W, B, O, F = gx.shape
        if F != self.nsdiff[1]:
            raise MpvError("gx has F={}, the model was loaded with F={}".format(F, self.nsdiff[1]))
        if y0hat is not None and tuple(y0hat.shape) != tuple(gx.shape):
            raise MpvError("y0hat shape {} != gx shape {}".format(tuple(y0hat.shape), tuple(gx.shape)))
        if noise is not None:
            need = W * self.nsdiff[0] * B * n_samples * O * F
            if noise.numel() != need:
                raise MpvError("noise has {} elements, the draw order needs {}".format(noise.numel(), need))
        if gemm not in GEMM_MODES:
            raise MpvError("gemm must be one of {}".format(sorted(GEMM_MODES)))
        if not want_samples and not summary:
            raise MpvError("nothing to compute: want_samples=False and summary=False")
        out, samples, mpv, pm, feat = self._sample_out((W, B, O, F, n_samples), want_samples, summary, per_feature, std, mean)
        shape = SampleShape(W, B, O, n_samples, chunk, GEMM_MODES[gemm], seed & (2 ** 64 - 1), offset & (2 ** 64 - 1))
        self._bind_stream()
        self._check(self.lib.mpv_nsdiff_sample(self._h, C.byref(shape), _dev_ptr(y0hat, self.device, "y0hat"),
                                               _dev_ptr(gx, self.device, "gx"), _dev_ptr(noise, self.device, "noise"),
                                               C.byref(out)), "mpv_nsdiff_sample")
        return (samples, mpv, pm, feat) if summary else samples

    # ---- TMDM sampler ----------------------------------------------------------------
    def load_tmdm(self, steps: int, nf: int, lin_w: Sequence[torch.Tensor], lin_b: Sequence[torch.Tensor],
                  embed: Sequence[torch.Tensor], eps_w, eps_b, coef: torch.Tensor):
        """embed tables may carry the reference's T+1 rows (tmdm_model.py:25); rows 0..T-1 are uploaded."""
        keep = [_host_f32(t) for t in (*lin_w, *lin_b)] + [_host_f32(t[:steps]) for t in embed] + \
               [_host_f32(t) for t in (eps_w, eps_b, coef)]
        want = [(HIDDEN, 2 * nf), (HIDDEN, HIDDEN), (HIDDEN, HIDDEN), (HIDDEN,), (HIDDEN,), (HIDDEN,),
                (steps, HIDDEN), (steps, HIDDEN), (steps, HIDDEN), (nf, HIDDEN), (nf,), (steps, COEF_COLS)]
        for t, shp in zip(keep, want):
            if tuple(t.shape) != shp:
                raise MpvError("TMDM weight has shape {}, expected {}".format(tuple(t.shape), shp))
        w = NsdiffWeights()
        for i in range(3):
            w.lin_w[i], w.lin_b[i], w.embed[i] = _fptr(keep[i]), _fptr(keep[3 + i]), _fptr(keep[6 + i])
        w.eps_w, w.eps_b = _fptr(keep[9]), _fptr(keep[10])
        self._bind_stream()
        self._check(self.lib.mpv_tmdm_load(self._h, C.byref(NsdiffDesc(steps, nf)), C.byref(w), _fptr(keep[11])),
                    "mpv_tmdm_load")
        self.tmdm = (steps, nf)

    def sample_tmdm(self, y0hat: torch.Tensor, n_samples: int, chunk: int, noise: Optional[torch.Tensor] = None,
                    seed: int = 0, offset: int = 0, gemm: str = "f16x3", want_samples: bool = True, summary: bool = False,
                    per_feature: bool = False, std=None, mean=None):
        """y0hat [W, B, L, F] on device -> samples [W, B, L, F, S]; summary / want_samples as in `sample`."""
        if self.tmdm is None:
            raise MpvError("TMDM weights not loaded")
        W, B, L, F = y0hat.shape
        if F != self.tmdm[1]:
            raise MpvError("y0hat has F={}, the model was loaded with F={}".format(F, self.tmdm[1]))
        if noise is not None:
            need = W * self.tmdm[0] * B * n_samples * L * F
            if noise.numel() != need:
                raise MpvError("noise has {} elements, the draw order needs {}".format(noise.numel(), need))
        if gemm not in ("f16x3", "f32"):
            raise MpvError("the TMDM sampler supports gemm 'f16x3' and 'f32'")
        if gemm == "f16x3" and F > 3:
            gemm = "f32"                                       # the split-f16 kernel is built for nf <= 3
        if not want_samples and not summary:
            raise MpvError("nothing to compute: want_samples=False and summary=False")
        out, samples, mpv, pm, feat = self._sample_out((W, B, L, F, n_samples), want_samples, summary, per_feature, std, mean)
        shape = SampleShape(W, B, L, n_samples, chunk, GEMM_MODES[gemm], seed & (2 ** 64 - 1), offset & (2 ** 64 - 1))
        self._bind_stream()
        self._check(self.lib.mpv_tmdm_sample(self._h, C.byref(shape), _dev_ptr(y0hat, self.device, "y0hat"),
                                             _dev_ptr(noise, self.device, "noise"), C.byref(out)), "mpv_tmdm_sample")
        return (samples, mpv, pm, feat) if summary else samples

    # ---- DiffusionTS -------------------------------------------------------------------
    def load_dts(self, seq_len: int, nf: int, d_model: int, n_heads: int, n_enc: int, n_dec: int, mlp_times: int, timesteps: int,
                 combine_taps: int, tensors: Sequence[torch.Tensor], ada_tables: torch.Tensor, rfftfreq: torch.Tensor, poly: torch.Tensor,
                 gemm: str = "f16x3"):
        keep = [_host_f32(t) for t in tensors]
        ada, fr, po = _host_f32(ada_tables), _host_f32(rfftfreq), _host_f32(poly)
        if tuple(ada.shape) != (timesteps, n_enc + 2 * n_dec, 2 * d_model):
            raise MpvError("ada_tables has shape {}, expected {}".format(tuple(ada.shape), (timesteps, n_enc + 2 * n_dec, 2 * d_model)))
        if fr.numel() != seq_len // 2 + 1 or tuple(po.shape) != (3, seq_len):
            raise MpvError("rfftfreq / poly tables do not match seq_len {}".format(seq_len))
        arr = (_F * len(keep))(*[_fptr(t) for t in keep])
        if gemm not in ("f16x3", "f32"):
            raise MpvError("DiffusionTS supports gemm_mode 'f16x3' and 'f32'")
        desc = DtsDesc(seq_len, nf, d_model, n_heads, n_enc, n_dec, mlp_times, timesteps, combine_taps, GEMM_MODES[gemm])
        self._bind_stream()
        self._check(self.lib.mpv_dts_load(self._h, C.byref(desc), arr, len(keep), _fptr(ada), _fptr(fr), _fptr(po)), "mpv_dts_load")
        self.dts = (seq_len, nf, d_model, n_heads, n_enc, n_dec, timesteps)

    def _dts_x(self, x):
        if self.dts is None:
            raise MpvError("DiffusionTS weights not loaded")
        if x.dim() != 3 or x.shape[1] != self.dts[0] or x.shape[2] != self.dts[1]:
            raise MpvError("x must be [B, {}, {}], got {}".format(self.dts[0], self.dts[1], tuple(x.shape)))

    def dts_forward(self, x: torch.Tensor, t: int):
        """(trend, season_error) = Transformer.forward(x [B, L, F] on device, t)."""
        self._dts_x(x)
        trend, season = torch.empty_like(x), torch.empty_like(x)
        self._bind_stream()
        self._check(self.lib.mpv_dts_forward(self._h, _dev_ptr(x, self.device, "x"), x.shape[0], int(t), _dev_ptr(trend, self.device, "trend"),
                                             _dev_ptr(season, self.device, "season")), "mpv_dts_forward")
        return trend, season

    def dts_vjp(self, x: torch.Tensor, t: int, g_out: torch.Tensor) -> torch.Tensor:
        """(d (trend + season_error) / d x)^T g_out, all [B, L, F] on device."""
        self._dts_x(x)
        if tuple(g_out.shape) != tuple(x.shape):
            raise MpvError("g_out shape {} != x shape {}".format(tuple(g_out.shape), tuple(x.shape)))
        gx = torch.empty_like(x)
        self._bind_stream()
        self._check(self.lib.mpv_dts_vjp(self._h, _dev_ptr(x, self.device, "x"), x.shape[0], int(t), _dev_ptr(g_out, self.device, "g_out"),
                                         _dev_ptr(gx, self.device, "g_x")), "mpv_dts_vjp")
        return gx

    def dts_debug(self, name: str, index: int, shape) -> torch.Tensor:
        out = torch.empty(shape, device=self.device, dtype=torch.float32)
        self._bind_stream()
        self._check(self.lib.mpv_dts_debug_tensor(self._h, name.encode(), int(index), _dev_ptr(out, self.device, "out"), out.numel()),
                    "mpv_dts_debug_tensor")
        return out

    def dts_sample(self, history: torch.Tensor, plan: torch.Tensor, n_samples: int, chunk: int, infill_coef: float,
                   noise: Optional[torch.Tensor] = None, seed: int = 0, rows_per_pass: int = 0) -> torch.Tensor:
        """history [W, V, windows, F] on device, plan [n_pairs, 10] (host) -> samples [W, V, pred_len, F, S] on device."""
        if self.dts is None:
            raise MpvError("DiffusionTS weights not loaded")
        W, V, windows, F = history.shape
        L = self.dts[0]
        if F != self.dts[1] or not (0 < windows < L):
            raise MpvError("history must be [W, V, windows < {}, {}], got {}".format(L, self.dts[1], tuple(history.shape)))
        plan = _host_f32(plan)
        if plan.dim() != 2 or plan.shape[1] != DTS_PLAN_COLS:
            raise MpvError("plan must be [n_pairs, {}]".format(DTS_PLAN_COLS))
        if noise is not None:
            draws = 1 + sum(2 + int(r[6]) for r in plan.tolist() if r[1] >= 0)
            need = (n_samples // chunk) * draws * W * chunk * V * L * F
            if noise.numel() != need:
                raise MpvError("noise has {} elements, the draw order needs {}".format(noise.numel(), need))
        out = torch.empty((W, V, L - windows, F, n_samples), device=self.device, dtype=torch.float32)
        shape = DtsSampleShape(W, V, windows, n_samples, chunk, plan.shape[0], float(infill_coef), int(rows_per_pass), seed & (2 ** 64 - 1))
        self._bind_stream()
        self._check(self.lib.mpv_dts_sample(self._h, C.byref(shape), _dev_ptr(history, self.device, "history"), _fptr(plan),
                                            _dev_ptr(noise, self.device, "noise"), _dev_ptr(out, self.device, "samples")), "mpv_dts_sample")
        return out

    # ---- DiffSTG -----------------------------------------------------------------------
    # ---- f(x): ns-Transformer mean backbone ---------------------------------------------------------------
    def load_nst(self, backbone):
        """Upload an ns-Transformer holder (models/Diffusion_model/NsDiff/mu_backbone.py::Model or TMDM's): its
        native_weight_list() gives (NstDesc fields, tensors in the ABI's order)."""
        fields, tensors = backbone.native_weight_list()
        d = NstDesc()
        for k, v in fields.items():
            if k == "p_hidden":
                for i, h in enumerate(v):
                    d.p_hidden[i] = int(h)
            else:
                setattr(d, k, float(v) if k == "spatial_slope" else int(v))
        keep = [None if t is None else _host_f32(t) for t in tensors]       # (None: a slot the graph-convolution kind does not use)
        arr = (_F * len(keep))(*[None if t is None else _fptr(t) for t in keep])
        self._check(self.lib.mpv_nst_load(self._h, C.byref(d), arr, len(keep)), "mpv_nst_load")
        self.nst = d
        self._nst_graph = None

    def nst_set_graph(self, edge_index: torch.Tensor, num_nodes: int):
        """Model_spatial: the graph of the spatial stage (edge_index [2, E]: row 0 = source, row 1 = target); kept until the next load."""
        if self.nst is None:
            raise MpvError("f(x) weights not loaded")
        e = torch.as_tensor(edge_index).detach().to("cpu", torch.int64).reshape(2, -1).contiguous()
        key = (int(num_nodes), e.shape[1], hash(e.numpy().tobytes()))
        if getattr(self, "_nst_graph", None) == key:
            return
        src, dst = e[0].contiguous(), e[1].contiguous()
        P64 = C.POINTER(C.c_int64)
        self._check(self.lib.mpv_nst_set_graph(self._h, int(num_nodes), e.shape[1], C.cast(src.data_ptr(), P64), C.cast(dst.data_ptr(), P64)),
                    "mpv_nst_set_graph")
        self._nst_graph = key

    def nst_forward(self, series: torch.Tensor, n_windows: int, first: int = 0, step: int = 1, mean: Optional[torch.Tensor] = None,
                    std: Optional[torch.Tensor] = None, full: bool = False, rows_per_pass: int = 0, gemm: str = "f16x3") -> torch.Tensor:
        """series [V, L, F] on device -> y_0_hat [n_windows, V, pred_len (or label_len + pred_len when full), F]."""
        if self.nst is None:
            raise MpvError("f(x) weights not loaded")
        if gemm not in ("f16x3", "f32"):
            raise MpvError("f(x) supports gemm 'f16x3' and 'f32'")
        if series.dim() != 3 or series.shape[2] != self.nst.enc_in:
            raise MpvError("series must be [V, L, {}], got {}".format(self.nst.enc_in, tuple(series.shape)))
        V, L, F = series.shape
        keep = self.nst.label_len + self.nst.pred_len if full else self.nst.pred_len
        out = torch.empty((n_windows, V, keep, F), device=self.device, dtype=torch.float32)
        m = None if mean is None else mean.to(self.device, torch.float32).contiguous()
        sd = None if std is None else std.to(self.device, torch.float32).contiguous()
        self._bind_stream()
        self._check(self.lib.mpv_nst_forward(self._h, _dev_ptr(series, self.device, "series"), V, L, int(n_windows), int(first), int(step),
                                             _dev_ptr(m, self.device, "mean"), _dev_ptr(sd, self.device, "std"), int(bool(full)), GEMM_MODES[gemm], int(rows_per_pass),
                                             _dev_ptr(out, self.device, "y0hat")), "mpv_nst_forward")
        return out

    def nst_debug(self, which: str, rows: int) -> torch.Tensor:
        """Intermediate of the last pass (validation): 'x_norm', 'stats', 'tau', 'delta', 'enc_out'; rows = window rows of that pass."""
        d = self.nst
        ids = {"x_norm": (0, (rows, d.seq_len, d.enc_in)), "stats": (1, (rows, 2, d.enc_in)), "tau": (2, (rows,)), "delta": (3, (rows, d.seq_len)),
               "enc_out": (4, (rows, d.seq_len, d.d_model)), "spatial": (5, (rows, max(1, d.fT_h), d.d_model))}
        idx, shape = ids[which]
        out = torch.empty(shape, device=self.device, dtype=torch.float32)
        self._bind_stream()
        self._check(self.lib.mpv_nst_debug_tensor(self._h, idx, _dev_ptr(out, self.device, "out"), out.numel()), "mpv_nst_debug_tensor")
        return out

    def _stg_steps(self, ugnet, T_top: int, n_temb: int, keep: list):
        """Marshal a UGnet holder's layout() into mpv_stg_step records (+ the TimeEmbedding table for t = 0..n_temb-1, reference
        ugnet.py:15-33); `keep` collects the host tensors the pointers refer to."""
        import math

        def h(t):
            if t is None:
                return None
            keep.append(_host_f32(t))
            return _fptr(keep[-1])

        steps = ugnet.layout()
        arr = (StgStep * len(steps))()
        for i, st in enumerate(steps):
            m = st["module"]
            if st["kind"] == "res":
                g = m.spatial.gnn
                tensors = [m.tcn1.conv.weight, m.tcn1.conv.bias,
                           m.tcn1.shortcut.weight if m.tcn1.shortcut is not None else None, m.tcn1.shortcut.bias if m.tcn1.shortcut is not None else None,
                           m.t_conv.weight, m.t_conv.bias, m.tcn2.conv.weight, m.tcn2.conv.bias, m.norm.weight, m.norm.bias,
                           m.downsampling.weight, m.downsampling.bias] + list(g.native_tensors()) + [m.upsampling.weight, m.upsampling.bias]
                has_sc = not isinstance(m.shortcut, torch.nn.Identity)
                tensors += [m.shortcut.weight if has_sc else None, m.shortcut.bias if has_sc else None]
                arr[i].kind, arr[i].c_in, arr[i].c_out, arr[i].T = int(g.native_kind()), m.c_in, m.c_out, m.T      # 0 gated, 10.. the other layers
                arr[i].pop, arr[i].push = int(st["pop"]), int(st["push"])
            else:
                c = m.conv.weight.shape[0]
                tensors = [m.conv.weight, m.conv.bias] + [None] * 22
                arr[i].kind, arr[i].c_in, arr[i].c_out = (1 if st["kind"] == "down" else 2), c, c
                arr[i].pop, arr[i].push = 0, int(st.get("push", False))
            for k, t in enumerate(tensors):
                arr[i].w[k] = h(t)
        # time lengths of the resampling steps follow from the sequence
        T_cur = T_top
        for i, st in enumerate(steps):
            if st["kind"] == "res":
                if arr[i].T != T_cur:
                    raise MpvError("UGnet layout: block {} expects T = {}, the sequence gives {}".format(i, arr[i].T, T_cur))
            else:
                arr[i].T = T_cur
                T_cur = (T_cur - 1) // 2 + 1 if st["kind"] == "down" else 2 * T_cur
        d_h = ugnet.d_h
        half = d_h // 2                                                            # TimeEmbedding, reference ugnet.py:15-33
        freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
        emb = torch.arange(n_temb).float()[:, None] * freq[None, :]
        temb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
        if d_h % 2 == 1:
            temb = torch.nn.functional.pad(temb, (0, 1, 0, 0))
        return arr, len(steps), _host_f32(temb)

    def load_stg(self, ugnet):
        """Upload a UGnet holder (models/Diffusion_model/DiffSTG/ugnet.py): its layout() gives the step sequence."""
        keep = []
        T_steps = int(ugnet.net_param["diffusion_steps"])
        arr, n_steps, temb = self._stg_steps(ugnet, 2 * ugnet.T, T_steps + 1, keep)
        d_h = ugnet.d_h
        head = [_host_f32(t) for t in (ugnet.x_proj.weight, ugnet.x_proj.bias, ugnet.out[0].weight, ugnet.out[0].bias, ugnet.out[1].weight,
                                        ugnet.out[1].bias)]
        head_arr = (_F * 6)(*[_fptr(t) for t in head])
        gemm = ugnet.net_param.get("gemm_mode", "f16x3")
        if gemm not in ("f16x3", "f32"):
            raise MpvError("DiffSTG supports gemm_mode 'f16x3' and 'f32'")
        desc = StgDesc(ugnet.T_h, ugnet.T_p, ugnet.F, d_h, int(ugnet.net_param["Td_h"]), T_steps, GEMM_MODES[gemm])
        self._bind_stream()
        self._check(self.lib.mpv_stg_load(self._h, C.byref(desc), arr, n_steps, head_arr, _fptr(temb)), "mpv_stg_load")
        self.stg = (ugnet.T_h, ugnet.T_p, ugnet.F)
        self._stg_graph = None

    # ---- NsDiff_spatial --------------------------------------------------------------------
    def load_nss(self, ugnet, coef: torch.Tensor, steps: int, gemm: str = "f16x3"):
        """Upload NsDiff's UGnet holder (models/Diffusion_model/NsDiff/ugnet.py) with the per-step posterior coefficient rows."""
        keep = []
        arr, n_steps, temb = self._stg_steps(ugnet, ugnet.T, int(steps), keep)
        coef = _host_f32(coef)
        if tuple(coef.shape) != (int(steps), COEF_COLS):
            raise MpvError("coef must be [{}, {}]".format(steps, COEF_COLS))
        head = [_host_f32(t) for t in (ugnet.x_proj.weight, ugnet.x_proj.bias, ugnet.out[0].weight, ugnet.out[0].bias, ugnet.out[1].weight, ugnet.out[1].bias,
                                        ugnet.lin4.weight, ugnet.lin4.bias, ugnet.sigma_lin.weight, ugnet.sigma_lin.bias, coef)]
        head_arr = (_F * 11)(*[_fptr(t) for t in head])
        if gemm not in ("f16x3", "f32"):
            raise MpvError("NsDiff_spatial supports gemm_mode 'f16x3' and 'f32'")
        desc = NssDesc(ugnet.T, ugnet.F, ugnet.d_h, int(ugnet.Td_h), int(steps), GEMM_MODES[gemm])
        self._bind_stream()
        self._check(self.lib.mpv_nss_load(self._h, C.byref(desc), arr, n_steps, head_arr, _fptr(temb)), "mpv_nss_load")
        self.nss = (ugnet.T, ugnet.F, int(steps))
        self._nss_graph = None

    def nss_set_graph(self, edge_index: torch.Tensor, num_nodes: int):
        if self.nss is None:
            raise MpvError("NsDiff_spatial weights not loaded")
        e = torch.as_tensor(edge_index).detach().to("cpu", torch.int64).reshape(2, -1).contiguous()
        key = (int(num_nodes), e.shape[1], hash(e.numpy().tobytes()))
        if self._nss_graph == key:
            return
        src, dst = e[0].contiguous(), e[1].contiguous()
        P64 = C.POINTER(C.c_int64)
        self._check(self.lib.mpv_nss_set_graph(self._h, int(num_nodes), e.shape[1], C.cast(src.data_ptr(), P64), C.cast(dst.data_ptr(), P64)),
                    "mpv_nss_set_graph")
        self._nss_graph = key

    def nss_forward(self, y: torch.Tensor, y0hat: torch.Tensor, gx: torch.Tensor, t: int):
        """(eps, sigma) = UGnet(y_t, y_0_hat, g, t) over the graph set by nss_set_graph; all [rows, pred_len, F] on device."""
        if self.nss is None:
            raise MpvError("NsDiff_spatial weights not loaded")
        if not (tuple(y.shape) == tuple(y0hat.shape) == tuple(gx.shape)) or y.dim() != 3 or y.shape[1] != self.nss[0] or y.shape[2] != self.nss[1]:
            raise MpvError("y, y0hat and gx must all be [rows, {}, {}]".format(self.nss[0], self.nss[1]))
        eps, sig = torch.empty_like(y), torch.empty_like(y)
        self._bind_stream()
        self._check(self.lib.mpv_nss_forward(self._h, _dev_ptr(y, self.device, "y"), _dev_ptr(y0hat, self.device, "y0hat"), _dev_ptr(gx, self.device, "gx"),
                                             y.shape[0], int(t), _dev_ptr(eps, self.device, "eps"), _dev_ptr(sig, self.device, "sigma")), "mpv_nss_forward")
        return eps, sig

    def nss_sample(self, y0hat: torch.Tensor, gx: torch.Tensor, edge_index: torch.Tensor, n_samples: int, chunk: int,
                   noise: Optional[torch.Tensor] = None, seed: int = 0, rows_per_pass: int = 0) -> torch.Tensor:
        """y0hat, gx [W, V, pred_len, F] on device -> samples [W, V, pred_len, F, n_samples] on device.  Windows run in blocks of at most
        `rows_per_pass` node rows (0: the library's default, MPV_NSS_ROWS overrides); the result does not depend on the blocking."""
        if self.nss is None:
            raise MpvError("NsDiff_spatial weights not loaded")
        if tuple(y0hat.shape) != tuple(gx.shape) or y0hat.dim() != 4 or y0hat.shape[2] != self.nss[0] or y0hat.shape[3] != self.nss[1]:
            raise MpvError("y0hat and gx must both be [W, V, {}, {}]".format(self.nss[0], self.nss[1]))
        W, V, O, F = y0hat.shape
        if chunk < 1 or n_samples < 1 or n_samples % chunk != 0:
            raise MpvError("n_samples must be a positive multiple of chunk (parallel_sample)")
        self.nss_set_graph(edge_index, V)
        if noise is not None:
            need = W * (n_samples // chunk) * self.nss[2] * V * chunk * O * F
            if noise.numel() != need:
                raise MpvError("noise has {} elements, the draw order needs {}".format(noise.numel(), need))
        out = torch.empty((W, V, O, F, n_samples), device=self.device, dtype=torch.float32)
        rows_per_pass = int(rows_per_pass or os.environ.get("MPV_NSS_ROWS", 0))
        if rows_per_pass < 0:
            raise MpvError("rows_per_pass must be >= 0")
        shape = NssSampleShape(W, V, n_samples, chunk, seed & (2 ** 64 - 1), rows_per_pass)
        self._bind_stream()
        self._check(self.lib.mpv_nss_sample(self._h, C.byref(shape), _dev_ptr(y0hat, self.device, "y0hat"), _dev_ptr(gx, self.device, "gx"),
                                            _dev_ptr(noise, self.device, "noise"), _dev_ptr(out, self.device, "samples")), "mpv_nss_sample")
        return out

    def stg_set_graph(self, edge_index: torch.Tensor, num_nodes: int):
        if self.stg is None:
            raise MpvError("DiffSTG weights not loaded")
        e = torch.as_tensor(edge_index).detach().to("cpu", torch.int64).reshape(2, -1).contiguous()
        key = (int(num_nodes), e.shape[1], hash(e.numpy().tobytes()))
        if self._stg_graph == key:
            return
        src, dst = e[0].contiguous(), e[1].contiguous()
        P64 = C.POINTER(C.c_int64)
        self._check(self.lib.mpv_stg_set_graph(self._h, int(num_nodes), e.shape[1], C.cast(src.data_ptr(), P64), C.cast(dst.data_ptr(), P64)),
                    "mpv_stg_set_graph")
        self._stg_graph = key

    def stg_forward(self, xt: torch.Tensor, x_masked: torch.Tensor, t: int) -> torch.Tensor:
        """eps = UGnet(xt, t, x_masked) over the graph set by stg_set_graph; xt, x_masked [rows, T, F] on device."""
        if self.stg is None:
            raise MpvError("DiffSTG weights not loaded")
        if tuple(xt.shape) != tuple(x_masked.shape) or xt.dim() != 3 or xt.shape[1] != self.stg[0] + self.stg[1] or xt.shape[2] != self.stg[2]:
            raise MpvError("xt and x_masked must both be [rows, {}, {}]".format(self.stg[0] + self.stg[1], self.stg[2]))
        out = torch.empty_like(xt)
        self._bind_stream()
        self._check(self.lib.mpv_stg_forward(self._h, _dev_ptr(xt, self.device, "xt"), _dev_ptr(x_masked, self.device, "x_masked"), xt.shape[0], int(t),
                                             _dev_ptr(out, self.device, "eps")), "mpv_stg_forward")
        return out

    def stg_sample(self, history: torch.Tensor, edge_index: torch.Tensor, plan: torch.Tensor, parallel: int, sequential: int,
                   noise: Optional[torch.Tensor] = None, seed: int = 0) -> torch.Tensor:
        """history [W, V, T_h, F] on device -> samples [W, V, T, F, parallel*sequential] on device."""
        if self.stg is None:
            raise MpvError("DiffSTG weights not loaded")
        W, V, T_h, F = history.shape
        if T_h != self.stg[0] or F != self.stg[2]:
            raise MpvError("history must be [W, V, {}, {}], got {}".format(self.stg[0], self.stg[2], tuple(history.shape)))
        self.stg_set_graph(edge_index, V)
        plan = plan.detach().to("cpu", torch.float64).contiguous()
        if plan.dim() != 2 or plan.shape[1] != STG_PLAN_COLS:
            raise MpvError("plan must be [n_steps, {}]".format(STG_PLAN_COLS))
        T = self.stg[0] + self.stg[1]
        if noise is not None:
            draws = 1 + int((plan[:, 1] > 0).sum())
            need = sequential * draws * W * parallel * V * T * F
            if noise.numel() != need:
                raise MpvError("noise has {} elements, the draw order needs {}".format(noise.numel(), need))
        out = torch.empty((W, V, T, F, parallel * sequential), device=self.device, dtype=torch.float32)
        shape = StgSampleShape(W, parallel, sequential, plan.shape[0], seed & (2 ** 64 - 1))
        self._bind_stream()
        self._check(self.lib.mpv_stg_sample(self._h, C.byref(shape), _dev_ptr(history, self.device, "history"),
                                            C.cast(plan.data_ptr(), C.POINTER(C.c_double)), _dev_ptr(noise, self.device, "noise"),
                                            _dev_ptr(out, self.device, "samples")), "mpv_stg_sample")
        return out

    # ---- synthetic input generation ---------------------------------------------------
    def biomass_simulate(self, row_ptr: torch.Tensor, col_idx: torch.Tensor, x0: torch.Tensor, ticks: int, burn_ticks: int,
                         stride: int, r: float, k: float, d: float, eta: float, dt: float, eps_start: float, eps_end: float,
                         seed: int) -> torch.Tensor:
        """CSR graph (int32, on device), x0 [n_traj, V] -> sub-sampled trajectories [n_traj, V, ceil(ticks/stride), 1]."""
        for name, t in (("row_ptr", row_ptr), ("col_idx", col_idx)):
            if t.device != self.device or t.dtype != torch.int32 or not t.is_contiguous():
                raise MpvError("{} must be a contiguous int32 tensor on {}".format(name, self.device))
        n_traj, V = x0.shape
        if row_ptr.numel() != V + 1:
            raise MpvError("row_ptr has {} entries, expected nodes + 1 = {}".format(row_ptr.numel(), V + 1))
        L = (ticks + stride - 1) // stride
        out = torch.empty((n_traj, V, L, 1), device=self.device, dtype=torch.float32)
        desc = BiomassDesc(n_traj, V, ticks, burn_ticks, stride, r, k, d, eta, dt, eps_start, eps_end, seed & (2 ** 64 - 1))
        self._bind_stream()
        self._check(self.lib.mpv_biomass_simulate(self._h, C.byref(desc), C.c_void_p(row_ptr.data_ptr()),
                                                  C.c_void_p(col_idx.data_ptr()), _dev_ptr(x0, self.device, "x0"),
                                                  _dev_ptr(out, self.device, "series_out")), "mpv_biomass_simulate")
        return out

    def network_sde_simulate(self, model: str, row_ptr: torch.Tensor, col_idx: torch.Tensor, x0: torch.Tensor, ticks: int, burn_ticks: int,
                             stride: int, p0: float, p1: float, p2: float, eta: float, dt: float, eps_start: float, eps_end: float,
                             seed: int) -> torch.Tensor:
        """Euler-Maruyama of the 'biomass' (p0 = r, p1 = k, p2 = d), 'SIS' (p2 = d) or 'neuronal' (p0 = tau, p1 = mu) network dynamics:
        CSR graph (int32, on device), x0 [n_traj, V] -> sub-sampled trajectories [n_traj, V, ceil(ticks/stride), 1]."""
        if model not in SDE_MODELS:
            raise MpvError("model must be one of {}".format(sorted(SDE_MODELS)))
        for name, t in (("row_ptr", row_ptr), ("col_idx", col_idx)):
            if t.device != self.device or t.dtype != torch.int32 or not t.is_contiguous():
                raise MpvError("{} must be a contiguous int32 tensor on {}".format(name, self.device))
        n_traj, V = x0.shape
        if row_ptr.numel() != V + 1:
            raise MpvError("row_ptr has {} entries, expected nodes + 1 = {}".format(row_ptr.numel(), V + 1))
        out = torch.empty((n_traj, V, (ticks + stride - 1) // stride, 1), device=self.device, dtype=torch.float32)
        desc = SdeDesc(SDE_MODELS[model], n_traj, V, ticks, burn_ticks, stride, p0, p1, p2, eta, dt, eps_start, eps_end, seed & (2 ** 64 - 1))
        self._bind_stream()
        self._check(self.lib.mpv_network_sde_simulate(self._h, C.byref(desc), C.c_void_p(row_ptr.data_ptr()), C.c_void_p(col_idx.data_ptr()),
                                                      _dev_ptr(x0, self.device, "x0"), _dev_ptr(out, self.device, "series_out")),
                    "mpv_network_sde_simulate")
        return out

    def slbp_simulate(self, x0: torch.Tensor, ticks: int, burn_ticks: int, stride: int, dt: float, n_start: float, n_end: float, seed: int,
                      **params) -> torch.Tensor:
        """The SLBP bream / pike SDE (reference dataset/SLBP_dynamic_gene.py): x0 [n_traj, 2] on device -> [n_traj, ceil(ticks/stride), 2];
        params override SLBP_PARAMS; the nutrient level ramps linearly from n_start to n_end over the recorded ticks."""
        par = dict(SLBP_PARAMS)
        unknown = set(params) - set(par)
        if unknown:
            raise MpvError("unknown SLBP parameter(s): {}".format(sorted(unknown)))
        par.update(params)
        if x0.dim() != 2 or x0.shape[1] != 2:
            raise MpvError("x0 must be [n_trajectories, 2]")
        n = x0.shape[0]
        out = torch.empty((n, (ticks + stride - 1) // stride, 2), device=self.device, dtype=torch.float32)
        desc = SlbpDesc(n, ticks, burn_ticks, stride, par["i_b"], par["i_p"], par["r"], par["H1"], par["H2"], par["H3"], par["H4"], par["c_b"],
                        par["c_p"], par["p_r"], par["c_e"], par["m_p"], par["K"], par["D"], dt, n_start, n_end, seed & (2 ** 64 - 1))
        self._bind_stream()
        self._check(self.lib.mpv_slbp_simulate(self._h, C.byref(desc), _dev_ptr(x0, self.device, "x0"), _dev_ptr(out, self.device, "series_out")),
                    "mpv_slbp_simulate")
        return out

    def chain_stats(self, reset: bool = False):
        """(tile_steps, safe_steps) of the split-f16 sampler since the last reset: reverse steps executed per 32-chain tile
        and how many of them fell through to the exact out-of-line routine."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.lib.mpv_chain_stats(self._h, C.byref(a), C.byref(b), int(reset)), "mpv_chain_stats")
        return a.value, b.value

    def philox_normals(self, seed: int, offset: int, n_chains: int) -> torch.Tensor:
        if self.nsdiff is None:
            raise MpvError("NsDiff weights not loaded")
        out = torch.empty((n_chains, self.nsdiff[0], self.nsdiff[1]), device=self.device, dtype=torch.float32)
        self._bind_stream()
        self._check(self.lib.mpv_nsdiff_philox_normals(self._h, seed, offset, n_chains, _dev_ptr(out, self.device, "out")),
                    "mpv_nsdiff_philox_normals")
        return out

    def reduce(self, samples: torch.Tensor, std=None, mean=None, per_feature: bool = False):
        """samples [W, R, F, S] (or [W, B, O, F, S]) -> (mpv [W], pred_mean [W], mpv_feat [W, F] | None)."""
        if samples.dim() == 5:
            samples = samples.reshape(samples.shape[0], -1, samples.shape[3], samples.shape[4])
        W, R, F, S = samples.shape
        mpv = torch.empty(W, device=self.device, dtype=torch.float32)
        pm = torch.empty(W, device=self.device, dtype=torch.float32)
        feat = torch.empty((W, F), device=self.device, dtype=torch.float32) if per_feature else None
        self._bind_stream()
        self._check(self.lib.mpv_reduce(self._h, _dev_ptr(samples, self.device, "samples"), W, R, F, S,
                                        _dev_ptr(std, self.device, "std"), _dev_ptr(mean, self.device, "mean"),
                                        _dev_ptr(mpv, self.device, "mpv"), _dev_ptr(pm, self.device, "pred_mean"),
                                        _dev_ptr(feat, self.device, "mpv_feat")), "mpv_reduce")
        return mpv, pm, feat

    def timing_enable(self, on: bool = True):
        self._check(self.lib.mpv_chain_timing_enable(self._h, int(on)), "mpv_chain_timing_enable")

    def timing(self):
        ms, n = C.c_double(), C.c_int64()
        self._check(self.lib.mpv_chain_timing(self._h, C.byref(ms), C.byref(n)), "mpv_chain_timing")
        return ms.value, n.value
