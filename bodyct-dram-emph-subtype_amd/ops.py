"""Tensor-level wrappers over the C ABI (include/dram_hip.h).

PyTorch is used here for device memory (``torch.empty``) and the current HIP stream
only; every computation is a hand-written kernel in libdram_hip.so.  Each wrapper
validates on the host that operand shapes/dtypes match what the kernel and its grid
assume *before* launching (a faulting kernel can reset the whole GPU host).

Activations are NDHWC float32: ``[B, D, H, W, C]``.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import threading
import weakref
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import DramConvDesc

Tensor = torch.Tensor


def _L():
    return _lib.load()


_TLS = threading.local()        # forward runs on the caller's thread, backward on autograd's device thread


@contextlib.contextmanager
def launch_scope(device):
    """Pin the launch device for a run of kernels: makes `device` torch's current device (so the
    stream handed to the library belongs to the GPU the operand pointers live on), caches the stream
    handle, and lets `_req` reject operands that live on any other GPU *before* a kernel is launched
    with foreign pointers (a page fault / GPU reset instead of a Python error)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("libdram_hip has no CPU path")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    prev = getattr(_TLS, "scope", None)
    with torch.cuda.device(idx):
        _TLS.scope = (idx, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        try:
            yield
        finally:
            _TLS.scope = prev


def tuning_env(name: str, default: str) -> str:
    """A/B and test switches (DRAM_WGRAD_STREAM, DRAM_BF16_S2, DRAM_STEM_BF16, DRAM_SIDE_PRIORITY and the kernel-variant
    switches read inside the library) count ONLY under DRAM_TUNING=1 (tests/conftest.py, tools/, bench.py set it): a
    stray DRAM_* variable cannot change which kernels the product path runs.  User-facing settings --
    DRAM_STORAGE, DRAM_RECOMPUTE, DRAM_DIST_FORCE -- are read directly (README)."""
    return os.environ.get(name, default) if os.environ.get("DRAM_TUNING", "0") == "1" else default


_SIDE: Dict[tuple, "torch.cuda.Stream"] = {}


def side_stream(device_index: int, which: int = 0) -> "torch.cuda.Stream":
    """The per-device side HIP streams of the engine (weight-gradient kernels run there, concurrently with the
    data-gradient chain on the caller's stream; `which` = 0, 1: consecutive layers alternate, so one layer's HBM-bound
    transforms run under the previous layer's matrix-bound GEMM)."""
    s = _SIDE.get((device_index, which))
    if s is None:
        prio = int(tuning_env("DRAM_SIDE_PRIORITY", "0"))     # A/B switch (tools): HIP stream priority of the side stream
        s = _SIDE[(device_index, which)] = torch.cuda.Stream(device=device_index, priority=prio)
    return s


@contextlib.contextmanager
def on_stream(stream: "torch.cuda.Stream"):
    """Launch the enclosed kernels (and make the enclosed allocations) on `stream`; inside a launch_scope the
    cached stream handle follows."""
    prev = getattr(_TLS, "scope", None)
    with torch.cuda.stream(stream):
        if prev is not None:
            _TLS.scope = (prev[0], ctypes.c_void_p(stream.cuda_stream))
        try:
            yield
        finally:
            _TLS.scope = prev


def _launch_device() -> int:
    sc = getattr(_TLS, "scope", None)
    return sc[0] if sc is not None else torch.cuda.current_device()


def _stream():
    sc = getattr(_TLS, "scope", None)
    if sc is not None:
        return sc[1]
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(rc: int, name: str):
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")


def _req(t: Tensor, name: str, dtype=torch.float32, shape=None):
    if not isinstance(t, Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a device tensor (libdram_hip has no CPU path)")
    if t.device.index != _launch_device():
        raise RuntimeError(f"{name}: lives on cuda:{t.device.index} but kernels launch on cuda:{_launch_device()} "
                           "(all operands of a call must be on the current device)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


BF16 = torch.bfloat16


def _act(t: Tensor, name: str, shape=None, like: Optional[Tensor] = None) -> str:
    """Validate an ACTIVATION tensor (float32, or bfloat16 on the bf16-storage path) and return the entry-point
    suffix of its storage type ("" | "_bf16").  `like`: another activation of the same call, whose type it must share."""
    if not isinstance(t, Tensor) or t.dtype not in (torch.float32, BF16):
        raise TypeError(f"{name}: expected a float32 or bfloat16 device tensor, got {getattr(t, 'dtype', type(t))}")
    _req(t, name, t.dtype, shape)
    if like is not None and like.dtype != t.dtype:
        raise TypeError(f"{name}: {t.dtype} but the call's other activations are {like.dtype}")
    return "_bf16" if t.dtype == BF16 else ""


def _fn(name: str, sfx: str):
    return getattr(_L(), name + sfx)


def cast(t: Tensor, dtype) -> Tensor:
    """float32 <-> bfloat16 copy of an activation tensor (round to nearest even) with the library's cast kernels."""
    if t.dtype == dtype:
        return t
    _act(t, "t")
    out = torch.empty(t.shape, device=t.device, dtype=dtype)
    if dtype == BF16:
        _chk(_L().dram_cast_f32_to_bf16(_p(t), _p(out), t.numel(), _stream()), "dram_cast_f32_to_bf16")
    elif dtype == torch.float32:
        _chk(_L().dram_cast_bf16_to_f32(_p(t), _p(out), t.numel(), _stream()), "dram_cast_bf16_to_f32")
    else:
        raise TypeError(f"cast: unsupported target {dtype}")
    return out


class KernelTimeline:
    """The library's own kernel timeline (dram_profile_*): every kernel launch bracketed by two
    hipEvents on its launch stream, tagged with family / executed MFMA FLOPs / direct-convolution FLOPs
    / algorithmic HBM bytes.  bench.py's roofline table is built from it."""

    def __init__(self, max_records: int = 1 << 16):
        self.max_records = int(max_records)

    def start(self):
        _chk(_L().dram_profile_start(self.max_records), "dram_profile_start")

    def stop(self):
        _chk(_L().dram_profile_stop(), "dram_profile_stop")

    def families(self):
        """{family name: dict(bound, launches, ms, mfma_flops, alg_flops, hbm_bytes, variants{variant: [launches, ms]})}
        (synchronises the device)"""
        L = _L()
        buf = (_lib.DramProfRecord * self.max_records)()
        n = L.dram_profile_read(buf, self.max_records)
        if n < 0:
            raise RuntimeError(f"dram_profile_read failed with code {n}")
        out = {}
        for i in range(n):
            r = buf[i]
            name = L.dram_profile_family_name(r.family).decode()
            d = out.setdefault(name, dict(bound="mfma" if L.dram_profile_family_is_mfma(r.family) else "hbm",
                                          launches=0, ms=0.0, mfma_flops=0.0, alg_flops=0.0, hbm_bytes=0.0,
                                          variants={}))
            d["launches"] += 1
            d["ms"] += r.ms
            d["mfma_flops"] += r.mfma_flops
            d["alg_flops"] += r.alg_flops
            d["hbm_bytes"] += r.hbm_bytes
            v = d["variants"].setdefault(int(r.variant), [0, 0.0])
            v[0] += 1
            v[1] += r.ms
        self.dropped = int(L.dram_profile_dropped())
        return out


class KernelProfiler:
    """HIP-event timing of whole convolution calls on torch's current stream (bench.py --detail: the
    per-layer table).  Records (family, algorithmic flops, start, end) per call; summary() after a sync."""

    def __init__(self):
        self.records = []

    def span(self, family: str, flops: float, detail: str = ""):
        return _Span(self, family, flops, detail)

    def by_launch(self):
        """{(family, detail): dict(launches, ms, flops)} -- per-geometry breakdown."""
        torch.cuda.synchronize()
        out = {}
        for fam, flops, a, b, detail in self.records:
            d = out.setdefault((fam, detail), dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += a.elapsed_time(b)
            d["flops"] += flops
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, flops, a, b, _ in self.records:
            d = out.setdefault(fam, dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += a.elapsed_time(b)
            d["flops"] += flops
        return out


class _Span:
    def __init__(self, prof, family, flops, detail=""):
        self.prof, self.family, self.flops, self.detail = prof, family, flops, detail

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.prof.records.append((self.family, self.flops, self.a, self.b, self.detail))


class _NoSpan:
    def __enter__(self):
        pass

    def __exit__(self, *exc):
        pass


_PROFILER: Optional[KernelProfiler] = None
_NOSPAN = _NoSpan()


def set_profiler(p: Optional[KernelProfiler]):
    global _PROFILER
    _PROFILER = p


def _span(family: str, flops: float, detail: str = ""):
    return _NOSPAN if _PROFILER is None else _PROFILER.span(family, flops, detail)


@dataclass(frozen=True)
class ConvGeom:
    """Forward-sense geometry of one convolution call."""
    B: int
    D: int
    H: int
    W: int
    Cin: int
    Cout: int
    k: int
    stride: int
    pad: int
    dil: int
    tol: int = 0        # DRAM_CONV_ROUNDING_TOLERANT (include/dram_hip.h): plan hint set by the engine for BasicBlock networks

    def out(self, n: int) -> int:
        return (n + 2 * self.pad - (self.dil * (self.k - 1) + 1)) // self.stride + 1

    @property
    def Do(self):
        return self.out(self.D)

    @property
    def Ho(self):
        return self.out(self.H)

    @property
    def Wo(self):
        return self.out(self.W)

    @property
    def taps(self):
        return self.k ** 3

    @property
    def flops(self) -> float:
        """algorithmic FLOPs of one pass (fwd, dgrad and wgrad each): 2*M*N*K"""
        return 2.0 * self.B * self.Do * self.Ho * self.Wo * self.Cout * self.Cin * self.taps

    @property
    def in_shape(self):
        return (self.B, self.D, self.H, self.W, self.Cin)

    @property
    def out_shape(self):
        return (self.B, self.Do, self.Ho, self.Wo, self.Cout)

    def desc(self) -> DramConvDesc:
        return DramConvDesc(self.B, self.D, self.H, self.W, self.Cin, self.Do, self.Ho, self.Wo, self.Cout,
                            self.k, self.stride, self.pad, self.dil, self.tol)


# --------------------------------------------------------------------------- conv
# Memory a CAPTURED launch bakes a pointer of (scratch buffers, device work lists) must outlive the graph, and must not
# be handed to anybody else meanwhile.  Whoever owns a capture passes a list to capture_keepalive() and keeps it as long
# as the graph (graph.GraphedTrainStep does); captures made without one pin such memory for the life of the process.
_CAPTURE_KEEP: List[object] = []
_CAPTURE_OWNER: Optional[List[object]] = None       # process-wide: backward runs on autograd's thread, not the caller's


@contextlib.contextmanager
def capture_keepalive(owner: List[object]):
    global _CAPTURE_OWNER
    prev, _CAPTURE_OWNER = _CAPTURE_OWNER, owner
    try:
        yield owner
    finally:
        _CAPTURE_OWNER = prev


def capture_id() -> int:
    """0 when the launch stream is not being captured into a hipGraph, else a number unique to that capture."""
    return int(_L().dram_stream_capture_id(_stream()))


def _keep_for_capture(obj):
    keep = _CAPTURE_OWNER if _CAPTURE_OWNER is not None else _CAPTURE_KEEP
    if not any(o is obj for o in keep[-16:]):
        keep.append(obj)


_WORKSPACE: Dict[tuple, Tensor] = {}
_CAPTURE_WS: Dict[tuple, "weakref.ref"] = {}        # (device, stream, capture id) -> the capture's own scratch (weak)


def _workspace(nbytes: int, device) -> Tensor:
    """Grow-only scratch shared by the conv passes that run back to back on ONE stream (one buffer per
    (device, launch stream): the weight-gradient stream has its own)."""
    device = torch.device(device)
    n = (nbytes + 3) // 4
    cid = capture_id()
    if cid:
        # capturing: a buffer of THIS capture's own (from the graph's private pool), owned by the capture's keep-alive
        # list -- never shared with an eager call or another capture on the same stream, never evicted: the graph
        # writes to it on every replay.  A buffer it outgrows stays alive too (earlier captured launches use it).
        key = (device, _stream().value, cid)
        ref = _CAPTURE_WS.get(key)
        ws = ref() if ref is not None else None
        if ws is None or ws.numel() < n:
            ws = torch.empty((n,), device=device, dtype=torch.float32)
            _keep_for_capture(ws)
            for k in [k for k, r in _CAPTURE_WS.items() if r() is None]:
                del _CAPTURE_WS[k]
            _CAPTURE_WS[key] = weakref.ref(ws)
        return ws
    key = (device, _stream().value)
    ws = _WORKSPACE.pop(key, None)                  # re-inserted below: the dict stays in least-recently-used order
    if ws is None or ws.numel() < n:
        ws = None
        ws = torch.empty((n,), device=device, dtype=torch.float32)
        while len(_WORKSPACE) >= 6:                 # caller stream + side stream + a few more; streams that went away
            _WORKSPACE.pop(next(iter(_WORKSPACE)))  # must not pin up to 0.7 GB each
    _WORKSPACE[key] = ws
    return ws


# environment switches the library's plan functions read (tests flip them between cases): part of the plan-cache key
_PLAN_ENV = ("DRAM_CONV_ALGO", "DRAM_WINO_TILING", "DRAM_MATH", "DRAM_W2D_V", "DRAM_IGEMM_V", "DRAM_IGEMM_V3_FORCE",
             "DRAM_WGRAD_V", "DRAM_BF16_NW", "DRAM_BF16_WGRAD", "DRAM_W2D_MARGIN", "DRAM_W2D_MARGIN_BIG", "DRAM_NN_STREAM",
             "DRAM_WGRAD_MARGIN")


class ConvPlan:
    """Everything the host needs to know about one convolution geometry, asked from the library ONCE per
    (geometry, plan-relevant environment): forward / weight-gradient algorithm, packed-weight leading
    dimensions, statistic rows, workspace sizes, cached-transform size -- ~12 host calls into the library
    (each of which re-derives tilings and reads the environment) instead of that many per launch."""
    __slots__ = ("desc", "dref", "desc_ovl", "dref_ovl", "algo", "walgo", "taps_f", "taps_b", "stat_rows", "ws_fwd", "ws_bwd", "ws_wgrad",
                 "v_elems", "ws_direct_wgrad", "bf16", "bf16_stat_rows", "bf16_ws_wgrad", "prologue")

    def __init__(self, g: "ConvGeom"):
        L = _L()
        self.desc = g.desc()
        self.dref = d = ctypes.byref(self.desc)
        # the same geometry with DRAM_CONV_BWD_OVERLAPPED (include/dram_hip.h): a data gradient launched while another
        # stream runs weight-gradient kernels (conv3d_bwd_data(..., overlapped=True))
        self.desc_ovl = g.desc()
        self.desc_ovl.flags |= 2
        self.dref_ovl = ctypes.byref(self.desc_ovl)
        self.algo = int(L.dram_conv_algo(d))
        self.walgo = int(L.dram_conv_wgrad_algo(d))
        if self.algo == 1:
            self.taps_f, self.taps_b = int(L.dram_wino_num_points(d)), int(L.dram_wino_num_points_bwd(d))
        else:
            self.taps_f = self.taps_b = 48 if self.algo == 2 else g.taps
        self.stat_rows = int({0: L.dram_conv_num_mtiles, 1: L.dram_wino_num_stat_rows, 2: L.dram_wino2d_num_stat_rows,
                              3: L.dram_conv1x1_num_stat_rows}[self.algo](d))
        self.ws_fwd = int(L.dram_wino_workspace(d, 0)) if self.algo == 1 else 0
        self.ws_bwd = int(L.dram_wino_workspace(d, 1)) if self.algo == 1 else 0
        self.v_elems = int(L.dram_wino_v_elems(d)) if (self.algo == 1 or self.walgo == 1) else 0
        self.ws_wgrad = {3: lambda: int(L.dram_conv1x1_bwd_weight_workspace(d)),
                         2: lambda: int(L.dram_wgrad_w2d_workspace(d)),
                         1: lambda: int(L.dram_wino_workspace(d, 2))}.get(self.walgo, lambda: 0)()
        self.ws_direct_wgrad = int(L.dram_conv3d_bwd_weight_workspace(d)) if (self.walgo == 0 or self.ws_wgrad == 0) else 0
        # bf16-storage path: the direct bf16-MFMA kernels take the 3x3x3 stride-1 and the 1x1x1 convolutions; the one
        # stride-2 convolution per network runs on them through its space-to-depth form (s2_geom), whatever is left
        # (odd extents) on the fp32 kernels above around cast passes
        self.prologue = bool(self.algo == 1 and L.dram_wino_prologue_supported(d))
        self.bf16 = bool(L.dram_conv_bf16_supported(d))
        self.bf16_stat_rows = int(L.dram_conv_bf16_num_stat_rows(d)) if self.bf16 else 0
        self.bf16_ws_wgrad = int(L.dram_conv3d_bwd_weight_bf16_workspace(d)) if self.bf16 else 0


_PLANS: Dict[tuple, ConvPlan] = {}


def conv_plan(g: "ConvGeom") -> ConvPlan:
    env = os.environ
    # the library reads the A/B switches under DRAM_TUNING=1 only (csrc/common.h tune_env): without it the plan depends
    # on the geometry alone (13 environment look-ups per call were 1.2 ms of a ResNet-50 step's host time)
    key = (g,) + tuple(env.get(k) for k in _PLAN_ENV) if env.get("DRAM_TUNING") == "1" else g
    p = _PLANS.get(key)
    if p is None:
        p = _PLANS[key] = ConvPlan(g)
    return p


def conv_algo(g: "ConvGeom") -> int:
    """Library plan for this geometry (dram_conv_algo): 0 direct implicit GEMM, 1 Winograd
    F(2x2x2,3x3x3) pipeline, 2 fused in-plane Winograd F(2x2,3x3) x direct-z, 3 1x1x1 GEMM."""
    return conv_plan(g).algo


def conv_use_wino(g: "ConvGeom") -> bool:
    return conv_algo(g) == 1


def packed_taps(g: "ConvGeom", bwd: bool = False) -> int:
    """Leading dimension of the packed weights the library's plan expects for g (forward operand wf, or
    with bwd=True the data-gradient operand wb: the Winograd pipeline may tile the two passes differently)."""
    p = conv_plan(g)
    return p.taps_b if bwd else p.taps_f


def s2_geom(g: "ConvGeom") -> Optional["ConvGeom"]:
    """bf16 storage: the stride-1 geometry that runs a stride-2 3x3x3 convolution on the bf16 kernels (space to depth:
    eight parity sub-lattices as 8 Cin channels at half the extents, embedded weights -- include/dram_hip.h,
    dram_s2d_bf16), or None (odd extents, another kernel size, or DRAM_BF16_S2=0: fp32 kernels around casts)."""
    if (g.k != 3 or g.stride != 2 or g.pad != 1 or g.dil != 1 or ((g.D | g.H | g.W) & 1) or g.Cin % 8
            or tuning_env("DRAM_BF16_S2", "1") == "0"):
        return None
    g8 = ConvGeom(g.B, g.D // 2, g.H // 2, g.W // 2, 8 * g.Cin, g.Cout, 3, 1, 1, 1, g.tol)
    return g8 if (conv_plan(g8).bf16 and tuple(g8.out_shape) == tuple(g.out_shape)) else None


def s2d(x: Tensor) -> Tensor:
    B, D, H, W, C = x.shape
    x8 = torch.empty((B, D // 2, H // 2, W // 2, 8 * C), device=x.device, dtype=BF16)
    _chk(_L().dram_s2d_bf16(_p(x), _p(x8), B, D, H, W, C, _stream()), "dram_s2d_bf16")
    return x8


def pack_conv_weight(w: Tensor, want_fwd=True, want_bwd=True, g: Optional["ConvGeom"] = None, dtype=torch.float32
                     ) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """[Cout,Cin,k,k,k] -> wf [taps,Cout,Cin], wb [taps,Cin,Cout]; for a geometry the library plans
    on the Winograd path the packed copies are the transformed weights (64 'taps', wb tap-flipped).
    dtype=bfloat16 (bf16-storage path): bf16 copies for the geometries the bf16 kernels take, the fp32 packing
    otherwise (the convolution then runs on the fp32 kernels around casts; the consumer looks at wf.dtype)."""
    _req(w, "w")
    Cout, Cin = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3] * w.shape[4]
    if dtype == BF16 and g is not None and conv_plan(g).bf16:
        wf = torch.empty((taps, Cout, Cin), device=w.device, dtype=BF16) if want_fwd else None
        wb = torch.empty((taps, Cin, Cout), device=w.device, dtype=BF16) if want_bwd else None
        _chk(_L().dram_pack_conv_weight_bf16(_p(w), _p(wf), _p(wb), Cout, Cin, taps, _stream()), "dram_pack_conv_weight_bf16")
        return wf, wb
    if dtype == BF16 and g is not None and s2_geom(g) is not None:      # stride 2: embedded weights of the stride-1 form
        w3 = torch.empty((Cout, 8 * Cin, 3, 3, 3), device=w.device, dtype=torch.float32)
        _chk(_L().dram_s2_embed_weight(_p(w), _p(w3), Cout, Cin, _stream()), "dram_s2_embed_weight")
        wf = torch.empty((taps, Cout, 8 * Cin), device=w.device, dtype=BF16) if want_fwd else None
        wb = torch.empty((taps, 8 * Cin, Cout), device=w.device, dtype=BF16) if want_bwd else None
        _chk(_L().dram_pack_conv_weight_bf16(_p(w3), _p(wf), _p(wb), Cout, 8 * Cin, taps, _stream()),
             "dram_pack_conv_weight_bf16")
        return wf, wb
    plan = conv_plan(g) if g is not None else None
    algo = plan.algo if plan else 0
    pt = plan.taps_f if plan else taps
    ptb = plan.taps_b if plan else taps
    wf = torch.empty((pt, Cout, Cin), device=w.device, dtype=torch.float32) if want_fwd else None
    wb = torch.empty((ptb, Cin, Cout), device=w.device, dtype=torch.float32) if want_bwd else None
    if algo == 1:
        _chk(_L().dram_wino_pack_weight(_p(w), _p(wf), _p(wb), plan.dref, _stream()),
             "dram_wino_pack_weight")
    elif algo == 2:
        _chk(_L().dram_wino2d_pack_weight(_p(w), _p(wf), _p(wb), Cout, Cin, _stream()), "dram_wino2d_pack_weight")
    else:
        _chk(_L().dram_pack_conv_weight(_p(w), _p(wf), _p(wb), Cout, Cin, taps, _stream()), "dram_pack_conv_weight")
    return wf, wb


# All bf16 weight packings of a training step in ONE launch (dram_pack_conv_weight_bf16_multi): the device work list is
# built once per set of weights (their addresses and shapes: parameters are updated in place) by an EAGER step and
# reused, also inside hipGraph captures, where a host->device copy is not allowed.
_PACK_TABLES: Dict[tuple, tuple] = {}


def pack_conv_weights_bf16_multi(weights: List[Tensor]):
    """[Cout,Cin,k,k,k] fp32 weights -> [(wf [taps,Cout,Cin], wb [taps,Cin,Cout])] bf16, views of one flat buffer
    written by one kernel; None when no work list exists yet and the stream is capturing (the caller packs one by one)."""
    import numpy as np
    if not weights:
        return []
    key = tuple((w.data_ptr(), tuple(w.shape)) for w in weights)
    ent = _PACK_TABLES.pop(key, None)               # (re-inserted below: least recently USED is evicted first)
    if ent is not None:
        _PACK_TABLES[key] = ent
    if ent is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        tab = np.zeros(len(weights), dtype=np.dtype([("w", "<u8"), ("off_f", "<i8"), ("off_b", "<i8"), ("Cout", "<i4"),
                                                     ("Cin", "<i4"), ("taps", "<i4"), ("pad", "<i4")]))
        assert tab.dtype.itemsize == ctypes.sizeof(_lib.DramPackRef)
        chunks, layout, off = [], [], 0
        for i, w in enumerate(weights):
            _req(w, "w")
            Cout, Cin = w.shape[0], w.shape[1]
            taps = w.shape[2] * w.shape[3] * w.shape[4]
            n = Cout * Cin * taps
            npad = (n + 63) // 64 * 64
            tab[i] = (w.data_ptr(), off, off + npad, Cout, Cin, taps, 0)
            layout.append((off, off + npad, n, taps, Cout, Cin))
            off += 2 * npad
            ntiles = _L().dram_pack_conv_weight_bf16_tiles(Cout, Cin, taps)
            _chk(min(ntiles, 0), "dram_pack_conv_weight_bf16_tiles")
            chunks.extend((i, 0, t) for t in range(ntiles))
        ch = np.array(chunks, dtype=np.dtype([("tensor", "<i4"), ("pad", "<i4"), ("offset", "<i8")]))
        dev = weights[0].device
        ent = (torch.from_numpy(tab.view(np.uint8).copy()).to(dev), torch.from_numpy(ch.view(np.uint8).copy()).to(dev),
               len(chunks), off, layout)
        while len(_PACK_TABLES) >= 8:
            _PACK_TABLES.pop(next(iter(_PACK_TABLES)))
        _PACK_TABLES[key] = ent
    if torch.cuda.is_current_stream_capturing():
        _keep_for_capture(ent)                      # the captured launch reads this work list on every replay
    table, chunks, nchunks, total, layout = ent
    flat = torch.empty((total,), device=weights[0].device, dtype=BF16)
    _chk(_L().dram_pack_conv_weight_bf16_multi(_p(table), _p(chunks), nchunks, _p(flat), float(total // 2), _stream()),
         "dram_pack_conv_weight_bf16_multi")
    return [(flat[of:of + n].view(taps, Cout, Cin), flat[ob:ob + n].view(taps, Cin, Cout))
            for of, ob, n, taps, Cout, Cin in layout]


# Packed forward weights of inference calls (no_grad): repacking / re-transforming every weight on every
# forward is pure overhead when the weights did not change.  Entries hang off the weight TENSOR OBJECT (weakly:
# they die with it) and hold a reference to the storage they were packed from, so that address cannot be
# recycled for another tensor while the entry lives; an entry is valid while the tensor still uses that storage,
# its torch version counter is unchanged and no raw-pointer writer (FusedAdam / FusedSGD bump WEIGHT_EPOCH)
# has run.

WEIGHT_EPOCH = 0
_PACKED: Dict[int, tuple] = {}      # id(weight) -> (weakref to the weight, {plan key: entry}); removed when it dies


def weights_changed():
    """Called by anything that rewrites parameters through raw pointers (FusedAdam / FusedSGD)."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def packed_forward_weight(w: Tensor, g: "ConvGeom", dtype=torch.float32) -> Tensor:
    slot = _PACKED.get(id(w))
    if slot is None or slot[0]() is not w:          # identity, never tensor ==
        wid = id(w)
        slot = _PACKED[wid] = (weakref.ref(w, lambda _r, wid=wid: _PACKED.pop(wid, None)), {})
    per = slot[1]
    key = (g, dtype) + tuple(os.environ.get(k) for k in _PLAN_ENV)
    ent = per.get(key)
    st = w.untyped_storage()
    if ent is not None and ent[0].data_ptr() == st.data_ptr() and ent[1] == (w.data_ptr(), w._version, WEIGHT_EPOCH):
        return ent[2]
    wf = pack_conv_weight(w, True, False, g, dtype)[0]
    per[key] = (st, (w.data_ptr(), w._version, WEIGHT_EPOCH), wf)
    return wf


def conv3d_fwd(x: Tensor, wf: Tensor, bias: Optional[Tensor], g: ConvGeom, want_stats: bool):
    y, stats, _ = conv3d_fwd_keep(x, wf, bias, g, want_stats, False)
    return y, stats


def conv_prologue_ok(g: ConvGeom, dtype=torch.float32) -> bool:
    """Can the convolution apply the producing unit's BatchNorm + ReLU on the way in (conv3d_fwd_keep(prologue=...))?
    The fp32 Winograd pipeline with F(4,3)^3 tiles; DRAM_BN_PROLOGUE=0 under DRAM_TUNING=1 switches it off (A/B)."""
    return dtype == torch.float32 and tuning_env("DRAM_BN_PROLOGUE", "1") != "0" and conv_plan(g).prologue


def conv3d_fwd_keep(x: Tensor, wf: Tensor, bias: Optional[Tensor], g: ConvGeom, want_stats: bool, keep: bool,
                    prologue: Optional[Tuple[Tensor, Tensor]] = None):
    """Forward conv; with keep=True on the Winograd path also returns the transformed input V
    (reused by conv3d_bwd_weight instead of transforming x again), else None.
    prologue=(scale, shift): x is the PRE-BatchNorm output of the producing unit and max(x*scale + shift, 0) is applied
    inside the input transform (only where conv_prologue_ok(g))."""
    plan = conv_plan(g)
    if prologue is not None and not (x.dtype == torch.float32 and plan.prologue):
        raise RuntimeError(f"conv3d_fwd_keep: no BatchNorm prologue for {g}")
    if _act(x, "x", g.in_shape):                       # bf16 storage
        if bias is not None:
            _req(bias, "bias", shape=(g.Cout,))
        if wf.dtype == BF16 and wf.shape[2] == 8 * g.Cin:      # stride 2 as stride 1 over the space-to-depth tensor
            return conv3d_fwd_keep(s2d(x), wf, bias, s2_geom(g), want_stats, False)
        if wf.dtype == BF16:
            _req(wf, "wf", BF16, (g.taps, g.Cout, g.Cin))
            y = torch.empty(g.out_shape, device=x.device, dtype=BF16)
            stats = torch.empty((plan.bf16_stat_rows, 2, g.Cout), device=x.device, dtype=torch.float32) if want_stats else None
            with _span("conv3_bf16_kernel", g.flops, f"fwd {g}"):
                _chk(_L().dram_conv3d_fwd_bf16(_p(x), _p(wf), _p(bias), _p(y), _p(stats), plan.dref, _stream()),
                     f"dram_conv3d_fwd_bf16{g}")
            return y, stats, None
        # geometry outside the bf16 kernels: fp32 kernels around casts (statistics of the fp32 result)
        y32, stats, _ = conv3d_fwd_keep(cast(x, torch.float32), wf, bias, g, want_stats, False)
        return cast(y32, BF16), stats, None
    algo, d = plan.algo, plan.dref
    _req(wf, "wf", shape=(plan.taps_f, g.Cout, g.Cin))
    if bias is not None:
        _req(bias, "bias", shape=(g.Cout,))
    y = torch.empty(g.out_shape, device=x.device, dtype=torch.float32)
    stats = None
    if want_stats:
        if plan.stat_rows <= 0:
            raise RuntimeError(f"dram_conv_num_mtiles rejected {g}")
        stats = torch.empty((plan.stat_rows, 2, g.Cout), device=x.device, dtype=torch.float32)
    if algo == 1:
        nbytes = plan.ws_fwd
        ws = _workspace(nbytes, x.device)
        v = None
        if keep:
            v = torch.empty((plan.v_elems,), device=x.device, dtype=torch.float32)
        with _span("conv_wino_kernels", g.flops, f"fwd {g}"):
            if prologue is not None:
                _req(prologue[0], "scale", shape=(g.Cin,))
                _req(prologue[1], "shift", shape=(g.Cin,))
                _chk(_L().dram_wino_conv3d_fwd_bn(_p(x), _p(prologue[0]), _p(prologue[1]), _p(wf), _p(bias), _p(y), _p(stats),
                                                  _p(v), d, _p(ws), nbytes, _stream()), f"dram_wino_conv3d_fwd_bn{g}")
            else:
                _chk(_L().dram_wino_conv3d_fwd(_p(x), _p(wf), _p(bias), _p(y), _p(stats), _p(v), d, _p(ws),
                                               nbytes, _stream()), f"dram_wino_conv3d_fwd{g}")
        return y, stats, v
    if algo == 3:                                    # 1x1x1: plain GEMM, wf [1, Cout, Cin] is the weight itself
        with _span("conv1x1_gemm", g.flops, f"fwd {g}"):
            _chk(_L().dram_conv1x1_fwd(_p(x), _p(wf), _p(bias), _p(y), _p(stats), d, _stream()),
                 f"dram_conv1x1_fwd{g}")
        return y, stats, None
    if algo == 2:
        with _span("conv_wino2d_kernel", g.flops, f"fwd {g}"):
            _chk(_L().dram_wino2d_conv3d_fwd(_p(x), _p(wf), _p(bias), _p(y), _p(stats), d, _stream()),
                 f"dram_wino2d_conv3d_fwd{g}")
        return y, stats, None
    with _span("conv_igemm_kernel", g.flops, f"fwd {g}"):
        _chk(_L().dram_conv3d_fwd(_p(x), _p(wf), _p(bias), _p(y), _p(stats), d, _stream()),
             f"dram_conv3d_fwd{g}")
    return y, stats, None


def conv3d_bwd_data(dy: Tensor, wb: Tensor, g: ConvGeom, add: Optional[Tensor] = None,
                    gate: Optional[Tensor] = None, overlapped: bool = False) -> Tensor:
    """overlapped: weight-gradient kernels run on another stream meanwhile (DRAM_CONV_BWD_OVERLAPPED: a hint for the
    Winograd pipeline's choice of GEMM form; results do not depend on it)."""
    plan = conv_plan(g)
    if _act(dy, "dy", g.out_shape):                    # bf16 storage
        if add is not None:
            _act(add, "add", g.in_shape, like=dy)
        if gate is not None:
            _act(gate, "gate", g.in_shape, like=dy)
        if wb.dtype == BF16 and wb.shape[1] == 8 * g.Cin:      # stride 2: gradient of the space-to-depth tensor, then back
            dx8 = conv3d_bwd_data(dy, wb, s2_geom(g))
            dx = torch.empty(g.in_shape, device=dy.device, dtype=BF16)
            _chk(_L().dram_d2s_bf16(_p(dx8), _p(add), _p(gate), _p(dx), g.B, g.D, g.H, g.W, g.Cin, _stream()),
                 "dram_d2s_bf16")
            return dx
        if wb.dtype == BF16:
            _req(wb, "wb", BF16, (g.taps, g.Cin, g.Cout))
            dx = torch.empty(g.in_shape, device=dy.device, dtype=BF16)
            with _span("conv3_bf16_kernel", g.flops, f"dgrad {g}"):
                _chk(_L().dram_conv3d_bwd_data_bf16(_p(dy), _p(wb), _p(dx), _p(add), _p(gate), plan.dref, _stream()),
                     f"dram_conv3d_bwd_data_bf16{g}")
            return dx
        f32 = torch.float32
        return cast(conv3d_bwd_data(cast(dy, f32), wb, g, None if add is None else cast(add, f32),
                                    None if gate is None else cast(gate, f32)), BF16)
    algo, d = plan.algo, plan.dref
    _req(wb, "wb", shape=(plan.taps_b, g.Cin, g.Cout))
    if add is not None:
        _req(add, "add", shape=g.in_shape)
    if gate is not None:
        _req(gate, "gate", shape=g.in_shape)
    dx = torch.empty(g.in_shape, device=dy.device, dtype=torch.float32)
    if algo == 1:
        nbytes = plan.ws_bwd
        ws = _workspace(nbytes, dy.device)
        with _span("conv_wino_kernels", g.flops, f"dgrad {g}"):
            _chk(_L().dram_wino_conv3d_bwd_data(_p(dy), _p(wb), _p(dx), _p(add), _p(gate),
                                                plan.dref_ovl if overlapped else d, _p(ws),
                                                nbytes, _stream()), f"dram_wino_conv3d_bwd_data{g}")
        return dx
    if algo == 3:
        with _span("conv1x1_gemm", g.flops, f"dgrad {g}"):
            _chk(_L().dram_conv1x1_bwd_data(_p(dy), _p(wb), _p(dx), _p(add), _p(gate), d, _stream()),
                 f"dram_conv1x1_bwd_data{g}")
        return dx
    if algo == 2:
        with _span("conv_wino2d_kernel", g.flops, f"dgrad {g}"):
            _chk(_L().dram_wino2d_conv3d_bwd_data(_p(dy), _p(wb), _p(dx), _p(add), _p(gate), d,
                                                  _stream()), f"dram_wino2d_conv3d_bwd_data{g}")
        return dx
    with _span("conv_igemm_kernel", g.flops, f"dgrad {g}"):
        _chk(_L().dram_conv3d_bwd_data(_p(dy), _p(wb), _p(dx), _p(add), _p(gate), d, _stream()),
             f"dram_conv3d_bwd_data{g}")
    return dx


def conv_bwd_bnstats_ok(g: ConvGeom, dtype=torch.float32) -> bool:
    """Can the data gradient of this convolution also take the BatchNorm-backward statistics of the unit in front of it
    (conv3d_bwd_data_bnstats)?  The fp32 Winograd pipeline; DRAM_BWD_BNSTATS=0 under DRAM_TUNING=1 switches it off (A/B)."""
    return (dtype == torch.float32 and tuning_env("DRAM_BWD_BNSTATS", "1") != "0" and conv_plan(g).algo == 1
            and g.Cin % 64 == 0)


def conv3d_bwd_data_bnstats(dy: Tensor, wb: Tensor, g: ConvGeom, bn_y: Tensor, mean: Tensor, invstd: Tensor,
                            scale: Tensor, shift: Tensor, overlapped: bool = False):
    """-> (dx, partial): the data gradient (bit-identical to conv3d_bwd_data) and, from its output transform, the rows
    bn_bwd_reduce(dx, None, bn_y, mean, invstd, True, scale, shift) would produce in a pass of its own -- dx is dz of the
    BatchNorm + ReLU unit in front of this convolution (reference med3d.py:121-124 backward), bn_y that unit's
    pre-BatchNorm output."""
    plan = conv_plan(g)
    if not conv_bwd_bnstats_ok(g, dy.dtype):
        raise RuntimeError(f"conv3d_bwd_data_bnstats: not available for {g}")
    _req(dy, "dy", shape=g.out_shape)
    _req(wb, "wb", shape=(plan.taps_b, g.Cin, g.Cout))
    _req(bn_y, "bn_y", shape=g.in_shape)
    for name, t in (("mean", mean), ("invstd", invstd), ("scale", scale), ("shift", shift)):
        _req(t, name, shape=(g.Cin,))
    rows = int(_L().dram_wino_num_stat_rows_bwd(plan.dref))
    if rows <= 0:
        raise RuntimeError(f"conv3d_bwd_data_bnstats: no statistic rows for {g}")
    dx = torch.empty(g.in_shape, device=dy.device, dtype=torch.float32)
    partial = torch.empty((rows, 2, g.Cin), device=dy.device, dtype=torch.float32)
    nbytes = plan.ws_bwd
    ws = _workspace(nbytes, dy.device)
    with _span("conv_wino_kernels", g.flops, f"dgrad+bnstats {g}"):
        _chk(_L().dram_wino_conv3d_bwd_data_bn(_p(dy), _p(wb), _p(dx), _p(bn_y), _p(mean), _p(invstd), _p(scale),
                                               _p(shift), _p(partial), plan.dref_ovl if overlapped else plan.dref,
                                               _p(ws), nbytes, _stream()),
             f"dram_wino_conv3d_bwd_data_bn{g}")
    return dx, partial


def conv3d_bwd_weight(x: Tensor, dy: Tensor, g: ConvGeom, out: Optional[Tensor] = None,
                      v_cache: Optional[Tensor] = None) -> Tensor:
    plan = conv_plan(g)
    d, walgo = plan.dref, plan.walgo
    shape = (g.Cout, g.Cin, g.k, g.k, g.k)
    dw = out if out is not None else torch.empty(shape, device=dy.device, dtype=torch.float32)
    _req(dw, "dw", shape=shape)
    if x is not None and _act(x, "x", g.in_shape):     # bf16 storage: the gradient itself is fp32
        _act(dy, "dy", g.out_shape, like=x)
        if plan.bf16:
            nbytes = plan.bf16_ws_wgrad
            ws = _workspace(nbytes, x.device)
            with _span("wgrad3_bf16_kernel+reduce", g.flops, f"wgrad {g}"):
                _chk(_L().dram_conv3d_bwd_weight_bf16(_p(x), _p(dy), _p(dw), d, _p(ws), nbytes, _stream()),
                     f"dram_conv3d_bwd_weight_bf16{g}")
            return dw
        g8 = s2_geom(g)
        if g8 is not None:                              # stride 2: gradient of the embedded weights, 27 of 216 slots kept
            dw3 = conv3d_bwd_weight(s2d(x), dy, g8)
            _chk(_L().dram_s2_extract_wgrad(_p(dw3), _p(dw), g.Cout, g.Cin, _stream()), "dram_s2_extract_wgrad")
            return dw
        return conv3d_bwd_weight(cast(x, torch.float32), cast(dy, torch.float32), g, out=dw)
    if x is None:                                     # the forward's transformed input stands in for x (pipeline only)
        if not (walgo == 1 and v_cache is not None and plan.ws_wgrad):
            raise RuntimeError(f"conv3d_bwd_weight: x missing and no cached Winograd-domain input for {g}")
    else:
        _req(x, "x", shape=g.in_shape)
    _req(dy, "dy", shape=g.out_shape)
    if walgo == 3:
        nbytes = plan.ws_wgrad
        ws = _workspace(max(nbytes, 4), x.device)
        with _span("conv1x1_gemm", g.flops, f"wgrad {g}"):
            _chk(_L().dram_conv1x1_bwd_weight(_p(x), _p(dy), _p(dw), d, _p(ws), nbytes, _stream()),
                 f"dram_conv1x1_bwd_weight{g}")
        return dw
    if walgo == 2:
        nbytes = plan.ws_wgrad
        ws = _workspace(nbytes, x.device)
        with _span("conv_wgrad_w2d_kernel+reduce", g.flops, f"wgrad {g}"):
            _chk(_L().dram_wgrad_w2d(_p(x), _p(dy), _p(dw), d, _p(ws), nbytes, _stream()),
                 f"dram_wgrad_w2d{g}")
        return dw
    if walgo == 1:
        nbytes = plan.ws_wgrad
        if nbytes:                                   # 0: this geometry's weight gradient stays on the direct path
            ws = _workspace(nbytes, dy.device)
            if v_cache is not None:
                _req(v_cache, "v_cache", shape=(plan.v_elems,))
            with _span("conv_wino_kernels", g.flops, f"wgrad {g}"):
                _chk(_L().dram_wino_conv3d_bwd_weight(_p(x), _p(v_cache), _p(dy), _p(dw), d, _p(ws),
                                                      nbytes, _stream()), f"dram_wino_conv3d_bwd_weight{g}")
            return dw
    nbytes = plan.ws_direct_wgrad
    if nbytes == 0:
        raise RuntimeError(f"dram_conv3d_bwd_weight: unsupported geometry {g}")
    ws = _workspace(nbytes, x.device)
    with _span("conv_wgrad_kernel+reduce", g.flops, f"wgrad {g}"):
        _chk(_L().dram_conv3d_bwd_weight(_p(x), _p(dy), _p(dw), d, _p(ws), nbytes, _stream()),
             f"dram_conv3d_bwd_weight{g}")
    return dw


# --------------------------------------------------------------------------- stem
def stem_out(n: int) -> int:
    return (n + 6 - 7) // 2 + 1


def stem_fwd(x: Tensor, w: Tensor, want_stats: bool, out_dtype=torch.float32):
    """x [B,D,H,W] (C=1), w [64,1,7,7,7] -> y [B,Do,Ho,Wo,64] (fp32 arithmetic; y stored as out_dtype)."""
    _req(x, "x")
    if x.dim() != 4:
        raise ValueError("stem_fwd: x must be [B,D,H,W]")
    _req(w, "w", shape=(64, 1, 7, 7, 7))
    B, D, H, W = x.shape
    Do, Ho, Wo = stem_out(D), stem_out(H), stem_out(W)
    if out_dtype not in (torch.float32, BF16):
        raise TypeError(f"stem_fwd: unsupported storage type {out_dtype}")
    y = torch.empty((B, Do, Ho, Wo, 64), device=x.device, dtype=out_dtype)
    stats = None
    if want_stats:
        nt = _L().dram_stem_num_tiles(B, Do, Ho, Wo)
        stats = torch.empty((nt, 2, 64), device=x.device, dtype=torch.float32)
    with _span("stem_fwd_kernel", 2.0 * B * Do * Ho * Wo * 64 * 343):
        sfx = "" if out_dtype != BF16 else ("_bf16mm" if tuning_env("DRAM_STEM_BF16", "1") != "0" else "_bf16")
        _chk(_fn("dram_stem_fwd", sfx)(_p(x), _p(w), _p(y), _p(stats), B, D, H, W, _stream()), "dram_stem_fwd" + sfx)
    return y, stats


def stem_bwd_weight(x: Tensor, dy: Tensor, out: Optional[Tensor] = None) -> Tensor:
    _req(x, "x")
    B, D, H, W = x.shape
    sfx = _act(dy, "dy", (B, stem_out(D), stem_out(H), stem_out(W), 64))
    nbytes = _L().dram_stem_bwd_weight_workspace(B, D, H, W)
    ws = torch.empty(((nbytes + 3) // 4,), device=x.device, dtype=torch.float32)
    dw = out if out is not None else torch.empty((64, 1, 7, 7, 7), device=x.device, dtype=torch.float32)
    _req(dw, "dw", shape=(64, 1, 7, 7, 7))
    with _span("stem_wgrad_kernel+reduce", 2.0 * dy.numel() * 343):
        if sfx and tuning_env("DRAM_STEM_BF16", "1") != "0":
            sfx = "_bf16mm"                          # bf16 matrix cores (the "_bf16" form: fp32 MFMA on the up-cast dy)
        _chk(_fn("dram_stem_bwd_weight", sfx)(_p(x), _p(dy), _p(dw), B, D, H, W, _p(ws), nbytes, _stream()),
             "dram_stem_bwd_weight" + sfx)
    return dw


# --------------------------------------------------------------------------- batch norm
FOLD_TICKET_DOUBLES = 256       # include/dram_hip.h DRAM_FOLD_TICKET_DOUBLES: per-call ticket words behind the stage rows


def reduce_partials(partial: Tensor, tail: Optional[float] = None, want_f32: bool = False):
    """[P,R,C] float32 -> [R,C] float64, ONE launch (dram_fold_partials).  With `tail` (SyncBN: the rank's element
    count) returns (flat [R*C+1] float64 whose last element is tail -- the buffer to all-reduce --, its [R,C] view);
    with want_f32 additionally a list of R float32 [C] tensors (the rows, separately allocated) written by the same
    kernel (appended to the result)."""
    _req(partial, "partial")
    Pn, R, C = partial.shape
    stages = _L().dram_fold_partials_stages(Pn)
    n = R * C + (1 if tail is not None else 0)
    buf = torch.empty((n + stages * R * C + (FOLD_TICKET_DOUBLES if stages > 1 else 0),), device=partial.device,
                      dtype=torch.float64)
    flat = buf[:n]
    # float copy: one tensor PER ROW (whole tensors, which autograd takes over as .grad without a copy; row views of
    # one [R, C] tensor are cloned by AccumulateGrad -- 108 copies per ResNet-50 step)
    f32 = [torch.empty((C,), device=partial.device, dtype=torch.float32) for _ in range(R)] if want_f32 else None
    if want_f32 and R > 2:
        raise ValueError("reduce_partials: the float copy supports at most two rows")
    _chk(_L().dram_fold_partials(_p(partial), _p(flat), _p(buf[n:]), _p(f32[0]) if f32 else None,
                                 _p(f32[1]) if (f32 and R == 2) else None, Pn, R, C,
                                 float(tail) if tail is not None else 0.0, int(tail is not None), _stream()),
         "dram_fold_partials")
    res = (flat, flat[:R * C].view(R, C)) if tail is not None else (flat.view(R, C),)
    if want_f32:
        res = res + (f32,)
    return res if len(res) > 1 else res[0]


def bn_fold_finalize(partial: Tensor, count: float, gamma: Tensor, beta: Tensor, running_mean: Tensor,
                     running_var: Tensor, momentum: float, eps: float):
    """Training-mode statistics of a single-process step in ONE launch: fold of the convolution epilogue's
    [P,2,C] partial sums + bn_finalize (running-statistic update included) -> (mean, invstd, scale, shift)."""
    _req(partial, "partial")
    Pn, R, C = partial.shape
    if R != 2:
        raise ValueError("bn_fold_finalize: expected [P,2,C] partial sums")
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _req(t, nm, shape=(C,))
    stages = _L().dram_fold_partials_stages(Pn)
    buf = torch.empty(((1 + stages) * 2 * C + (FOLD_TICKET_DOUBLES if stages > 1 else 0),), device=partial.device,
                      dtype=torch.float64)
    out = torch.empty((4, C), device=partial.device, dtype=torch.float32)
    _chk(_L().dram_bn_fold_finalize(_p(partial), _p(buf), _p(buf[2 * C:]), Pn, C, float(count), _p(gamma), _p(beta),
                                    _p(running_mean), _p(running_var), float(momentum), float(eps), 1, _p(out[0]),
                                    _p(out[1]), _p(out[2]), _p(out[3]), _stream()), "dram_bn_fold_finalize")
    return out[0], out[1], out[2], out[3]


def bn_finalize(sums: Optional[Tensor], count: float, gamma: Tensor, beta: Tensor, running_mean: Tensor,
                running_var: Tensor, momentum: float, eps: float, update_running: bool,
                count_dev: Optional[Tensor] = None):
    """count_dev: optional 1-element float64 device tensor (the all-reduced global count) read by the
    kernel instead of `count`."""
    C = gamma.numel()
    if count_dev is not None:
        _req(count_dev, "count_dev", dtype=torch.float64, shape=(1,))
    _req(gamma, "gamma", shape=(C,))
    _req(beta, "beta", shape=(C,))
    _req(running_mean, "running_mean", shape=(C,))
    _req(running_var, "running_var", shape=(C,))
    if sums is not None:
        _req(sums, "sums", dtype=torch.float64, shape=(2, C))
    out = torch.empty((4, C), device=gamma.device, dtype=torch.float32)
    mean, invstd, scale, shift = out[0], out[1], out[2], out[3]
    _chk(_L().dram_bn_finalize(_p(sums), float(count), _p(count_dev), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                               float(momentum), float(eps), int(update_running), _p(mean), _p(invstd), _p(scale),
                               _p(shift), C, _stream()), "dram_bn_finalize")
    return mean, invstd, scale, shift


def bn_apply(y: Tensor, scale: Tensor, shift: Tensor, residual: Optional[Tensor], rs: int, relu: bool) -> Tensor:
    sfx = _act(y, "y")
    B, D, H, W, C = y.shape
    _req(scale, "scale", shape=(C,))
    _req(shift, "shift", shape=(C,))
    Dr = Hr = Wr = Cr = 0
    if residual is not None:
        _act(residual, "residual", like=y)
        if residual.shape[0] != B:
            raise ValueError("bn_apply: residual batch mismatch")
        _, Dr, Hr, Wr, Cr = residual.shape
        if Cr > C or (D - 1) * rs >= Dr or (H - 1) * rs >= Hr or (W - 1) * rs >= Wr:
            raise ValueError(f"bn_apply: residual {tuple(residual.shape)} incompatible with {tuple(y.shape)} rs={rs}")
        # F.avg_pool3d(kernel_size=1, stride=rs) output size must equal ours
        if ((Dr - 1) // rs + 1, (Hr - 1) // rs + 1, (Wr - 1) // rs + 1) != (D, H, W):
            raise ValueError("bn_apply: strided residual does not match the output grid")
    z = torch.empty_like(y)
    _chk(_fn("dram_bn_apply", sfx)(_p(y), _p(scale), _p(shift), _p(residual), Dr, Hr, Wr, Cr, rs, _p(z), B, D, H, W, C,
                                   int(relu), _stream()), "dram_bn_apply")
    return z


def _rows(t: Tensor) -> int:
    return t.numel() // t.shape[-1]


def _mask_args(z, y, scale, shift, relu):
    """ReLU mask source of the BN backward kernels: the saved output z, or (z None) scale / shift to
    re-derive it from y."""
    C = y.shape[-1]
    if relu and z is not None:
        _act(z, "z", y.shape, like=y)
    elif relu:
        if scale is None or shift is None:
            raise ValueError("BN backward with ReLU needs z, or scale and shift")
        _req(scale, "scale", shape=(C,))
        _req(shift, "shift", shape=(C,))


def bn_bwd_reduce(dz: Tensor, z: Optional[Tensor], y: Tensor, mean: Tensor, invstd: Tensor, relu: bool,
                  scale: Optional[Tensor] = None, shift: Optional[Tensor] = None) -> Tensor:
    sfx = _act(y, "y")
    _act(dz, "dz", y.shape, like=y)
    _mask_args(z, y, scale, shift, relu)
    C = y.shape[-1]
    _req(mean, "mean", shape=(C,))
    _req(invstd, "invstd", shape=(C,))
    rows = _rows(y)
    nparts = _L().dram_colsum_nparts(rows, C)
    partial = torch.empty((nparts, 2, C), device=y.device, dtype=torch.float32)
    _chk(_fn("dram_bn_bwd_reduce", sfx)(_p(dz), _p(z), _p(y), _p(mean), _p(invstd), _p(scale), _p(shift), _p(partial), rows, C,
                                        int(relu), _stream()), "dram_bn_bwd_reduce")
    return partial


def bn_bwd_apply(dz: Tensor, z: Optional[Tensor], y: Tensor, mean: Tensor, invstd: Tensor, gamma: Tensor, sums: Tensor,
                 count: float, relu: bool, scale: Optional[Tensor] = None, shift: Optional[Tensor] = None,
                 want_colsum: bool = False, count_dev: Optional[Tensor] = None):
    """-> dy, or (dy, partial [P,1,C] column sums of dy) with want_colsum (None when C is unsupported there)."""
    C = y.shape[-1]
    if count_dev is not None:
        _req(count_dev, "count_dev", dtype=torch.float64, shape=(1,))
    sfx = _act(y, "y")
    _act(dz, "dz", y.shape, like=y)
    _mask_args(z, y, scale, shift, relu)
    colpart = None
    if want_colsum:
        npart = _L().dram_bn_bwd_apply_nparts(_rows(y), C)
        if npart >= 1:
            colpart = torch.empty((npart, 1, C), device=y.device, dtype=torch.float32)
    _req(sums, "sums", dtype=torch.float64, shape=(2, C))
    _req(gamma, "gamma", shape=(C,))
    dy = torch.empty_like(y)
    _chk(_fn("dram_bn_bwd_apply", sfx)(_p(dz), _p(z), _p(y), _p(mean), _p(invstd), _p(gamma), _p(scale), _p(shift), _p(sums),
                                       float(count), _p(count_dev), _p(dy), _p(colpart), _rows(y), C, int(relu), _stream()),
         "dram_bn_bwd_apply")
    return (dy, colpart) if want_colsum else dy


def colsum(a: Tensor) -> Tensor:
    """[..., C] -> partial [P,1,C]."""
    sfx = _act(a, "a")
    C = a.shape[-1]
    rows = _rows(a)
    nparts = _L().dram_colsum_nparts(rows, C)
    partial = torch.empty((nparts, 1, C), device=a.device, dtype=torch.float32)
    _chk(_fn("dram_colsum", sfx)(_p(a), _p(partial), rows, C, _stream()), "dram_colsum")
    return partial


def add(a: Tensor, b: Tensor) -> Tensor:
    _req(a, "a")
    _req(b, "b", shape=a.shape)
    out = torch.empty_like(a)
    _chk(_L().dram_add(_p(a), _p(b), _p(out), a.numel(), _stream()), "dram_add")
    return out


# --------------------------------------------------------------------------- pool / up-projection
def pool_out(n: int) -> int:
    return (n + 2 - 3) // 2 + 1


def maxpool_fwd(x: Tensor):
    sfx = _act(x, "x")
    B, D, H, W, C = x.shape
    shape = (B, pool_out(D), pool_out(H), pool_out(W), C)
    y = torch.empty(shape, device=x.device, dtype=x.dtype)
    am = torch.empty(shape, device=x.device, dtype=torch.uint8)
    _chk(_fn("dram_maxpool_fwd", sfx)(_p(x), _p(y), _p(am), B, D, H, W, C, _stream()), "dram_maxpool_fwd")
    return y, am


def bn_maxpool_fwd(y: Tensor, scale: Tensor, shift: Tensor):
    """Stem (med3d.py:272-275): z = relu(y*scale + shift), max-pool 3/2/1 of z and its taps in ONE pass over y --
    bit-identical to bn_apply(...) + maxpool_fwd(...).  -> (z, pooled, argmax)."""
    sfx = _act(y, "y")
    B, D, H, W, C = y.shape
    _req(scale, "scale", shape=(C,))
    _req(shift, "shift", shape=(C,))
    shape = (B, pool_out(D), pool_out(H), pool_out(W), C)
    z = torch.empty_like(y)
    pooled = torch.empty(shape, device=y.device, dtype=y.dtype)
    am = torch.empty(shape, device=y.device, dtype=torch.uint8)
    _chk(_fn("dram_bn_maxpool_fwd", sfx)(_p(y), _p(scale), _p(shift), _p(z), _p(pooled), _p(am), B, D, H, W, C, _stream()),
         "dram_bn_maxpool_fwd")
    return z, pooled, am


def maxpool_bwd(dy: Tensor, argmax: Tensor, in_shape, add_: Optional[Tensor] = None) -> Tensor:
    B, D, H, W, C = in_shape
    oshape = (B, pool_out(D), pool_out(H), pool_out(W), C)
    sfx = _act(dy, "dy", oshape)
    _req(argmax, "argmax", dtype=torch.uint8, shape=oshape)
    add_stride = C
    if add_ is not None:
        # a dense tensor shaped like dx, or a channel slice [..., c0:c0+C] of a dense wider tensor (read in place)
        if (not add_.is_cuda or add_.dtype != dy.dtype or tuple(add_.shape) != tuple(in_shape)
                or add_.stride(-1) != 1):
            raise ValueError("maxpool_bwd: add must be a device tensor of dy's type shaped like the pooled input")
        add_stride = add_.stride(3)
        want = (D * H * W * add_stride, H * W * add_stride, W * add_stride, add_stride, 1)
        if tuple(add_.stride()) != want or add_stride % 4 or add_.data_ptr() % (4 * add_.element_size()):
            raise ValueError("maxpool_bwd: add must be dense or an aligned channel slice of a dense tensor")
    dx = torch.empty(in_shape, device=dy.device, dtype=dy.dtype)
    _chk(_fn("dram_maxpool_bwd", sfx)(_p(dy), _p(argmax), _p(add_), add_stride, _p(dx),
                                      B, D, H, W, C, _stream()), "dram_maxpool_bwd")
    return dx


def upcat_fwd(src: Tensor, skip: Tensor) -> Tensor:
    sfx = _act(src, "src")
    _act(skip, "skip", like=src)
    B, Ds, Hs, Ws, Cu = src.shape
    Bk, Dk, Hk, Wk, Ck = skip.shape
    if Bk != B or Dk < 2 * Ds or Hk < 2 * Hs or Wk < 2 * Ws:
        raise ValueError(f"upcat_fwd: skip {tuple(skip.shape)} smaller than upsampled {tuple(src.shape)}")
    cat = torch.empty((B, 2 * Ds, 2 * Hs, 2 * Ws, Cu + Ck), device=src.device, dtype=src.dtype)
    _chk(_fn("dram_upcat_fwd", sfx)(_p(src), _p(skip), _p(cat), B, Ds, Hs, Ws, Cu, Dk, Hk, Wk, Ck, _stream()),
         "dram_upcat_fwd")
    return cat


def up_fwd(src: Tensor) -> Tensor:
    """x2 trilinear up-sampling (align_corners) alone: the first Cu channels of upcat_fwd(src, skip), bit for bit,
    without the concatenation (the convolution behind it reads the skip tensor as its second source: conv3d_fwd_cat)."""
    sfx = _act(src, "src")
    B, Ds, Hs, Ws, Cu = src.shape
    up = torch.empty((B, 2 * Ds, 2 * Hs, 2 * Ws, Cu), device=src.device, dtype=src.dtype)
    _chk(_fn("dram_upcat_fwd", sfx)(_p(src), None, _p(up), B, Ds, Hs, Ws, Cu, 2 * Ds, 2 * Hs, 2 * Ws, 0, _stream()),
         "dram_upcat_fwd (up only)")
    return up


def conv_cat_ok(g: "ConvGeom", C0: int, dtype=torch.float32) -> bool:
    """Can the convolution take its input as two channel blocks [C0 | g.Cin - C0] (conv3d_fwd_cat)?  The fp32 Winograd
    pipeline with F(4,3)^3 tiles, both blocks multiples of 64; DRAM_CONV_CAT=0 under DRAM_TUNING=1 switches it off (A/B)."""
    return (dtype == torch.float32 and tuning_env("DRAM_CONV_CAT", "1") != "0" and conv_plan(g).prologue
            and C0 >= 64 and C0 % 64 == 0 and (g.Cin - C0) >= 64 and (g.Cin - C0) % 64 == 0)


def conv3d_fwd_cat(x0: Tensor, x1: Tensor, wf: Tensor, bias: Optional[Tensor], g: "ConvGeom", want_stats: bool, keep: bool):
    """Forward convolution of concat([x0, x1], channels) WITHOUT the concatenated tensor (reference med3d.py:87 feeding
    :67): each source's input transform fills its channel range of the Winograd-domain image.  Returns (y, stats, V) like
    conv3d_fwd_keep; bit-identical to it on the concatenation."""
    plan = conv_plan(g)
    C0, C1 = x0.shape[-1], x1.shape[-1]
    if not conv_cat_ok(g, C0, x0.dtype) or C0 + C1 != g.Cin:
        raise RuntimeError(f"conv3d_fwd_cat: not available for {g} with blocks {C0} | {C1}")
    _req(x0, "x0", shape=g.in_shape[:4] + (C0,))
    _req(x1, "x1", shape=g.in_shape[:4] + (C1,))
    _req(wf, "wf", shape=(plan.taps_f, g.Cout, g.Cin))
    if bias is not None:
        _req(bias, "bias", shape=(g.Cout,))
    y = torch.empty(g.out_shape, device=x0.device, dtype=torch.float32)
    stats = torch.empty((plan.stat_rows, 2, g.Cout), device=x0.device, dtype=torch.float32) if want_stats else None
    nbytes = plan.ws_fwd
    ws = _workspace(nbytes, x0.device)
    v = torch.empty((plan.v_elems,), device=x0.device, dtype=torch.float32) if keep else None
    with _span("conv_wino_kernels", g.flops, f"fwd {g}"):
        _chk(_L().dram_wino_conv3d_fwd_cat(_p(x0), C0, _p(x1), C1, _p(wf), _p(bias), _p(y), _p(stats), _p(v), plan.dref, _p(ws),
                                           nbytes, _stream()), f"dram_wino_conv3d_fwd_cat{g}")
    return y, stats, v


def upcat_bwd(dcat: Tensor, src_shape, skip_shape, need_src=True, need_skip=True):
    B, Ds, Hs, Ws, Cu = src_shape
    _, Dk, Hk, Wk, Ck = skip_shape
    sfx = _act(dcat, "dcat", (B, 2 * Ds, 2 * Hs, 2 * Ws, Cu + Ck))
    dsrc = torch.empty(src_shape, device=dcat.device, dtype=dcat.dtype) if need_src else None
    dskip = torch.empty(skip_shape, device=dcat.device, dtype=dcat.dtype) if need_skip else None
    _chk(_fn("dram_upcat_bwd", sfx)(_p(dcat), _p(dsrc), _p(dskip), B, Ds, Hs, Ws, Cu, Dk, Hk, Wk, Ck, _stream()),
         "dram_upcat_bwd")
    return dsrc, dskip


# --------------------------------------------------------------------------- us1.0 without the up-sampled tensor
def upmix_mode() -> int:
    """0 = never, 1 = where it pays (default), 2 = wherever the geometry allows (tests): DRAM_UPMIX under DRAM_TUNING=1."""
    return int(tuning_env("DRAM_UPMIX", "1"))


def upmix_split_weight(w: Tensor, Cu: int) -> Tuple[Tensor, Tensor]:
    """w [Co, Cu+Cs, 3,3,3] -> (wlo [27*Co, Cu, 1,1,1]: the low-resolution mixing GEMM's weight, row t*Co + co;
    ws [Co, Cs, 3,3,3]: the skip channels' convolution weight).  csrc/upmix.hip."""
    _req(w, "w")
    Co, Ct = w.shape[0], w.shape[1]
    Cs = Ct - Cu
    if tuple(w.shape[2:]) != (3, 3, 3) or Cu < 1 or Cs < 1:
        raise ValueError(f"upmix_split_weight: weight {tuple(w.shape)}, Cu={Cu}")
    wlo = torch.empty((27 * Co, Cu, 1, 1, 1), device=w.device, dtype=torch.float32)
    ws = torch.empty((Co, Cs, 3, 3, 3), device=w.device, dtype=torch.float32)
    _chk(_L().dram_upmix_split_weight(_p(w), _p(wlo), _p(ws), Co, Cu, Cs, _stream()), "dram_upmix_split_weight")
    return wlo, ws


def upmix_merge_wgrad(dwlo: Tensor, dws: Tensor, out: Optional[Tensor] = None) -> Tensor:
    Co, Cs = dws.shape[0], dws.shape[1]
    Cu = dwlo.shape[1]
    _req(dwlo, "dwlo", shape=(27 * Co, Cu, 1, 1, 1))
    _req(dws, "dws", shape=(Co, Cs, 3, 3, 3))
    dw = out if out is not None else torch.empty((Co, Cu + Cs, 3, 3, 3), device=dws.device, dtype=torch.float32)
    _req(dw, "dw", shape=(Co, Cu + Cs, 3, 3, 3))
    _chk(_L().dram_upmix_merge_wgrad(_p(dwlo), _p(dws), _p(dw), Co, Cu, Cs, _stream()), "dram_upmix_merge_wgrad")
    return dw


def upmix_gather_fwd(b: Tensor, base: Optional[Tensor], Co: int, want_stats: bool, out_dtype=None):
    """b [B,Ds,Hs,Ws,27*Co] (tap-major: column t*Co + co) -> y [B,2Ds,2Hs,2Ws,Co] = sum over the 27 taps of the
    trilinearly up-sampled tap images, shifted by the tap (zero outside the volume), + base (written IN PLACE into
    base when given).  Three separable passes; intermediates fp32.  Returns (y, BatchNorm partial sums or None)."""
    bf = _act(b, "b") != ""
    B, Ds, Hs, Ws, N = b.shape
    if N != 27 * Co:
        raise ValueError(f"upmix_gather_fwd: {N} columns for Co={Co}")
    dev = b.device
    if base is not None:
        _act(base, "base", (B, 2 * Ds, 2 * Hs, 2 * Ws, Co))
        y = base
    else:
        y = torch.empty((B, 2 * Ds, 2 * Hs, 2 * Ws, Co), device=dev, dtype=out_dtype or b.dtype)
    L = _L()
    with _span("upmix_gather", 0.0, f"fwd {tuple(b.shape)}"):
        q1 = torch.empty((B, Ds, Hs, 2 * Ws, 9, Co), device=dev, dtype=torch.float32)
        _chk(L.dram_upmix_axis_fwd(_p(b), int(bf), _p(q1), B * Ds * Hs, Ws, 9, Co, _stream()), "dram_upmix_axis_fwd[x]")
        q2 = torch.empty((B, Ds, 2 * Hs, 2 * Ws, 3, Co), device=dev, dtype=torch.float32)
        _chk(L.dram_upmix_axis_fwd(_p(q1), 0, _p(q2), B * Ds, Hs, 2 * Ws * 3, Co, _stream()), "dram_upmix_axis_fwd[y]")
        del q1
        nvox = B * 8 * Ds * Hs * Ws
        stats = torch.empty((L.dram_upmix_stat_rows(nvox), 2, Co), device=dev, dtype=torch.float32) if want_stats else None
        _chk(L.dram_upmix_axis_fwd_final(_p(q2), _p(base), _p(y), int(y.dtype == BF16), _p(stats), B, Ds, 4 * Hs * Ws, Co,
                                         _stream()), "dram_upmix_axis_fwd_final")
    return y, stats


def upmix_gather_bwd(dy: Tensor, out_dtype=None) -> Tensor:
    """Transpose of upmix_gather_fwd: dy [B,2Ds,2Hs,2Ws,Co] -> h [B,Ds,Hs,Ws,27*Co] (storage type of dy unless given)."""
    gbf = _act(dy, "dy") != ""
    B, Do, Ho, Wo, Co = dy.shape
    if (Do | Ho | Wo) & 1:
        raise ValueError(f"upmix_gather_bwd: odd extents {tuple(dy.shape)}")
    Ds, Hs, Ws = Do // 2, Ho // 2, Wo // 2
    dev = dy.device
    odt = out_dtype or dy.dtype
    L = _L()
    with _span("upmix_gather", 0.0, f"bwd {tuple(dy.shape)}"):
        g2 = torch.empty((B, Ds, Ho, Wo, 3, Co), device=dev, dtype=torch.float32)
        _chk(L.dram_upmix_axis_bwd(_p(dy), int(gbf), _p(g2), 0, B, Ds, Ho * Wo, Co, _stream()), "dram_upmix_axis_bwd[z]")
        g1 = torch.empty((B, Ds, Hs, Wo, 9, Co), device=dev, dtype=torch.float32)
        _chk(L.dram_upmix_axis_bwd(_p(g2), 0, _p(g1), 0, B * Ds, Hs, Wo * 3, Co, _stream()), "dram_upmix_axis_bwd[y]")
        del g2
        h = torch.empty((B, Ds, Hs, Ws, 27 * Co), device=dev, dtype=odt)
        _chk(L.dram_upmix_axis_bwd(_p(g1), 0, _p(h), int(odt == BF16), B * Ds * Hs, Ws, 9, Co, _stream()),
             "dram_upmix_axis_bwd[x]")
    return h


def upproject(dense: Tensor, ess: Tensor, size):
    """dense [B,D,H,W] -> out [B,Do,Ho,Wo] = trilinear(align_corners)(dense)*ess, partial [B,nblk]."""
    _req(dense, "dense")
    B, D, H, W = dense.shape
    Do, Ho, Wo = size
    _req(ess, "ess", shape=(B, Do, Ho, Wo))
    nblk = _L().dram_upproject_nblk(Do * Ho * Wo)
    out = torch.empty((B, Do, Ho, Wo), device=dense.device, dtype=torch.float32)
    partial = torch.empty((B, nblk), device=dense.device, dtype=torch.float32)
    _chk(_L().dram_upproject(_p(dense), _p(ess), _p(out), _p(partial), B, D, H, W, Do, Ho, Wo, _stream()),
         "dram_upproject")
    return out, partial


# --------------------------------------------------------------------------- heads / losses
def head_fwd(x: Tensor, w: Tensor, bias: Tensor, lungs: Optional[Tensor], sigmoid: bool):
    """x [B,D,H,W,32] (float32 or bfloat16); w [NO,32]; lungs None or [B,Dl,Hl,Wl] full-res mask.  The dense maps
    and the pooling partial sums are float32 on both storage paths."""
    sfx = _act(x, "x")
    B, D, H, W, C = x.shape
    if C != 32:
        raise ValueError("head_fwd: the head consumes the 32-channel us3 output")
    NO = w.shape[0]
    _req(w, "w", shape=(NO, 32))
    _req(bias, "bias", shape=(NO,))
    Dl = Hl = Wl = 0
    if lungs is not None:
        _req(lungs, "lungs")
        if lungs.dim() != 4 or lungs.shape[0] != B:
            raise ValueError("head_fwd: lungs must be [B,Dl,Hl,Wl]")
        _, Dl, Hl, Wl = lungs.shape
    nblk = _L().dram_head_nblk(D * H * W)
    dense = torch.empty((B, NO, D, H, W), device=x.device, dtype=torch.float32)
    partial = torch.empty((B, nblk, NO + 1), device=x.device, dtype=torch.float32)
    _chk(_fn("dram_head_fwd", sfx)(_p(x), _p(w), _p(bias), _p(lungs), Dl, Hl, Wl, _p(dense), _p(partial), B, D, H, W, NO,
                                   int(sigmoid), _stream()), "dram_head_fwd")
    return dense, partial


def head_bwd(x: Tensor, w: Tensor, dense: Optional[Tensor], gdense: Optional[Tensor], gpool: Tensor,
             lungs: Optional[Tensor], sigmoid: bool):
    sfx = _act(x, "x")
    B, D, H, W, C = x.shape
    NO = w.shape[0]
    _req(w, "w", shape=(NO, 32))
    _req(gpool, "gpool", shape=(B, NO))
    if dense is not None:
        _req(dense, "dense", shape=(B, NO, D, H, W))
    if gdense is not None:
        _req(gdense, "gdense", shape=(B, NO, D, H, W))
    Dl = Hl = Wl = 0
    if lungs is not None:
        _req(lungs, "lungs")
        _, Dl, Hl, Wl = lungs.shape
    nparts = _L().dram_head_bwd_nparts(D * H * W)
    dx = torch.empty_like(x)
    wpartial = torch.empty((B * nparts, NO, 33), device=x.device, dtype=torch.float32)
    _chk(_fn("dram_head_bwd", sfx)(_p(x), _p(w), _p(dense), _p(gdense), _p(gpool), _p(lungs), Dl, Hl, Wl, _p(dx),
                                   _p(wpartial), B, D, H, W, NO, int(sigmoid), _stream()), "dram_head_bwd")
    return dx, wpartial


def segloss_fwd(cle: Tensor, pse: Tensor, lungs: Tensor, ems: Tensor, binary: Tensor,
                smoothness: float = 0.85) -> Tensor:
    """cle/pse [B,D,H,W]; lungs/ems [B,Dl,Hl,Wl]; binary [B] -> partial [nblk,6].  smoothness: the in-mask
    weight of metrics.py:24 (0.85 at the models.py:529 call site)."""
    _req(cle, "cle")
    B, D, H, W = cle.shape
    _req(pse, "pse", shape=cle.shape)
    _req(lungs, "lungs")
    _req(ems, "ems", shape=lungs.shape)
    _req(binary, "binary", shape=(B,))
    _, Dl, Hl, Wl = lungs.shape
    nblk = _L().dram_segloss_nblk(B * D * H * W)
    partial = torch.empty((nblk, 6), device=cle.device, dtype=torch.float32)
    _chk(_L().dram_segloss_fwd(_p(cle), _p(pse), _p(lungs), _p(ems), _p(binary), Dl, Hl, Wl, _p(partial), B, D, H, W,
                               float(smoothness), _stream()), "dram_segloss_fwd")
    return partial


def segloss_bwd(cle: Tensor, pse: Tensor, lungs: Tensor, ems: Tensor, binary: Tensor, coef: Tensor,
                smoothness: float = 0.85):
    B, D, H, W = cle.shape
    _req(coef, "coef", shape=(8,))
    _, Dl, Hl, Wl = lungs.shape
    gcle = torch.empty_like(cle)
    gpse = torch.empty_like(pse)
    _chk(_L().dram_segloss_bwd(_p(cle), _p(pse), _p(lungs), _p(ems), _p(binary), Dl, Hl, Wl, _p(coef), _p(gcle),
                               _p(gpse), B, D, H, W, float(smoothness), _stream()), "dram_segloss_bwd")
    return gcle, gpse


def regloss_tail(partial: Tensor, reg_cle: Tensor, reg_pse: Tensor, cle_labels: Tensor, pse_labels: Tensor,
                 cle_w: Tensor, pse_w: Tensor, cle_bands: Tensor, pse_bands: Tensor, voxels_total: int,
                 smooth: float, beta: float, gamma: float):
    """The O(B) tail of models.py:549-574 in one launch (csrc/head_loss.hip regloss_tail_kernel) ->
    out [5] (loss, loss_cle, loss_pse, mul, seg), coef [8] for segloss_bwd, greg [2,B] = d loss / d reg_outs."""
    _req(partial, "partial")
    B = reg_cle.shape[0]
    for t, n in ((reg_cle, "reg_cle"), (reg_pse, "reg_pse"), (cle_w, "cle_w"), (pse_w, "pse_w")):
        _req(t, n, shape=(B,))
    for t, n in ((cle_labels, "cle_labels"), (pse_labels, "pse_labels")):
        _req(t, n, dtype=torch.int64, shape=(B,))
    _req(cle_bands, "cle_bands")
    _req(pse_bands, "pse_bands")
    dev = partial.device
    out = torch.empty(5, device=dev, dtype=torch.float32)
    coef = torch.empty(8, device=dev, dtype=torch.float32)
    greg = torch.empty((2, B), device=dev, dtype=torch.float32)
    _chk(_L().dram_regloss_tail(_p(partial), partial.shape[0], _p(reg_cle), _p(reg_pse), _p(cle_labels),
                                _p(pse_labels), _p(cle_w), _p(pse_w), _p(cle_bands), cle_bands.shape[0],
                                _p(pse_bands), pse_bands.shape[0], B, float(voxels_total), float(smooth), float(beta),
                                float(gamma), _p(out), _p(coef), _p(greg), _stream()), "dram_regloss_tail")
    return out, coef, greg


# --------------------------------------------------------------------------- optimizer
def adam_multi(table: Tensor, chunks: Tensor, nchunks: int, lr, b1, b2, eps, wd, bc1, bc2, grad_scale):
    _chk(_L().dram_adam_multi(_p(table), _p(chunks), nchunks, lr, b1, b2, eps, wd, bc1, bc2, grad_scale, _stream()),
         "dram_adam_multi")


def adam_multi_dev(table: Tensor, chunks: Tensor, nchunks: int, hyper: Tensor):
    """hyper: float32[7] device tensor {lr, b1, b2, eps, wd, grad_scale, step} (graph-replayable form)."""
    _req(hyper, "hyper", shape=(7,))
    _chk(_L().dram_adam_multi_dev(_p(table), _p(chunks), nchunks, _p(hyper), _stream()), "dram_adam_multi_dev")


def sgd_multi(table: Tensor, chunks: Tensor, nchunks: int, lr, momentum, wd, first_step, grad_scale):
    _chk(_L().dram_sgd_multi(_p(table), _p(chunks), nchunks, lr, momentum, wd, int(first_step), grad_scale,
                             _stream()), "dram_sgd_multi")
