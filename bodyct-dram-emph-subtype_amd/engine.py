"""Forward / backward executor of the Med3D-ResNet + dRAM decoder on libdram_hip.

This is the host-side "tape": it walks the network of reference med3d.py:270-285 /
:369-388 layer by layer, launching the hand-written kernels (ops.py) on NDHWC buffers,
and walks it back for the gradients.  Autograd only sees the whole network as one
Function (med3d.py of this package); PyTorch provides memory, streams and collectives.

Fusion boundaries
  conv  -> pre-BN output y + per-tile BN statistics (conv epilogue)
  BN-apply + residual (identity or detached shortcut-A) + ReLU -> z   (one pass)
  backward: BN-reduce (1 pass), BN-apply (1 pass), wgrad, dgrad with the identity-
  shortcut gradient `dz*(z>0)` folded into the dgrad epilogue.
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional

import torch

from . import ops
from .ops import ConvGeom

Tensor = torch.Tensor

ARCHS = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
}
EXPANSION = {"basic": 1, "bottleneck": 4}
STAGES = ((64, 1, 1), (128, 2, 1), (256, 1, 2), (512, 1, 4))  # planes, stride, dilation (med3d.py:207-213)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class _State:
    """Per-call state: parameter/buffer dicts, flags, saved contexts, gradient sink."""

    def __init__(self, P: Dict[str, Tensor], training: bool, need_grad: bool, dist=None, recompute: bool = False,
                 storage=torch.float32):
        self.P = P
        self.storage = storage        # storage type of the activations: float32, or bfloat16 (`--precision bf16`)
        self.training = training
        self.need_grad = need_grad
        self.dist = dist
        self.recompute = bool(recompute) and need_grad
        self.grads: Dict[str, Tensor] = {}
        self.nbt: List[Tensor] = []
        self.deferred = None          # (ctx, dy) of the unit whose weight gradient is still to be launched
        self.side = None              # second HIP stream for the weight-gradient kernels (set by backward)

        self.convs: List[tuple] = []  # (weight name, geometry) in call order, recorded for the pre-pack of later steps
        self.packed: Dict[str, tuple] = {}   # weight name -> (wf, wb) packed ahead on the side stream
        self.packed_ready = None      # event: every entry of `packed` is complete


class Engine:
    def __init__(self, net: str, head: str):
        if net not in ARCHS:
            raise NotImplementedError(net)
        if head not in ("cls", "reg"):
            raise NotImplementedError(head)
        self.net, self.head = net, head
        self.kind, self.layers = ARCHS[net]
        self.e = EXPANSION[self.kind]
        # plan hint (include/dram_hip.h DRAM_CONV_ROUNDING_TOLERANT): the BasicBlock networks hand a layer's rounding
        # error on with little amplification, so the cheaper convolution estimate may win on every layer
        self.tol = 1 if self.kind == "basic" else 0
        self._conv_lists: Dict[tuple, list] = {}     # input shape -> [(weight name, geometry)] of a forward
        self._inflight: List[torch.cuda.Event] = []  # end-of-backward events of the steps the host has issued
        self.throttle_wait_s = 0.0                   # host time spent waiting in _throttle (bench.py reports it)
        self.graph_streams = 1                       # branches of a hipGraph capture of the step (graph.GraphedTrainStep)

    # ------------------------------------------------------------------ BN helpers
    def _bn_fwd(self, st: _State, y: Tensor, sp: Optional[Tensor], bnp: str, residual, rs, relu=True, pool=False,
                apply=True):
        """pool: the stem's form -- BatchNorm-apply + ReLU + max-pool in one pass; returns (z, pooled, taps) first.
        apply=False: statistics / scale / shift only, z = None (the consumer applies them on its way in)."""
        P = st.P
        count = float(y.numel() // y.shape[-1])
        count_dev = None
        if st.training:
            if st.dist is not None and st.dist.sync_bn:
                # SyncBN (train.py:101): [sum, sum^2, count] summed over ranks in ONE in-place all-reduce;
                # the global count never leaves the device
                flat, sums = ops.reduce_partials(sp, tail=count)
                st.dist.all_reduce_stats(flat)
                count_dev = flat[-1:]
                mean, invstd, scale, shift = ops.bn_finalize(sums, count, P[bnp + ".weight"], P[bnp + ".bias"],
                                                             P[bnp + ".running_mean"], P[bnp + ".running_var"],
                                                             BN_MOMENTUM, BN_EPS, True, count_dev)
            else:                                       # one launch: fold of the epilogue's partial sums + finalize
                mean, invstd, scale, shift = ops.bn_fold_finalize(sp, count, P[bnp + ".weight"], P[bnp + ".bias"],
                                                                  P[bnp + ".running_mean"], P[bnp + ".running_var"],
                                                                  BN_MOMENTUM, BN_EPS)
            st.nbt.append(P[bnp + ".num_batches_tracked"])      # incremented together at the end of forward
        else:
            mean, invstd, scale, shift = ops.bn_finalize(None, 1.0, P[bnp + ".weight"], P[bnp + ".bias"],
                                                         P[bnp + ".running_mean"], P[bnp + ".running_var"],
                                                         BN_MOMENTUM, BN_EPS, False)
        if not apply:
            z = None
        elif pool:
            z = ops.bn_maxpool_fwd(y, scale, shift)       # (z, pooled, taps)
        else:
            z = ops.bn_apply(y, scale, shift, residual, rs, relu)
        # without a residual the backward pass re-derives the ReLU mask from y (scale, shift) instead of reading z
        return z, mean, invstd, (count, count_dev), (None if residual is not None else (scale, shift))

    def _bn_bwd(self, st: _State, c: dict, dz: Tensor) -> Tensor:
        bnp = c["bn"]
        zmask, (sc, sh) = (c["z"], (None, None)) if c.get("ss") is None else (None, c["ss"])
        part = c.pop("part", None)      # taken by the data gradient that produced dz (_conv_bn_bwd, nxt), or a pass here
        if part is None:
            part = ops.bn_bwd_reduce(dz, zmask, c["y"], c["mean"], c["invstd"], True, sc, sh)
        # parameter gradients use the LOCAL sums (DDP averages them afterwards), the input
        # gradient the all-reduced ones -- torch SyncBatchNorm semantics.
        sums, sf = ops.reduce_partials(part, want_f32=True)     # (the float copy: the two parameter gradients, its rows)
        st.grads[bnp + ".weight"] = sf[1]
        st.grads[bnp + ".bias"] = sf[0]
        count, count_dev = c["count"]
        if st.dist is not None and st.dist.sync_bn:
            # C3: the sums travel while the previous unit's weight-gradient kernel runs
            work = st.dist.all_reduce_stats_async(sums)
            self._flush_wgrad(st)
            work.wait()
        if c.get("b"):    # the convolution in front has a bias: its gradient = column sums of dy, taken on the way
            dy, colpart = ops.bn_bwd_apply(dz, zmask, c["y"], c["mean"], c["invstd"], st.P[bnp + ".weight"], sums,
                                           count, True, sc, sh, want_colsum=True, count_dev=count_dev)
            if colpart is None:
                colpart = ops.colsum(dy)
            st.grads[c["b"]] = ops.reduce_partials(colpart)[0].float()
            return dy
        return ops.bn_bwd_apply(dz, zmask, c["y"], c["mean"], c["invstd"], st.P[bnp + ".weight"], sums, count,
                                True, sc, sh, count_dev=count_dev)

    # ------------------------------------------------------------------ conv + BN unit
    def _conv_bn_fwd(self, st: _State, x, wname: str, bname: Optional[str], bnp: str, k: int, stride: int,
                     pad: int, dil: int, residual=None, rs: int = 1, xr=None, defer_z: bool = False):
        """xr: how to RE-DERIVE x in backward (activation-recompute mode): ('bn', ctx of the unit whose BN+ReLU output
        x is) or ('upcat', src, skip).  With st.recompute the context then holds neither x nor the cached
        Winograd-domain image of x, and a unit without residual does not hold its z (its consumer re-derives it).
        x may be a DEFERRED input {'y', 'ss'}: the pre-BatchNorm output of the producing unit whose BatchNorm-apply + ReLU
        was left to this consumer (reference med3d.py:121-124: bn, relu, conv).  Where the convolution can apply them
        on its way in (the fp32 Winograd pipeline, ops.conv_prologue_ok) the activation tensor in between is never
        written; elsewhere it is materialised here, as before.  defer_z: leave THIS unit's BatchNorm-apply to its (only)
        consumer the same way -- the first return value is then such a deferred input."""
        w = st.P[wname]
        pro = None
        split = None
        if isinstance(x, tuple):                        # (x0, x1): the input as two channel blocks, never concatenated
            split = x
            x = None
            B, D, H, W, _ = split[0].shape
            Cin = split[0].shape[4] + split[1].shape[4]
        elif isinstance(x, dict):
            B, D, H, W, Cin = x["y"].shape
            g = ConvGeom(B, D, H, W, Cin, w.shape[0], k, stride, pad, dil, self.tol)
            if k == 3 and ops.conv_prologue_ok(g, x["y"].dtype):
                pro, x = x["ss"], x["y"]
            else:
                x = ops.bn_apply(x["y"], x["ss"][0], x["ss"][1], None, 1, True)
        if split is None:
            B, D, H, W, Cin = x.shape
        g = ConvGeom(B, D, H, W, Cin, w.shape[0], k, stride, pad, dil, self.tol)
        st.convs.append((wname, g))
        if st.need_grad:
            pre = st.packed.pop(wname, None)
            if pre is not None and pre[2] == g:
                if st.packed_ready is not None:         # first consumer: order the caller's stream after the pre-pack
                    torch.cuda.current_stream().wait_event(st.packed_ready)
                    st.packed_ready = None
                wf, wb = pre[0], pre[1]
            else:
                wf, wb = ops.pack_conv_weight(w, True, True, g, st.storage)
        else:                                           # inference: packed / transformed once per weight version
            wf, wb = ops.packed_forward_weight(w, g, st.storage), None
        if split is not None:
            y, sp, v = ops.conv3d_fwd_cat(split[0], split[1], wf, st.P[bname] if bname else None, g, st.training,
                                          st.need_grad and not st.recompute)
        else:
            y, sp, v = ops.conv3d_fwd_keep(x, wf, st.P[bname] if bname else None, g, st.training,
                                           st.need_grad and not st.recompute, prologue=pro)
        defer_z = defer_z and residual is None
        z, mean, invstd, count, ss = self._bn_fwd(st, y, sp, bnp, residual, rs, apply=not defer_z)
        c = None
        if st.need_grad:
            if (pro is not None or split is not None) and xr is None:
                raise RuntimeError("a fused BatchNorm prologue / a split input needs the recipe of the input (xr)")
            # (pro: x is the producer's y, not the input; split: the concatenated input was never built)
            drop_x = (st.recompute and xr is not None) or pro is not None or split is not None
            c = dict(x=None if drop_x else x, xr=xr if drop_x else None, y=y,
                     z=None if (st.recompute and residual is None) else z, mean=mean, invstd=invstd, g=g, wb=wb,
                     count=count, w=wname, b=bname, bn=bnp, v=v, ss=ss)
        return (dict(y=y, ss=ss) if defer_z else z), c

    def _conv_bn_bwd(self, st: _State, c: dict, dz: Tensor, need_dx=True, add=None, gate=None, nxt: Optional[dict] = None):
        """nxt: the context of the unit whose BatchNorm + ReLU output is this convolution's (only) input -- the returned
        gradient is that unit's dz.  Where the data gradient can take that unit's backward statistics on its way out
        (ops.conv_bwd_bnstats_ok) they are left in nxt['part'] and its _bn_bwd skips the pass over dz and y."""
        dy = self._bn_bwd(st, c, dz)
        if st.side is not None:
            # second stream: concurrent with the data-gradient chain (and, data parallel, with the host-visible
            # wait for the next unit's statistic all-reduce: the GPU keeps executing this kernel meanwhile)
            self._wgrad_side(st, c, dy)
        elif st.dist is not None:
            # data parallel on ONE stream: this unit's weight gradient is launched inside the NEXT unit's statistic
            # all-reduce (see _bn_bwd), hiding that latency-bound collective behind a long kernel
            st.deferred = (c, dy)
        else:
            self._wgrad(st, c, dy)
        if not need_dx:
            return None
        if (nxt is not None and add is None and gate is None and nxt.get("ss") is not None
                and ops.conv_bwd_bnstats_ok(c["g"], dy.dtype)):
            dx, nxt["part"] = ops.conv3d_bwd_data_bnstats(dy, c["wb"], c["g"], nxt["y"], nxt["mean"], nxt["invstd"],
                                                          *nxt["ss"], overlapped=st.side is not None)
            return dx
        # (overlapped: this data gradient shares the device with the weight-gradient kernels of the second stream)
        return ops.conv3d_bwd_data(dy, c["wb"], c["g"], add, gate, overlapped=st.side is not None)

    @staticmethod
    def _recipe_inputs(c: dict):
        """tensors the re-derivation of c's input reads"""
        xr = c.get("xr")
        if xr is None:
            return ()
        if xr[0] == "bn":
            return (xr[1]["y"],) + tuple(xr[1]["ss"])
        return (xr[1], xr[2])

    def _input_of(self, c: dict) -> Tensor:
        """The unit's input: saved, or (activation recompute) re-derived with the kernels that produced it in the
        forward pass -- bit-identical to the tensor the forward convolution read."""
        if c["x"] is not None:
            return c["x"]
        xr = c["xr"]
        if xr[0] == "bn":
            sc, sh = xr[1]["ss"]
            return ops.bn_apply(xr[1]["y"], sc, sh, None, 1, True)
        return ops.upcat_fwd(xr[1], xr[2])

    def _wgrad(self, st: _State, c: dict, dy: Tensor):
        out = st.dist.grad_out(c["w"]) if st.dist is not None else None
        if c.get("kind") == "upmix":
            # low-resolution mixing GEMM's weight gradient (taps as output channels) + the skip channels' 3x3x3 one,
            # merged into the reference layout [Co][Cu+Cs][27]
            dwlo = ops.conv3d_bwd_weight(c["src"], c["h"], c["g_lo"])
            dws = ops.conv3d_bwd_weight(c["skip"], dy, c["g_s"])
            st.grads[c["w"]] = ops.upmix_merge_wgrad(dwlo, dws, out=out)
            c["h"] = None
        else:
            v = c.get("v")
            # (a unit that applied its producer's BatchNorm on the way in holds no input tensor: the pipeline's weight
            # gradient takes the forward's transformed input instead; anything else re-derives the input)
            # (ops.conv3d_bwd_weight's own predicate: a cached image, the pipeline's plan AND its workspace -- a geometry
            # whose TN plan fails reports no workspace and runs the direct weight gradient, which needs the input)
            plan = ops.conv_plan(c["g"])
            from_v = v is not None and plan.walgo == 1 and plan.ws_wgrad > 0
            xin = c["x"] if (c["x"] is not None or from_v) else self._input_of(c)
            st.grads[c["w"]] = ops.conv3d_bwd_weight(xin, dy, c["g"], out=out, v_cache=v)
            c["v"] = None                               # release the cached Winograd-domain input
        if st.dist is not None:
            names = [c["w"], c["bn"] + ".weight", c["bn"] + ".bias"] + ([c["b"]] if c["b"] else [])
            st.dist.grads_ready(st.grads, names)

    def _wgrad_side(self, st: _State, c: dict, dy: Tensor):
        """Weight gradient on the engine's second stream, concurrent with the data-gradient chain that continues
        on the caller's stream: the two are independent (both only read dy), and the chain's BN-backward /
        Winograd-transform / pooling kernels are HBM-bound while the weight-gradient GEMMs are matrix-bound, so
        co-resident workgroups of the two streams use different pipes of a CU."""
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        side = st.side
        arena = st.dist.grad_out(c["w"]) if st.dist is not None else None     # written by the side stream's kernel
        for t in (c.get("x"), dy, c.get("v"), arena, c.get("src"), c.get("skip"), c.get("h")) + self._recipe_inputs(c):
            if t is not None:
                t.record_stream(side)         # the allocator must not recycle them under the side stream's kernels
        with ops.on_stream(side):
            side.wait_event(ready)
            # data parallel: grads_ready() inside _wgrad launches the arena all-reduce from THIS stream context,
            # so the collective is ordered after the weight-gradient kernels that fill the range
            self._wgrad(st, c, dy)
        # allocated from the side stream's pool, consumed (autograd accumulation, optimizer, zero_grad) on the caller's
        st.grads[c["w"]].record_stream(main)

    def _flush_wgrad(self, st: _State):
        if st.deferred is not None:
            (c, dy), st.deferred = st.deferred, None
            self._wgrad(st, c, dy)

    # ------------------------------------------------------------------ residual blocks
    def _block_fwd(self, st, x, p, planes, stride, dil, has_ds):
        e = self.e
        if self.kind == "basic":
            # (z1 has ONE consumer, conv2: its BatchNorm-apply + ReLU is deferred to that convolution's way in)
            z1, c1 = self._conv_bn_fwd(st, x, p + ".conv1.weight", None, p + ".bn1", 3, stride, dil, dil, defer_z=True)
            z2, c2 = self._conv_bn_fwd(st, z1, p + ".conv2.weight", None, p + ".bn2", 3, 1, dil, dil,
                                       residual=x, rs=stride if has_ds else 1, xr=("bn", c1))
            return z2, (c1, c2, has_ds)
        z1, c1 = self._conv_bn_fwd(st, x, p + ".conv1.weight", None, p + ".bn1", 1, 1, 0, 1, defer_z=True)
        z2, c2 = self._conv_bn_fwd(st, z1, p + ".conv2.weight", None, p + ".bn2", 3, stride, dil, dil, xr=("bn", c1))
        z3, c3 = self._conv_bn_fwd(st, z2, p + ".conv3.weight", None, p + ".bn3", 1, 1, 0, 1,
                                   residual=x, rs=stride if has_ds else 1, xr=("bn", c2))
        del e
        return z3, (c1, c2, c3, has_ds)

    def _block_bwd(self, st, ctx, dz_out, extra_add=None, need_dx=True):
        """dz_out: gradient w.r.t. the block output (post-ReLU).  Identity shortcut:
        dx = dgrad(conv1) + dz_out*(z_out>0); shortcut A is detached (med3d.py:110): no term."""
        *cs, has_ds = ctx
        last = cs[-1]
        d = dz_out
        for i in range(len(cs) - 1, 0, -1):
            d = self._conv_bn_bwd(st, cs[i], d, nxt=cs[i - 1])
        if has_ds:
            return self._conv_bn_bwd(st, cs[0], d, need_dx=need_dx, add=extra_add, gate=None)
        if extra_add is not None:
            raise NotImplementedError("extra gradient into an identity-shortcut block")
        return self._conv_bn_bwd(st, cs[0], d, need_dx=need_dx, add=dz_out, gate=last["z"])

    # ------------------------------------------------------------------ decoder block
    def _upmix_geoms(self, src, skip, w):
        """(low-resolution mixing geometry, skip-convolution geometry) when the first decoder convolution can run
        without the up-sampled tensor (csrc/upmix.hip), else None: no crop (skip is exactly twice the source), channel
        counts the GEMM / convolution kernels take, and -- unless forced -- an up-sampled operand wide enough to pay
        (the gather costs 28.5 FMAs + ~13 elements of traffic per output element whatever Cu; ResNet-18: 512, -50: 2 048)."""
        mode = ops.upmix_mode()
        B, Ds, Hs, Ws, Cu = src.shape
        Co, Cs = w.shape[0], skip.shape[4]
        if mode == 0 or tuple(skip.shape[1:4]) != (2 * Ds, 2 * Hs, 2 * Ws) or w.shape[1] != Cu + Cs:
            return None
        if Cu % 64 or Cs % 32 or Co % 32 or Co > 128:
            return None
        if mode == 1 and (Cu < 256 or (B * Ds * Hs * Ws) % 256 or B * Ds * Hs * Ws < 2048):
            return None                                 # the mixing GEMM wants whole 256-row tiles and a filled chip
        g_lo = ConvGeom(B, Ds, Hs, Ws, Cu, 27 * Co, 1, 1, 0, 1, self.tol)
        g_s = ConvGeom(B, 2 * Ds, 2 * Hs, 2 * Ws, Cs, Co, 3, 1, 1, 1, self.tol)
        return g_lo, g_s

    def _upmix_fwd(self, st, src, skip, wname, bname, bnp, geoms):
        """conv_blocks[0] of a decoder block on (src, skip) directly: y = gather(W_lo . src) + conv_s(skip) + bias."""
        g_lo, g_s = geoms
        w = st.P[wname]
        Cu, Co = src.shape[4], w.shape[0]
        wlo, ws = ops.upmix_split_weight(w, Cu)
        wf_lo, wb_lo = ops.pack_conv_weight(wlo, True, st.need_grad, g_lo, st.storage)
        wf_s, wb_s = ops.pack_conv_weight(ws, True, st.need_grad, g_s, st.storage)
        b, _ = ops.conv3d_fwd(src, wf_lo, None, g_lo, False)
        ysk, _ = ops.conv3d_fwd(skip, wf_s, st.P[bname], g_s, False)
        y, sp = ops.upmix_gather_fwd(b, ysk, Co, st.training)
        del b
        z, mean, invstd, count, ss = self._bn_fwd(st, y, sp, bnp, None, 1, apply=False)     # left to conv_blocks[1]
        c = None
        if st.need_grad:
            c = dict(kind="upmix", src=src, skip=skip, y=y, z=None, mean=mean, invstd=invstd,
                     g_lo=g_lo, g_s=g_s, wb_lo=wb_lo, wb_s=wb_s, count=count, w=wname, b=bname, bn=bnp, ss=ss, h=None)
        return dict(y=y, ss=ss), c

    def _upmix_bwd(self, st, c, dz):
        """-> (gradient w.r.t. the low-resolution source, gradient w.r.t. the skip tensor)"""
        dy = self._bn_bwd(st, c, dz)
        c["h"] = ops.upmix_gather_bwd(dy)               # read by the weight gradient (side stream) and the data gradient
        h = c["h"]
        if st.side is not None:
            self._wgrad_side(st, c, dy)
        elif st.dist is not None:
            st.deferred = (c, dy)
        else:
            self._wgrad(st, c, dy)
        dsrc = ops.conv3d_bwd_data(h, c["wb_lo"], c["g_lo"])
        dskip = ops.conv3d_bwd_data(dy, c["wb_s"], c["g_s"])
        return dsrc, dskip

    def _up_fwd(self, st, src, skip, p):
        geoms = self._upmix_geoms(src, skip, st.P[f"{p}.conv_blocks.0.0.weight"])
        if geoms is not None:
            za, ca = self._upmix_fwd(st, src, skip, f"{p}.conv_blocks.0.0.weight", f"{p}.conv_blocks.0.0.bias",
                                     f"{p}.conv_blocks.0.1", geoms)
            zb, cb = self._conv_bn_fwd(st, za, f"{p}.conv_blocks.1.0.weight", f"{p}.conv_blocks.1.0.bias",
                                       f"{p}.conv_blocks.1.1", 3, 1, 1, 1, xr=("bn", ca))
            if st.recompute and cb is not None:
                cb["z"] = zb
            return zb, (ca, cb, tuple(src.shape), tuple(skip.shape))
        # the up-sampled + concatenated input of conv_blocks[0] (med3d.py:86-87) is NOT built where the convolution can
        # take two sources (fp32 pipeline, F(4,3)^3 tiles, no crop): only the up-sampled half is written, the skip tensor
        # is read in place -- half the pass, 0.5 GB less on config 1's us2.  Elsewhere: the materialised concatenation.
        B, Ds, Hs, Ws, Cu = src.shape
        wc = st.P[f"{p}.conv_blocks.0.0.weight"]
        gc = ConvGeom(B, 2 * Ds, 2 * Hs, 2 * Ws, wc.shape[1], wc.shape[0], 3, 1, 1, 1, self.tol)
        if (tuple(skip.shape[1:4]) == (2 * Ds, 2 * Hs, 2 * Ws) and wc.shape[1] == Cu + skip.shape[4]
                and ops.conv_cat_ok(gc, Cu, src.dtype)):
            cat = (ops.up_fwd(src), skip)
        else:
            cat = ops.upcat_fwd(src, skip)
        za, ca = self._conv_bn_fwd(st, cat, f"{p}.conv_blocks.0.0.weight", f"{p}.conv_blocks.0.0.bias",
                                   f"{p}.conv_blocks.0.1", 3, 1, 1, 1, xr=("upcat", src, skip), defer_z=True)
        del cat
        zb, cb = self._conv_bn_fwd(st, za, f"{p}.conv_blocks.1.0.weight", f"{p}.conv_blocks.1.0.bias",
                                   f"{p}.conv_blocks.1.1", 3, 1, 1, 1, xr=("bn", ca))
        if st.recompute and cb is not None:
            cb["z"] = zb                                # a block output: the next stage reads it (kept)
        return zb, (ca, cb, tuple(src.shape), tuple(skip.shape))

    def _up_bwd(self, st, ctx, dz, skip_view=False):
        ca, cb, src_shape, skip_shape = ctx
        dza = self._conv_bn_bwd(st, cb, dz, nxt=ca)
        if ca.get("kind") == "upmix":
            return self._upmix_bwd(st, ca, dza)
        dcat = self._conv_bn_bwd(st, ca, dza)
        if skip_view and tuple(skip_shape[1:4]) == tuple(dcat.shape[1:4]):
            # no crop: the consumer reads the skip half of dcat in place (a channel-slice view) instead of a copy
            dsrc, _ = ops.upcat_bwd(dcat, src_shape, skip_shape, True, False)
            return dsrc, dcat[..., src_shape[4]:]
        return ops.upcat_bwd(dcat, src_shape, skip_shape, True, True)

    # ------------------------------------------------------------------ whole network
    def forward(self, P: Dict[str, Tensor], x: Tensor, lungs: Optional[Tensor], training: bool, need_grad: bool,
                dist=None, recompute: bool = False, storage=torch.float32):
        """x [B,1,D,H,W] (NCDHW == NDHW for C=1), lungs None or [B,1,D,H,W] float.
        Returns (dense_list, outs_list, saved-or-None).  Every kernel is launched on x's device (its
        current stream); operands on any other device are rejected before launch.
        storage: float32 (the reference's default arithmetic) or bfloat16 -- activations and saved tensors in bf16,
        products of bf16 operands accumulated in fp32, statistics / parameters / weight gradients / dense head
        outputs in fp32 (the reference under `--precision bf16`, train.py:46)."""
        if storage not in (torch.float32, torch.bfloat16):
            raise NotImplementedError(f"activation storage type {storage}")
        with ops.launch_scope(x.device):
            if need_grad:
                self._throttle()
            return self._forward(P, x, lungs, training, need_grad, dist, recompute, storage)

    def backward(self, saved: dict, g_dense: List[Optional[Tensor]], g_outs: List[Optional[Tensor]]):
        with ops.launch_scope(saved["dense"].device):
            out = self._backward(saved, g_dense, g_outs)
            if not torch.cuda.is_current_stream_capturing():
                ev = torch.cuda.Event()
                ev.record()
                self._inflight.append(ev)
            return out

    def _throttle(self):
        """The host issues an eager step in a quarter of the time the GPU needs for it.  Left alone (no .item() in the
        training loop) it runs many steps ahead, and every tensor that crossed to the second stream (record_stream)
        stays unavailable to the caching allocator until the GPU has caught up: the pool grows step after step and the
        hipMalloc / hipFree traffic made config 1 swing between 39 and 69 ms per step.  So: at most ONE step issued
        ahead of the one the GPU is executing -- the forward of step k + 1 waits (on the host) for the backward of step
        k - 1.  The GPU never idles for it; DRAM_INFLIGHT (DRAM_TUNING=1) changes the depth, 0 = as if the loop read
        the loss every step."""
        if torch.cuda.is_current_stream_capturing():
            return
        depth = int(ops.tuning_env("DRAM_INFLIGHT", "1"))
        if len(self._inflight) > depth:
            t0 = time.perf_counter()
            while len(self._inflight) > depth:
                self._inflight.pop(0).synchronize()
            self.throttle_wait_s += time.perf_counter() - t0

    def _two_streams(self) -> bool:
        """Weight-gradient kernels and the per-step weight packing on the engine's second stream?  Eager steps: yes.
        While the step is being captured into a hipGraph: when the capture asks for it (`graph_streams` = 2, set by
        graph.GraphedTrainStep(streams=2); DRAM_GRAPH_STREAMS=2 under DRAM_TUNING=1 forces it) -- the side stream forks from
        the capturing stream by wait_stream / event and rejoins it before the step ends, so the graph has two branches.
        Measured: it buys nothing -- config 1 37.11 / 37.24 / 37.38 ms with one branch, 37.16 / 37.23 /
        37.07 with two (alternating runs on one box; 3.7 GB more), config 0 8.58 vs 8.81-8.85, config 2 in fp32 37.61 vs 37.93;
        the packing alone as a branch: config 0 8.52 vs 8.9, ResNet-50 bf16 15.41 vs 15.74.  Captured steps default to ONE
        branch; what the capture is for is a step time that does not depend on the host (a cold host issues the eager step
        in 23.7 instead of 6.9 ms and leaves gaps: 42.3 instead of 37.0 ms)."""
        if ops.tuning_env("DRAM_WGRAD_STREAM", "1") == "0":
            return False
        if not torch.cuda.is_current_stream_capturing():
            return True
        return self.graph_streams == 2 or ops.tuning_env("DRAM_GRAPH_STREAMS", "1") == "2"

    def _prepack(self, st: _State, key):
        """Training steps repack / re-transform every convolution weight (21 launches for ResNet-18, independent
        of the activations).  From the second step of an input shape on they all run on the engine's second
        stream at the start of forward, under the stem convolution, instead of in front of each layer."""
        plan = self._conv_lists.get(key)
        if plan is None:
            return
        # bf16 storage: every plain bf16 packing (3x3x3 stride 1, 1x1x1) in ONE launch -- also on a single stream and
        # inside a hipGraph capture (54 launches of ~10 us per ResNet-50 step otherwise)
        multi = [(wname, g) for wname, g in plan if st.storage == torch.bfloat16 and ops.conv_plan(g).bf16]
        if not self._two_streams():
            if multi:
                packed = ops.pack_conv_weights_bf16_multi([st.P[wname] for wname, _ in multi])
                if packed is not None:
                    for (wname, g), (wf, wb) in zip(multi, packed):
                        st.packed[wname] = (wf, wb, g)
            return
        side = ops.side_stream(key[-1])
        main = torch.cuda.current_stream()
        side.wait_stream(main)                         # the optimizer's update of the weights precedes the packing
        with ops.on_stream(side):
            done = set()
            if multi:
                packed = ops.pack_conv_weights_bf16_multi([st.P[wname] for wname, _ in multi])
                if packed is not None:
                    packed[0][0].record_stream(main)   # (views of ONE flat buffer)
                    for (wname, g), (wf, wb) in zip(multi, packed):
                        st.packed[wname] = (wf, wb, g)
                        done.add(wname)
            for wname, g in plan:
                if wname in done:
                    continue
                wf, wb = ops.pack_conv_weight(st.P[wname], True, True, g, st.storage)
                wf.record_stream(main)
                wb.record_stream(main)
                st.packed[wname] = (wf, wb, g)
            st.packed_ready = torch.cuda.Event()
            st.packed_ready.record(side)

    def _forward(self, P, x, lungs, training, need_grad, dist, recompute=False, storage=torch.float32):
        if need_grad and not training:
            raise NotImplementedError("gradients through eval-mode BatchNorm are not part of the hot path")
        st = _State(P, training, need_grad, dist, recompute, storage)
        B, _, D, H, W = x.shape
        shape_key = (tuple(x.shape), storage, x.device.index)
        if need_grad:
            self._prepack(st, shape_key)
        x4 = x.reshape(B, D, H, W)
        lungs4 = None if lungs is None else lungs.reshape(B, *lungs.shape[-3:]).contiguous()

        y0, sp0 = ops.stem_fwd(x4, P["conv1.weight"], training, storage)
        # stem BatchNorm-apply + ReLU + max-pool: one pass (K9, med3d.py:272-275)
        (xs, xp, amax), mean0, invstd0, count0, ss0 = self._bn_fwd(st, y0, sp0, "bn1", None, 1, pool=True)

        h = xp
        inplanes = 64
        block_ctx = []
        feats = []
        for li, ((planes, stride, dil), nblk) in enumerate(zip(STAGES, self.layers)):
            for bi in range(nblk):
                s = stride if bi == 0 else 1
                has_ds = bi == 0 and (stride != 1 or inplanes != planes * self.e)
                h, c = self._block_fwd(st, h, f"layer{li + 1}.{bi}", planes, s, dil, has_ds)
                block_ctx.append(c)
                inplanes = planes * self.e
            feats.append(h)
        x1, x4f = feats[0], feats[3]
        xup1, cu1 = self._up_fwd(st, x4f, x1, "us1")
        xup2, cu2 = self._up_fwd(st, xup1, xs, "us2")
        xup3, cu3 = self._conv_bn_fwd(st, xup2, "us3.0.weight", "us3.0.bias", "us3.1", 3, 1, 1, 1)

        n0, n1 = P["fcs.0.weight"].shape[0], P["fcs.1.weight"].shape[0]
        NO = n0 + n1
        hw = torch.cat([P["fcs.0.weight"].reshape(n0, 32), P["fcs.1.weight"].reshape(n1, 32)], 0).contiguous()
        hb = torch.cat([P["fcs.0.bias"], P["fcs.1.bias"]], 0).contiguous()
        sig = self.head == "reg"
        dense, partial = ops.head_fwd(xup3, hw, hb, lungs4 if sig else None, sig)
        sums = partial.sum(1)  # [B, NO+1]   (tiny: glue)
        denom = sums[:, NO]
        if sig:
            outs = [sums[:, 0] / denom, sums[:, 1] / denom]
        else:
            pooled = sums[:, :NO] / denom[:, None]
            outs = [pooled[:, :n0].contiguous(), pooled[:, n0:].contiguous()]
        dense_list = [dense[:, :n0], dense[:, n0:]]
        saved = None
        if need_grad:
            saved = dict(st=st, x4=x4, y0=y0, xs=xs, mean0=mean0, invstd0=invstd0, count0=count0, ss0=ss0, amax=amax,
                         blocks=block_ctx, cu1=cu1, cu2=cu2, cu3=cu3, xup3=xup3, hw=hw, dense=dense, lungs4=lungs4,
                         denom=denom, n0=n0, n1=n1, xs_shape=tuple(xs.shape))
        if need_grad:
            self._conv_lists[shape_key] = list(st.convs)
            st.packed.clear()
        if st.nbt:
            torch._foreach_add_(st.nbt, 1)               # one launch for all num_batches_tracked counters
        return dense_list, outs, saved

    def _backward(self, saved, g_dense, g_outs):
        st: _State = saved["st"]
        if st.dist is not None:
            st.dist.begin_backward(saved["dense"].device)
        if self._two_streams():
            st.side = ops.side_stream(saved["dense"].device.index)

        n0, n1 = saved["n0"], saved["n1"]
        NO = n0 + n1
        dense = saved["dense"]
        B = dense.shape[0]
        sig = self.head == "reg"
        dev = dense.device
        # pooled-score gradient coefficients
        gpool = torch.zeros((B, NO), device=dev, dtype=torch.float32)
        denom = saved["denom"]
        if sig:
            if g_outs[0] is not None:
                gpool[:, 0] = g_outs[0] / denom
            if g_outs[1] is not None:
                gpool[:, 1] = g_outs[1] / denom
        else:
            if g_outs[0] is not None:
                gpool[:, :n0] = g_outs[0] / denom[:, None]
            if g_outs[1] is not None:
                gpool[:, n0:] = g_outs[1] / denom[:, None]
        gd = None
        if g_dense[0] is not None or g_dense[1] is not None:
            gd = torch.zeros_like(dense)
            if g_dense[0] is not None:
                gd[:, :n0] = g_dense[0]
            if g_dense[1] is not None:
                gd[:, n0:] = g_dense[1]
        dxup3, wpart = ops.head_bwd(saved["xup3"], saved["hw"], dense if sig else None, gd, gpool,
                                    saved["lungs4"] if sig else None, sig)
        wg = ops.reduce_partials(wpart.reshape(wpart.shape[0], 1, NO * 33)).reshape(NO, 33).float()
        st.grads["fcs.0.weight"] = wg[:n0, :32].reshape(n0, 32, 1, 1, 1).contiguous()
        st.grads["fcs.0.bias"] = wg[:n0, 32].contiguous()
        st.grads["fcs.1.weight"] = wg[n0:, :32].reshape(n1, 32, 1, 1, 1).contiguous()
        st.grads["fcs.1.bias"] = wg[n0:, 32].contiguous()
        if st.dist is not None:
            st.dist.grads_ready(st.grads, ["fcs.0.weight", "fcs.0.bias", "fcs.1.weight", "fcs.1.bias"])

        dxup2 = self._conv_bn_bwd(st, saved["cu3"], dxup3, nxt=saved["cu2"][1])
        dxup1, dskip_stem = self._up_bwd(st, saved["cu2"], dxup2, skip_view=True)   # consumed by maxpool_bwd
        d, dskip_x1 = self._up_bwd(st, saved["cu1"], dxup1)

        # walk the residual stages backwards; x1 (end of layer1) also feeds the us1 skip
        blocks = saved["blocks"]
        idx = len(blocks)
        for li in (3, 2, 1, 0):
            nblk = self.layers[li]
            for bi in reversed(range(nblk)):
                idx -= 1
                extra = dskip_x1 if (li == 1 and bi == 0) else None
                d = self._block_bwd(st, blocks[idx], d, extra_add=extra)
        self._flush_wgrad(st)
        # d = gradient w.r.t. the max-pooled stem output
        dxs = ops.maxpool_bwd(d, saved["amax"], saved["xs_shape"], dskip_stem)
        c0 = dict(z=saved["xs"], y=saved["y0"], mean=saved["mean0"], invstd=saved["invstd0"], count=saved["count0"],
                  bn="bn1", ss=saved["ss0"])
        dy0 = self._bn_bwd(st, c0, dxs)
        st.grads["conv1.weight"] = ops.stem_bwd_weight(saved["x4"], dy0,
                                                       out=st.dist.grad_out("conv1.weight") if st.dist else None)
        if st.side is not None:
            torch.cuda.current_stream().wait_stream(st.side)      # every weight gradient is final from here on
        if st.dist is not None:
            st.dist.grads_ready(st.grads, ["conv1.weight", "bn1.weight", "bn1.bias"])
            st.dist.finish(st.grads)
        return st.grads


def forward_decisions(saved: dict) -> Dict[str, Tensor]:
    """The piecewise-linear decisions one training forward took, from its saved state: for every
    ReLU the on/off mask (bool, NCDHW, keyed by the state_dict prefix of the BatchNorm in front of it)
    and under 'maxpool' the window tap each max-pool output took (uint8 (kz*3+ky)*3+kx, NCDHW).
    Inspection / verification API: lets a checker evaluate a reference backward on exactly the linear
    piece this forward ran on (tests/test_network_gpu.py), so a ReLU input within rounding of zero
    cannot hide -- or fake -- a gradient error."""
    def mask(z):
        return (z > 0).permute(0, 4, 1, 2, 3).contiguous()

    out = {"bn1": mask(saved["xs"]), "maxpool": saved["amax"].permute(0, 4, 1, 2, 3).contiguous()}
    units = []
    for blk in saved["blocks"]:
        units.extend(blk[:-1])
    for cu in (saved["cu1"], saved["cu2"]):
        units.extend(cu[:2])
    units.append(saved["cu3"])
    for c in units:
        if c["z"] is not None:
            out[c["bn"]] = mask(c["z"])
        else:
            # z was not kept (activation recompute) or never written (its BatchNorm-apply ran as the consumer's
            # prologue): re-derived with the kernel the engine itself uses -- the SAME fma, bit for bit (a torch
            # expression may round the product before the add and flip a tie)
            sc, sh = c["ss"]
            with ops.launch_scope(c["y"].device):
                out[c["bn"]] = mask(ops.bn_apply(c["y"], sc, sh, None, 1, True))
    return out
