"""CPU oracle for the Med3D-ResNet + dRAM train-step hot path.

TEST INFRASTRUCTURE ONLY.  This file is a from-scratch CPU restatement (plain
``torch`` CPU ops, functional style, no ``nn.Module``) of the algorithm the
reference runs on its hot path.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the
checker.  The product path (``bodyct-dram-emph-subtype_amd``) never imports it
and fails loudly when the HIP library is missing.

Parity pinning: the reference ships no tests/golden vectors, so
this oracle is pinned by fixtures generated in the build container by importing
the reference itself (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` checks every function below against them.

Each function cites the reference file:line it restates (paths relative to the
reference repo root).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ---------------------------------------------------------------------------
# architecture table (med3d.py:391-425: factories -> block type + layer counts)
# ---------------------------------------------------------------------------
ARCHS = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
}
EXPANSION = {"basic": 1, "bottleneck": 4}
# (planes, stride, dilation) per stage -- med3d.py:207-213 / :306-312
STAGES = ((64, 1, 1), (128, 2, 1), (256, 1, 2), (512, 1, 4))
BN_EPS = 1e-5        # nn.BatchNorm3d default (med3d.py:12)
BN_MOMENTUM = 0.1


def split_arch(name: str) -> Tuple[str, str]:
    """'resnet18segreg' -> ('resnet18', 'reg')."""
    assert name.startswith("resnet") and name[-6:-3] == "seg", name
    return name[:-6], name[-3:]


# ---------------------------------------------------------------------------
# primitive ops
# ---------------------------------------------------------------------------
def batch_norm(x: Tensor, sd: Dict[str, Tensor], prefix: str, train: bool,
               new_stats: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """nn.BatchNorm3d (med3d.py:121,124,153,156,158,203,227).

    train: biased batch variance for normalisation; running stats updated with
    momentum 0.1 using the UNBIASED variance.  eval: running stats.
    """
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if not train:
        return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)
    # torch's own batch-norm op (what nn.BatchNorm3d calls); running stats go to copies
    rm2, rv2 = rm.detach().clone(), rv.detach().clone()
    out = F.batch_norm(x, rm2, rv2, w, b, True, BN_MOMENTUM, BN_EPS)
    if new_stats is not None:
        new_stats[prefix + ".running_mean"] = rm2
        new_stats[prefix + ".running_var"] = rv2
        new_stats[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    return out


def batch_norm_explicit(x: Tensor, w: Tensor, b: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """The same training-mode arithmetic spelled out (used by kernel-level tests):
    returns (out, batch mean, biased batch variance)."""
    dims = [0, 2, 3, 4]
    mean = x.mean(dims)
    var = x.var(dims, unbiased=False)
    shape = (1, -1, 1, 1, 1)
    xhat = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS)
    return xhat * w.view(shape) + b.view(shape), mean, var


def shortcut_a(x: Tensor, planes: int, stride: int) -> Tensor:
    """downsample_basic_block (med3d.py:103-112): strided subsample, zero channel
    pad, and -- crucially -- DETACHED from autograd (``Variable(... out.data ...)``)."""
    out = x.detach()[:, :, ::stride, ::stride, ::stride]
    pad = planes - out.shape[1]
    if pad > 0:
        out = torch.cat([out, out.new_zeros(out.shape[0], pad, *out.shape[2:])], dim=1)
    return out


def crop_concat(t1: Tensor, t2: Tensor) -> Tensor:
    """crop_concat_5d (med3d.py:39-48): centre-crop t2 to t1's DHW, cat [t1, t2]."""
    sl = [slice(None), slice(None)]
    for a, b in zip(t1.shape[2:], t2.shape[2:]):
        o = int(math.ceil((b - a) / 2))
        sl.append(slice(o, a + o))
    return torch.cat([t1, t2[tuple(sl)]], dim=1)


def upsample2_trilinear(x: Tensor) -> Tensor:
    """nn.Upsample(scale_factor=2, 'trilinear', align_corners=True) (med3d.py:83)."""
    return F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=True)


# ---------------------------------------------------------------------------
# storage rounding (test device: the reference under Lightning's `--precision bf16`, train.py:46, with the rounding
# points of the build's bf16 STORAGE path -- DESIGN.md section 4b -- instead of torch autocast's)
# ---------------------------------------------------------------------------
_STORAGE = None          # None, or the dtype activations / convolution weights are rounded to where the build stores them


class _RoundST(torch.autograd.Function):
    """Round to the storage type and back; the gradient passes straight through (the build's backward treats a stored
    activation as exact)."""

    @staticmethod
    def forward(ctx, t, dtype):
        return t.to(dtype).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


def q(t: Tensor) -> Tensor:
    """Identity, or -- inside forward(..., storage=torch.bfloat16) -- one rounding to the storage type: applied to the
    network input, to every convolution weight (the packed bf16 operand), to every convolution output after its bias
    (the stored pre-BatchNorm tensor; batch statistics are taken from the ROUNDED values), to every BatchNorm + residual
    + ReLU output and to the up-sampled tensor.  Everything else (statistics, affine parameters, heads) stays exact."""
    return t if _STORAGE is None else _RoundST.apply(t, _STORAGE)


# ---------------------------------------------------------------------------
# pinned decisions (test device: removes ReLU / max-pool tie-flips from gradient comparisons)
# ---------------------------------------------------------------------------
def relu(x: Tensor, pins: Optional[Dict[str, Tensor]] = None, key: str = "") -> Tensor:
    """nn.ReLU.  With ``pins`` the on/off decision of every element is FORCED to ``pins[key]``
    (bool, same shape) instead of being re-derived from ``x > 0``: the piecewise-linear network is
    evaluated on the linear piece another implementation chose, so an input within rounding of zero
    cannot land on different sides in the two implementations and the gradient comparison measures
    arithmetic error only.  ``key`` is the state_dict prefix of the BatchNorm in front of the ReLU."""
    if pins is None:
        return F.relu(x)
    if pins.get("__record__"):          # recording run: plain ReLU, decisions written into `pins`
        pins[key] = x.detach() > 0
        if pins["__record__"] == "pre":  # ... and the values the decisions were taken on (decision audits)
            pins["pre:" + key] = x.detach()
        return F.relu(x)
    return x * pins[key].to(x.dtype)


def max_pool3(x: Tensor, pins: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """nn.MaxPool3d(3, 2, 1) (med3d.py:206,275).  With ``pins`` the window element taken is
    ``pins['maxpool']`` (tap index (kz*3+ky)*3+kx into the window starting at 2*o-1, the same
    encoding ATen's scan order induces) instead of the arg-max."""
    if pins is None:
        return F.max_pool3d(x, 3, 2, 1)
    if pins.get("__record__"):
        out, idx = F.max_pool3d(x, 3, 2, 1, return_indices=True)
        D, H, W = x.shape[-3:]
        zo, yo, xo = torch.meshgrid(*[torch.arange(n) for n in out.shape[-3:]], indexing="ij")
        kz, ky, kx = idx // (H * W) - (2 * zo - 1), (idx // W) % H - (2 * yo - 1), idx % W - (2 * xo - 1)
        pins["maxpool"] = ((kz * 3 + ky) * 3 + kx).to(torch.uint8)
        return out
    am = pins["maxpool"]
    Do, Ho, Wo = am.shape[-3:]
    xp = F.pad(x, (1, 1, 1, 1, 1, 1))
    out = 0.0
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                tap = (kz * 3 + ky) * 3 + kx
                sl = xp[:, :, kz:kz + 2 * Do:2, ky:ky + 2 * Ho:2, kx:kx + 2 * Wo:2]
                out = out + sl * (am == tap).to(x.dtype)
    return out


# ---------------------------------------------------------------------------
# blocks
# ---------------------------------------------------------------------------
def basic_block(x, sd, p, planes, stride, dil, has_ds, train, ns, pins=None):
    """BasicBlock.forward (med3d.py:129-144)."""
    out = q(F.conv3d(x, q(sd[p + ".conv1.weight"]), None, stride, dil, dil))
    out = q(relu(batch_norm(out, sd, p + ".bn1", train, ns), pins, p + ".bn1"))
    out = q(F.conv3d(out, q(sd[p + ".conv2.weight"]), None, 1, dil, dil))
    out = batch_norm(out, sd, p + ".bn2", train, ns)
    res = shortcut_a(x, planes, stride) if has_ds else x
    return q(relu(out + res, pins, p + ".bn2"))


def bottleneck(x, sd, p, planes, stride, dil, has_ds, train, ns, pins=None):
    """Bottleneck.forward (med3d.py:164-184)."""
    out = q(F.conv3d(x, q(sd[p + ".conv1.weight"])))
    out = q(relu(batch_norm(out, sd, p + ".bn1", train, ns), pins, p + ".bn1"))
    out = q(F.conv3d(out, q(sd[p + ".conv2.weight"]), None, stride, dil, dil))
    out = q(relu(batch_norm(out, sd, p + ".bn2", train, ns), pins, p + ".bn2"))
    out = q(F.conv3d(out, q(sd[p + ".conv3.weight"])))
    out = batch_norm(out, sd, p + ".bn3", train, ns)
    res = shortcut_a(x, planes * 4, stride) if has_ds else x
    return q(relu(out + res, pins, p + ".bn3"))


def up_block(inputs, cats, sd, p, nconv, train, ns, pins=None):
    """UpsampleConvBlock5d.forward (med3d.py:85-89): up2 -> crop_concat (upsampled
    channels FIRST) -> nconv x (conv3+bias -> BN -> ReLU)."""
    x = crop_concat(q(upsample2_trilinear(inputs)), cats)
    for i in range(nconv):
        cb = f"{p}.conv_blocks.{i}"
        x = q(F.conv3d(x, q(sd[cb + ".0.weight"]), sd[cb + ".0.bias"], 1, 1))
        x = q(relu(batch_norm(x, sd, cb + ".1", train, ns), pins, cb + ".1"))
    return x


# ---------------------------------------------------------------------------
# whole network
# ---------------------------------------------------------------------------
def forward(sd: Dict[str, Tensor], x: Tensor, lungs: Optional[Tensor], arch: str,
            train: bool = True, new_stats: Optional[Dict[str, Tensor]] = None,
            taps: Optional[Dict[str, Tensor]] = None, pins: Optional[Dict[str, Tensor]] = None,
            storage: Optional[torch.dtype] = None):
    """ResNetSegCls.forward (med3d.py:270-285) / ResNetSegReg.forward (:369-388).

    ``storage`` (optional, e.g. torch.bfloat16): evaluate the SAME graph with activations and convolution weights
    rounded once to that type wherever the build's bf16 storage path stores them (see ``q``), in whatever precision
    ``sd`` / ``x`` carry -- the yardstick of the bf16 path that does not inherit bf16's own conditioning.

    ``sd``: state_dict-keyed tensors (reference key names).  Returns
    (dense_outs, outs) exactly like the reference.  ``taps`` (optional dict)
    receives named intermediate activations for layer-level tests.  ``pins`` (optional) forces
    every ReLU / max-pool decision (see ``relu`` / ``max_pool3``).
    """
    global _STORAGE
    prev, _STORAGE = _STORAGE, storage
    try:
        return _forward(sd, x, lungs, arch, train, new_stats, taps, pins)
    finally:
        _STORAGE = prev


def _forward(sd, x, lungs, arch, train, new_stats, taps, pins):
    net, head = split_arch(arch)
    kind, layers = ARCHS[net]
    e = EXPANSION[kind]
    blk = basic_block if kind == "basic" else bottleneck
    B = x.shape[0]
    ns = new_stats

    x = q(F.conv3d(q(x), q(sd["conv1.weight"]), None, 2, 3))            # :272 / :371
    x = q(relu(batch_norm(x, sd, "bn1", train, ns), pins, "bn1"))       # :273-274
    xp = max_pool3(x, pins)                                             # :275
    feats = []
    h = xp
    inplanes = 64
    for li, ((planes, stride, dil), nblk) in enumerate(zip(STAGES, layers)):
        for bi in range(nblk):
            s = stride if bi == 0 else 1
            has_ds = bi == 0 and (stride != 1 or inplanes != planes * e)  # :244
            h = blk(h, sd, f"layer{li + 1}.{bi}", planes, s, dil, has_ds, train, ns, pins)
            inplanes = planes * e
        feats.append(h)
    x1, x4 = feats[0], feats[3]
    xup1 = up_block(x4, x1, sd, "us1", 2, train, ns, pins)              # :280 / :379
    xup2 = up_block(xup1, x, sd, "us2", 2, train, ns, pins)             # :281 / :380 (skip = post-ReLU stem)
    xup3 = q(F.conv3d(xup2, q(sd["us3.0.weight"]), sd["us3.0.bias"], 1, 1))   # :282 / :381
    xup3 = q(relu(batch_norm(xup3, sd, "us3.1", train, ns), pins, "us3.1"))
    if taps is not None:
        taps.update(stem=x, xp=xp, x1=feats[0], x2=feats[1], x3=feats[2], x4=x4,
                    xup1=xup1, xup2=xup2, xup3=xup3)
    nheads = 2
    if head == "cls":
        dense = [F.conv3d(xup3, sd[f"fcs.{i}.weight"], sd[f"fcs.{i}.bias"]) for i in range(nheads)]
        outs = [d.mean(dim=(2, 3, 4)).view(B, -1) for d in dense]       # :284 adaptive_avg_pool3d(.,1)
        return dense, outs
    dense = [torch.sigmoid(F.conv3d(xup3, sd[f"fcs.{i}.weight"], sd[f"fcs.{i}.bias"]))
             for i in range(nheads)]                                     # :382
    if lungs is None:
        lg = torch.ones_like(x)                                          # :383-384 (64-ch; broadcast == mean)
    else:
        lg = F.interpolate(lungs, xup3.shape[-3:], mode="nearest")       # :386
    outs = [(d * lg).view(B, -1).sum(-1) / lg.view(B, -1).sum(-1) for d in dense]  # :387
    return dense, outs


# ---------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------
# dataset.py:99-112
CLE_RATIO_MAP = {0: (0.0, 0.01), 1: (0.01, 0.05), 2: (0.05, 0.1), 3: (0.1, 0.2), 4: (0.2, 0.3), 5: (0.3, 1.0001)}
PSE_RATIO_MAP = {0: (0.0, 0.01), 1: (0.01, 0.05), 2: (0.05, 1.0001)}
BETA, GAMMA = 0.7338, 0.2578          # models.py:414-415


def regression_labels(cls_targets: Sequence[int], ratio_map, tightness: float = 1.0) -> Tensor:
    """_generate_regression_labels (models.py:495-510)."""
    bands = []
    for c in cls_targets:
        lb, ub = ratio_map[int(c)]
        if lb < 1e-7:
            bands.append((0.0, 0.0))
        else:
            m = (lb + ub) / 2.0
            span = (ub - lb) * tightness / 2.0
            bands.append((m - span, m + span))
    return torch.tensor(bands, dtype=torch.float32)


def interval_regression_loss(outs: Tensor, reg_targets: Tensor, weights: Tensor) -> Tensor:
    """_interval_regression_loss (models.py:512-521)."""
    n = torch.cat([outs.unsqueeze(1), reg_targets], dim=1)
    n = BETA * n ** GAMMA
    K = (0.5 * (n[:, 2] - n[:, 1])) ** 2
    unh = (n[:, 0] - (n[:, 2] + n[:, 1]) / 2.0) ** 2 - K
    return (10.0 * F.relu(unh) * weights).sum()


def dice_coef(y: Tensor, y_hat: Tensor, smooth: float = 1e-7) -> Tensor:
    """dice_coef (metrics.py:33-37), smooth=1e-7 per models.py:412."""
    inter = (y_hat.reshape(-1) * y.reshape(-1)).sum()
    return (2.0 * inter + smooth) / (y.sum() + y_hat.sum() + smooth)


def balanced_bce(y: Tensor, y_hat: Tensor, mask: Optional[Tensor], smoothness: float = 0.65,
                 eps: float = 1e-6) -> Tensor:
    """BinaryCrossEntropy.__call__ (metrics.py:10-30)."""
    t = y.float()
    p = y_hat
    alpha = (1.0 - t.sum() / t.shape[0]).clamp(0.3, 0.7)
    pt = p * t + (1.0 - p) * (1.0 - t)
    w = alpha * t + (1.0 - alpha) * (1.0 - t)
    ptc = pt.clamp(eps, 1.0 - eps)
    if mask is not None:
        nll = -1.0 * (smoothness * torch.log(ptc) * w * mask + torch.log(ptc) * w * (1.0 - mask))
    else:
        nll = -smoothness * torch.log(ptc) * w
    return nll.sum() / w.sum()


def segmentation_loss(dense_cle: Tensor, dense_pse: Tensor, ems: Tensor, lungs: Tensor):
    """_segmentation_loss (models.py:523-531)."""
    mul = dice_coef(dense_cle * lungs, dense_pse * lungs)
    p = torch.clamp(dense_cle + dense_pse, 0.0, 1.0)
    seg = balanced_bce(ems, p, lungs, smoothness=0.85)
    return mul, seg


def reg_train_loss(dense_outs, reg_outs, lungs, ems, cle_labels, pse_labels,
                   cle_w: Tensor, pse_w: Tensor):
    """Train branch of ScanRegLightningModule.shared_step (models.py:549-574).

    lungs/ems: f32 [B,1,D,H,W]; labels int64 [B]; cle_w/pse_w: per-SAMPLE class
    weights (models.py:556-561).  Returns (loss, parts dict).
    """
    B = lungs.shape[0]
    cle_t = regression_labels(cle_labels.tolist(), CLE_RATIO_MAP)
    pse_t = regression_labels(pse_labels.tolist(), PSE_RATIO_MAP)
    loss_cle = interval_regression_loss(reg_outs[0], cle_t, cle_w)
    loss_pse = interval_regression_loss(reg_outs[1], pse_t, pse_w)
    binary = torch.logical_or(cle_labels > 0, pse_labels > 0).long()
    size = dense_outs[0].shape[-3:]
    seg_labels = F.interpolate(ems * binary.float().view(B, 1, 1, 1, 1), size, mode="nearest").detach()
    lung_labels = F.interpolate(lungs, size=size, mode="nearest")
    mul, seg = segmentation_loss(dense_outs[0], dense_outs[1], seg_labels, lung_labels)
    loss = loss_cle + loss_pse + 2.0 * mul + seg
    return loss, dict(loss_cle=loss_cle, loss_pse=loss_pse, mul_loss=mul, seg_loss=seg)


def cls_train_loss(cls_outs, cle_labels, pse_labels, cle_cw: Tensor, pse_cw: Tensor):
    """Train branch of ScanCLSLightningModule.shared_step (models.py:248-258)."""
    loss_cle = F.cross_entropy(cls_outs[0], cle_labels, weight=cle_cw)
    loss_pse = F.cross_entropy(cls_outs[1], pse_labels, weight=pse_cw)
    return loss_cle + loss_pse, dict(loss_cle=loss_cle, loss_pse=loss_pse)


def ratio_to_label(ratios: Tensor, ratio_map) -> Tensor:
    """_ratio_to_label (models.py:533-537)."""
    out = []
    for r in ratios.tolist():
        out.append([k for k, (lo, hi) in ratio_map.items() if lo <= r < hi][0])
    return torch.tensor(out, dtype=torch.long)


# ---------------------------------------------------------------------------
# predict-time up-projection (models.py:430-450)
# ---------------------------------------------------------------------------
def predict_upproject(dense: Tensor, size, ess: Tensor, lungs: Tensor):
    """K18: trilinear(align_corners) resize to scan grid x ess mask; percentage =
    per-sample sum / lungs.sum() over the WHOLE batch (models.py:438-441)."""
    up = F.interpolate(dense, size=size, mode="trilinear", align_corners=True) * ess
    pct = up.view(up.shape[0], -1).sum(-1) / lungs.sum()
    return up, pct


# ---------------------------------------------------------------------------
# deterministic input transforms (models.py:59-63)
# ---------------------------------------------------------------------------
def prepare_image(scan: Tensor, target_size, from_span=(-1150.0, -300.0)) -> Tensor:
    """IntensityWindow (functional.py:13-26) -> Standardize (intensity_transforms.py:108-111)
    -> Interpolate(align_corners=True, only_in_plane=True) (spatial_transforms.py:55-75)."""
    img = scan.float()
    img = torch.clamp(img, min=from_span[0], max=from_span[1])
    img = (img - from_span[0]) / (from_span[1] - from_span[0])
    img = img - img.mean()
    img = img / img.std()
    data = F.interpolate(img[None], size=tuple(target_size[1:]), mode="bilinear", align_corners=True)
    idx = torch.linspace(0, scan.shape[0] - 1, target_size[0]).long()
    return data[:, idx][0]


def prepare_mask(mask: Tensor, target_size) -> Tensor:
    """Interpolate.apply_to_mask (spatial_transforms.py:77-98): nearest in-plane + depth select."""
    data = F.interpolate(mask[None].float(), size=tuple(target_size[1:]), mode="nearest")
    idx = torch.linspace(0, mask.shape[0] - 1, target_size[0]).long()
    return data[:, idx][0].type(mask.dtype)


# ---------------------------------------------------------------------------
# optimizers (torch.optim.Adam / SGD semantics; models.py:385-394, 689-698)
# ---------------------------------------------------------------------------
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8, wd: float = 0.0):
    """One torch.optim.Adam update (no amsgrad), in place on p, m, v.  ``step`` is
    the 1-based step count AFTER increment."""
    if wd != 0.0:
        g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def sgd_step(p: Tensor, g: Tensor, buf: Optional[Tensor], lr: float, momentum: float = 0.0,
             wd: float = 0.0, first: bool = False):
    """torch.optim.SGD update (dampening 0, no nesterov).  train.py:25,27 args."""
    if wd != 0.0:
        g = g + wd * p
    if momentum != 0.0:
        if first:
            buf.copy_(g)
        else:
            buf.mul_(momentum).add_(g)
        g = buf
    p.add_(g, alpha=-lr)


# ---------------------------------------------------------------------------
# N-rank DDP + SyncBN emulation
# ---------------------------------------------------------------------------
def ddp_emulated_grads(sd, xs: List[Tensor], lungs: List[Optional[Tensor]], arch: str, loss_fn,
                       pins: Optional[Dict[str, Tensor]] = None, dtype=None):
    """Gradients an N-rank DDP+SyncBatchNorm run produces (train.py:70,100-103):
    BN statistics over the concatenated batch, loss = mean over ranks of the loss
    computed from rank r's slice only.  ``loss_fn(rank, dense_slice, outs_slice)``.
    ``pins``: forced ReLU / max-pool decisions of the concatenated batch (see ``relu``);
    ``dtype``: evaluate in this precision (default: the state dict's).
    """
    N = len(xs)
    sizes = [t.shape[0] for t in xs]
    x = torch.cat(xs, 0)
    lg = None if lungs[0] is None else torch.cat(lungs, 0)
    if dtype is not None:
        sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        x = x.to(dtype)
        lg = None if lg is None else lg.to(dtype)
    leaves = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v)
              for k, v in sd.items()}
    dense, outs = forward(leaves, x, lg, arch, train=True, pins=pins)
    total = 0.0
    off = 0
    for r, b in enumerate(sizes):
        d_r = [d[off:off + b] for d in dense]
        o_r = [o[off:off + b] for o in outs]
        total = total + loss_fn(r, d_r, o_r) / N
        off += b
    total.backward()
    return {k: v.grad for k, v in leaves.items() if isinstance(v, Tensor) and v.requires_grad}, total.detach()


# ---------------------------------------------------------------------------
# predict post-processing (processor.py:34-38, 111-143)
# ---------------------------------------------------------------------------
def paste_resampled(dense: Tensor, crop, original_size) -> Tensor:
    """processor.py:115-122: dense [D,H,W] -> trilinear(align_corners) to the crop size -> pasted into zeros of
    the original grid.  crop = [[z0,z1],[y0,y1],[x0,x1]]."""
    recon = tuple(int(c[1]) - int(c[0]) for c in crop)
    up = F.interpolate(dense[None, None], size=recon, mode="trilinear", align_corners=True)[0, 0]
    full = torch.zeros(tuple(int(s) for s in original_size), dtype=up.dtype)
    full[tuple(slice(int(c[0]), int(c[1])) for c in crop)] = up
    return full


def window_u8(full: Tensor) -> Tensor:
    """utils.windowing(full, from_span=(0, 1)) (utils.py:28-37, to_span (0, 255)) + .astype(np.uint8) (processor.py:143)."""
    w = full.double().clamp(0.0, 1.0) / 1.0 * 255.0
    return w.to(torch.uint8)            # float -> uint8 truncates, like numpy astype


def severity_label(ratio: float, ratio_map) -> int:
    """processor.ratio_to_label (processor.py:34-38)."""
    return [k for k, (lo, hi) in ratio_map.items() if lo <= ratio < hi][0]


# ---------------------------------------------------------------------------
# epoch-end bookkeeping (models.py:287-317, 367-379)
# ---------------------------------------------------------------------------
def dedup_by_index(indices: Tensor, *cols: Tensor):
    """models.py:303-309: np.unique(indices, return_index=True) keeps the FIRST occurrence of every sample index
    (sorted by index); accuracies (models.py:300-301) are taken BEFORE the de-duplication."""
    order = torch.argsort(indices, stable=True)
    s = indices[order]
    first = torch.ones_like(s, dtype=torch.bool)
    first[1:] = s[1:] != s[:-1]
    keep = order[first]
    return indices[keep], tuple(c[keep] for c in cols)


def update_class_weights(weights: Tensor, y_true: Tensor, y_pred: Tensor) -> Tensor:
    """models.py:367-377: per-class accuracy = diag / row sums of sklearn's confusion_matrix (labels = sorted
    union of the values that occur); new weights = w * (1 - acc), renormalised to sum 1."""
    labels = torch.unique(torch.cat([y_true, y_pred]))
    acc = []
    for c in labels.tolist():
        row = y_true == c
        acc.append(float(((y_pred == c) & row).sum()) / float(row.sum()))
    w = weights.double() * (1.0 - torch.tensor(acc, dtype=torch.float64))
    return w / w.sum()


# ---------------------------------------------------------------------------
# train-time augmentations with given parameters (models.py:66-74)
# ---------------------------------------------------------------------------
def gaussian_additive(img: Tensor, sigma: float, noise: Tensor) -> Tensor:
    """GaussianAddictive.apply_to_image (intensity_transforms.py:163-177); `noise` = the torch.randn draw."""
    d_min, d_max = img.min(), img.max()
    d_range = d_max - d_min
    r = (img - d_min) / float(d_range + 1e-7) + sigma * noise
    r = r.clamp(0.0, 1.0)
    return r * d_range + d_min


def _frac_box(center, size, shape):
    """the integer box of BoxMaskOut / CropAndResize (intensity_transforms.py:226-235, spatial_transforms.py:172-177)"""
    return [(max(0, int(mc * ds) - int(ms * ds) // 2), min(int(mc * ds) + (int(ms * ds) - int(ms * ds) // 2), ds))
            for mc, ds, ms in zip(center, shape, size)]


def box_mask_out(img: Tensor, centers, sizes) -> Tensor:
    """BoxMaskOut.apply_to_image (intensity_transforms.py:220-237), assign_value 0."""
    out = img.clone()
    for c, s in zip(centers, sizes):
        out[tuple(slice(a, b) for a, b in _frac_box(c, s, img.shape))] = 0
    return out


def flip(t: Tensor, dims) -> Tensor:
    """Flip.apply (spatial_transforms.py:121-125)."""
    return torch.flip(t, dims=list(dims))


def crop_and_resize(t: Tensor, center, size, mask: bool = False) -> Tensor:
    """CropAndResize.apply (spatial_transforms.py:169-197) + functional.roi_align (functional.py:68-94):
    affine_grid over the normalised box (axes reversed to x,y,z), grid_sample with zero padding; images
    'bilinear' (trilinear) align_corners=True, masks nearest align_corners=False."""
    box = torch.tensor(_frac_box(center, size, t.shape), dtype=torch.float32) / torch.tensor(t.shape, dtype=torch.float32)[:, None]
    box = box.flip(0)                                   # (z,y,x) -> (x,y,z)
    theta = torch.cat([torch.diag(box[:, 1] - box[:, 0]), (-1.0 + box.sum(-1))[:, None]], dim=-1)[None]
    grid = F.affine_grid(theta, (1, 1) + tuple(t.shape), align_corners=False)
    out = F.grid_sample(t[None, None].float(), grid, mode="nearest" if mask else "bilinear", padding_mode="zeros",
                        align_corners=False if mask else True)
    return out[0, 0].to(t.dtype)
