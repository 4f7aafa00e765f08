"""Generate golden fixtures by running the REFERENCE itself on CPU.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference repo ships no tests or golden vectors, so parity is
pinned by what this script records: it imports the reference's ``med3d.py`` and
``metrics.py`` unmodified and ``models.py`` with its missing third-party imports
(pytorch_lightning, hydra, SimpleITK, ...) stubbed in ``sys.modules``, runs them
on seeded inputs, and writes inputs' seeds + expected outputs to ``*.npz``.
Only data is committed -- never reference source.
"""
import os
import sys
import types
import enum

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import med3d as ref_med3d      # noqa: E402
import metrics as ref_metrics  # noqa: E402


# ---------------------------------------------------------------------------
def make_inputs(seed, shape, with_lungs=True):
    """Shared recipe (also used by tests): deterministic CPU generator."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    lungs = (torch.rand(*shape, generator=g) > 0.3).float() if with_lungs else None
    return x, lungs


def head_weights(seed, B):
    g = torch.Generator().manual_seed(seed + 77)
    return [torch.randn(B, 6, generator=g), torch.randn(B, 3, generator=g), torch.randn(B, generator=g),
            torch.randn(B, generator=g)]


def net_case(factory, model_seed, in_seed, shape, with_lungs=True, adam_steps=2, lr=1e-3):
    torch.manual_seed(model_seed)
    kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
    m = getattr(ref_med3d, factory)(**kw)
    rec = {}
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    rec["wsum"] = np.array([float(v.double().sum()) for k, v in sd0.items() if v.is_floating_point()])
    rec["wabs"] = np.array([float(v.double().abs().sum()) for k, v in sd0.items() if v.is_floating_point()])
    x, lungs = make_inputs(in_seed, shape, with_lungs)
    B = shape[0]
    hw = head_weights(in_seed, B)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    names = [n for n, _ in m.named_parameters()]
    for step in range(adam_steps):
        opt.zero_grad()
        dense, outs = m(x, lungs)
        if factory.endswith("cls"):
            loss = (outs[0] * hw[0]).sum() + (outs[1] * hw[1]).sum()
        else:
            # scan-level scores + a dense term so gradients also enter through the dRAM maps
            loss = (outs[0] * hw[2]).sum() + (outs[1] * hw[3]).sum() + 0.1 * (dense[0] * dense[1]).mean()
        loss.backward()
        if step == 0:
            rec["dense0"] = dense[0].detach().numpy()
            rec["dense1"] = dense[1].detach().numpy()
            rec["out0"] = outs[0].detach().numpy()
            rec["out1"] = outs[1].detach().numpy()
            rec["loss"] = np.array(float(loss))
            rec["gnorm"] = np.array([float(p.grad.double().norm()) for p in m.parameters()])
            rec["gsum"] = np.array([float(p.grad.double().sum()) for p in m.parameters()])
            for n, p in m.named_parameters():
                if n in ("conv1.weight", "fcs.0.weight", "fcs.1.bias", "us3.0.weight", "bn1.weight",
                         "layer2.0.bn1.bias", "us1.conv_blocks.1.0.weight"):
                    rec["grad:" + n] = p.grad.numpy().copy()
            sd1 = m.state_dict()
            for k in ("bn1.running_mean", "bn1.running_var", "us3.1.running_mean", "us3.1.running_var",
                      "layer4.1.bn2.running_var", "us2.conv_blocks.0.1.running_mean"):
                if k in sd1:
                    rec["stat:" + k] = sd1[k].numpy().copy()
        opt.step()
    rec["psum_after"] = np.array([float(p.double().sum()) for p in m.parameters()])
    rec["conv1_after"] = dict(m.named_parameters())["conv1.weight"].detach().numpy().copy()
    rec["fc0_after"] = dict(m.named_parameters())["fcs.0.weight"].detach().numpy().copy()
    # eval-mode forward with the running stats accumulated so far
    m.eval()
    with torch.no_grad():
        dense, outs = m(x, lungs)
    rec["eval_out0"] = outs[0].numpy()
    rec["eval_out1"] = outs[1].numpy()
    rec["eval_dense0"] = dense[0].numpy()
    rec["meta"] = np.array([model_seed, in_seed, adam_steps] + list(shape))
    rec["lr"] = np.array(lr)
    rec["names"] = np.array(names)
    return rec


# ---------------------------------------------------------------------------
def import_ref_models():
    """Import reference models.py with its absent third-party deps stubbed."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _LM(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    class _Stage(str, enum.Enum):
        TRAINING = "train"
        VALIDATING = "validate"
        TESTING = "test"
        PREDICTING = "predict"

    pl = mod("pytorch_lightning", LightningModule=_LM, LightningDataModule=object)
    mod("pytorch_lightning.loggers", TensorBoardLogger=object)
    mod("pytorch_lightning.trainer", )
    mod("pytorch_lightning.trainer.states", RunningStage=_Stage)
    mod("pytorch_lightning.callbacks", ModelCheckpoint=object)
    pl.loggers = sys.modules["pytorch_lightning.loggers"]
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=object)
    import importlib
    for n in ("SimpleITK", "cv2", "seaborn", "hydra", "hydra.utils", "omegaconf", "torchmetrics",
              "matplotlib", "matplotlib.pyplot", "matplotlib.backends", "matplotlib.backends.backend_agg"):
        try:
            importlib.import_module(n)
        except Exception:
            mod(n)
    if not hasattr(sys.modules["omegaconf"], "OmegaConf"):
        sys.modules["omegaconf"].OmegaConf = object
    if not hasattr(sys.modules["matplotlib.backends.backend_agg"], "FigureCanvasAgg"):
        sys.modules["matplotlib.backends.backend_agg"].FigureCanvasAgg = object
        sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    torch.Tensor.cuda = lambda self, *a, **k: self      # models.py:250,252,509,558,561 hard-code .cuda()
    import models as ref_models
    return ref_models


def loss_case(ref_models, seed, shape, cle_labels, pse_labels):
    """Reference dRAM loss terms (models.py:495-531, metrics.py) on random maps."""
    g = torch.Generator().manual_seed(seed)
    B, D, H, W = shape
    dense_cle = torch.rand(B, 1, D // 2, H // 2, W // 2, generator=g).requires_grad_(True)
    dense_pse = torch.rand(B, 1, D // 2, H // 2, W // 2, generator=g).requires_grad_(True)
    lungs = (torch.rand(B, 1, D, H, W, generator=g) > 0.4).float()
    ems = ((torch.rand(B, 1, D, H, W, generator=g) > 0.8).float() * lungs)
    cw = torch.rand(B, generator=g) + 0.1
    pw = torch.rand(B, generator=g) + 0.1

    class A:
        model_arch = "med3ddram18"
        lr = 1e-4
    ref_models.get_model_by_name = lambda name: torch.nn.Identity()
    mod = ref_models.ScanRegLightningModule(A())
    ds = ref_models._DATASET_CLASS
    cle = torch.tensor(cle_labels)
    pse = torch.tensor(pse_labels)
    lg = torch.nn.functional.interpolate(lungs, dense_cle.shape[-3:], mode="nearest")
    reg0 = (dense_cle * lg).view(B, -1).sum(-1) / lg.view(B, -1).sum(-1)
    reg1 = (dense_pse * lg).view(B, -1).sum(-1) / lg.view(B, -1).sum(-1)
    t0 = mod._generate_regression_labels(cle, ds.cle_ratio_map)
    t1 = mod._generate_regression_labels(pse, ds.pse_ratio_map)
    l0 = mod._interval_regression_loss(reg0, t0, cw)
    l1 = mod._interval_regression_loss(reg1, t1, pw)
    binary = torch.logical_or(cle > 0, pse > 0).long()
    seg_labels = torch.nn.functional.interpolate(ems * binary.float().view(B, 1, 1, 1, 1), dense_cle.shape[-3:],
                                                 mode="nearest").detach()
    mul, seg = mod._segmentation_loss(dense_cle, dense_pse, seg_labels, lg)
    loss = l0 + l1 + 2.0 * mul + seg
    loss.backward()
    return dict(meta=np.array([seed, B, D, H, W]), cle=np.array(cle_labels), pse=np.array(pse_labels),
                t0=t0.numpy(), t1=t1.numpy(), cw=cw.numpy(), pw=pw.numpy(),
                l0=np.array(float(l0)), l1=np.array(float(l1)), mul=np.array(float(mul)), seg=np.array(float(seg)),
                loss=np.array(float(loss)), g_cle=dense_cle.grad.numpy(), g_pse=dense_pse.grad.numpy(),
                pred0=mod._ratio_to_label(reg0.detach(), ds.cle_ratio_map).numpy(),
                pred1=mod._ratio_to_label(reg1.detach(), ds.pse_ratio_map).numpy(),
                reg0=reg0.detach().numpy(), reg1=reg1.detach().numpy())


def transform_case(ref_models):
    """Deterministic input pipeline of SubtypeDataModule._transform (models.py:59-63) through the
    reference's own IntensityWindow / Standardize / Interpolate classes."""
    g = torch.Generator().manual_seed(31)
    scan = (torch.rand(20, 37, 45, generator=g) * 1400.0 - 1300.0)          # HU-like values
    mask = torch.rand(20, 37, 45, generator=g) > 0.5
    target = (12, 24, 32)
    tr = [ref_models.IntensityWindow(from_span=(-1150, -300), to_span=(0, 1), output_dtype=torch.float32),
          ref_models.Standardize(), ref_models.Interpolate(target, None, align_corners=True)]
    img = scan.clone()
    for t_ in tr:
        img = t_.apply_to_image(img)
    m = tr[2].apply_to_mask(mask.clone())
    return dict(scan=scan.numpy(), mask=mask.numpy(), target=np.array(target), image_out=img.numpy(),
                mask_out=m.numpy())


def block_cases():
    """Per-block fixtures: crop_concat, shortcut-A detach,
    UpsampleConvBlock5d."""
    rec = {}
    g = torch.Generator().manual_seed(5)
    t1 = torch.randn(1, 2, 4, 6, 8, generator=g)
    t2 = torch.randn(1, 3, 6, 9, 8, generator=g)
    rec["cc_t1"], rec["cc_t2"] = t1.numpy(), t2.numpy()
    rec["cc_out"] = ref_med3d.crop_concat_5d(t1, t2).numpy()
    x = torch.randn(2, 4, 6, 6, 6, generator=g).requires_grad_(True)
    ds = ref_med3d.downsample_basic_block(x, 8, 2)
    rec["ds_x"], rec["ds_out"] = x.detach().numpy(), ds.numpy()
    rec["ds_requires_grad"] = np.array(int(ds.requires_grad))
    torch.manual_seed(11)
    blk = ref_med3d.BasicBlock(4, 8, stride=2, dilation=1,
                               downsample=lambda t: ref_med3d.downsample_basic_block(t, 8, 2))
    blk.train()
    y = blk(x)
    y.square().sum().backward()
    rec["bb_out"] = y.detach().numpy()
    rec["bb_dx"] = x.grad.numpy().copy()
    rec["bb_w1"] = blk.conv1.weight.detach().numpy()
    rec["bb_w2"] = blk.conv2.weight.detach().numpy()
    rec["bb_dw1"] = blk.conv1.weight.grad.numpy()
    torch.manual_seed(12)
    up = ref_med3d.UpsampleConvBlock5d([6, 4], [4, 4], 1, 2, (3, 3), True, (1, 1), norm_method="bn",
                                       act_method="relu", dropout=0.0)
    up.train()
    a = torch.randn(1, 4, 3, 4, 5, generator=g)
    b = torch.randn(1, 2, 6, 8, 10, generator=g)
    rec["up_a"], rec["up_b"] = a.numpy(), b.numpy()
    rec["up_out"] = up(a, b).detach().numpy()
    for k, v in up.state_dict().items():
        rec["up_sd:" + k] = v.numpy()
    return rec


def main():
    torch.set_num_threads(8)
    cases = [
        ("resnet18segreg", 0, 100, (1, 1, 16, 32, 32), True),
        ("resnet18segreg", 1, 101, (2, 1, 24, 32, 40), True),
        ("resnet18segreg", 2, 102, (1, 1, 16, 16, 16), False),
        ("resnet18segcls", 3, 103, (2, 1, 16, 32, 32), True),
        ("resnet34segcls", 4, 104, (1, 1, 16, 32, 32), True),
        ("resnet34segreg", 5, 105, (1, 1, 16, 16, 32), True),
        ("resnet50segcls", 6, 106, (1, 1, 16, 32, 32), True),
        ("resnet50segreg", 7, 107, (2, 1, 16, 16, 32), True),
    ]
    if "--skip-nets" in sys.argv:
        cases = []
    for i, (f, ms, ins, shape, wl) in enumerate(cases):
        rec = net_case(f, ms, ins, shape, wl)
        rec["factory"] = np.array(f)
        rec["with_lungs"] = np.array(int(wl))
        path = os.path.join(OUT, f"net_{i}_{f}.npz")
        np.savez_compressed(path, **rec)
        print("wrote", path, {k: float(np.ravel(v)[0]) for k, v in rec.items() if k in ("out0", "out1", "loss")})
    # survey anchor
    torch.manual_seed(0)
    m = ref_med3d.resnet18segreg()
    x = torch.randn(1, 1, 16, 32, 32)
    lungs = (torch.rand(1, 1, 16, 32, 32) > 0.3).float()
    _, o = m(x, lungs)
    print("anchor", [float(t) for t in o])
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **block_cases())
    rm = import_ref_models()
    lc = [loss_case(rm, 200, (2, 16, 32, 32), [3, 0], [1, 0]),
          loss_case(rm, 201, (3, 8, 16, 24), [0, 5, 2], [2, 0, 1]),
          loss_case(rm, 202, (1, 8, 16, 16), [0], [0])]
    for i, rec in enumerate(lc):
        np.savez_compressed(os.path.join(OUT, f"loss_{i}.npz"), **rec)
        print("loss case", i, float(rec["loss"]))
    np.savez_compressed(os.path.join(OUT, "transforms.npz"), **transform_case(rm))
    print("wrote transforms.npz")


if __name__ == "__main__":
    main()
