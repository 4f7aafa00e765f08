"""GPU: whole-network parity of the drop-in modules (HIP path through the C ABI) against
(a) the golden fixtures recorded from the reference and (b) the CPU oracle run here.

Tolerances
  * class logits, regression scores, dRAM volumes, loss, BN running stats: max-relative
    <= 1e-3 against the reference's golden values (BASELINE.json north_star, fp32).
  * gradients: train-mode BN over tiny batches makes the fp32 gradient itself chaotic -- the
    CPU fp32 oracle is 6e-6 .. 3e-3 away from the same oracle evaluated in fp64, depending on
    the case (tools/grad_diag.py).  The bar is therefore accuracy-relative: per tensor,
    err(HIP, fp64) <= 4 * err(CPU fp32, fp64) + 1e-4 + FLIP (relative L2), i.e. the HIP path is
    as accurate as the reference's own arithmetic; plus gradient norms within 3e-2 of the golden.
    FLIP: a ReLU input within fp32 rounding of zero can land on either side (measured with
    tools/mask_diag.py on net_4: ONE of 65,536 decisions in xup3 differs, HIP 6.9e-6 vs 0.0);
    one flip moves every upstream weight gradient by O(1/sqrt(voxels per channel)): ~1-2e-2 at
    the 16x32x32 golden sizes (2,048 voxels), so FLIP = 3e-2 there and 5e-3 at 64x128x128.
  * decoder conv biases sit in front of a BatchNorm: their true gradient is 0, both sides
    compute rounding noise (SURVEY.md §7 parity traps) -> only |g| is bounded, and parameters
    after Adam steps (which turn that noise and every near-zero gradient into +-lr moves) are
    compared at 1e-2 relative L2.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_loss, head_weights, make_inputs, rel_l2
from oracle import med3d_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NET_FILES = sorted(glob.glob(os.path.join(GOLDEN, "net_*.npz")))
OUT_TOL = 1e-3
NORM_TOL = 3e-2
FLIP_TINY, FLIP_MID = 3e-2, 5e-3


def is_noise_param(name):
    """conv bias directly followed by BatchNorm (decoder convs): analytically zero gradient."""
    return name.endswith(".0.bias") and name.startswith("us")


def build(factory, seed):
    from bodyct_dram_emph_subtype_amd import med3d
    torch.manual_seed(seed)
    kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
    return getattr(med3d, factory)(**kw)


def assert_close_rel(a, b, tol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
    assert err < tol, f"{what}: max-rel {err:.3e} >= {tol}"


# every golden network on the library's own plan, and once more with the Winograd path forced
# onto every 3x3x3 stride-1 convolution it supports (DRAM_CONV_ALGO=2)
# ... and, on three of them, with the coarsest tiling (F(4,3) on all three axes) forced as well
NET_RUNS = [(p, "") for p in NET_FILES] + [(p, "2") for p in NET_FILES] + [(p, "2:4,4,4") for p in NET_FILES[:3]]


@pytest.mark.parametrize("path,algo", NET_RUNS,
                         ids=[os.path.basename(p)[:-4] + ("-winograd" + a[1:].replace(":", "-F").replace(",", "") if a else "")
                              for p, a in NET_RUNS])
def test_train_step_matches_reference_golden(path, algo, monkeypatch):
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam
    if algo:
        monkeypatch.setenv("DRAM_CONV_ALGO", algo.split(":")[0])
        if ":" in algo:
            monkeypatch.setenv("DRAM_WINO_TILING", algo.split(":")[1])
    g = np.load(path)
    factory = str(g["factory"])
    shape = tuple(int(v) for v in g["meta"][3:])
    m = build(factory, int(g["meta"][0]))
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.to(DEV).train()
    x, lungs = make_inputs(int(g["meta"][1]), shape, bool(int(g["with_lungs"])))
    hw = [t.to(DEV) for t in head_weights(int(g["meta"][1]), shape[0])]
    xd = x.to(DEV)
    ld = None if lungs is None else lungs.to(DEV)
    opt = FusedAdam(m.parameters(), lr=float(g["lr"]))
    names = [n for n, _ in m.named_parameters()]

    # CPU oracle on the same weights in fp32 and fp64 (the accuracy yardstick for gradients)
    def oracle_grads(dtype):
        lv = {k: (v.clone().to(dtype).requires_grad_(True) if k in names
                  else (v.clone().to(dtype) if v.is_floating_point() else v.clone())) for k, v in sd0.items()}
        od, oo = orc.forward(lv, x.to(dtype), None if lungs is None else lungs.to(dtype), factory, train=True)
        golden_loss(factory, od, oo, [t.cpu().to(dtype) for t in hw]).backward()
        return {n: lv[n].grad.double() for n in names}
    g32, g64 = oracle_grads(torch.float32), oracle_grads(torch.float64)

    # eval-mode forward on the initial weights (running stats 0/1), must not touch the buffers
    m.eval()
    with torch.no_grad():
        de, oe = m(xd, ld)
        dr, orf = orc.forward(sd0, x, lungs, factory, train=False)
    for a, b in zip(oe + de, orf + dr):
        assert_close_rel(a.cpu(), b, OUT_TOL, "eval-mode output")
    assert int(m.state_dict()["bn1.num_batches_tracked"]) == 0
    m.train()

    for step in range(int(g["meta"][2])):
        opt.zero_grad()
        dense, outs = m(xd, ld)
        loss = golden_loss(factory, dense, outs, hw)
        loss.backward()
        if step == 0:
            assert tuple(dense[0].shape) == tuple(g["dense0"].shape)
            assert_close_rel(dense[0].detach().cpu(), g["dense0"], OUT_TOL, "dense0")
            assert_close_rel(dense[1].detach().cpu(), g["dense1"], OUT_TOL, "dense1")
            assert_close_rel(outs[0].detach().cpu(), g["out0"], OUT_TOL, "out0")
            assert_close_rel(outs[1].detach().cpu(), g["out1"], OUT_TOL, "out1")
            assert abs(float(loss) - float(g["loss"])) < OUT_TOL * max(1.0, abs(float(g["loss"])))
            worst = (0.0, "")
            for i, (n, p) in enumerate(m.named_parameters()):
                gh = p.grad.double().cpu()
                if is_noise_param(n):
                    assert float(gh.norm()) < 1e-4 and float(g["gnorm"][i]) < 1e-4, n
                    continue
                assert abs(float(gh.norm()) / float(g["gnorm"][i]) - 1.0) < NORM_TOL, (n, float(gh.norm()))
                e_hip, e_cpu = rel_l2(gh, g64[n]), rel_l2(g32[n], g64[n])
                assert e_hip <= 4.0 * e_cpu + 1e-4 + FLIP_TINY, f"{n}: hip-vs-fp64 {e_hip:.2e}, cpu32-vs-fp64 {e_cpu:.2e}"
                worst = max(worst, (e_hip, n))
            print(f"[{factory}] worst gradient error vs fp64 oracle: {worst}")
            sd = m.state_dict()
            for k in g.files:
                if k.startswith("stat:"):
                    assert np.allclose(sd[k[5:]].cpu().numpy(), g[k], rtol=OUT_TOL, atol=1e-5), k
            assert int(sd["bn1.num_batches_tracked"]) == 1
        opt.step()
    sd = m.state_dict()
    # Adam turns every near-zero gradient into a +-lr move (exact optimizer arithmetic is
    # checked in test_fused_adam_and_sgd_match_torch): only a loose bound is meaningful here
    # (measured 0.03-0.062 across the kernel plans: the value is set by which near-zero gradient components flip sign)
    assert rel_l2(sd["conv1.weight"].cpu(), g["conv1_after"]) < 8e-2
    assert rel_l2(sd["fcs.0.weight"].cpu(), g["fc0_after"]) < 8e-2
    # (the golden eval-mode outputs after these two Adam steps are not compared: by then the
    #  parameters differ by the +-lr noise moves above; eval-mode parity is checked on the
    #  initial weights at the top of this test)


def test_survey_anchor():
    """manual_seed(0) -> resnet18segreg -> randn/rand inputs."""
    from bodyct_dram_emph_subtype_amd import med3d
    torch.manual_seed(0)
    m = med3d.resnet18segreg()
    x = torch.randn(1, 1, 16, 32, 32)
    lungs = (torch.rand(1, 1, 16, 32, 32) > 0.3).float()
    m = m.to(DEV).train()
    _, outs = m(x.to(DEV), lungs.to(DEV))
    assert abs(float(outs[0]) - 0.56180799) < 1e-3 * 0.56 and abs(float(outs[1]) - 0.43881506) < 1e-3 * 0.44


def test_fused_adam_and_sgd_match_torch():
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam, FusedSGD
    torch.manual_seed(0)
    shapes = [(64, 1, 7, 7, 7), (128,), (33, 17), (70001,)]
    for cls_f, cls_r, kw in ((FusedAdam, torch.optim.Adam, dict(lr=1e-3)),
                             (FusedAdam, torch.optim.Adam, dict(lr=1e-2, weight_decay=0.01)),
                             (FusedSGD, torch.optim.SGD, dict(lr=0.1, momentum=0.9, weight_decay=1e-4)),
                             (FusedSGD, torch.optim.SGD, dict(lr=0.1))):
        ref = [torch.randn(s).requires_grad_(True) for s in shapes]
        dev = [r.detach().clone().to(DEV).requires_grad_(True) for r in ref]
        o_r, o_d = cls_r(ref, **kw), cls_f(dev, **kw)
        sched = torch.optim.lr_scheduler.ExponentialLR(o_d, gamma=0.95)
        sched_r = torch.optim.lr_scheduler.ExponentialLR(o_r, gamma=0.95)
        for step in range(4):
            for r, d in zip(ref, dev):
                gr = torch.randn(r.shape)
                r.grad, d.grad = gr.clone(), gr.to(DEV)
            o_r.step()
            o_d.step()
            sched.step()
            sched_r.step()
        for r, d in zip(ref, dev):
            assert torch.allclose(d.detach().cpu(), r.detach(), rtol=2e-5, atol=2e-6), (cls_f.__name__, kw)
    st = o_d.state_dict()
    assert "state" in st and "param_groups" in st


def test_no_cpu_fallback():
    from bodyct_dram_emph_subtype_amd import med3d, ops
    m = med3d.resnet18segreg()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 16, 16, 16))
    with pytest.raises(RuntimeError):
        ops.add(torch.zeros(4), torch.zeros(4))


def _synthetic(B, dims, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, *dims, generator=g)
    D, H, W = dims
    z = (torch.arange(D).float() - (D - 1) / 2) / (0.4 * D)
    y = (torch.arange(H).float() - (H - 1) / 2) / (0.35 * H)
    xx = (torch.arange(W).float() - (W - 1) / 2) / (0.4 * W)
    lung = ((z[:, None, None] ** 2 + y[None, :, None] ** 2 + xx[None, None, :] ** 2) <= 1.0).float()
    return x, lung[None, None].expand(B, 1, D, H, W).contiguous()


def test_mid_size_train_step_vs_oracle():
    """resnet18segreg, 1x64x128x128 (BASELINE configs[0] volume): full dRAM train loss through the
    fused loss kernels; outputs 1e-3 vs the fp32 oracle, gradients accuracy-relative vs fp64."""
    from bodyct_dram_emph_subtype_amd import med3d, models
    factory, dims = "resnet18segreg", (64, 128, 128)
    torch.manual_seed(5)
    m = med3d.resnet18segreg()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    x, lungs = _synthetic(1, dims, 11)
    ems = ((x < -1.0).float() * lungs)
    cle, pse = torch.tensor([3]), torch.tensor([1])
    cw, pw = torch.tensor([0.3]), torch.tensor([0.6])

    def oracle(dtype):
        lv = {k: (v.clone().to(dtype).requires_grad_(True) if k in names
                  else (v.clone().to(dtype) if v.is_floating_point() else v.clone())) for k, v in sd0.items()}
        d, o = orc.forward(lv, x.to(dtype), lungs.to(dtype), factory, train=True)
        loss, parts = orc.reg_train_loss(d, o, lungs.to(dtype), ems.to(dtype), cle, pse, cw.to(dtype), pw.to(dtype))
        loss.backward()
        return [t.detach() for t in d], [t.detach() for t in o], loss.detach(), parts, {n: lv[n].grad.double() for n in names}
    d32, o32, l32, parts32, g32 = oracle(torch.float32)
    _, _, _, _, g64 = oracle(torch.float64)
    md = m.to(DEV).train()
    dd, od = md(x.to(DEV), lungs.to(DEV))
    loss, parts = models.reg_train_loss(dd, od, lungs.to(DEV), ems.to(DEV), cle.to(DEV), pse.to(DEV), cw.to(DEV),
                                        pw.to(DEV))
    loss.backward()
    for a, b in zip(od, o32):
        assert_close_rel(a.detach().cpu(), b, OUT_TOL, "regression score")
    for a, b in zip(dd, d32):
        assert_close_rel(a.detach().cpu(), b, OUT_TOL, "dRAM volume")
    for k in parts32:
        assert abs(float(parts[k]) - float(parts32[k])) < OUT_TOL * max(1.0, abs(float(parts32[k]))), k
    assert abs(float(loss) - float(l32)) < OUT_TOL * max(1.0, abs(float(l32)))
    for n, p in md.named_parameters():
        if is_noise_param(n):
            continue
        e_hip, e_cpu = rel_l2(p.grad.double().cpu(), g64[n]), rel_l2(g32[n], g64[n])
        assert e_hip <= 4.0 * e_cpu + 1e-4 + FLIP_MID, f"{n}: hip-vs-fp64 {e_hip:.2e}, cpu32-vs-fp64 {e_cpu:.2e}"


@pytest.mark.slow
def test_full_size_forward_config1_vs_oracle():
    """BASELINE configs[1] at full size (resnet18segcls, 2x1x128x256x256): class logits and the
    dense maps against the CPU oracle forward; plus determinism (bitwise equal re-run)."""
    from bodyct_dram_emph_subtype_amd import med3d
    torch.manual_seed(0)
    m = med3d.resnet18segcls(n_classes=[6, 3])
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    x, lungs = _synthetic(2, (128, 256, 256), 1234)
    with torch.no_grad():
        d_ref, o_ref = orc.forward(sd0, x, lungs, "resnet18segcls", train=True)
    md = m.to(DEV).train()
    with torch.no_grad():
        d1, o1 = md(x.to(DEV), lungs.to(DEV))
        d2, o2 = md(x.to(DEV), lungs.to(DEV))
    for a, b in zip(o1, o_ref):
        assert_close_rel(a.cpu(), b, OUT_TOL, "class logits")
    assert_close_rel(d1[0].cpu(), d_ref[0], OUT_TOL, "dense cle map")
    assert torch.equal(o1[0], o2[0]) and torch.equal(d1[1], d2[1])      # no atomics anywhere: reproducible
