"""GPU: whole-network parity of the drop-in modules (HIP path through the C ABI) against
(a) the golden fixtures recorded from the reference and (b) the CPU oracle run here.

Tolerances
  * class logits, regression scores, dRAM volumes, loss, BN running stats: max-relative
    <= 1e-3 against the reference's golden values (BASELINE.json north_star, fp32).
  * gradients, DECISION-PINNED: the network is piecewise linear in its ReLU / max-pool decisions and a
    ReLU input within fp32 rounding of zero can land on either side in two implementations (one such
    flip moves every upstream gradient by O(1/sqrt(voxels))), which used to force a 3e-2 allowance
    that could also hide a real halo bug.  Instead the HIP forward exports the decisions it took
    (engine.forward_decisions: every ReLU mask + the max-pool taps) and the oracle backward is
    evaluated in fp64 on exactly that linear piece (oracle.forward(pins=...)).  What is left is
    summation-order rounding only: per-tensor relative L2 <= GRAD_TOL = 1e-4 for every parameter of
    every golden network, on the library's plan and with the Winograd paths forced (where the CPU fp32
    oracle itself is further than that from fp64 on the same pinned piece: 3x its distance, at most 5e-4).
    (Gradient norms are additionally compared with the reference's golden values at 3e-2: that
    number contains the reference's own fp32 decisions and is a sanity bound, not the parity bar.)
  * decoder conv biases sit in front of a BatchNorm: their true gradient is 0, both sides
    compute rounding noise (SURVEY.md §7 parity traps) -> only |g| is bounded.
  * parameters after two Adam steps: Adam turns every near-zero gradient component into a +-lr move,
    so the sign noise of those components dominates: relative L2 <= 8e-2 (a smoke bound; the exact
    optimizer arithmetic is held to 2e-5 against torch.optim in test_fused_adam_and_sgd_match_torch).
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
OUT_HEADROOM = 8e-4      # full-size outputs: the 1e-3 bar with 20 % of headroom, so a plan change that eats the margin fails loudly
NORM_TOL = 3e-2
GRAD_TOL = 1e-4          # decision-pinned gradients vs the fp64 oracle, per tensor, relative L2
DENSE_L2_TOL = 1e-4      # dense maps vs the reference's recorded ones, relative L2 (element-wise, unlike the max-norm bar)


def is_noise_param(name):
    """conv bias directly followed by BatchNorm (decoder convs): analytically zero gradient."""
    return name.endswith(".0.bias") and name.startswith("us")


def build(factory, seed):
    from bodyct_dram_emph_subtype_amd import med3d
    torch.manual_seed(seed)
    kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
    return getattr(med3d, factory)(**kw)


def grad_tol(e_cpu32):
    """GRAD_TOL, or -- where the reference's own fp32 arithmetic is further than that from fp64 on the same
    pinned piece (54 BN layers over 16-voxel statistics in the ResNet-50 fixtures) -- 3x that distance,
    never more than 5e-4."""
    return min(max(GRAD_TOL, 3.0 * e_cpu32), 5e-4)


def pinned_decisions(out):
    """ReLU masks + max-pool taps of the forward that produced `out` (before its backward frees them)."""
    from bodyct_dram_emph_subtype_amd.engine import forward_decisions
    return {k: v.cpu() for k, v in forward_decisions(out.grad_fn.saved_state).items()}


FLIP_FRAC = 2e-5         # share of ReLU decisions that may differ from the free-running fp32 oracle's
FLIP_MARGIN = 5e-5       # ... and only where the oracle's pre-activation is this close to zero (x the layer's max |value|)


def audit_decisions(pins_hip, sd0, x, lungs, factory):
    """Decision INDEPENDENCE check.  The gradient comparisons below run the oracle on the HIP forward's own ReLU /
    max-pool decisions; a kernel that produced wrong activations would pin the oracle to its own mistake.  So
    the free-running fp32 oracle (its own decisions, the reference's arithmetic) is evaluated too, and every ReLU
    decision the HIP forward took differently must sit on an oracle pre-activation within FLIP_MARGIN x max|value|
    of zero -- a rounding tie, not a wrong value -- and such ties must be rare (<= FLIP_FRAC of all decisions;
    measured on the 8 golden networks: 0-9 of 0.2-2.5 million, margins <= 1.2e-5).
    Max-pool taps may differ only where the two candidate values are equally close.  Returns (flipped ReLU
    decisions, ReLU decisions, differing max-pool taps, max-pool outputs, worst flip margin)."""
    rec = {"__record__": "pre"}
    with torch.no_grad():
        orc.forward(dict(sd0), x, lungs, factory, train=True, pins=rec)
    flips = total = 0
    worst = 0.0
    for k, mask in pins_hip.items():
        if k == "maxpool":
            continue
        diff = mask != rec[k]
        total += mask.numel()
        n = int(diff.sum())
        if n:
            pre = rec["pre:" + k]
            margin = float(pre[diff].abs().max() / pre.abs().max())
            worst = max(worst, margin)
            assert margin <= FLIP_MARGIN, f"{k}: {n} ReLU decisions differ from the fp32 oracle at |pre|/max up to {margin:.2e}"
        flips += n
    assert flips <= max(2, FLIP_FRAC * total), f"{flips} of {total} ReLU decisions differ from the free-running fp32 oracle"
    mp = int((pins_hip["maxpool"] != rec["maxpool"]).sum())
    assert mp <= max(2, FLIP_FRAC * pins_hip["maxpool"].numel()), f"{mp} max-pool taps differ"
    return flips, total, mp, pins_hip["maxpool"].numel(), worst


def assert_close_rel(a, b, tol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
    assert err < tol, f"{what}: max-rel {err:.3e} >= {tol}"
    return float(err)


# every golden network on the library's own plan, and once more with the Winograd path forced
# onto every 3x3x3 stride-1 convolution it supports (DRAM_CONV_ALGO=2)
# ... and, on three of them, with the coarsest tiling (F(4,3) on all three axes) forced as well
# ... and with the first decoder convolution forced onto the low-resolution-mixing path (csrc/upmix.hip; the library's
# own plan takes it from a 256-voxel low-resolution grid on, which the 16x32x32 fixtures do not reach)
NET_RUNS = ([(p, "") for p in NET_FILES] + [(p, "2") for p in NET_FILES] + [(p, "2:4,4,4") for p in NET_FILES[:3]]
            + [(p, "u") for p in NET_FILES])


@pytest.mark.parametrize("path,algo", NET_RUNS,
                         ids=[os.path.basename(p)[:-4] + ("-upmix" if a == "u" else
                                                          "-winograd" + a[1:].replace(":", "-F").replace(",", "") if a else "")
                              for p, a in NET_RUNS])
def test_train_step_matches_reference_golden(path, algo, monkeypatch):
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam
    monkeypatch.setenv("DRAM_UPMIX", "2" if algo == "u" else "0")
    if algo == "u":
        algo = ""
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

    def oracle_grads_pinned(pins, dt=torch.float64):
        """oracle backward (fp64: the yardstick; fp32: the reference arithmetic's own rounding) on the linear
        piece the HIP forward ran on"""
        lv = {k: (v.clone().to(dt).requires_grad_(True) if k in names
                  else (v.clone().to(dt) if v.is_floating_point() else v.clone())) for k, v in sd0.items()}
        od, oo = orc.forward(lv, x.to(dt), None if lungs is None else lungs.to(dt), factory, train=True, pins=pins)
        golden_loss(factory, od, oo, [t.cpu().to(dt) for t in hw]).backward()
        return {n: lv[n].grad for n in names}

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
        if step == 0:
            pins = pinned_decisions(dense[0])
            flips = audit_decisions(pins, sd0, x, lungs, factory)
            g64, g32 = oracle_grads_pinned(pins), oracle_grads_pinned(pins, torch.float32)
        loss = golden_loss(factory, dense, outs, hw)
        loss.backward()
        if step == 0:
            assert tuple(dense[0].shape) == tuple(g["dense0"].shape)
            assert_close_rel(dense[0].detach().cpu(), g["dense0"], OUT_TOL, "dense0")
            assert_close_rel(dense[1].detach().cpu(), g["dense1"], OUT_TOL, "dense1")
            assert_close_rel(outs[0].detach().cpu(), g["out0"], OUT_TOL, "out0")
            assert_close_rel(outs[1].detach().cpu(), g["out1"], OUT_TOL, "out1")
            assert abs(float(loss) - float(g["loss"])) < OUT_TOL * max(1.0, abs(float(g["loss"])))
            # element-wise (relative L2) on the dense maps too: the max-norm bar alone would let a few wrong voxels through
            assert rel_l2(dense[0].detach().cpu(), g["dense0"]) < DENSE_L2_TOL and rel_l2(dense[1].detach().cpu(), g["dense1"]) < DENSE_L2_TOL
            worst = (0.0, "")
            for i, (n, p) in enumerate(m.named_parameters()):
                gh = p.grad.double().cpu()
                if is_noise_param(n):
                    assert float(gh.norm()) < 1e-4 and float(g["gnorm"][i]) < 1e-4, n
                    continue
                assert abs(float(gh.norm()) / float(g["gnorm"][i]) - 1.0) < NORM_TOL, (n, float(gh.norm()))
                e_hip, e_cpu = rel_l2(gh, g64[n]), rel_l2(g32[n], g64[n])
                worst = max(worst, (e_hip, e_cpu, n))
                assert e_hip <= grad_tol(e_cpu), f"{n}: hip vs decision-pinned fp64 oracle {e_hip:.2e} (CPU fp32: {e_cpu:.2e})"
            # ... and against the gradient TENSORS the reference itself recorded (its own fp32 decisions; seven per
            # network, make_golden.py:75-78): with no decision flipped the two ran on the same linear piece and must
            # agree like the pinned comparison; otherwise each flipped tie moves the gradient by O(1/sqrt(voxels))
            gold_worst = (0.0, "")
            for k in g.files:
                if k.startswith("grad:") and not is_noise_param(k[5:]):
                    e = rel_l2(dict(m.named_parameters())[k[5:]].grad.cpu(), g[k])
                    gold_worst = max(gold_worst, (e, k[5:]))
                    # (measured: 2e-5 ... 5e-5 with no flip; 1e-3 ... 3.5e-2 with 1-9 flipped ties of ~1e6 decisions)
                    bar = grad_tol(rel_l2(g32[k[5:]], g64[k[5:]])) if (flips[0] == 0 and flips[2] == 0) else 2 * NORM_TOL
                    assert e <= bar, f"{k[5:]}: hip vs the reference's recorded gradient {e:.2e} > {bar:.1e} (flips {flips[:4]})"
            print(f"[{factory}{' ' + algo if algo else ''}] worst gradient error vs decision-pinned fp64 oracle "
                  f"(hip, cpu-fp32, tensor): {worst}; vs the reference's recorded gradients {gold_worst}; "
                  f"decisions differing from the free fp32 oracle: relu {flips[0]}/{flips[1]} (worst margin {flips[4]:.1e}), "
                  f"maxpool {flips[2]}/{flips[3]}")
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


def _oracle_backward(sd0, names, factory, x, lungs, pins, upstream, dtype):
    """decision-pinned oracle forward, then backward of the GIVEN upstream gradients of (dense0, dense1, out0, out1)"""
    lv = {k: (v.clone().to(dtype).requires_grad_(True) if k in names
              else (v.clone().to(dtype) if v.is_floating_point() else v.clone())) for k, v in sd0.items()}
    d, o = orc.forward(lv, x.to(dtype), None if lungs is None else lungs.to(dtype), factory, train=True, pins=pins)
    pairs = [(t, u.to(dtype)) for t, u in zip(d + o, upstream) if u is not None]
    torch.autograd.backward([t for t, _ in pairs], [u for _, u in pairs])
    return {n: lv[n].grad.double() for n in names}


def _dram_loss_checks(models, dd, od, lungs, ems, cle, pse, cw, pw):
    """dRAM loss through the fused loss kernels: value + parts at 1e-3 against the fp32 oracle evaluated at the
    SAME dense maps / scores; its gradient FIELDS (what the loss kernels hand to the network backward) against
    fp64 at the same point.  At a random initialisation sigmoid(.) ~ 0.5 puts cle+pse on the clamp(., 0, 1) kink
    of models.py:527 and BCE's 1/(1-p) (metrics.py:18-24, eps = 1e-6) makes the reference's own fp32 gradient
    field ~1e-2 from fp64 there (tools/grad_pairs.py), so the per-voxel fields are compared where that term is
    conditioned (|1 - (cle+pse)| > 1e-3): relative L2 <= 2e-4; the pooled-score gradients at 1e-5."""
    loss, parts = models.reg_train_loss(dd, od, lungs.to(DEV), ems.to(DEV), cle.to(DEV), pse.to(DEV), cw.to(DEV), pw.to(DEV))
    ups = [u.detach().cpu() for u in torch.autograd.grad(loss, dd + od, retain_graph=True)]
    at = [t.detach().cpu() for t in dd + od]
    l32, parts32 = orc.reg_train_loss(at[:2], at[2:], lungs, ems, cle, pse, cw, pw)
    for k in parts32:
        assert abs(float(parts[k]) - float(parts32[k])) < OUT_TOL * max(1.0, abs(float(parts32[k]))), k
    assert abs(float(loss) - float(l32)) < OUT_TOL * max(1.0, abs(float(l32)))
    leaf = [t.double().requires_grad_(True) for t in at]
    orc.reg_train_loss(leaf[:2], leaf[2:], lungs.double(), ems.double(), cle, pse, cw.double(), pw.double())[0].backward()
    ok = ((at[0] + at[1]).double() - 1.0).abs() > 1e-3
    for i in (0, 1):
        assert rel_l2(ups[i][ok], leaf[i].grad[ok]) <= 2e-4, f"dense-gradient field {i}"
    for i in (2, 3):
        assert rel_l2(ups[i], leaf[i].grad) <= 1e-5, f"score gradient {i}"
    return loss, ups


R50_DRAM_GRAD_TOL = 1.5e-3
MID_GRAD_TOL = 2e-4


@pytest.mark.parametrize("factory", ["resnet34segcls", "resnet18segreg", "resnet50segreg"])
def test_mid_size_train_step_vs_oracle(factory):
    """1x64x128x128 (BASELINE configs[0] volume).  resnet34segcls IS configs[0] (conf/med3d.yaml:1, batch 1, class-
    weighted CE of reference models.py:253-258); the two regression networks run the
    full dRAM train loss through the fused loss kernels, on the
    library's OWN plan: S2 = 8x16x16 = 2,048 voxels, so for ResNet-50 the 1x1x1 convolutions run as plain
    GEMMs (plan 3) and the 2304->64 decoder convolution runs the Winograd pipeline unforced.
    Outputs 1e-3 vs the fp32 oracle; loss + its gradient fields (see _dram_loss_checks); then the network backward
    of exactly those fields: HIP vs the decision-pinned fp64 oracle given the same upstream, every parameter
    <= MID_GRAD_TOL (ResNet-18 / -34) or R50_DRAM_GRAD_TOL (absolute bars; the CPU fp32 oracle's own distance from fp64
    on the same piece is printed next to it: 2e-5 (R18) / 7e-5 (R50))."""
    from bodyct_dram_emph_subtype_amd import med3d, models
    dims = (64, 128, 128)
    cls = factory.endswith("cls")
    torch.manual_seed(5)
    m = getattr(med3d, factory)(**(dict(n_classes=[6, 3]) if cls else {}))
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    x, lungs = _synthetic(1, dims, 11)
    ems = ((x < -1.0).float() * lungs)
    cle, pse = torch.tensor([3]), torch.tensor([1])
    cw, pw = torch.tensor([0.3]), torch.tensor([0.6])
    with torch.no_grad():
        d32, o32 = orc.forward(dict(sd0), x, lungs, factory, train=True)
    md = m.to(DEV).train()
    dd, od = md(x.to(DEV), lungs.to(DEV))
    pins = pinned_decisions(dd[0])
    worst_out = 0.0
    for a, b in zip(od, o32):
        worst_out = max(worst_out, assert_close_rel(a.detach().cpu(), b, OUT_TOL, "class logits" if cls else "regression score"))
    for a, b in zip(dd, d32):
        worst_out = max(worst_out, assert_close_rel(a.detach().cpu(), b, OUT_TOL, "dense map"))
    if cls:
        cwt, pwt = torch.tensor([.1, .2, .1, .3, .2, .1]), torch.tensor([.5, .2, .3])
        loss = models.cls_train_loss(od, cle.to(DEV), pse.to(DEV), cwt.to(DEV), pwt.to(DEV))[0]
        ups = [None, None] + [u.detach().cpu() for u in torch.autograd.grad(loss, od, retain_graph=True)]
        l_ref = orc.cls_train_loss([t.detach().cpu() for t in od], cle, pse, cwt, pwt)[0]
        assert abs(float(loss) - float(l_ref)) < OUT_TOL * max(1.0, abs(float(l_ref)))
    else:
        loss, ups = _dram_loss_checks(models, dd, od, lungs, ems, cle, pse, cw, pw)
    loss.backward()
    g64 = _oracle_backward(sd0, names, factory, x, lungs, pins, ups, torch.float64)
    g32 = _oracle_backward(sd0, names, factory, x, lungs, pins, ups, torch.float32)
    worst = (0.0, 0.0, "")
    table = []
    for n, p in md.named_parameters():
        if is_noise_param(n):
            continue
        e, e_cpu = rel_l2(p.grad.double().cpu(), g64[n]), rel_l2(g32[n], g64[n])
        worst = max(worst, (e, e_cpu, n))
        table.append((e, e_cpu, n))
        # ABSOLUTE bars (the CPU oracle's own distance from fp64 moves with its thread count: a bar that floats with
        # it is not one): 2e-4 for the BasicBlock networks -- the full-size bar; measured 5.3-5.6e-5 while layer1 ran the
        # fused in-plane kernels and 1.2e-4 (layer1.0.bn2.weight, dRAM loss) on the round-5 plan, whose F(4,3)^3 tiles
        # round 10 x more per layer there.
        # (ResNet-50: below.)
        # ResNet-50 under the dRAM loss: 1.5e-3 for every tensor, as in rounds 2-4.  Its 54 BatchNorm backward passes each
        # cancel g - mean(g) - xhat * mean(g * xhat) on an outlier-dominated upstream field, and the error of the data-gradient
        # chain that arrives at the early layers moves with the rounding realisation: worst tensor 1.45e-4 on one build of
        # round 5, 8.1-8.2e-4 (conv1.weight / layer1.0.conv1.weight: everything upstream of layer1 alike) on two others that
        # differ from it only in the ORDER of the forward BatchNorm partial sums; 8.4-9.5e-4 in rounds 2-4.  (An attempt to
        # hold 3e-4 with three named stem outliers failed on the second build.)
        bar = R50_DRAM_GRAD_TOL if factory.startswith("resnet50") else MID_GRAD_TOL
        assert e <= bar, f"{n}: hip {e:.2e} vs decision-pinned fp64 oracle (CPU fp32: {e_cpu:.2e})"
    table.sort(reverse=True)
    print(f"[{factory} 1x64x128x128] per-tensor gradient errors, largest five (hip, cpu-fp32, tensor): {table[:5]}")
    print(f"[{factory} 1x64x128x128, {'CE' if cls else 'dRAM'} loss] worst output vs the fp32 oracle {worst_out:.2e}; worst gradient "
          f"error vs decision-pinned fp64 oracle (hip, cpu-fp32, tensor): {worst}")


FULL_CASES = {
    # id: (factory, batch, (D, H, W), loss)
    1: ("resnet18segcls", 2, (128, 256, 256), "ce"),       # BASELINE configs[1] as specified
    2: ("resnet18segreg", 1, (128, 256, 256), "dram"),     # configs[2] network + loss (fp32 storage; batch 1 halves the oracle's time)
    3: ("resnet50segreg", 1, (128, 256, 256), "smooth"),   # configs[3]: ResNet-50 + dRAM head, 1 volume per GPU
    5: ("resnet50segreg", 1, (128, 224, 288), "dram"),     # the reference's own defaults: train.py:21,30,42
}


@pytest.mark.slow
@pytest.mark.parametrize("config", sorted(FULL_CASES))
def test_full_size_train_step_vs_oracle(config):
    """One FULL-SIZE train step of BASELINE configs[1] (resnet18segcls, class-weighted CE), of configs[2]'s network
    and loss (resnet18segreg, dRAM loss; fp32 storage; batch 1), of configs[3]'s per-GPU work (resnet50segreg, one
    1x128x256x256 volume; smooth scalar objective through scores and dRAM maps -- the dRAM loss of a randomly
    initialised ResNet-50 is dominated by its clamp kink, see _dram_loss_checks) and of the reference's own default
    job (med3ddram50 at 1x128x224x288, reference train.py:21,30,42: S2 = 16x28x36, so the dilated stages run ragged
    F(4,3) Winograd tiles) -- the shapes bench.py times, so the kernels compared are the ones the plan picks at
    full size (4x4x4 Winograd tilings with >= 512 tiles, the two-workgroup in-plane Winograd variant, the
    slab-split TN GEMMs, the 1x1x1 GEMM plan, the tiled upsample+concat).
    Yardstick: the fp64 oracle on the HIP forward's own ReLU / max-pool decisions.  (The fp32 CPU oracle is not
    usable as one at this size: its weight-gradient sums over 10^6-10^7 voxels sit 1e-3 ... 1.4e-1 from fp64
    -- us3.0.weight 14 % -- where the HIP path, with double-precision statistic folds and blocked fp32
    accumulation, sits at 2-6e-5; tools/grad_pairs.py)
    Pooled outputs, dense maps and loss at 1e-3; the upstream gradients of (dense, outs) the loss hands back are
    propagated by both sides and EVERY parameter gradient must agree to 2e-4 relative L2 (ResNet-50 with the dRAM
    loss: 3x the fp32 oracle's own distance is not affordable here, so 2e-3 as at mid size); a second forward must
    reproduce the first bit for bit (no atomics anywhere)."""
    from bodyct_dram_emph_subtype_amd import med3d, models
    factory, B, dims, loss_kind = FULL_CASES[config]
    torch.manual_seed(0)
    kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
    m = getattr(med3d, factory)(**kw)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    x, lungs = _synthetic(B, dims, 1234)
    ems = ((x < -1.0).float() * lungs)
    cle, pse = torch.tensor([4, 1])[:B], torch.tensor([0, 2])[:B]

    md = m.to(DEV).train()
    with torch.no_grad():
        d_first, o_first = md(x.to(DEV), lungs.to(DEV))
    for k in [k for k in sd0 if "running" in k or "num_batches" in k]:      # undo the running-stat update of that run
        md.state_dict()[k].copy_(sd0[k])
    dd, od = md(x.to(DEV), lungs.to(DEV))
    assert torch.equal(o_first[0], od[0]) and torch.equal(d_first[1], dd[1])
    del d_first, o_first
    pins = pinned_decisions(dd[0])
    outs_hip = [o.detach().cpu() for o in od]
    dense_hip = [d.detach().cpu() for d in dd]
    if loss_kind == "ce":
        cwt, pwt = torch.full((6,), 1 / 6), torch.full((3,), 1 / 3)
        loss = models.cls_train_loss(od, cle.to(DEV), pse.to(DEV), cwt.to(DEV), pwt.to(DEV))[0]
        ups = [None, None] + [u.detach().cpu() for u in torch.autograd.grad(loss, od, retain_graph=True)]
        l_ref = orc.cls_train_loss(outs_hip, cle, pse, cwt, pwt)[0]
        assert abs(float(loss) - float(l_ref)) < OUT_TOL * max(1.0, abs(float(l_ref)))
    elif loss_kind == "smooth":
        loss = od[0].sum() * 0.7 - od[1].sum() * 1.3 + 0.1 * (dd[0] * dd[1]).mean()
        ups = [u.detach().cpu() for u in torch.autograd.grad(loss, dd + od, retain_graph=True)]
    else:
        loss, ups = _dram_loss_checks(models, dd, od, lungs, ems, cle, pse, torch.tensor([0.3, 0.2])[:B],
                                      torch.tensor([0.6, 0.1])[:B])
    loss.backward()
    torch.cuda.synchronize()
    got = {n: p.grad.double().cpu() for n, p in md.named_parameters()}
    loss_hip = float(loss)
    del md, dd, od, loss
    torch.cuda.empty_cache()

    dt = torch.float64
    lv = {k: (v.to(dt).requires_grad_(True) if k in names else (v.to(dt) if v.is_floating_point() else v))
          for k, v in sd0.items()}
    d, o = orc.forward(lv, x.to(dt), lungs.to(dt), factory, train=True, pins=pins)
    worst_out = 0.0
    for a, b in zip(outs_hip + dense_hip, o + d):
        worst_out = max(worst_out, assert_close_rel(a, b.detach(), OUT_HEADROOM, "pooled output / dense map"))
    pairs = [(t, u.to(dt)) for t, u in zip(d + o, ups) if u is not None]
    torch.autograd.backward([t for t, _ in pairs], [u for _, u in pairs])
    worst = (0.0, "")
    bar = 2e-3 if (factory.startswith("resnet50") and loss_kind == "dram") else 2e-4
    for n in names:
        if is_noise_param(n):
            continue
        e = rel_l2(got[n], lv[n].grad)
        worst = max(worst, (e, n))
        assert e <= bar, f"{n}: full-size gradient vs decision-pinned fp64 oracle {e:.2e}"
    print(f"[config {config} full size] loss {loss_hip:.6f}; worst output (pooled / dense, max-rel vs the pinned fp64 oracle) "
          f"{worst_out:.2e} (bar {OUT_HEADROOM:g}); worst gradient vs decision-pinned fp64 oracle {worst}")
    if config == 1:
        # decision INDEPENDENCE at full size (the headline configuration, where the full-size-only launch shapes run):
        # the free-running fp32 oracle's own ReLU / max-pool decisions against the ones the HIP forward took -- every
        # difference must be a rounding tie (audit_decisions), so the pinned comparison above cannot have pinned the
        # oracle to a wrong activation
        del lv, d, o, pairs
        flips, total, mp, nmp, margin = audit_decisions(pins, sd0, x, lungs, factory)
        print(f"[config 1 full size] {flips} of {total} ReLU decisions and {mp} of {nmp} max-pool taps differ from the "
              f"free-running fp32 oracle; worst flip margin {margin:.2e} x max|pre-activation|")


def test_inference_weight_cache_tracks_weight_identity_and_updates():
    """no_grad forwards reuse packed / Winograd-transformed weights.  The cache must never serve another tensor's
    packing (a freed weight's address recycled by a new model) nor a stale one (in-place torch update, fused
    optimizer step through raw pointers, load_state_dict)."""
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam
    x, lungs = make_inputs(3, (1, 1, 16, 32, 32))

    def check(m, what):
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        with torch.no_grad():
            d, o = m(x.to(DEV), lungs.to(DEV))
            d, o = m(x.to(DEV), lungs.to(DEV))            # second call: served from the cache
            dr, orf = orc.forward(sd, x, lungs, "resnet18segreg", train=False)
        assert_close_rel(o[0].cpu(), orf[0], OUT_TOL, what)
        assert_close_rel(d[1].cpu(), dr[1], OUT_TOL, what)

    for seed in (1, 2, 3):                                 # models die between iterations: addresses get recycled
        m = build("resnet18segreg", seed).to(DEV).eval()
        check(m, f"fresh model {seed}")
    with torch.no_grad():
        m.layer3[0].conv1.weight.mul_(1.5)                 # torch in-place update (version counter)
    check(m, "after in-place update")
    m.train()
    opt = FusedAdam(m.parameters(), lr=1e-2)
    dd, od = m(x.to(DEV), lungs.to(DEV))
    (od[0].sum() + od[1].sum()).backward()
    opt.step()                                             # raw-pointer update (WEIGHT_EPOCH)
    m.eval()
    check(m, "after a fused optimizer step")
    m.load_state_dict(build("resnet18segreg", 7).state_dict())
    check(m, "after load_state_dict")


@pytest.mark.slow
@pytest.mark.parametrize("recompute,storage", [(False, "f32"), (True, "f32"), (True, "bf16")],
                         ids=["keep", "recompute", "bf16-recompute"])
def test_config4_geometry_train_step_properties(recompute, storage):
    """BASELINE configs[4] geometry (resnet50segreg, 1x1x256x512x512 -- 8x the voxels of the headline shape): in fp32
    storage without / with activation recompute (~182 / 131 GB of HBM) and AS SPECIFIED -- bf16 storage with
    activation checkpointing (the per-GPU work of the 8-GPU job).  No CPU oracle finishes at this size, so the step is held
    to size-independent properties: finite loss and gradients for every parameter, dense maps inside [0, 1], the
    pooled scores equal to the lung-masked mean of the dense maps they come from (recomputed with torch from the
    returned volumes), BN running statistics moved, and a second identical step from the same state reproducing
    loss and gradients BIT FOR BIT (32-bit offsets, tile counts and workspaces all exercised at 8x scale)."""
    from bodyct_dram_emph_subtype_amd import med3d, models
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs the 288 GB of an MI355X")
    dims = (256, 512, 512)
    torch.manual_seed(0)
    m = med3d.resnet50segreg().to(DEV).train()
    m.activation_recompute = recompute
    if storage == "bf16":
        m.storage_dtype = torch.bfloat16
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    g = torch.Generator(device=DEV).manual_seed(1234)
    x = torch.randn(1, 1, *dims, device=DEV, generator=g)
    D, H, W = dims
    z = (torch.arange(D, device=DEV).float() - (D - 1) / 2) / (0.4 * D)
    y = (torch.arange(H, device=DEV).float() - (H - 1) / 2) / (0.35 * H)
    xx = (torch.arange(W, device=DEV).float() - (W - 1) / 2) / (0.4 * W)
    lungs = ((z[:, None, None] ** 2 + y[None, :, None] ** 2 + xx[None, None, :] ** 2) <= 1.0).float()[None, None].contiguous()
    ems = ((x < -1.0).float() * lungs)
    cle, pse = torch.tensor([3], device=DEV), torch.tensor([1], device=DEV)
    cw, pw = torch.tensor([0.3], device=DEV), torch.tensor([0.6], device=DEV)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}

    def step():
        m.load_state_dict(sd0)
        m.zero_grad(set_to_none=True)
        dense, outs = m(x, lungs)
        loss, _ = models.reg_train_loss(dense, outs, lungs, ems, cle, pse, cw, pw)
        loss.backward()
        return dense, outs, loss.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}

    dense, outs, loss, grads = step()
    assert torch.isfinite(loss)
    assert all(bool(torch.isfinite(v).all()) for v in grads.values())
    assert float(grads["conv1.weight"].norm()) > 0 and float(grads["layer4.2.conv3.weight"].norm()) > 0
    lg = torch.nn.functional.interpolate(lungs, size=dense[0].shape[-3:], mode="nearest")
    for d, o in zip(dense, outs):
        assert float(d.min()) >= 0.0 and float(d.max()) <= 1.0      # sigmoid saturates to exactly 1.0 in fp32
        ref = (d.double() * lg).sum() / lg.double().sum()
        assert abs(float(o) - float(ref)) < 1e-5 * abs(float(ref))      # (the dense maps are fp32 on both storage paths)
    assert float((m.state_dict()["bn1.running_mean"] - sd0["bn1.running_mean"]).abs().max()) > 0
    peak = torch.cuda.max_memory_allocated() / 1e9
    del dense, outs
    _, _, loss2, grads2 = step()
    assert torch.equal(loss, loss2)
    for n in grads:
        assert torch.equal(grads[n], grads2[n]), n
    print(f"[configs[4] geometry, {storage} storage, activation recompute {recompute}] loss {float(loss):.6f}, peak HBM {peak:.0f} GB")


@pytest.mark.parametrize("factory,shape", [("resnet18segreg", (2, 1, 16, 32, 32)), ("resnet50segcls", (1, 1, 16, 32, 32)),
                                           ("resnet34segreg", (1, 1, 32, 64, 64))])
def test_activation_recompute_is_bit_identical_and_smaller(factory, shape):
    """module.activation_recompute = True: backward re-derives intra-block BN+ReLU outputs, the upsample+concat
    tensors and the Winograd-domain images instead of keeping them (the build-side "activation checkpointing" of
    BASELINE configs[4]).  Same kernels on the same inputs -> every gradient, output and running statistic must
    be BIT-identical to the default mode; the memory held between forward and backward must be smaller."""
    x, lungs = make_inputs(17, shape)
    res = {}
    for mode in (False, True):
        m = build(factory, 4).to(DEV).train()
        m.activation_recompute = mode
        import gc
        gc.collect()                      # (reference cycles of earlier tests would be collected -- and their tensors freed -- mid-forward)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        dense, outs = m(x.to(DEV), lungs.to(DEV))
        held = torch.cuda.memory_allocated() - base
        pins = pinned_decisions(dense[0])
        loss = (outs[0].sum() * 0.7 - outs[1].sum() * 1.3 + 0.1 * (dense[0] * dense[0]).mean())
        loss.backward()
        torch.cuda.synchronize()
        res[mode] = (loss.detach().cpu(), {n: p.grad.cpu() for n, p in m.named_parameters()},
                     {k: v.cpu() for k, v in m.state_dict().items() if "running" in k}, held, pins)
    assert torch.equal(res[False][0], res[True][0])
    for n in res[False][1]:
        assert torch.equal(res[False][1][n], res[True][1][n]), n
    for k in res[False][2]:
        assert torch.equal(res[False][2][k], res[True][2][k]), k
    for k in res[False][4]:
        assert torch.equal(res[False][4][k], res[True][4][k]), k          # exported decisions agree too
    assert res[True][3] < res[False][3], (res[True][3], res[False][3])      # 8-18 % at fixture size, 28 % at configs[4]
    print(f"[{factory} {shape}] held between forward and backward: {res[False][3] / 1e6:.1f} MB -> {res[True][3] / 1e6:.1f} MB")
