"""GPU: the LightningModule mirrors (models.py) -- train/val steps, predict-time dRAM
up-projection, configure_optimizers -- against the CPU oracle restating reference
models.py:236-276, :430-450, :539-592, and the metrics drop-ins."""
import os
import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from oracle import med3d_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _batch(B, dims, seed):
    g = torch.Generator().manual_seed(seed)
    image = torch.randn(B, *dims, generator=g)
    lung = (torch.rand(B, *dims, generator=g) > 0.35)
    em = ((image < -0.5) & lung)
    ess = ((image < 0.0) & lung)
    return {"image": image, "lung_mask": lung, "em_mask": em, "ess_mask": ess,
            "cls_label": torch.randint(0, 6, (B,), generator=g), "pse_label": torch.randint(0, 3, (B,), generator=g),
            "index": torch.arange(B).unsqueeze(-1), "uid": [f"case{i}" for i in range(B)]}


def _to(batch, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def test_reg_module_train_step_and_optimizer():
    from bodyct_dram_emph_subtype_amd import models
    torch.manual_seed(7)
    mod = models.ScanRegLightningModule(models.make_args("med3ddram18", lr=1e-4))
    sd0 = {k: v.clone() for k, v in mod.model.state_dict().items()}
    mod.cle_class_weights = torch.tensor([0.3, 0.1, 0.2, 0.1, 0.2, 0.1])
    mod.pse_class_weights = torch.tensor([0.5, 0.2, 0.3])
    batch = _batch(2, (16, 32, 32), 3)
    mod = mod.to(DEV).train()
    (opt,), (sched,) = mod.configure_optimizers()
    out = mod.training_step(_to(batch, DEV), 0)
    out["loss"].backward()
    opt.step()
    sched.step()
    assert abs(opt.param_groups[0]["lr"] - 0.95e-4) < 1e-12        # ExponentialLR(gamma=0.95)
    # oracle: same weights, reference loss assembly (models.py:549-574)
    x = batch["image"].unsqueeze(1)
    lungs = batch["lung_mask"].unsqueeze(1).float()
    ems = batch["em_mask"].unsqueeze(1).float()
    dense, outs = orc.forward(sd0, x, lungs, "resnet18segreg", train=True)
    cw = mod.cle_class_weights[batch["cls_label"]]
    pw = mod.pse_class_weights[batch["pse_label"]]
    loss, parts = orc.reg_train_loss(dense, outs, lungs, ems, batch["cls_label"], batch["pse_label"], cw, pw)
    assert abs(float(out["loss"]) - float(loss)) < 1e-3 * max(1.0, abs(float(loss)))
    assert torch.equal(out["pred_cle_labels"].cpu(), orc.ratio_to_label(outs[0], orc.CLE_RATIO_MAP))
    assert torch.equal(out["pred_pse_labels"].cpu(), orc.ratio_to_label(outs[1], orc.PSE_RATIO_MAP))
    # validation step: no grad, no loss key, BN in eval mode does not touch the running stats
    mod.eval()
    nb = int(mod.model.state_dict()["bn1.num_batches_tracked"])
    vout = mod.validation_step(_to(batch, DEV), 0)
    assert "loss" not in vout and int(mod.model.state_dict()["bn1.num_batches_tracked"]) == nb


def test_cls_module_train_step():
    from bodyct_dram_emph_subtype_amd import models
    torch.manual_seed(8)
    mod = models.ScanCLSLightningModule(models.make_args("med3d18", lr=1e-4))
    sd0 = {k: v.clone() for k, v in mod.model.state_dict().items()}
    batch = _batch(2, (16, 32, 32), 4)
    mod = mod.to(DEV).train()
    out = mod.training_step(_to(batch, DEV), 0)
    out["loss"].backward()
    x = batch["image"].unsqueeze(1)
    dense, outs = orc.forward(sd0, x, batch["lung_mask"].unsqueeze(1).float(), "resnet18segcls", train=True)
    loss, _ = orc.cls_train_loss(outs, batch["cls_label"], batch["pse_label"], torch.full((6,), 1 / 6),
                                 torch.full((3,), 1 / 3))
    assert abs(float(out["loss"]) - float(loss)) < 1e-3 * max(1.0, abs(float(loss)))
    assert torch.equal(out["pred_cle_labels"].cpu(), outs[0].argmax(-1))
    g = dict(mod.model.named_parameters())["fcs.0.bias"].grad
    assert g is not None and float(g.abs().sum()) > 0


def test_predict_step_up_projection():
    """models.py:430-450: eval forward, trilinear(align_corners) to the scan grid x ess,
    percentages over lungs.sum() of the WHOLE batch."""
    from bodyct_dram_emph_subtype_amd import models
    torch.manual_seed(9)
    mod = models.ScanRegLightningModule(models.make_args("med3ddram18"))
    sd0 = {k: v.clone() for k, v in mod.model.state_dict().items()}
    batch = _batch(2, (16, 32, 32), 5)
    mod = mod.to(DEV).eval()
    res = mod.predict_step(_to(batch, DEV), 0)
    x = batch["image"].unsqueeze(1)
    lungs = batch["lung_mask"].unsqueeze(1).float()
    ess = batch["ess_mask"].unsqueeze(1).float()
    dense, _ = orc.forward(sd0, x, lungs, "resnet18segreg", train=False)
    for name, d in (("cle", dense[0]), ("pse", dense[1])):
        up, pct = orc.predict_upproject(d, x.shape[-3:], ess, lungs)
        assert tuple(res[f"{name}_dense_outs"].shape) == tuple(up.shape)
        assert rel_l2(res[f"{name}_dense_outs"].cpu(), up) < 1e-3
        assert np.allclose(res[f"{name}_precentages"].cpu().numpy(), pct.numpy(), rtol=1e-3)
    assert res["uids"] == batch["uid"]


def test_metrics_dropins():
    from bodyct_dram_emph_subtype_amd import metrics
    g = torch.Generator().manual_seed(2)
    a = torch.rand(2, 1, 4, 8, 8, generator=g)
    b = torch.rand(2, 1, 4, 8, 8, generator=g)
    mask = (torch.rand(2, 1, 4, 8, 8, generator=g) > 0.4).float()
    tgt = (torch.rand(2, 1, 4, 8, 8, generator=g) > 0.7).float()
    d = metrics.BinaryDice(1e-7)(a.to(DEV), b.to(DEV))
    assert abs(float(d) - float(orc.dice_coef(a, b))) < 1e-5
    # the reference's call signatures (metrics.py:10, :33, :40-47), any smooth / smoothness, mask optional,
    # differentiable w.r.t. the predictions
    for smooth in (1e-7, 1.0):
        ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        d = metrics.BinaryDice(smooth)(ad, bd)
        r = orc.dice_coef(ac, bc, smooth)
        assert abs(float(d) - float(r)) < 1e-5
        d.backward()
        r.backward()
        assert rel_l2(ad.grad.cpu(), ac.grad) < 1e-4 and rel_l2(bd.grad.cpu(), bc.grad) < 1e-4
    assert abs(float(metrics.dice_coef(a.to(DEV), b.to(DEV), 0.5)) - float(orc.dice_coef(a, b, 0.5))) < 1e-5
    p = torch.clamp(0.4 * a + 0.4 * b, 0, 1)
    for m, kw in ((mask, dict(smoothness=0.85)), (mask, {}), (None, {}), (None, dict(smoothness=0.3))):
        pd = p.to(DEV).requires_grad_(True)
        pc = p.clone().requires_grad_(True)
        args = (tgt.to(DEV), pd) + ((m.to(DEV),) if m is not None else ())
        bce = metrics.BinaryCrossEntropy()(*args, **kw)          # exactly the models.py:529 call form
        ref = orc.balanced_bce(tgt, pc, m, smoothness=kw.get("smoothness", 0.65))
        assert abs(float(bce) - float(ref)) < 1e-5 * max(1.0, abs(float(ref))), (kw, float(bce), float(ref))
        bce.backward()
        ref.backward()
        assert rel_l2(pd.grad.cpu(), pc.grad) < 1e-4


def test_trainer_harness_two_epochs_and_resume(tmp_path):
    """End to end through train.run_training_job on synthetic batches: per-epoch checkpoints named
    like Lightning's '{epoch:02d}', loss finite, full resume continues at the next epoch."""
    import os
    from bodyct_dram_emph_subtype_amd import train
    argv = ["--model_arch", "med3ddram18", "--model_path", str(tmp_path), "--num_samples", "2", "--batch_size", "1",
            "--target_size", "16", "32", "32", "--max_epochs", "2", "--log_every_n_steps", "1"]
    mod = train.run_training_job(argv)
    ck = tmp_path / "subtyping_med3ddram18" / "checkpoints"
    assert sorted(os.listdir(ck)) == ["epoch=00.ckpt", "epoch=01.ckpt"]
    c1 = torch.load(ck / "epoch=01.ckpt", map_location="cpu", weights_only=False)
    assert c1["epoch"] == 1 and c1["global_step"] == 4
    assert abs(c1["optimizer_states"][0]["param_groups"][0]["lr"] - 1e-4 * 0.95 ** 2) < 1e-12
    assert all(torch.isfinite(v).all() for v in c1["state_dict"].values() if v.is_floating_point())
    mod2 = train.run_training_job(argv[:-4] + ["--max_epochs", "3", "--reload_only_weights", "0"])
    assert sorted(os.listdir(ck))[-1] == "epoch=02.ckpt"
    assert int(mod2.model.state_dict()["bn1.num_batches_tracked"]) == 6      # 3 epochs x 2 steps, buffers resumed


# ------------------------------------------------------------------ GPU input transforms (models.py:59-63)
def test_input_transforms_match_reference_golden():
    from conftest import GOLDEN
    import os
    from bodyct_dram_emph_subtype_amd import transforms as T
    z = np.load(os.path.join(GOLDEN, "transforms.npz"))
    tgt = tuple(int(v) for v in z["target"])
    img = T.prepare_image(torch.from_numpy(z["scan"]).cuda(), tgt).cpu()
    ref = torch.from_numpy(z["image_out"])
    assert img.shape == ref.shape
    assert float((img - ref).abs().max()) <= 2e-5 * float(ref.abs().max())     # fp32 z-score + bilinear
    msk = T.prepare_mask(torch.from_numpy(z["mask"]).cuda(), tgt).cpu()
    assert msk.dtype == torch.bool and torch.equal(msk, torch.from_numpy(z["mask_out"]))


@pytest.mark.parametrize("src,tgt", [((16, 40, 56), (16, 40, 56)),       # identity size
                                      ((9, 33, 47), (24, 64, 96)),        # up-sampling, ragged
                                      ((70, 128, 96), (32, 48, 40)),      # down-sampling
                                      ((1, 8, 8), (8, 16, 16))])          # single slice
def test_input_transforms_match_oracle(src, tgt):
    from bodyct_dram_emph_subtype_amd import transforms as T
    g = torch.Generator().manual_seed(sum(src) + sum(tgt))
    scan = torch.rand(src, generator=g) * 1600.0 - 1400.0
    lung = torch.rand(src, generator=g) > 0.4
    lab = torch.randint(0, 5, src, generator=g).to(torch.int16)
    out = T.prepare_sample({"image": scan.cuda(), "lung_mask": lung.cuda(), "lesion_mask": lab.cuda(),
                            "cls_label": 3}, tgt)
    ref = orc.prepare_image(scan, tgt)
    assert float((out["image"].cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(out["lung_mask"].cpu(), orc.prepare_mask(lung, tgt))
    assert out["lesion_mask"].dtype == torch.int16
    assert torch.equal(out["lesion_mask"].cpu(), orc.prepare_mask(lab, tgt))
    assert out["cls_label"] == 3


def test_processor_resample_paste_matches_reference_fixture():
    """processor.py:111-129, :143 as one gather kernel, against the fixture recorded from the reference statements."""
    from bodyct_dram_emph_subtype_amd import processor
    g = np.load(os.path.join(GOLDEN, "processor.npz"))
    dense = torch.from_numpy(g["dense"][0]).to(DEV)
    f32, u8 = processor.resample_paste(dense, torch.from_numpy(g["crop"]), torch.from_numpy(g["original"]), True, True)
    assert tuple(f32.shape) == tuple(g["original"])
    assert float((f32.cpu() - torch.from_numpy(g["full"])).abs().max()) < 2e-6
    outside = torch.from_numpy(g["full"]) == 0
    assert torch.equal(f32.cpu()[outside], torch.zeros(int(outside.sum())))          # exact zeros outside the crop box
    d = (u8.cpu().int() - torch.from_numpy(g["full_u8"]).int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-3                # truncation at an integer boundary
    with pytest.raises(ValueError):
        processor.resample_paste(dense, [[0, 50], [0, 10], [0, 10]], (40, 61, 75))


def test_augmentations_match_reference_fixture():
    """models.py:66-74 with given parameters: each prefix of the chain noise -> boxes -> flip -> crop-resize through
    the ONE fused kernel equals the reference classes applied one after the other."""
    from bodyct_dram_emph_subtype_amd.transforms import AugmentParams, augment_image, augment_mask
    g = np.load(os.path.join(GOLDEN, "augment.npz"))
    img, mask = torch.from_numpy(g["image"]).to(DEV), torch.from_numpy(g["mask"]).to(DEV)
    torch.manual_seed(int(g["noise_seed"]))
    noise = torch.randn(g["image"].shape).to(DEV)
    cen, siz = [tuple(c) for c in g["box_centers"]], [tuple(c) for c in g["box_sizes"]]
    cc, cs = tuple(g["crop_center"]), tuple(g["crop_size"])
    chain = [(AugmentParams(noise_sigma=float(g["noise_sigma"])), "after_noise"),
             (AugmentParams(noise_sigma=float(g["noise_sigma"]), box_centers=cen, box_sizes=siz), "after_box"),
             (AugmentParams(noise_sigma=float(g["noise_sigma"]), box_centers=cen, box_sizes=siz, flip_dims=(2, 0)), "after_flip"),
             (AugmentParams(noise_sigma=float(g["noise_sigma"]), box_centers=cen, box_sizes=siz, flip_dims=(2, 0),
                            crop_center=cc, crop_size=cs), "after_crop")]
    for ap, key in chain:
        out = augment_image(img, ap, noise).cpu()
        # the resampling stage recomputes the affine grid in fp32 (coordinates differ from ATen's in the last bit:
        # values of magnitude ~4 move by a few 1e-6)
        assert float((out - torch.from_numpy(g[key])).abs().max()) < (2e-5 if key == "after_crop" else 2e-6), key
    m = augment_mask(mask, AugmentParams(flip_dims=(2, 0))).cpu()
    assert torch.equal(m, torch.from_numpy(g["mask_after_flip"]))
    m = augment_mask(mask, chain[-1][0]).cpu()
    assert torch.equal(m, torch.from_numpy(g["mask_after_crop"]))
    assert augment_mask(mask, AugmentParams(noise_sigma=0.05)) is mask                # image-only transforms leave masks alone


def test_graphed_train_step_equals_eager_steps():
    """graph.GraphedTrainStep: forward + CE loss + backward + FusedAdam(capturable) captured in one hipGraph and
    replayed.  Weights, BN running statistics, num_batches_tracked and Adam state after 3 replays (with an lr
    change by ExponentialLR in between) must equal the eager capturable path bit for bit, and that path must
    match plain FusedAdam to optimizer-arithmetic tolerance."""
    from bodyct_dram_emph_subtype_amd import med3d
    from bodyct_dram_emph_subtype_amd.graph import GraphedTrainStep
    from bodyct_dram_emph_subtype_amd.models import cls_train_loss
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randn(2, 1, 16, 32, 32, generator=g).to(DEV), (torch.rand(2, 1, 16, 32, 32, generator=g) > 0.3).float().to(DEV),
                torch.randint(0, 6, (2,), generator=g).to(DEV), torch.randint(0, 3, (2,), generator=g).to(DEV)) for _ in range(4)]
    cw, pw = torch.full((6,), 1 / 6, device=DEV), torch.full((3,), 1 / 3, device=DEV)

    def run(mode):
        torch.manual_seed(11)
        m = med3d.resnet18segcls(n_classes=[6, 3]).to(DEV).train()
        opt = FusedAdam(m.parameters(), lr=1e-3, capturable=(mode != "plain"))
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5)

        def loss_fn(image, lung, cle, pse):
            return cls_train_loss(m(image, lung)[1], cle, pse, cw, pw)[0]

        def eager(b):
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(*b)
            loss.backward()
            opt.step()
            return loss.detach().clone()
        losses = []
        if mode.startswith("graph"):
            # the constructor runs 2 eager warm-up steps on the example batch: mirror them in the other modes
            # ("graph2": the capture keeps the weight-gradient side stream as a second branch of the graph)
            step = GraphedTrainStep(m, opt, loss_fn, batches[0], warmup=2, streams=2 if mode == "graph2" else 1)
            assert m._engine.graph_streams == 1          # (the setting is the capture's, not the engine's)
        else:
            eager(batches[0]); eager(batches[0])
            step = lambda *b: eager(b)
        for i, b in enumerate(batches[1:]):
            losses.append(step(*b).clone())
            if i == 0:
                sched.step()                       # lr 1e-3 -> 5e-4 must reach the captured update
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        st = opt.state_dict()["state"]
        return losses, sd, st

    l_g, sd_g, st_g = run("graph")
    l_g2, sd_g2, st_g2 = run("graph2")
    for a, b in zip(l_g, l_g2):
        assert torch.equal(a, b)
    for k in sd_g:
        assert torch.equal(sd_g[k], sd_g2[k]), k
    l_e, sd_e, st_e = run("eager-capturable")
    l_p, sd_p, st_p = run("plain")
    print("graph", [float(v) for v in l_g], "eager-capturable", [float(v) for v in l_e], "plain", [float(v) for v in l_p])
    for a, b in zip(l_g, l_e):
        assert torch.equal(a, b)
    for k in sd_e:
        assert torch.equal(sd_g[k], sd_e[k]), k
    assert int(sd_g["bn1.num_batches_tracked"]) == 5
    assert float(st_g[0]["step"]) == float(st_e[0]["step"]) == 5.0
    assert torch.equal(st_g[0]["exp_avg"].cpu(), st_e[0]["exp_avg"].cpu())
    # the device-side bias corrections (float betas, double pow in the kernel) against the host-side ones of the
    # plain optimizer: first compared loss (after two updates) within 5e-4; later steps diverge chaotically
    # (Adam's normalised steps flip ReLU decisions), so they are not compared
    assert abs(float(l_e[0]) - float(l_p[0])) < 5e-4 * abs(float(l_p[0]))
    # one update from identical state: parameter-for-parameter agreement
    def one_step(capturable):
        torch.manual_seed(11)
        m = med3d.resnet18segcls(n_classes=[6, 3]).to(DEV).train()
        opt = FusedAdam(m.parameters(), lr=1e-3, capturable=capturable)
        loss = cls_train_loss(m(batches[0][0], batches[0][1])[1], batches[0][2], batches[0][3], cw, pw)[0]
        loss.backward()
        opt.step()
        return {k: v.detach().cpu() for k, v in m.state_dict().items()}
    a, b = one_step(True), one_step(False)
    for k in a:
        if a[k].is_floating_point():
            assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-7), k


def test_eval_forward_after_graph_replays_sees_the_updated_weights():
    """The replayed Adam update rewrites every weight through raw pointers (no torch version counter moves): the
    packed / Winograd-transformed weights cached for no_grad forwards must be invalidated by the replay.  An eval
    forward after N replays has to equal, bit for bit, the eval forward of a FRESH module loaded with the same
    state_dict -- and differ from the eval forward before the replays."""
    from bodyct_dram_emph_subtype_amd import med3d
    from bodyct_dram_emph_subtype_amd.graph import GraphedTrainStep
    from bodyct_dram_emph_subtype_amd.models import cls_train_loss
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(5)
    image = torch.randn(2, 1, 16, 32, 32, generator=g).to(DEV)
    lung = (torch.rand(2, 1, 16, 32, 32, generator=g) > 0.3).float().to(DEV)
    cle, pse = torch.randint(0, 6, (2,), generator=g).to(DEV), torch.randint(0, 3, (2,), generator=g).to(DEV)
    cw, pw = torch.full((6,), 1 / 6, device=DEV), torch.full((3,), 1 / 3, device=DEV)
    torch.manual_seed(12)
    m = med3d.resnet18segcls(n_classes=[6, 3]).to(DEV).train()
    opt = FusedAdam(m.parameters(), lr=1e-2, capturable=True)
    step = GraphedTrainStep(m, opt, lambda i, l, c, p: cls_train_loss(m(i, l)[1], c, p, cw, pw)[0], (image, lung, cle, pse))

    def eval_out(mod):
        mod.eval()
        with torch.no_grad():
            dense, outs = mod(image, lung)
        mod.train()
        return [t.clone() for t in dense + outs]
    before = eval_out(m)               # fills the packed-weight cache
    for _ in range(3):
        step(image, lung, cle, pse)
    after = eval_out(m)
    torch.cuda.synchronize()
    fresh = med3d.resnet18segcls(n_classes=[6, 3]).to(DEV)
    fresh.load_state_dict(m.state_dict())
    want = eval_out(fresh)
    for a, w in zip(after, want):
        assert torch.equal(a, w), "eval forward after graph replays used stale packed weights"
    assert not torch.equal(before[2], after[2])


def test_steps_issued_without_host_sync_are_throttled_and_equal_synchronised_steps():
    """A training loop that never reads the loss: the engine lets the host issue at most one step ahead of the one
    the GPU executes (engine._throttle; the optimizer's pointer table goes up through pinned memory, asynchronously,
    so nothing else blocks the host) -- and six such steps end in exactly the weights, running statistics and Adam
    state of six steps with a device synchronisation after each."""
    from bodyct_dram_emph_subtype_amd import med3d
    from bodyct_dram_emph_subtype_amd.models import cls_train_loss
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(5)
    batches = [(torch.randn(2, 1, 16, 32, 32, generator=g).to(DEV), torch.randint(0, 6, (2,), generator=g).to(DEV),
                torch.randint(0, 3, (2,), generator=g).to(DEV)) for _ in range(6)]
    cw, pw = torch.full((6,), 1 / 6, device=DEV), torch.full((3,), 1 / 3, device=DEV)

    def run(sync):
        torch.manual_seed(13)
        m = med3d.resnet18segcls(n_classes=[6, 3]).to(DEV).train()
        opt = FusedAdam(m.parameters(), lr=1e-3)
        most = 0
        for image, cle, pse in batches:
            opt.zero_grad(set_to_none=True)
            cls_train_loss(m(image, None)[1], cle, pse, cw, pw)[0].backward()
            opt.step()
            most = max(most, len(m._engine._inflight))
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        state = {k: v.clone() for k, v in m.state_dict().items()}
        state.update({f"adam{i}.{k}": v.clone() for i, p in enumerate(m.parameters())
                      for k, v in opt.state[p].items() if k != "step"})
        return state, most

    free, most = run(False)
    ref, _ = run(True)
    assert most <= 2                       # the step being issued + one ahead
    for k in ref:
        assert torch.equal(free[k], ref[k]), k
