"""Golden fixtures for the SURVEY.md §8(f) rows, produced by running the REFERENCE's own code on CPU
(build container only; /root/reference does not exist on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_f.py

  processor.npz  -- processor.py:111-129 (resample the dRAM to the crop size, paste into the original grid,
                    uint8 windowing through the reference's utils.windowing) and processor.py:34-38 /
                    :130-136 (ratio_to_label severity score), called / restated statement by statement
  epoch_end.npz  -- models.py:287-317 shared_epoch_end (all-gather + de-duplication by sample index, run on a
                    1-rank gloo group) and models.py:367-379 (dynamic class-weight update), by calling the
                    reference ScanCLSLightningModule methods on a bare instance with plotting / csv stubbed
  augment.npz    -- models.py:66-74 train-time augmentations through the reference's own GaussianAddictive /
                    BoxMaskOut / Flip / CropAndResize classes with FIXED parameters (their randomness is only
                    in get_params; the noise tensor comes from torch.manual_seed)
Only data is committed -- never reference source.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from make_golden import import_ref_models  # noqa: E402  (also puts /root/reference on sys.path)

rm = import_ref_models()


def mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ------------------------------------------------------------------ processor.py
mod("pytorch_lightning.strategies", DDPStrategy=object)
sys.modules["matplotlib"].use = lambda *a, **k: None
import processor as ref_proc   # noqa: E402
import utils as ref_utils      # noqa: E402
from dataset import COPDGeneSubtyping  # noqa: E402


def processor_case():
    g = torch.Generator().manual_seed(77)
    dense = torch.rand(1, 16, 24, 32, generator=g)            # one sample of predict_step's cle_dense_outs [1,D,H,W]
    crop = torch.tensor([[5, 33], [11, 58], [3, 70]])         # crop_slice [[z0,z1],[y0,y1],[x0,x1]]
    original = (40, 61, 75)
    # processor.py:115-122, statement by statement
    recon_size = tuple(s[1].item() - s[0].item() for s in crop)
    up = torch.nn.functional.interpolate(dense.unsqueeze(0), size=recon_size, mode="trilinear", align_corners=True)
    up_np = up.squeeze(0).squeeze(0).cpu().numpy()
    full = np.zeros(original)
    full[tuple([slice(s[0].item(), s[1].item()) for s in crop])] = up_np
    full_w = ref_utils.windowing(full, from_span=(0, 1)).astype(np.uint8)     # processor.py:143
    pcts = np.array([0.0, 0.004, 0.01, 0.0499, 0.05, 0.12, 0.2, 0.29999, 0.3, 0.77, 1.0])
    cle_scores = np.array([ref_proc.ratio_to_label(float(p), COPDGeneSubtyping.cle_ratio_map) for p in pcts])
    pse_scores = np.array([ref_proc.ratio_to_label(float(p), COPDGeneSubtyping.pse_ratio_map) for p in pcts])
    return dict(dense=dense.numpy(), crop=crop.numpy(), original=np.array(original), full=full.astype(np.float32),
                full_u8=full_w, pcts=pcts, cle_scores=cle_scores, pse_scores=pse_scores)


# ------------------------------------------------------------------ epoch end
def epoch_end_case():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29641")
    dist.init_process_group("gloo", rank=0, world_size=1)
    rm.plot_confusion_matrix_from_data = lambda *a, **k: None
    rm.plot_to_numpy_array = lambda *a, **k: np.zeros((2, 2, 3), dtype=np.uint8)
    rm.save_image = lambda *a, **k: None
    seen = {}

    class _M(rm.ScanCLSLightningModule):
        def __init__(self):                      # bare instance: no network needed for the epoch-end logic
            torch.nn.Module.__init__(self)

        def _log_csv(self, p0, p1, y0, y1, idx, phase):
            seen.update(pred_cle=p0.copy(), pred_pse=p1.copy(), cle=y0.copy(), pse=y1.copy(), indices=idx.copy())

        def log(self, name, value, **k):
            seen[name] = float(value)

        tb_logger = property(lambda self: types.SimpleNamespace(experiment=types.SimpleNamespace(add_image=lambda **k: None)))

    m = _M()
    tmp = tempfile.mkdtemp()
    train_ds = types.SimpleNamespace(cle_class_weights=np.array([0.2, 0.1, 0.15, 0.25, 0.1, 0.2]),
                                     pse_class_weights=np.array([0.5, 0.2, 0.3]))
    m.trainer = types.SimpleNamespace(default_root_dir=tmp, current_epoch=3,
                                      datamodule=types.SimpleNamespace(datasets={rm.TRAIN_PHASE: train_ds}))
    g = torch.Generator().manual_seed(9)
    n = 40
    # step outputs with DUPLICATED sample indices (the distributed sampler pads ranks to equal length)
    idx = torch.cat([torch.randperm(30, generator=g), torch.randint(0, 30, (n - 30,), generator=g)])
    cle = torch.randint(0, 6, (30,), generator=g)[idx]
    pse = torch.randint(0, 3, (30,), generator=g)[idx]
    pc = torch.where(torch.rand(n, generator=g) < 0.6, cle, torch.randint(0, 6, (n,), generator=g))
    pp = torch.where(torch.rand(n, generator=g) < 0.7, pse, torch.randint(0, 3, (n,), generator=g))
    outs = [dict(pred_cle_labels=pc[i:i + 8], cle_labels=cle[i:i + 8], pred_pse_labels=pp[i:i + 8],
                 pse_labels=pse[i:i + 8], index=idx[i:i + 8]) for i in range(0, n, 8)]
    w0 = dict(cle=train_ds.cle_class_weights.copy(), pse=train_ds.pse_class_weights.copy())
    m.shared_epoch_end(outs, rm.TRAIN_PHASE)
    dist.destroy_process_group()
    rec = dict(index=idx.numpy(), cle=cle.numpy(), pse=pse.numpy(), pred_cle=pc.numpy(), pred_pse=pp.numpy(),
               w_cle_before=w0["cle"], w_pse_before=w0["pse"],
               w_cle_after=np.asarray(train_ds.cle_class_weights), w_pse_after=np.asarray(train_ds.pse_class_weights),
               acc_cle=np.array(seen[f"epoch_{rm.TRAIN_PHASE}_acc_cle"]), acc_pse=np.array(seen[f"epoch_{rm.TRAIN_PHASE}_acc_pse"]))
    for k in ("pred_cle", "pred_pse", "cle", "pse", "indices"):
        rec["dedup_" + k] = seen[k]
    return rec


# ------------------------------------------------------------------ augmentations
def augment_case():
    g = torch.Generator().manual_seed(123)
    img = torch.randn(12, 20, 28, generator=g)
    mask = (torch.rand(12, 20, 28, generator=g) > 0.6).float()
    rec = dict(image=img.numpy(), mask=mask.numpy())
    ga = rm.GaussianAddictive(p=1.0, always_apply=True)
    ga.params = {"sigma": 0.045}
    torch.manual_seed(4242)                                   # the transform draws torch.randn(data.shape) itself
    a1 = ga.apply_to_image(img.clone())
    rec.update(noise_sigma=np.array(0.045), noise_seed=np.array(4242), after_noise=a1.numpy())
    bm = rm.BoxMaskOut(p=1.0, always_apply=True, n_masks=(1, 10))
    centers = [(0.3, 0.5, 0.7), (0.62, 0.25, 0.41), (0.8, 0.8, 0.2)]
    sizes = [(0.3, 0.2, 0.25), (0.2, 0.35, 0.1), (0.5, 0.1, 0.3)]   # larger than the reference's range: visible boxes at this size
    bm.params = {"n_masks": 3, "mask_centers": centers, "mask_sizes": sizes}
    a2 = bm.apply_to_image(a1)
    rec.update(box_centers=np.array(centers), box_sizes=np.array(sizes), after_box=a2.numpy())
    fl = rm.Flip(1.0, True, dim=(1, 3))
    fl.params = {"combs": [2, 0]}
    a3, m3 = fl.apply_to_image(a2), fl.apply_to_mask(mask)
    rec.update(flip_dims=np.array([2, 0]), after_flip=a3.numpy(), mask_after_flip=m3.numpy())
    cr = rm.CropAndResize(1.0, True, (0.45, 0.55), (0.95, 1.0), align_corners=True)
    cc, cs = (0.47, 0.53, 0.5), (0.96, 0.99, 0.95)
    cr.params = {"crop_center": cc, "crop_size": cs}
    a4, m4 = cr.apply_to_image(a3), cr.apply_to_mask(m3)
    rec.update(crop_center=np.array(cc), crop_size=np.array(cs), after_crop=a4.numpy(), mask_after_crop=m4.numpy())
    return rec


if __name__ == "__main__":
    torch.set_num_threads(8)
    np.savez_compressed(os.path.join(HERE, "processor.npz"), **processor_case())
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **augment_case())
    np.savez_compressed(os.path.join(HERE, "epoch_end.npz"), **epoch_end_case())
    print("wrote processor.npz, augment.npz, epoch_end.npz")
