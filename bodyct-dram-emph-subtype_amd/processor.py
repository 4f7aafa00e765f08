"""Predict-path post-processing of the reference's ``processor.py`` (grand-challenge entrypoint) on the GPU.

reference processor.py:97-177: merge the ``predict_step`` outputs, resize every dRAM volume to its lung-crop
size (trilinear, align_corners=True), paste it into a zero volume of the original scan grid (:111-129),
derive the severity score from the lesion percentage (:34-38, :130-136) and write the json reports
(:160-177).  The resize + paste (+ the uint8 windowing of :143) is one gather kernel over the original grid
(``dram_resample_paste``); ``.mha`` writing (SimpleITK, :146-158) stays with the caller.
"""
from __future__ import annotations

if not __package__:          # imported top-level (this directory on sys.path): bind to the package, see _dropin.py
    import _dropin
    __package__ = _dropin.adopt(__name__)

import json
from typing import Dict, List, Optional, Sequence

import torch

from .models import CLE_RATIO_MAP, PSE_RATIO_MAP
from .ops import _L, _chk, _p, _req, _stream


def ratio_to_label(ratio: float, ratio_mapping: Dict[int, tuple]) -> int:
    """processor.py:34-38: the class whose [lo, hi) band holds the ratio."""
    for label, (lo, hi) in ratio_mapping.items():
        if lo <= ratio < hi:
            return label
    raise IndexError(f"ratio {ratio} is outside every band")       # the reference raises IndexError here too


def resample_paste(dense: torch.Tensor, crop_slice, original_size: Sequence[int], want_f32: bool = True,
                   want_u8: bool = False):
    """dense [D,H,W] (one sample of predict_step's *_dense_outs) -> (full f32 [Do,Ho,Wo] | None, full uint8 | None).
    crop_slice [[z0,z1],[y0,y1],[x0,x1]] (tensor or nested list); processor.py:111-129, :143."""
    dense = dense.float().contiguous()
    _req(dense, "dense")
    if dense.dim() != 3:
        raise ValueError("resample_paste: dense must be [D,H,W]")
    cs = [[int(v) for v in row] for row in (crop_slice.tolist() if torch.is_tensor(crop_slice) else crop_slice)]
    Do, Ho, Wo = (int(v) for v in (original_size.tolist() if torch.is_tensor(original_size) else original_size))
    (z0, z1), (y0, y1), (x0, x1) = cs
    if not (0 <= z0 < z1 <= Do and 0 <= y0 < y1 <= Ho and 0 <= x0 < x1 <= Wo):
        raise ValueError(f"crop_slice {cs} does not fit the original size {(Do, Ho, Wo)}")
    D, H, W = dense.shape
    of = torch.empty((Do, Ho, Wo), device=dense.device, dtype=torch.float32) if want_f32 else None
    ob = torch.empty((Do, Ho, Wo), device=dense.device, dtype=torch.uint8) if want_u8 else None
    _chk(_L().dram_resample_paste(_p(dense), _p(of), _p(ob), D, H, W, z1 - z0, y1 - y0, x1 - x0, z0, y0, x0, Do, Ho, Wo,
                                  _stream()), "dram_resample_paste")
    return of, ob


def build_outputs(predictions: List[dict], want_u8: bool = True) -> List[dict]:
    """processor.py:102-145 for a list of predict_step outputs: per scan the pasted CLE / PSE volumes (uint8 like
    the written .mha, and/or float) and the metrics entry of the results json."""
    results = []
    for out in predictions:
        B = out["cle_dense_outs"].shape[0]
        for b in range(B):
            vols = {}
            for name in ("cle", "pse"):
                d = out[f"{name}_dense_outs"][b]
                d = d.reshape(d.shape[-3:])
                f32, u8 = resample_paste(d, out["crop_slices"][b], out["original_size"][b], want_f32=not want_u8,
                                         want_u8=want_u8)
                vols[name] = u8 if want_u8 else f32
            cle_p, pse_p = float(out["cle_precentages"][b]), float(out["pse_precentages"][b])
            metrics = {"cle_severity_score": "{:d}".format(ratio_to_label(cle_p, CLE_RATIO_MAP)),
                       "cle_lesion_percentage_per_lung": "{:.3f}".format(cle_p),
                       "pse_severity_score": "{:d}".format(ratio_to_label(pse_p, PSE_RATIO_MAP)),
                       "pse_lesion_percentage_per_lung": "{:.3f}".format(pse_p)}
            uid = out["uids"][b] if out.get("uids") is not None else None
            results.append({"entity": uid, "metrics": metrics, "error_messages": [], "full_cle": vols["cle"],
                            "full_pse": vols["pse"]})
    return results


def write_reports(results: List[dict], centrilobular_json: Optional[str] = None, paraseptal_json: Optional[str] = None,
                  output_json: Optional[str] = None):
    """processor.py:160-177: the two single-scan score files and the results list."""
    m = results[0]["metrics"]
    if centrilobular_json:
        with open(centrilobular_json, "w") as f:
            f.write(json.dumps({"score": int(float(m["cle_severity_score"])),
                                "percentage": float(m["cle_lesion_percentage_per_lung"])}))
    if paraseptal_json:
        with open(paraseptal_json, "w") as f:
            f.write(json.dumps({"score": int(float(m["pse_severity_score"])),
                                "percentage": float(m["pse_lesion_percentage_per_lung"])}))
    if output_json:
        with open(output_json, "w") as f:
            f.write(json.dumps([{k: r[k] for k in ("entity", "metrics", "error_messages")} for r in results]))
