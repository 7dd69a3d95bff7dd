"""Distance between two runs of the same frames -- used to measure and gate how far results sit from a reference
built the way its authors build it (GCC -O3 with FMA contraction, real Sophus: oracle variants `fma` + `sophus_quat`),
SURVEY.md 8(d) "Parity statement".  Inputs are (coords, x, y) of Morton-sorted blocks and vertex / normal images."""
from __future__ import annotations

import numpy as np


def map_distance(a_blocks, b_blocks, relative_x: bool = False) -> dict:
    ca, xa, ya = a_blocks[:3]
    cb, xb, yb = b_blocks[:3]
    ka = {tuple(c) for c in ca.tolist()}
    kb = {tuple(c) for c in cb.tolist()}
    common = ka & kb
    res = {"blocks_a": len(ka), "blocks_b": len(kb), "block_set_symdiff_frac": len(ka ^ kb) / max(1, len(ka | kb))}
    ia = np.array([i for i, c in enumerate(ca.tolist()) if tuple(c) in common], dtype=np.int64)
    ib = np.array([i for i, c in enumerate(cb.tolist()) if tuple(c) in common], dtype=np.int64)
    xa, ya, xb, yb = xa[ia], ya[ia], xb[ib], yb[ib]   # both lists are Morton-sorted: common blocks line up
    dx = np.abs(xa.astype(np.float64) - xb.astype(np.float64))
    if relative_x:
        dx = dx / np.maximum(1.0, np.maximum(np.abs(xa), np.abs(xb)))
    n = max(1, dx.size)
    res.update(voxels=int(dx.size),
               x_bit_identical_frac=float((xa.view(np.uint32) == xb.view(np.uint32)).sum() / n),
               x_gt_1e6_frac=float((dx > 1e-6).sum() / n), x_gt_1e5_frac=float((dx > 1e-5).sum() / n),
               x_gt_1e3_frac=float((dx > 1e-3).sum() / n), x_max=float(dx.max()) if dx.size else 0.0,
               y_differs_frac=float((ya != yb).sum() / n))
    return res


def raycast_distance(v_a, n_a, v_b, n_b, voxel: float) -> dict:
    hit_a, hit_b = n_a[..., 0] != -2, n_b[..., 0] != -2
    both = hit_a & hit_b
    d = np.linalg.norm(v_a.astype(np.float64) - v_b.astype(np.float64), axis=-1)[both] / voxel
    res = {"hits_a": int(hit_a.sum()), "hits_b": int(hit_b.sum()), "hitmask_disagree_frac": float((hit_a != hit_b).sum() / hit_a.size),
           "vertex_bit_identical_frac": float((v_a.view(np.uint32) == v_b.view(np.uint32)).all(axis=-1)[both].mean()) if both.any() else 1.0}
    if d.size:
        res.update(vert_le_0p1_vox_frac=float((d <= 0.1).mean()), vert_le_0p5_vox_frac=float((d <= 0.5).mean()),
                   vert_le_2_vox_frac=float((d <= 2.0).mean()),
                   vert_err_vox_median=float(np.median(d)), vert_err_vox_p90=float(np.percentile(d, 90)),
                   vert_err_vox_p99=float(np.percentile(d, 99)), vert_err_vox_p999=float(np.percentile(d, 99.9)), vert_err_vox_max=float(d.max()))
        cosang = np.clip((n_a[both].astype(np.float64) * n_b[both].astype(np.float64)).sum(-1), -1, 1)
        ang = np.degrees(np.arccos(cosang))
        res.update(normal_deg_p99=float(np.percentile(ang, 99)), normal_deg_p999=float(np.percentile(ang, 99.9)), normal_gt_0p1deg_frac=float((ang > 0.1).mean()))
    return res


# SURVEY.md 8(d) acceptance tolerances ("no worse than the reference's own FMA noise floor")
def check_survey_tolerances(m: dict, r: dict) -> list:
    bad = []
    if m["block_set_symdiff_frac"] > 1e-3: bad.append(("block_set_symdiff_frac", m["block_set_symdiff_frac"]))
    if m["x_gt_1e5_frac"] > 1e-4: bad.append(("x_gt_1e5_frac", m["x_gt_1e5_frac"]))
    if m["y_differs_frac"] > 1e-5: bad.append(("y_differs_frac", m["y_differs_frac"]))
    if r["hitmask_disagree_frac"] > 2e-3: bad.append(("hitmask_disagree_frac", r["hitmask_disagree_frac"]))
    if r.get("vert_le_0p1_vox_frac", 1) < 0.90: bad.append(("vert_le_0p1_vox_frac", r["vert_le_0p1_vox_frac"]))
    if r.get("vert_le_0p5_vox_frac", 1) < 0.99: bad.append(("vert_le_0p5_vox_frac", r["vert_le_0p5_vox_frac"]))
    if r.get("vert_le_2_vox_frac", 1) < 0.999: bad.append(("vert_le_2_vox_frac", r["vert_le_2_vox_frac"]))
    return bad
