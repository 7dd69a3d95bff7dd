"""Shared driver for the parity tests: runs the same frames through the CPU oracle and the HIP
path (through the C ABI) and compares map state and raycast output."""
from __future__ import annotations

import numpy as np

from oracle.binding import OraclePipeline
from supereight_amd.pipeline import DenseSLAMPipeline
from supereight_amd.synthetic import make_stream


def run_both(field, W, H, N, dim, mu, frames, holes=True, max_blocks=0, on_frame=None, negative_fy=False, stream_kind="room"):
    stream = make_stream(stream_kind, W, H, dim, holes=holes, negative_fy=negative_fy) if stream_kind == "room" else make_stream(stream_kind, W, H, dim, holes=holes)
    cpu = OraclePipeline(field, N, dim, W, H)
    gpu = DenseSLAMPipeline((W, H), N, dim, field_type=field, max_blocks=max_blocks)
    cpu.count_stats(True)
    out = []
    for f in range(frames):
        depth = stream.depth(f)
        pose = stream.pose(f)
        gpu.set_depth(depth)
        gpu.setPose(pose)
        ran_i_g = gpu.integration(stream.k, 1, mu, f)
        ran_r_g = gpu.raycasting(stream.k, mu, f)
        ran_i_c = cpu.integrate(depth, pose, stream.k, mu, f)
        ran_r_c, v_c, n_c = cpu.raycast(pose, stream.k, mu, f)
        assert ran_i_g == ran_i_c and ran_r_g == ran_r_c
        rec = {"frame": f, "raycast": ran_r_c}
        if ran_r_c:
            v_g, n_g = gpu.vertex_normal()
            rec.update(v_c=v_c, n_c=n_c, v_g=v_g, n_g=n_g)
        if on_frame:
            on_frame(f, cpu, gpu, rec)
        out.append(rec)
    return cpu, gpu, out


def compare_maps(cpu, gpu):
    """Returns a dict of mismatch statistics between oracle and HIP map state (blocks sorted by key)."""
    cc, cx, cy, ca = cpu.blocks()
    gc, gx, gy, ga = gpu.blocks()
    res = {"blocks_cpu": len(cc), "blocks_gpu": len(gc)}
    same_set = cc.shape == gc.shape and bool((cc == gc).all())
    res["same_block_set"] = same_set
    if same_set:
        res["x_mismatch"] = int((cx.view(np.uint32) != gx.view(np.uint32)).sum())
        res["y_mismatch"] = int((cy.view(np.uint32) != gy.view(np.uint32)).sum())
        res["active_mismatch"] = int((ca != ga).sum())
        res["x_maxabs"] = float(np.abs(cx - gx).max()) if cx.size else 0.0
        res["voxels"] = int(cx.size)
    ncode, nside, nx, ny = cpu.nodes()
    gcode, gside, gnx, gny = gpu.nodes()
    res["nodes_cpu"], res["nodes_gpu"] = len(ncode), len(gcode)
    same_nodes = ncode.shape == gcode.shape and bool((ncode == gcode).all()) and bool((nside == gside).all())
    res["same_node_set"] = same_nodes
    if same_nodes:
        res["node_x_mismatch"] = int((nx.view(np.uint32) != gnx.view(np.uint32)).sum())
        res["node_y_mismatch"] = int((ny.view(np.uint32) != gny.view(np.uint32)).sum())
    return res


def compare_raycast(rec, voxel):
    v_c, n_c, v_g, n_g = rec["v_c"], rec["n_c"], rec["v_g"], rec["n_g"]
    hit_c = n_c[..., 0] != -2
    hit_g = n_g[..., 0] != -2
    both = hit_c & hit_g
    d = np.linalg.norm(v_c.astype(np.float64) - v_g.astype(np.float64), axis=-1)[both] / voxel
    res = {
        "pixels": int(hit_c.size),
        "hits_cpu": int(hit_c.sum()), "hits_gpu": int(hit_g.sum()),
        "hitmask_mismatch": int((hit_c != hit_g).sum()),
        "vertex_bit_mismatch_px": int((v_c.view(np.uint32) != v_g.view(np.uint32)).any(axis=-1).sum()),
        "normal_bit_mismatch_px": int((n_c.view(np.uint32) != n_g.view(np.uint32)).any(axis=-1).sum()),
    }
    if d.size:
        res.update(vert_err_vox_p90=float(np.percentile(d, 90)), vert_err_vox_p99=float(np.percentile(d, 99)),
                   vert_err_vox_p999=float(np.percentile(d, 99.9)), vert_err_vox_max=float(d.max()))
    return res
