"""Parity outside the comfort zone: the "ICL-like stress" stream of supereight_amd/synthetic.py --
a room larger than the volume on two sides (surfaces clipped by the cube, depth points and allocation band steps outside
it, rays ended by the cube exit or the far plane), a pillar and a sphere cut by the x = 0 face (depth discontinuities),
~1.3 cm and 2 deg per frame with a +-90 deg pan there and back (thousands of new keys per frame at 1024^3, blocks leave the
frustum -> active(false) and come back), sigma = 1 mm sensor noise, 2 % holes, depths beyond farPlane, ICL-NUIM intrinsics
(negative fy) -- through the CPU oracle and through the C ABI, compared bit for bit.

Out-of-volume semantics (documented in oracle/se_oracle.cpp, Octree::count_oob, and DESIGN.md 2): the allocation scan skips
steps outside the volume (kfusion/alloc_impl.hpp:92-96, guarded in the reference); the read paths of the raycast
(get_fine / fetch / the cached get, octree.hpp:357-408,439-458) are NOT guarded in the reference: their tree walk wraps
modulo size, which returns "not allocated" unless a block exists at the wrapped position, in which case the reference
indexes that block's array out of bounds (undefined).  Oracle and HIP path return "not allocated" for every such read.
`oob` counts them, `oob_ub` the undefined kind: the tests assert oob > 0 (the regime is really exercised) and
oob_ub == 0 (the reference itself is defined on every read of the stream), i.e. there is no `oob == 0` crutch any more.
"""
import json

import numpy as np
import pytest

from oracle.binding import OFUSION, SDF
from tests.parity_util import compare_maps, compare_raycast, run_both

pytestmark = pytest.mark.gpu

CASES = [
    # name, field, W, H, N, mu, frames, max_blocks
    ("stress_sdf_640x480_512", SDF, 640, 480, 512, 0.1, 36, 0),            # BASELINE configs[0] geometry
    ("stress_sdf_640x480_1024", SDF, 640, 480, 1024, 0.1, 32, 0),          # configs[2]: allocation churn (1-3 k new keys per frame)
    ("stress_ofusion_640x480_512", OFUSION, 640, 480, 512, 0.008, 32, 0),
    ("stress_ofusion_640x480_1024", OFUSION, 640, 480, 1024, 0.008, 30, 0),
    # the whole pan there and back (0 -> +90 -> -90 deg -> ...): almost every block leaves the frustum and is found again
    ("stress_sdf_320x240_512_long", SDF, 320, 240, 512, 0.1, 150, 0),
    ("stress_sdf_320x240_512_long_pooled", SDF, 320, 240, 512, 0.1, 150, 1 << 15),
    ("stress_ofusion_320x240_512_long", OFUSION, 320, 240, 512, 0.02, 110, 0),
]


@pytest.mark.parametrize("name,field,W,H,N,mu,frames,max_blocks", CASES, ids=[c[0] for c in CASES])
def test_stress_stream_parity(name, field, W, H, N, mu, frames, max_blocks):
    dim = 4.8
    seen = {"min_active_frac": 1.0, "max_new": 0, "prev_blocks": 0, "raycasts": 0}

    def on_frame(f, cpu, gpu, rec):
        # every frame: raycast images bit for bit (cheap); block bookkeeping for the regime assertions below
        if rec["raycast"]:
            r = compare_raycast(rec, dim / N)
            assert r["hitmask_mismatch"] == 0 and r["vertex_bit_mismatch_px"] == 0 and r["normal_bit_mismatch_px"] == 0, (f, r)
            seen["raycasts"] += 1
        nb, _ = cpu.counts()
        seen["max_new"] = max(seen["max_new"], nb - seen["prev_blocks"]) if f > 3 else seen["max_new"]
        seen["prev_blocks"] = nb
        if f % 10 == 9 or f == frames - 1:
            m = compare_maps(cpu, gpu)
            assert m["same_block_set"] and m["same_node_set"], (f, m)
            assert m["x_mismatch"] == 0 and m["y_mismatch"] == 0 and m["active_mismatch"] == 0, (f, m)
            assert m["node_x_mismatch"] == 0 and m["node_y_mismatch"] == 0, (f, m)
            act = cpu.blocks()[3]
            seen["min_active_frac"] = min(seen["min_active_frac"], float(act.mean()))

    cpu, gpu, recs = run_both(field, W, H, N, dim, mu, frames, max_blocks=max_blocks, on_frame=on_frame, stream_kind="stress")
    st = cpu.stats()
    print(name, json.dumps({"blocks": cpu.counts()[0], "oob": st["oob"], "oob_ub": st["oob_ub"], "truncated": st["truncated"],
                            "max_new_blocks_per_frame": seen["max_new"], "min_active_frac": round(seen["min_active_frac"], 3)}))
    assert seen["raycasts"] == frames - 3
    assert st["truncated"] == 0          # the reference's key buffer never saturated -> its block set is deterministic
    assert st["oob"] > 0                 # samples did leave the volume: the regime is exercised ...
    assert st["oob_ub"] == 0             # ... and the reference itself is defined on every one of them
    assert seen["max_new"] > (50 if N == 512 else 300)            # centimetre-scale motion: real allocation churn after the start-up frames
    if frames >= 100:
        assert seen["min_active_frac"] < 0.5                       # most of the map out of view at some point (deactivated), then found again
    cpu.close()
    gpu.close()


@pytest.mark.parametrize("field,mu,frames,max_blocks", [(SDF, 0.1, 48, 0), (OFUSION, 0.02, 40, 0), (SDF, 0.1, 48, 1 << 15), (OFUSION, 0.02, 40, 1 << 15)],
                         ids=["sdf", "ofusion", "sdf-pooled", "ofusion-pooled"])
def test_stress_stream_pipelined(field, mu, frames, max_blocks):
    """The same stream enqueued back to back without synchronisation (scan of frame f+1 beside the raycast of frame f,
    alternating key lists, occupancy bits published by the sweep): final map bit-exact, and the raycast of EVERY frame
    bit-exact -- each frame's vertex / normal images are copied into a device ring on the main stream (no host sync), so
    the raycasts that ran with the next frame's scan beside them are the ones compared (with pooled bricks
    that scan is writing the index those raycasts read; the last frame alone has no scan beside it).  Pooled maps: the
    serial schedule (SE_HIP_POOLED_OVERLAP=0) must give the same images."""
    import os
    import torch
    from oracle.binding import OraclePipeline
    from supereight_amd.pipeline import DenseSLAMPipeline
    from supereight_amd.synthetic import StressStream, to_colmajor
    W, H, N, dim = 320, 240, 512, 4.8
    s = StressStream(W, H, dim)
    depths = [s.depth(f) for f in range(frames)]
    poses = [s.pose(f) for f in range(frames)]
    dev = torch.from_numpy(np.stack(depths)).cuda()
    k = np.ascontiguousarray(s.k, np.float32)

    def run(expect_overlap=True):
        # pooled bricks (r03): their raycast reads the index that the next frame's scan, running beside it, is writing (se_block_entry)
        gpu = DenseSLAMPipeline((W, H), N, dim, field_type=field, max_blocks=max_blocks)
        assert gpu.scan_overlaps() == expect_overlap
        ring = torch.zeros((frames, 2, H, W, 3), dtype=torch.float32, device="cuda")
        assert gpu.image_tile_bytes(H) == ring[0].numel() * 4
        for f in range(frames):
            gpu.frame(dev[f].data_ptr(), to_colmajor(poses[f]), k, mu, f)
            # r04: se_hip_frame defers the raycast to the next frame call, which launches it in one kernel with that frame's scan
            # (k_raycast_scan); any other call -- this copy -- launches it first.  Every third frame is copied, so the stream mixes
            # fused launches (two of three frames) with stand-alone raycasts beside a side-stream scan
            if f > 2 and (f % 3 == 0 or f == frames - 1):
                gpu.pack_image_tile(ring[f].data_ptr(), H)     # device-to-device on the pipeline's main stream, behind this frame's raycast
        gpu.sync()
        return gpu, ring.cpu().numpy()

    gpu, ring = run()
    cpu = OraclePipeline(field, N, dim, W, H)
    worst = {"hitmask_mismatch": 0, "vertex_bit_mismatch_px": 0, "normal_bit_mismatch_px": 0}
    hits = 0
    for f in range(frames):
        cpu.integrate(depths[f], poses[f], s.k, mu, f)
        ran, v_c, n_c = cpu.raycast(poses[f], s.k, mu, f)
        if ran and (f % 3 == 0 or f == frames - 1):
            r = compare_raycast({"v_c": v_c, "n_c": n_c, "v_g": ring[f, 0], "n_g": ring[f, 1]}, dim / N)
            hits += r["hits_gpu"]
            for kk in worst:
                worst[kk] = max(worst[kk], r[kk])
            assert r["hitmask_mismatch"] == 0 and r["vertex_bit_mismatch_px"] == 0 and r["normal_bit_mismatch_px"] == 0, (f, r)
    assert hits > 300 * (frames - 3), hits
    m = compare_maps(cpu, gpu)
    assert m["same_block_set"] and m["same_node_set"], m
    assert m["x_mismatch"] == 0 and m["y_mismatch"] == 0 and m["active_mismatch"] == 0, m
    assert m["node_x_mismatch"] == 0 and m["node_y_mismatch"] == 0, m
    v_g, n_g = gpu.vertex_normal()
    assert (v_g.view(np.uint32) == ring[frames - 1, 0].view(np.uint32)).all() and (n_g.view(np.uint32) == ring[frames - 1, 1].view(np.uint32)).all()
    cpu.close(); gpu.close()
    if max_blocks:
        os.environ["SE_HIP_POOLED_OVERLAP"] = "0"
        try:
            gpu2, ring2 = run(expect_overlap=False)
        finally:
            del os.environ["SE_HIP_POOLED_OVERLAP"]
        assert (ring2.view(np.uint32) == ring.view(np.uint32)).all()
        gpu2.close()
