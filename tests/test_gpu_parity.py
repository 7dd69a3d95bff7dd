"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded
synthetic frames.  Bit-exactness is the gate for map state (integer block / node sets, float
TSDF / weight / log-odds values): both sides evaluate IEEE binary32 in the same order with FMA
contraction off.  Raycast output is gated the same way; the looser SURVEY 8(d) tolerances are
printed for information.
"""
import json

import numpy as np
import pytest

from oracle.binding import OFUSION, SDF
from tests.parity_util import compare_maps, compare_raycast, run_both

pytestmark = pytest.mark.gpu

CASES = [
    # name, field, W, H, N, dim, mu, frames
    ("sdf_160x120_256", SDF, 160, 120, 256, 2.4, 0.1, 6),
    ("sdf_640x480_512", SDF, 640, 480, 512, 4.8, 0.1, 6),       # BASELINE.json configs[1]
    ("sdf_320x240_1024", SDF, 320, 240, 1024, 4.8, 0.1, 5),     # configs[2] geometry, smaller image
    ("sdf_640x480_1024", SDF, 640, 480, 1024, 4.8, 0.1, 4),     # configs[2] geometry at full image size
    ("sdf_160x120_2048", SDF, 160, 120, 2048, 4.8, 0.1, 4),     # configs[3] volume: two occupancy levels beyond the LDS-staged ones
    ("ofusion_160x120_256", OFUSION, 160, 120, 256, 2.4, 0.02, 6),
    ("ofusion_640x480_512", OFUSION, 640, 480, 512, 4.8, 0.008, 5),   # configs[4], the reference's own ofusion mu (Makefile:39)
    # the widest band whose key buffer the reference does not truncate at this size (mu = 0.1 saturates it on frame 0):
    # long free-space rays, coarse (leaf-1 / leaf-2) octants hit hard
    ("ofusion_640x480_512_mu005", OFUSION, 640, 480, 512, 4.8, 0.05, 4),
    # ICL-NUIM camera convention of configs[0] / configs[2] (-k 481.2,-480,320,240: negative fy)
    ("icl_like_sdf_320x240_512", SDF, 320, 240, 512, 4.8, 0.1, 5),
    ("icl_like_ofusion_160x120_256", OFUSION, 160, 120, 256, 2.4, 0.02, 5),
]


@pytest.mark.parametrize("name,field,W,H,N,dim,mu,frames", CASES, ids=[c[0] for c in CASES])
def test_stream_parity(name, field, W, H, N, dim, mu, frames):
    cpu, gpu, recs = run_both(field, W, H, N, dim, mu, frames, negative_fy=name.startswith("icl_like"))
    m = compare_maps(cpu, gpu)
    print(name, "map:", json.dumps(m))
    assert cpu.stats()["oob"] == 0          # no sample left the volume -> reference behaviour is defined
    assert cpu.stats()["truncated"] == 0    # the key buffer never saturated -> the reference's block set is deterministic
    assert m["same_block_set"], m
    assert m["same_node_set"], m
    assert m["x_mismatch"] == 0 and m["y_mismatch"] == 0, m
    assert m["active_mismatch"] == 0, m
    assert m["node_x_mismatch"] == 0 and m["node_y_mismatch"] == 0, m
    voxel = dim / N
    for rec in recs:
        if not rec["raycast"]:
            continue
        r = compare_raycast(rec, voxel)
        print(name, "frame", rec["frame"], "raycast:", json.dumps(r))
        assert r["hitmask_mismatch"] == 0, r
        assert r["vertex_bit_mismatch_px"] == 0, r
        assert r["normal_bit_mismatch_px"] == 0, r
    cpu.close()
    gpu.close()


def test_gating_and_empty_input():
    """frame gates of DenseSLAMSystem.cpp:195,209 and an all-zero depth image (nothing allocated)."""
    from supereight_amd.pipeline import DenseSLAMPipeline
    from supereight_amd.synthetic import intrinsics, pose
    p = DenseSLAMPipeline((64, 48), 128, 1.2)
    p.set_depth(np.zeros((48, 64), np.float32))
    p.setPose(pose(0, 1.2))
    k = intrinsics(64)
    assert p.integration(k, 2, 0.1, 0) is True       # frame <= 3 always integrates
    assert p.integration(k, 2, 0.1, 5) is False      # 5 % 2 != 0
    assert p.integration(k, 2, 0.1, 6) is True
    assert p.raycasting(k, 0.1, 2) is False and p.raycasting(k, 0.1, 3) is True
    assert p.counts() == (0, 1)                       # only the root node
    v, n = p.vertex_normal()
    assert (v == 0).all() and (n[..., 0] == -2).all() and (n[..., 1:] == 0).all()
    # a pose or intrinsics with a NaN / infinity is refused by every stage call (the reference would fuse garbage): nothing is enqueued
    from supereight_amd.pipeline import SeHipError
    bad = pose(0, 1.2).copy(); bad[1, 3] = np.nan
    p.setPose(bad)
    for call in (lambda: p.integration(k, 1, 0.1, 7), lambda: p.raycasting(k, 0.1, 7), lambda: p.raycasting_deferred(k, 0.1, 7)):
        with pytest.raises(SeHipError, match="non-finite"):
            call()
    p.setPose(pose(0, 1.2))
    with pytest.raises(SeHipError, match="non-finite"):
        p.integration(np.asarray([np.inf, 60.0, 32.0, 24.0], np.float32), 1, 0.1, 7)
    assert p.counts() == (0, 1) and p.integration(k, 1, 0.1, 7) is True
    p.close()


def test_mm2meters_fused_upload():
    """se_hip_upload_depth_mm == mm2metersKernel (preprocessing.cpp:161-188) incl. subsampling."""
    from supereight_amd.pipeline import DenseSLAMPipeline, SeHipError
    from supereight_amd.synthetic import intrinsics, pose, render_depth_mm
    W, H, N, dim = 80, 60, 128, 2.4
    mm = render_depth_mm(0, 2 * W, 2 * H, dim)
    ref = (mm[::2, ::2].astype(np.float32) / np.float32(1000.0))
    a = DenseSLAMPipeline((W, H), N, dim)
    b = DenseSLAMPipeline((W, H), N, dim)
    a.set_depth_mm(mm)
    b.set_depth(ref)
    k = intrinsics(W)
    for q in (a, b):
        q.setPose(pose(0, dim))
        q.integration(k, 1, 0.1, 0)
    ca, xa, ya, _ = a.blocks()
    cb, xb, yb, _ = b.blocks()
    assert len(ca) > 0 and (ca == cb).all() and (xa == xb).all() and (ya == yb).all()
    with pytest.raises(SeHipError):
        a.set_depth_mm(np.zeros((H + 1, W), np.uint16))   # "Invalid ratio."
    a.close()
    b.close()


@pytest.mark.parametrize("streaming", [False, True], ids=["two-queue", "one-queue"])
@pytest.mark.parametrize("field,W,H,N,dim,mu,frames", [(SDF, 320, 240, 512, 4.8, 0.1, 14), (OFUSION, 160, 120, 256, 2.4, 0.02, 10)], ids=["sdf", "ofusion"])
def test_pipelined_stream_parity(field, W, H, N, dim, mu, frames, streaming):
    """The parity cases above download the raycast after every frame, i.e. they synchronise, and a synchronised caller gets
    the serial schedule (scan on the main stream).  Here the frames are enqueued back to back from device-resident depth
    images, as bench.py does: the allocation scan of frame f+1 runs beside the raycast of frame f -- on the scan stream, released
    by the host gate (two-queue), or in the same launch (one-queue, se_hip_set_streaming) --, the two key lists alternate, the
    occupancy bits are published by the sweep.  The final map and the last raycast must still be the oracle's, bit for bit
    (every frame's raycast of such a stream: tests/test_gpu_stress_parity.py::test_stress_stream_pipelined)."""
    import torch
    from oracle.binding import OraclePipeline
    from supereight_amd.pipeline import DenseSLAMPipeline
    from supereight_amd.synthetic import SyntheticStream, to_colmajor
    s = SyntheticStream(W, H, dim)
    depths = [s.depth(f) for f in range(frames)]
    poses = [s.pose(f) for f in range(frames)]
    dev = torch.from_numpy(np.stack(depths)).cuda()
    gpu = DenseSLAMPipeline((W, H), N, dim, field_type=field, streaming=streaming)
    assert gpu.scan_overlaps() and gpu.frame_is_fused() == streaming
    k = np.ascontiguousarray(s.k, np.float32)
    for f in range(frames):
        gpu.frame(dev[f].data_ptr(), to_colmajor(poses[f]), k, mu, f)       # no synchronisation between frames
    assert gpu.launch_counts()["fused"] == (frames - 4 if streaming else 0)
    cpu = OraclePipeline(field, N, dim, W, H)
    for f in range(frames):
        cpu.integrate(depths[f], poses[f], s.k, mu, f)
        _, v_c, n_c = cpu.raycast(poses[f], s.k, mu, f)
    m = compare_maps(cpu, gpu)
    assert m["same_block_set"] and m["same_node_set"], m
    assert m["x_mismatch"] == 0 and m["y_mismatch"] == 0 and m["active_mismatch"] == 0, m
    assert m["node_x_mismatch"] == 0 and m["node_y_mismatch"] == 0, m
    v_g, n_g = gpu.vertex_normal()
    r = compare_raycast({"v_c": v_c, "n_c": n_c, "v_g": v_g, "n_g": n_g}, dim / N)
    assert r["hits_gpu"] > 1000 and r["hitmask_mismatch"] == 0 and r["vertex_bit_mismatch_px"] == 0 and r["normal_bit_mismatch_px"] == 0, r
    cpu.close(); gpu.close()


@pytest.mark.parametrize("max_blocks", [0, 1 << 13], ids=["dense", "pooled"])
def test_weight_saturated_blocks_stay_bit_exact(max_blocks):
    """Weights saturate at maxweight = 100 (kfusion/mapping_impl.hpp:58-61): 120 frames of the slow room stream -- every block in view
    from frame 0 has all 512 weights at 100 from frame 99 on -- against the oracle: maps after frames 99, 100, 101 and 119 and the raycast
    of the last frame, bit for bit; the run must actually contain saturated blocks."""
    W, H, N, dim, mu, frames = 160, 120, 256, 4.8, 0.1, 120
    seen = {"sat_blocks": 0}

    def on_frame(f, cpu, gpu, rec):
        if f in (99, 100, 101, frames - 1):
            m = compare_maps(cpu, gpu)
            assert m["same_block_set"] and m["x_mismatch"] == 0 and m["y_mismatch"] == 0 and m["active_mismatch"] == 0, (f, m)
            if f == frames - 1:
                _, _, cy, _ = cpu.blocks()
                seen["sat_blocks"] = int((cy.reshape(-1, 512) == 100.0).all(axis=1).sum())

    cpu, gpu, recs = run_both(SDF, W, H, N, dim, mu, frames, max_blocks=max_blocks, on_frame=on_frame)
    assert seen["sat_blocks"] > 100, seen
    r = compare_raycast(recs[-1], dim / N)
    assert r["hits_gpu"] > 1000 and r["hitmask_mismatch"] == 0 and r["vertex_bit_mismatch_px"] == 0 and r["normal_bit_mismatch_px"] == 0, r
    cpu.close(); gpu.close()


def _look(position, yaw_deg=0.0, pitch_deg=0.0, roll_deg=0.0):
    """Camera->world pose: yaw about y, then pitch about x, then roll about z, at `position` (metres)."""
    y, p, r = np.deg2rad([yaw_deg, pitch_deg, roll_deg])
    Ry = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(p), -np.sin(p)], [0, np.sin(p), np.cos(p)]])
    Rz = np.array([[np.cos(r), -np.sin(r), 0], [np.sin(r), np.cos(r), 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Ry @ Rx @ Rz
    T[:3, 3] = position
    return T.astype(np.float32)


@pytest.mark.parametrize("field,W,H,N,dim,mu,frames", [(SDF, 320, 240, 512, 4.8, 0.1, 8), (SDF, 160, 120, 1024, 4.8, 0.1, 5), (OFUSION, 160, 120, 256, 2.4, 0.02, 8)],
                         ids=["sdf512", "sdf1024", "ofusion256"])
def test_raycast_from_unusual_viewpoints(field, W, H, N, dim, mu, frames):
    """The stream's own camera sits inside the room and looks along +z.  The raycast's per-tile start distance (se_beam_start) and its
    first-leaf search have corner cases that camera never reaches: rays that enter the volume from outside, rays that leave it at
    once, origins inside allocated blocks, a far plane that ends the ray before the first allocated block, views along the volume's
    diagonal and against the integration direction.  Same map on both sides, then raycastKernel (kfusion/rendering_impl.hpp:40-123)
    from each of these poses: bit for bit."""
    cpu, gpu, _ = run_both(field, W, H, N, dim, mu, frames)
    from supereight_amd.synthetic import intrinsics
    k = intrinsics(W)
    c = np.array([0.5, 0.5, 0.5]) * dim
    views = {
        "behind_the_volume": _look(c + [0, 0, -0.8 * dim]),                       # enters through z = 0 after 0.3 dim of nothing
        "beside_the_volume": _look(c + [-0.7 * dim, 0, 0], yaw_deg=90),           # enters through x = 0
        "above_looking_down": _look(c + [0, -0.75 * dim, 0], pitch_deg=-90),      # enters through y = 0 (rows run along world z)
        "far_away": _look(c + [0, 0, -6.0 * dim]),                                # far plane (4 m) ends every ray before the volume
        "turned_around": _look(np.array([0.34, 0.5, 0.24]) * dim, yaw_deg=180),   # back wall at 0.19 dim, against the integration direction
        "diagonal_from_corner": _look(np.array([0.02, 0.02, 0.02]) * dim, yaw_deg=45, pitch_deg=-35),
        "rolled_and_pitched": _look(np.array([0.4, 0.45, 0.3]) * dim, yaw_deg=-25, pitch_deg=20, roll_deg=30),
        "nose_on_the_wall": _look(np.array([0.5, 0.5, 0.93]) * dim),              # origin inside the wall's band, surface 0.02 dim ahead
        "inside_the_sphere": _look(np.array([0.5, 0.5, 0.62]) * dim),
        "grazing_the_floor": _look(np.array([0.3, 0.94, 0.2]) * dim, yaw_deg=10, pitch_deg=-2),
        "looking_out": _look(np.array([0.5, 0.5, 0.99]) * dim),                   # beyond the back wall, every ray leaves the volume at once
    }
    total_hits = 0
    for i, (name, view) in enumerate(views.items()):
        frame = 100 + i
        gpu.setPose(view)
        assert gpu.raycasting(k, mu, frame)
        ran, v_c, n_c = cpu.raycast(view, k, mu, frame)
        assert ran
        v_g, n_g = gpu.vertex_normal()
        r = compare_raycast({"v_c": v_c, "n_c": n_c, "v_g": v_g, "n_g": n_g}, dim / N)
        print(name, json.dumps(r))
        assert r["hitmask_mismatch"] == 0 and r["vertex_bit_mismatch_px"] == 0 and r["normal_bit_mismatch_px"] == 0, (name, r)
        if name in ("far_away", "looking_out"):
            assert r["hits_gpu"] == 0, (name, r)
        total_hits += r["hits_gpu"]
    assert total_hits > 5000      # the sweep looked at surfaces, not only at nothing
    cpu.close(); gpu.close()


@pytest.mark.parametrize("streaming", [False, True], ids=["eager", "one-queue"])
def test_pinned_host_input_is_read_in_place(streaming):
    """se_hip_set_pinned_input: images in page-locked memory (se_hip_host_alloc) are not copied into the handle's ring -- the frame's first kernel reads them
    over PCIe where the caller keeps them.  Same frames as uint16 millimetres three ways -- device-resident metres, pageable host images (copied), pinned
    host images (in place, a ring of four caller buffers: the contract is 'unmodified until three further uploads') -- must give the same map and the same
    last raycast, bit for bit; a float image goes the same way; a pageable image on a pinned-input handle is still copied (the caller may overwrite it at once)."""
    import torch
    from supereight_amd.pipeline import DenseSLAMPipeline
    from supereight_amd.synthetic import SyntheticStream
    W, H, N, dim, mu, frames = 320, 240, 256, 2.4, 0.1, 14
    s = SyntheticStream(W, H, dim)
    mm = [np.minimum(np.round(s.depth(f) * 1000.0), 65535).astype(np.uint16) for f in range(frames)]
    metres = [m.astype(np.float32) / np.float32(1000.0) for m in mm]        # mm2metersKernel (preprocessing.cpp:161-188)
    dev = torch.from_numpy(np.stack(metres)).cuda()
    k = np.ascontiguousarray(s.k, np.float32)

    def run(kind):
        p = DenseSLAMPipeline((W, H), N, dim, streaming=streaming)
        ring = []
        if kind.startswith("pinned"):
            p.set_pinned_input(True)
            ring = [p.pinned_image(np.uint16 if kind == "pinned-mm" else np.float32) for _ in range(4)]
        scratch = np.empty((H, W), np.uint16)
        for f in range(frames):
            if kind == "device":
                p.set_depth_device(dev[f].data_ptr())
            elif kind == "pageable":
                scratch[:] = mm[f]
                p.set_depth_mm(scratch)
                scratch[:] = 0                                               # copied: the caller's buffer is free at once
            elif kind == "pageable-on-pinned-handle":
                p.set_pinned_input(True)
                scratch[:] = mm[f]
                p.set_depth_mm(scratch)
                scratch[:] = 0
            elif kind == "pinned-mm":
                ring[f % 4][:] = mm[f]
                p.set_depth_mm(ring[f % 4])
            else:
                ring[f % 4][:] = metres[f]
                p.set_depth(ring[f % 4])
            p.setPose(s.pose(f))
            p.integration(k, 1, mu, f)
            p.raycasting_deferred(k, mu, f)
        v, n = p.vertex_normal()
        out = (p.blocks(), v, n, p.launch_counts())
        p.close()
        return out

    ref = run("device")
    assert len(ref[0][0]) > 500 and (ref[2][..., 0] != -2).sum() > 1000
    if streaming:
        assert ref[3]["fused"] == frames - 4
    for kind in ("pageable", "pageable-on-pinned-handle", "pinned-mm", "pinned-float"):
        got = run(kind)
        for a, b in zip(ref[0], got[0]):
            assert a.shape == b.shape and (a.view(np.uint8) == b.view(np.uint8)).all(), kind
        assert (ref[1].view(np.uint32) == got[1].view(np.uint32)).all() and (ref[2].view(np.uint32) == got[2].view(np.uint32)).all(), kind
        if streaming:
            assert got[3]["fused"] == frames - 4, kind
