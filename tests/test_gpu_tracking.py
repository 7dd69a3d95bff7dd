"""ICP tracking on the device against the oracle -- the full SLAM loop
(preprocessing -> tracking -> integration -> raycasting) with poses estimated by the tracker instead of
injected.  Pyramid, TrackData, reduction sums, pose updates and the accept / reject decision are
compared bit for bit (oracle and HIP path share one summation order and one LLT / SE3-exp definition)."""
import numpy as np
import pytest

from oracle.binding import OFUSION, SDF, OraclePipeline, oracle_tracking
from supereight_amd.pipeline import DenseSLAMPipeline
from supereight_amd.synthetic import SyntheticStream

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("field,W,H,N,dim,mu,frames", [(SDF, 320, 240, 256, 4.8, 0.1, 9), (SDF, 640, 480, 512, 4.8, 0.1, 7),
                                                         (OFUSION, 320, 240, 256, 4.8, 0.02, 7), (SDF, 320, 240, 256, 4.8, 0.1, -7)],
                         ids=["sdf-320", "sdf-640", "ofusion-320", "sdf-320-icl-negative-fy"])
def test_slam_loop_with_tracking(field, W, H, N, dim, mu, frames):
    negative_fy, frames = frames < 0, abs(frames)       # ICL-NUIM convention: vertex2normalKernel<true> (DenseSLAMSystem.cpp:160-163)
    s = SyntheticStream(W, H, dim, negative_fy=negative_fy)
    cpu = OraclePipeline(field, N, dim, W, H)
    gpu = DenseSLAMPipeline((W, H), N, dim, field_type=field)
    pose_c = s.pose(0).copy()
    gpu.setPose(s.pose(0))
    v_c = n_c = rp_c = None
    tracked_frames = 0
    for f in range(frames):
        depth = s.depth(f)
        gpu.set_depth(depth)
        if f >= 4:   # the reference apps track from frame 1 on; the first frames here build a map with GT poses
            ok_c, pose_c, track_c, red_c, it_c = oracle_tracking(depth, s.k, pose_c, rp_c, v_c, n_c, 1e-5, (10, 5, 4))
            ok_g = gpu.tracking(s.k, 1e-5, 1, f, (10, 5, 4))
            track_g, red_g, it_g = gpu.track_data()
            assert ok_g == ok_c and it_g == it_c
            assert (track_g["result"] == track_c["result"]).all()
            good = track_c["result"] == 1
            assert (track_g["error"][good].view(np.uint32) == track_c["error"][good].view(np.uint32)).all()
            assert (track_g["J"][good].view(np.uint32) == track_c["J"][good].view(np.uint32)).all()
            assert (red_g.view(np.uint32) == red_c.view(np.uint32)).all(), (red_g, red_c)
            assert (gpu.getPose().view(np.uint32) == pose_c.view(np.uint32)).all()
            assert ok_c and 0.5 * W * H < red_c[28]          # most pixels are inliers on this scene
            tracked_frames += 1
        else:
            pose_c = s.pose(f).copy()
            gpu.setPose(pose_c)
        cpu.integrate(depth, pose_c, s.k, mu, f)
        gpu.integration(s.k, 1, mu, f)
        ran, vv, nn = cpu.raycast(pose_c, s.k, mu, f)
        gpu.raycasting(s.k, mu, f)
        if ran:
            v_c, n_c, rp_c = vv, nn, pose_c.copy()
            v_g, n_g = gpu.vertex_normal()
            assert (v_g.view(np.uint32) == v_c.view(np.uint32)).all() and (n_g.view(np.uint32) == n_c.view(np.uint32)).all()
    assert tracked_frames == frames - 4
    # the tracker follows the camera: pose error stays within a fraction of a voxel / of the per-frame motion
    err = np.abs(pose_c - s.pose(frames - 1)).max()
    print("pose error after", tracked_frames, "tracked frames:", err)
    assert err < 0.02
    cc, cx, cy, ca = cpu.blocks()
    gc, gx, gy, ga = gpu.blocks()
    assert (cc == gc).all() and (cx.view(np.uint32) == gx.view(np.uint32)).all() and (cy.view(np.uint32) == gy.view(np.uint32)).all()
    cpu.close(); gpu.close()


def test_slam_loop_with_tracking_on_the_stress_stream():
    """the tracker on a stream it has to work on.  The ICL-like stress scene (clipped room, occluders, 1 mm
    sensor noise, 2 % holes, depths beyond the far plane) entered at path position 36 -- where the view is inside the volume;
    from its own start the reference's checkPoseKernel rejects every frame, < 15 % inliers -- and sampled at a quarter of the
    stress speed (3 mm + 0.5 deg per frame, a 30 Hz hand-held sensor): 40 tracked frames, every one compared bit for bit with
    the oracle (decision, iteration count, the 32 sums, pose), then the maps."""
    from supereight_amd.synthetic import StressStream
    W, H, N, dim, mu, frames = 320, 240, 256, 4.8, 0.1, 44
    s = StressStream(W, H, dim, time_scale=0.25, start=36.0)
    cpu = OraclePipeline(SDF, N, dim, W, H)
    gpu = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
    pose_c = s.pose(0).copy()
    gpu.setPose(pose_c)
    v_c = n_c = rp_c = None
    tracked, worst_gt, iters = 0, 0.0, []
    for f in range(frames):
        depth = s.depth(f)
        gpu.set_depth(depth)
        if f >= 4:
            ok_c, pose_c, track_c, red_c, it_c = oracle_tracking(depth, s.k, pose_c, rp_c, v_c, n_c, 1e-5, (10, 5, 4))
            ok_g = gpu.tracking(s.k, 1e-5, 1, f, (10, 5, 4))
            track_g, red_g, it_g = gpu.track_data()
            assert ok_g == ok_c and it_g == it_c, (f, ok_g, ok_c, it_g, it_c)
            assert (track_g["result"] == track_c["result"]).all()
            assert (red_g.view(np.uint32) == red_c.view(np.uint32)).all(), (f, red_g, red_c)
            assert (gpu.getPose().view(np.uint32) == pose_c.view(np.uint32)).all(), f
            tracked += int(ok_c)
            iters.append(it_c)
            worst_gt = max(worst_gt, float(np.abs(pose_c[:3, 3] - s.pose(f)[:3, 3]).max()))
        else:
            pose_c = s.pose(f).copy()
            gpu.setPose(pose_c)
        cpu.integrate(depth, pose_c, s.k, mu, f)
        gpu.integration(s.k, 1, mu, f)
        ran, vv, nn = cpu.raycast(pose_c, s.k, mu, f)
        gpu.raycasting(s.k, mu, f)
        if ran:
            v_c, n_c, rp_c = vv, nn, pose_c.copy()
            v_g, n_g = gpu.vertex_normal()
            assert (v_g.view(np.uint32) == v_c.view(np.uint32)).all() and (n_g.view(np.uint32) == n_c.view(np.uint32)).all(), f
    print(f"stress stream, {tracked} of {frames - 4} frames accepted, iterations {min(iters)}..{max(iters)}, worst distance to the ground truth {worst_gt:.4f} m")
    assert tracked == frames - 4          # checkPoseKernel accepts every frame ...
    assert worst_gt < 0.08                # ... and the estimate stays with the camera (the reference's ICP jitters by centimetres on this noisy, clipped scene)
    cc, cx, cy, ca = cpu.blocks()
    gc, gx, gy, ga = gpu.blocks()
    assert (cc == gc).all() and (cx.view(np.uint32) == gx.view(np.uint32)).all() and (cy.view(np.uint32) == gy.view(np.uint32)).all() and (ca == ga).all()
    cpu.close(); gpu.close()


def _tracked_loop(W, H, N, dim, mu, frames, one_call, lookahead=None, odd=False):
    """The reference's loop with tracking on (se_apps/src/benchmark.cpp:115-150) on the device; returns what a caller can observe per frame."""
    import os
    import torch
    s = SyntheticStream(W, H, dim)
    old = os.environ.get("SE_HIP_ICP_LOOKAHEAD")
    if lookahead is not None:
        os.environ["SE_HIP_ICP_LOOKAHEAD"] = str(lookahead)
    try:
        p = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
    finally:
        if lookahead is not None:
            if old is None: del os.environ["SE_HIP_ICP_LOOKAHEAD"]
            else: os.environ["SE_HIP_ICP_LOOKAHEAD"] = old
    dev = torch.from_numpy(np.stack([s.depth(f) for f in range(frames)])).cuda()
    p.setPose(s.pose(0))
    out = []
    for f in range(frames):
        if f <= 3:
            p.setPose(s.pose(f))
        if one_call and f > 3:
            r = p.frame_tracked(dev[f].data_ptr(), s.k, mu, f)
            flags = (bool(r & 4), bool(r & 1), bool(r & 2))
        else:
            p.set_depth_device(dev[f].data_ptr())
            tracked = p.tracking(s.k, 1e-5, 1, f) if f > 3 else False
            integrated = p.integration(s.k, 1, mu, f) if (tracked or f <= 3) else False
            flags = (tracked, integrated, p.raycasting(s.k, mu, f))
        track, red, it = p.track_data() if f > 3 else (None, None, 0)
        v, n = p.vertex_normal()
        out.append((flags, p.getPose(), it, red, None if track is None else track["result"].copy(), v, n))
    blocks = p.blocks()
    p.close()
    return out, blocks


def test_frame_tracked_is_the_four_calls():
    """se_hip_frame_tracked (one FFI call per frame, scan chained behind the ICP) == set_depth_device + tracking + integration + raycasting,
    and pruning a converged level's launches (SE_HIP_ICP_LOOKAHEAD, default 2) == enqueuing every iteration up front (0): poses, iteration
    counts, sums, tracking_result_, images and the map, bit for bit."""
    W, H, N, dim, mu, frames = 320, 240, 256, 4.8, 0.1, 10
    ref, blocks_ref = _tracked_loop(W, H, N, dim, mu, frames, one_call=False, lookahead=0)
    assert sum(1 for r in ref if r[0][0]) == frames - 4
    for one_call, look in ((True, 2), (False, 2), (True, 1), (True, 0)):
        got, blocks = _tracked_loop(W, H, N, dim, mu, frames, one_call=one_call, lookahead=look)
        for f, (a, b) in enumerate(zip(ref, got)):
            assert a[0] == b[0], (f, a[0], b[0])
            assert (a[1].view(np.uint32) == b[1].view(np.uint32)).all(), f
            assert a[2] == b[2], (f, a[2], b[2])
            if a[3] is not None:
                assert (a[3].view(np.uint32) == b[3].view(np.uint32)).all() and (a[4] == b[4]).all(), f
            assert (a[5].view(np.uint32) == b[5].view(np.uint32)).all() and (a[6].view(np.uint32) == b[6].view(np.uint32)).all(), f
        for x, y in zip(blocks_ref, blocks):
            assert (x.view(np.uint8) == y.view(np.uint8)).all()


@pytest.mark.parametrize("pyramid", [(6,), (5, 4), (4, 3, 2, 2), (0, 5, 4)], ids=["1-level", "2-levels", "4-levels", "no-fine-level"])
def test_tracking_with_other_pyramids(pyramid):
    """k_depth_pyramid with one / two levels (no l1 / l2), k_half_sample for the levels beyond the third, and a pyramid whose finest level has no
    iteration (tracking_result_ then comes from level 1): decision, iteration count, sums, TrackData and pose against the oracle, bit for bit."""
    W, H, N, dim, mu = 320, 240, 256, 4.8, 0.1
    s = SyntheticStream(W, H, dim)
    cpu = OraclePipeline(SDF, N, dim, W, H)
    gpu = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
    v_c = n_c = rp_c = None
    for f in range(5):
        depth, pose = s.depth(f), s.pose(f)
        gpu.set_depth(depth); gpu.setPose(pose)
        cpu.integrate(depth, pose, s.k, mu, f); gpu.integration(s.k, 1, mu, f)
        ran, vv, nn = cpu.raycast(pose, s.k, mu, f); gpu.raycasting(s.k, mu, f)
        if ran:
            v_c, n_c, rp_c = vv, nn, pose.copy()
    depth = s.depth(5)
    gpu.set_depth(depth)
    ok_c, pose_c, track_c, red_c, it_c = oracle_tracking(depth, s.k, s.pose(4), rp_c, v_c, n_c, 1e-5, pyramid)
    ok_g = gpu.tracking(s.k, 1e-5, 1, 5, pyramid)
    track_g, red_g, it_g = gpu.track_data()
    assert ok_g == ok_c and it_g == it_c and it_c > 0
    assert (red_g.view(np.uint32) == red_c.view(np.uint32)).all()
    assert (gpu.getPose().view(np.uint32) == pose_c.view(np.uint32)).all()
    lvl = next(i for i, n in enumerate(pyramid) if n > 0)              # tracking_result_ holds the finest level that ran (row stride = W)
    w, h = W >> lvl, H >> lvl
    assert (track_g["result"][:h, :w] == track_c["result"][:h, :w]).all()
    good = track_c["result"][:h, :w] == 1
    assert (track_g["J"][:h, :w][good].view(np.uint32) == track_c["J"][:h, :w][good].view(np.uint32)).all()
    cpu.close(); gpu.close()


def test_frame_tracked_does_not_integrate_a_rejected_frame():
    """benchmark.cpp:141-147: integration only if tracking succeeded (or frame <= 3); the raycast runs either way, from the restored pose."""
    import torch
    W, H, N, dim, mu = 160, 120, 128, 2.4, 0.1
    s = SyntheticStream(W, H, dim)
    p = DenseSLAMPipeline((W, H), N, dim)
    dev = torch.from_numpy(np.stack([s.depth(f) for f in range(6)])).cuda()
    for f in range(4):
        p.setPose(s.pose(f))
        r = p.frame_tracked(dev[f].data_ptr(), s.k, mu, f)          # frames 0..3: tracking cannot succeed before the first raycast (frame 3), integration runs anyway
        assert r & 1 and not (r & 4) and bool(r & 2) == (f > 2), (f, r)
    before = [x.copy() for x in p.blocks()]
    far = p.getPose().copy(); far[:3, 3] += 1.0                     # nothing overlaps from here: checkPoseKernel rejects and restores
    p.setPose(far)
    r = p.frame_tracked(dev[4].data_ptr(), s.k, mu, 4)
    assert r == 2, r                                                # not tracked, not integrated, raycast ran
    assert (p.getPose() == far).all()
    after = p.blocks()
    for x, y in zip(before, after):
        assert x.shape == y.shape and (x.view(np.uint8) == y.view(np.uint8)).all()
    p.setPose(s.pose(3))
    assert p.frame_tracked(dev[5].data_ptr(), s.k, mu, 5, tracking_rate=2) == 2     # frame % tracking_rate != 0: tracking() returns false
    p.close()


def test_depth_pyramid_of_an_odd_sized_image():
    """k_depth_pyramid (copy + two half-samplings in one launch) on a size that is not a multiple of 4: the scalar path and the
    guards of the last column / row, against the oracle's halfSampleRobustImageKernel."""
    from oracle.binding import oracle_half_sample
    W, H, N, dim = 322, 242, 128, 4.8
    s = SyntheticStream(W, H, dim)
    p = DenseSLAMPipeline((W, H), N, dim)
    for f in range(4):
        p.set_depth(s.depth(f)); p.setPose(s.pose(f))
        p.integration(s.k, 1, 0.1, f); p.raycasting(s.k, 0.1, f)
    depth = s.depth(4)
    p.set_depth(depth)
    p.tracking(s.k, 1e-5, 1, 4)
    l0, l1, l2 = p.scaled_depth(0), p.scaled_depth(1), p.scaled_depth(2)
    assert (l0.view(np.uint32) == depth.view(np.uint32)).all()
    h1 = oracle_half_sample(depth, 0.1 * 3, 1)
    h2 = oracle_half_sample(h1, 0.1 * 3, 1)
    assert l1.shape == h1.shape and l2.shape == h2.shape
    assert (l1.view(np.uint32) == h1.view(np.uint32)).all() and (l2.view(np.uint32) == h2.view(np.uint32)).all()
    p.close()


def test_tracking_gate_and_rejection():
    W, H, N, dim = 160, 120, 128, 2.4
    s = SyntheticStream(W, H, dim)
    p = DenseSLAMPipeline((W, H), N, dim)
    for f in range(4):
        p.set_depth(s.depth(f)); p.setPose(s.pose(f))
        p.integration(s.k, 1, 0.1, f); p.raycasting(s.k, 0.1, f)
    p.set_depth(s.depth(4))
    assert p.tracking(s.k, 1e-5, 2, 5) is False            # frame % tracking_rate != 0
    before = p.getPose()
    far = before.copy(); far[:3, 3] += 1.0                   # a pose from which nothing overlaps: checkPoseKernel must restore it
    p.setPose(far)
    assert p.tracking(s.k, 1e-5, 1, 4) is False
    assert (p.getPose() == far).all()
    p.close()


def test_bilateral_filter_in_front_of_tracking():
    """preprocessing(..., filterInput=true): bilateralFilterKernel feeds the tracking pyramid.  The filter
    calls expf, whose last bit belongs to the C library the reference happens to be linked with; the HIP
    kernel uses the correctly rounded value.  Stated tolerance: filtered depth within 1e-6 relative (zeros
    exactly preserved), tracked pose within 1e-5 of the oracle's."""
    from oracle.binding import oracle_bilateral_filter
    W, H, N, dim, mu = 320, 240, 256, 4.8, 0.1
    s = SyntheticStream(W, H, dim)
    cpu = OraclePipeline(SDF, N, dim, W, H)
    gpu = DenseSLAMPipeline((W, H), N, dim, field_type=SDF)
    v_c = n_c = rp_c = None
    for f in range(4):
        depth, pose = s.depth(f), s.pose(f)
        gpu.set_depth(depth); gpu.setPose(pose)
        cpu.integrate(depth, pose, s.k, mu, f); gpu.integration(s.k, 1, mu, f)
        ran, vv, nn = cpu.raycast(pose, s.k, mu, f); gpu.raycasting(s.k, mu, f)
        if ran:
            v_c, n_c, rp_c = vv, nn, pose.copy()
    depth = s.depth(4)
    gpu.set_depth(depth)
    gpu.filter_depth(True)
    filt_c = oracle_bilateral_filter(depth)
    ok_c, pose_c, track_c, red_c, it_c = oracle_tracking(filt_c, s.k, s.pose(3), rp_c, v_c, n_c, 1e-5, (10, 5, 4))
    ok_g = gpu.tracking(s.k, 1e-5, 1, 4, (10, 5, 4))
    filt_g = gpu.scaled_depth(0)
    assert ((filt_g == 0) == (depth == 0)).all() and ((filt_c == 0) == (depth == 0)).all()
    assert np.abs(filt_c - depth).max() > 1e-4          # the filter does something on the quantised depth
    nz = depth != 0
    rel = np.abs(filt_g[nz] - filt_c[nz]) / filt_c[nz]
    same = (filt_g.view(np.uint32) == filt_c.view(np.uint32)).mean()
    print(f"bilateral filter: {100 * same:.3f} % of pixels bit-identical, max relative difference {rel.max():.2e}")
    assert rel.max() < 1e-6 and same > 0.9
    assert ok_g == ok_c and ok_c
    assert np.abs(gpu.getPose() - pose_c).max() < 1e-5
    track_g, red_g, it_g = gpu.track_data()
    assert (track_g["result"] == track_c["result"]).mean() > 0.999
    # switched off again: bit-exact path
    gpu.filter_depth(False)
    gpu.setPose(s.pose(3))
    ok_c2, pose_c2, *_ = oracle_tracking(depth, s.k, s.pose(3), rp_c, v_c, n_c, 1e-5, (10, 5, 4))
    gpu.tracking(s.k, 1e-5, 1, 4, (10, 5, 4))
    assert (gpu.getPose().view(np.uint32) == pose_c2.view(np.uint32)).all()
    assert (gpu.scaled_depth(0).view(np.uint32) == depth.view(np.uint32)).all()
    cpu.close(); gpu.close()
