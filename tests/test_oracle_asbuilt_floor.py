"""CPU: the oracle's noise-floor variant really is a different arithmetic (FMA contraction on, Sophus quaternion round
trip) and stays inside loose versions of the SURVEY 8(d) tolerances against the plain oracle on a small stream.  The
full-size measurement against the HIP path is tests/test_gpu_asbuilt_tolerance.py."""
import numpy as np

from oracle.binding import OFUSION, SDF, OraclePipeline, load
from supereight_amd.synthetic import SyntheticStream
from tests.asbuilt_util import map_distance, raycast_distance


def _run(field, mu, fma):
    W, H, N, dim = 160, 120, 256, 2.4
    o = OraclePipeline(field, N, dim, W, H, fma=fma)
    s = SyntheticStream(W, H, dim)
    for f in range(5):
        d, p = s.depth(f), s.pose(f)
        o.integrate(d, p, s.k, mu, f)
        _, v, n = o.raycast(p, s.k, mu, f)
    b = o.blocks()
    o.close()
    return b, v, n, dim / N


def test_fma_variant_is_built_with_contraction():
    assert load().so_fp_contract() == 0
    assert load(fma=True).so_fp_contract() == 1


def test_quaternion_round_trip_changes_only_last_bits():
    lib = load()
    a = _run(SDF, 0.1, False)
    lib.so_set_sophus_quat(1)
    try:
        b = _run(SDF, 0.1, False)
    finally:
        lib.so_set_sophus_quat(0)
    m = map_distance(a[0], b[0])
    assert m["block_set_symdiff_frac"] == 0
    assert 0.2 < m["x_bit_identical_frac"] < 1.0      # it IS another arithmetic ...
    assert m["x_gt_1e5_frac"] < 1e-4                  # ... a few ulps away


def test_noise_floor_small_stream():
    for field, mu in ((SDF, 0.1), (OFUSION, 0.02)):
        fma = load(fma=True)
        fma.so_set_sophus_quat(1)
        try:
            a = _run(field, mu, False)
            b = _run(field, mu, True)
        finally:
            fma.so_set_sophus_quat(0)
        m = map_distance(a[0], b[0], relative_x=(field == OFUSION))
        r = raycast_distance(a[1], a[2], b[1], b[2], a[3])
        assert m["block_set_symdiff_frac"] <= 1e-3, m
        assert m["x_bit_identical_frac"] < 1.0, m
        assert m["x_gt_1e5_frac"] <= (1e-4 if field == SDF else 2e-2), m
        assert r["hitmask_disagree_frac"] <= 5e-3, r
        assert r["vert_le_0p5_vox_frac"] >= 0.98, r
