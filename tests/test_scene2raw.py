"""ICL-NUIM scene files -> SLAMBench .raw (the reference's se_tools/scene2raw.cpp): ray length -> z-depth conversion."""
import numpy as np

from supereight_amd.rawio import ICL_SCENE_K, icl_ray_length_to_depth_mm, read_raw, scene2raw


def test_ray_length_to_z_depth(tmp_path):
    h, w = 480, 640
    fx, fy, u0, v0 = (float(v) for v in ICL_SCENE_K)
    # a fronto-parallel wall at z = 2.5 m: ray length = z * sqrt(x^2 + y^2 + 1)
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    norm = np.sqrt(((u - u0) / fx) ** 2 + ((v - v0) / fy) ** 2 + 1.0)
    dist = 2.5 * norm
    mm = icl_ray_length_to_depth_mm(dist)
    assert mm.dtype == np.uint16 and mm.shape == (h, w)
    assert set(np.unique(mm)) <= {2499, 2500}                  # truncation of 2500 -/+ a few ulp
    # the scalar recipe of scene2raw.cpp:97-108 on a few pixels
    for (uu, vv) in ((0, 0), (639, 479), (320, 240), (17, 400)):
        a = np.float64((np.float32(uu) - np.float32(u0)) / np.float32(fx))
        b = np.float64((np.float32(vv) - np.float32(v0)) / np.float32(fy))
        want = int(dist[vv, uu] * 1000 / np.sqrt(a * a + b * b + 1))
        assert int(mm[vv, uu]) == want
    # a scene directory of two frames
    for i in range(2):
        np.savetxt(tmp_path / f"scene_00_{i:04d}.depth", (dist + i).reshape(1, -1), fmt="%.6f")
    out = str(tmp_path / "scene.raw")
    assert scene2raw(str(tmp_path), out) == 2
    frames = list(read_raw(out))
    assert len(frames) == 2 and np.abs(frames[0].astype(int) - mm.astype(int)).max() <= 1     # (text round trip at 1e-6 m)
    assert frames[1][240, 320] in (3499, 3500)
