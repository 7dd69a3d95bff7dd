"""The C++ DenseSLAMSystem mirror (include/se/DenseSLAMSystem.h) driven by examples/denseslam_raw on a
SLAMBench .raw stream must reproduce the ctypes path bit for bit (same library underneath)."""
import os
import subprocess

import numpy as np
import pytest

from supereight_amd.pipeline import OFUSION, SDF, DenseSLAMPipeline
from supereight_amd.rawio import frame_stride, read_raw, write_raw
from supereight_amd.synthetic import SyntheticStream, render_depth_mm

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("field,exe,mu", [(SDF, "denseslam_raw", 0.1), (OFUSION, "denseslam_raw_ofusion", 0.02)], ids=["sdf", "ofusion"])
def test_cpp_mirror_matches_ctypes_path(tmp_path, field, exe, mu):
    binary = os.path.join(ROOT, "examples", exe)
    if not os.path.exists(binary):
        import __graft_entry__ as g
        g.build_examples()
    W, H, N, dim, frames = 160, 120, 256, 2.4, 5
    s = SyntheticStream(W, H, dim, holes=False)
    mm = [render_depth_mm(f, W, H, dim) for f in range(frames)]
    poses = np.stack([s.pose(f) for f in range(frames)]).astype(np.float32)
    raw, pf, out = str(tmp_path / "scene.raw"), str(tmp_path / "poses.bin"), str(tmp_path / "out.bin")
    write_raw(raw, mm)
    assert os.path.getsize(raw) == frames * frame_stride(W, H)
    assert all((a == b).all() for a, b in zip(read_raw(raw), mm))
    poses.tofile(pf)
    r = subprocess.run([binary, raw, pf, str(N), str(dim), str(mu), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    p = DenseSLAMPipeline((W, H), N, dim, field_type=field)
    for f in range(frames):
        p.set_depth_mm(mm[f]); p.setPose(poses[f])
        p.integration(s.k, 1, mu, f)
        p.raycasting(s.k, mu, f)
    v, n = p.vertex_normal()
    nb, nn = p.counts()
    data = np.fromfile(out, np.uint8)
    hdr = data[:16].view(np.int32)
    assert tuple(hdr) == (W, H, nb, frames)
    vn = data[16:].view(np.float32)
    assert (vn[: W * H * 3].view(np.uint32) == v.reshape(-1).view(np.uint32)).all()
    assert (vn[W * H * 3:].view(np.uint32) == n.reshape(-1).view(np.uint32)).all()
    assert f"blocks {nb} nodes {nn}" in r.stdout
    # getMap() as a host se::Octree (include/se/octree.hpp): complete, consistent, and its save() writes the bytes of
    # se_hip_save_map; a second DenseSLAMSystem restored with loadMap() raycasts the same images
    assert f"octree blocks {nb} nodes {nn} fetch_bad 0" in r.stdout and "reload_identical 1" in r.stdout
    assert "coarse_checked 0" not in r.stdout
    ref = str(tmp_path / "ref.bin")
    p.save(ref)
    assert open(out + ".octree", "rb").read() == open(ref, "rb").read()
    assert open(out + ".devmap", "rb").read() == open(ref, "rb").read()
    p.close()
