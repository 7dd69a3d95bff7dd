"""Octree::save byte layout.  CPU: the oracle's dump parses with the documented layout and
round-trips its own map.  GPU: se_hip_save_map writes the same content (compared as key-sorted tables)."""
import numpy as np
import pytest

from oracle.binding import OFUSION, SDF, OraclePipeline
from supereight_amd.mapio import load_octree
from supereight_amd.synthetic import SyntheticStream

W, H, N, DIM = 80, 60, 128, 2.4


def build_cpu(field, mu, frames=4):
    s = SyntheticStream(W, H, DIM)
    o = OraclePipeline(field, N, DIM, W, H)
    for f in range(frames):
        o.integrate(s.depth(f), s.pose(f), s.k, mu, f)
    assert o.stats()["truncated"] == 0
    return o


@pytest.mark.parametrize("field,name,mu", [(SDF, "sdf", 0.1), (OFUSION, "ofusion", 0.02)], ids=["sdf", "ofusion"])
def test_oracle_dump_layout(tmp_path, field, name, mu):
    o = build_cpu(field, mu)
    path = str(tmp_path / "test.bin")
    assert o.save(path)
    d = load_octree(path, name)
    c, x, y, a = o.blocks()
    code, side, nx, ny = o.nodes()
    assert d["size"] == N and abs(d["dim"] - DIM) < 1e-6
    b = np.sort(d["blocks"], order="code")
    assert (b["coords"] == c).all() and (b["voxels"]["x"] == x).all() and (b["voxels"]["y"].astype(np.float32) == y).all()
    n = np.sort(d["nodes"], order="code")
    assert (n["code"] == code).all() and (n["side"] == side.astype(np.int32)).all() and (n["value"]["x"] == nx).all()
    expected = 16 + len(n) * (12 + 8 * (8 if field == SDF else 16)) + 8 + len(b) * (20 + 512 * (8 if field == SDF else 16))
    import os
    assert os.path.getsize(path) == expected


@pytest.mark.gpu
@pytest.mark.parametrize("field,name,mu", [(SDF, "sdf", 0.1), (OFUSION, "ofusion", 0.02)], ids=["sdf", "ofusion"])
def test_hip_dump_equals_oracle_dump(tmp_path, field, name, mu):
    from supereight_amd.pipeline import DenseSLAMPipeline
    o = build_cpu(field, mu)
    s = SyntheticStream(W, H, DIM)
    p = DenseSLAMPipeline((W, H), N, DIM, field_type=field)
    for f in range(4):
        p.set_depth(s.depth(f)); p.setPose(s.pose(f))
        p.integration(s.k, 1, mu, f)
    pa, pb = str(tmp_path / "cpu.bin"), str(tmp_path / "gpu.bin")
    assert o.save(pa)
    p.save(pb)
    a, b = load_octree(pa, name), load_octree(pb, name)
    assert a["size"] == b["size"] and a["dim"] == b["dim"]
    for key in ("nodes", "blocks"):
        ta, tb = np.sort(a[key], order="code"), b[key]
        assert (np.diff(tb["code"].astype(np.int64)) > 0).all()          # the HIP dump is key-sorted
        assert ta.tobytes() == tb.tobytes()
    p.close()
