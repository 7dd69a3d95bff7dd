"""SURVEY 8(f-4), second half: marching cubes (DenseSLAMSystem::dump_mesh).

CPU part: the triangle table (include/se_mc_table.h: the standard published table, pinned against the
reference's literals by tests/test_mc_table_reference.py) is checked for what a marching-cubes table must
guarantee -- triangles only on crossed edges, consistently oriented surfaces, closed on fields without
ambiguous faces -- and the oracle's restatement of
se::algorithms::marching_cube is checked on the analytic scene.  GPU part: the HIP kernel produces the
oracle's triangle set bit for bit."""
import os
import re
import sys

import numpy as np
import pytest

from oracle.binding import OFUSION, SDF, OraclePipeline
from supereight_amd.synthetic import SyntheticStream, surface_distance

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNER = np.array([(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)])
EDGE = [(0, 1), (1, 2), (2, 3), (0, 3), (4, 5), (5, 6), (6, 7), (4, 7), (0, 4), (1, 5), (2, 6), (3, 7)]


def load_table():
    """include/se_mc_table.h expanded the way se_mc_expand() does it."""
    text = open(os.path.join(ROOT, "include", "se_mc_table.h")).read()
    body = text[text.index("SE_MC_PACKED[256]"):text.index("};")]
    packed = re.findall(r'"([0-9a-b]*)"', body)
    assert len(packed) == 256
    t = -np.ones((256, 16), np.int64)
    for c, p in enumerate(packed):
        assert len(p) % 3 == 0 and len(p) <= 15
        t[c, :len(p)] = [int(ch, 16) for ch in p]
    return t


def test_table_triangles_use_crossed_edges_only():
    t = load_table()
    for case in range(256):
        inside = [(case >> i) & 1 for i in range(8)]
        crossed = {e for e, (a, b) in enumerate(EDGE) if inside[a] != inside[b]}
        used = {int(e) for e in t[case] if e >= 0}
        assert used == crossed, case                       # every crossed edge carries a vertex, no other edge does
        assert (t[case] >= 0).sum() % 3 == 0
    assert (t[0] == -1).all() and (t[255] == -1).all()


def _mesh_grid(field, t):
    """Marching cubes of a scalar grid with the table; vertices identified by their grid edge."""
    n = field.shape[0]
    tris = []
    for x in range(n - 1):
        for y in range(n - 1):
            for z in range(n - 1):
                c = CORNER + (x, y, z)
                case = sum(1 << i for i in range(8) if field[tuple(c[i])] < 0)
                row = t[case]
                for k in range(0, 15, 3):
                    if row[k] < 0:
                        break
                    tri = []
                    for e in row[k:k + 3]:
                        a, b = c[EDGE[e][0]], c[EDGE[e][1]]
                        tri.append((tuple(np.minimum(a, b)), tuple(np.maximum(a, b))))
                    tris.append(tri)
    return tris


def test_table_gives_closed_oriented_surfaces_on_random_fields():
    t = load_table()
    rng = np.random.default_rng(3)
    n = 9
    field = rng.standard_normal((n, n, n))                 # white noise: all 256 cases, ambiguous faces included
    field[0], field[-1], field[:, 0], field[:, -1], field[:, :, 0], field[:, :, -1] = (1.0,) * 6   # outside on the border -> closed
    tris = _mesh_grid(field, t)
    assert len(tris) > 500
    from collections import Counter
    directed = Counter()
    for tri in tris:
        for i in range(3):
            directed[(tri[i], tri[(i + 1) % 3])] += 1
    assert all(v == 1 for v in directed.values())                          # no edge used twice in the same direction
    assert all((b, a) in directed for (a, b) in directed)                  # every edge has its opposite: closed, oriented
    # orientation: normals point towards the outside (positive field), checked on a sphere
    g = np.indices((13, 13, 13)).transpose(1, 2, 3, 0) - 6.0
    sphere = np.linalg.norm(g, axis=-1) - 4.3
    for tri in _mesh_grid(sphere, t):
        p = []
        for a, b in tri:
            fa, fb = sphere[a], sphere[b]
            p.append(np.array(a) + (0 - fa) / (fb - fa) * (np.array(b) - np.array(a)))
        nrm = np.cross(p[1] - p[0], p[2] - p[1])
        assert np.dot(nrm, (p[0] + p[1] + p[2]) / 3 - 6.0) > 0


@pytest.mark.parametrize("field,mu", [(SDF, 0.1), (OFUSION, 0.02)], ids=["sdf", "ofusion"])
def test_oracle_mesh_lies_on_the_analytic_surface(field, mu):
    W, H, N, dim = 160, 120, 256, 2.4
    s = SyntheticStream(W, H, dim)
    o = OraclePipeline(field, N, dim, W, H)
    for f in range(4):
        o.integrate(s.depth(f), s.pose(f), s.k, mu, f)
    m = o.mesh()
    assert m.shape[0] > 20000
    d = surface_distance(m.reshape(-1, 3), dim)
    voxel = dim / N
    assert d.mean() < 0.5 * voxel and np.percentile(d, 99) < 1.5 * voxel
    assert (m > 0).all() and (m <= dim).all()              # checkVertex
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("field,mu,N", [(SDF, 0.1, 256), (OFUSION, 0.02, 256), (SDF, 0.1, 512)], ids=["sdf", "ofusion", "sdf-512"])
def test_gpu_mesh_equals_oracle(field, mu, N, tmp_path):
    from supereight_amd.pipeline import DenseSLAMPipeline
    W, H, dim = 160, 120, 2.4
    s = SyntheticStream(W, H, dim)
    o = OraclePipeline(field, N, dim, W, H)
    g = DenseSLAMPipeline((W, H), N, dim, field_type=field)
    for f in range(4):
        depth, pose = s.depth(f), s.pose(f)
        o.integrate(depth, pose, s.k, mu, f)
        g.set_depth(depth); g.setPose(pose); g.integration(s.k, 1, mu, f)
    mo, mg = o.mesh(), g.mesh()
    assert mo.shape == mg.shape and mo.shape[0] > 20000

    def canon(m):
        b = np.ascontiguousarray(m.reshape(-1, 9)).view(np.uint32)
        return b[np.lexsort(b.T[::-1])]
    assert (canon(mo) == canon(mg)).all()                  # same triangles, bit for bit (order is unspecified)
    # dump_mesh: writeVtkMesh's layout
    path = str(tmp_path / "mesh.vtk")
    g.dump_mesh(path)
    lines = open(path).read().split("\n")
    assert lines[:4] == ["# vtk DataFile Version 1.0", "vtk mesh generated from KFusion", "ASCII", "DATASET POLYDATA"]
    n = mg.shape[0]
    assert lines[4] == f"POINTS {3 * n} FLOAT" and lines[5 + 3 * n] == f"POLYGONS {n} {4 * n}"
    assert lines[6 + 3 * n] == "3 0 1 2" and lines[5 + 3 * n + n] == f"3 {3 * n - 3} {3 * n - 2} {3 * n - 1}"
    pts = np.array([[float(v) for v in ln.split()] for ln in lines[5:5 + 3 * n]], np.float64)
    assert np.abs(np.sort(pts, axis=0) - np.sort(mg.reshape(-1, 3).astype(np.float64), axis=0)).max() < 1e-5 * dim
    # pooled bricks: same mesh
    gp = DenseSLAMPipeline((W, H), N, dim, field_type=field, max_blocks=20000)
    s2 = SyntheticStream(W, H, dim)
    for f in range(4):
        gp.set_depth(s2.depth(f)); gp.setPose(s2.pose(f)); gp.integration(s2.k, 1, mu, f)
    assert (canon(gp.mesh()) == canon(mg)).all()
    o.close(); g.close(); gp.close()
