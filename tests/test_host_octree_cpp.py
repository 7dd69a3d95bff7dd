"""CPU test of the header-only host octree that DenseSLAMSystem::getMap() materialises (include/se/octree.hpp): a small
C++ program builds a one-block tree with the vectors of the reference's serialise tests (io_unittest.cpp:57-128), checks
the read interface (fetch / fetch_octant / get / get_fine, octree.hpp:340-478) and saves it; the files must parse as the
reference's Octree::save layout (octree.hpp:898-914) with exactly those records."""
import os
import subprocess

from supereight_amd.mapio import load_octree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _morton(x, y, z):
    k = 0
    for i in range(21):
        k |= ((x >> i) & 1) << (3 * i) | ((y >> i) & 1) << (3 * i + 1) | ((z >> i) & 1) << (3 * i + 2)
    return k


def test_host_octree_read_interface_and_save_layout(tmp_path):
    exe = str(tmp_path / "host_octree_kats")
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "host_octree_kats.cpp"), "-o", exe], check=True, capture_output=True)
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr
    for field, ybytes in (("sdf", 4), ("ofusion", 8)):
        path = str(tmp_path / f"{field}.bin")
        d = load_octree(path, field)
        assert d["size"] == 512 and d["dim"] == 5.0
        nodes, blocks = d["nodes"], d["blocks"]
        assert len(nodes) == 6 and len(blocks) == 1
        for l, n in enumerate(nodes):
            mask = ~((512 >> l) - 1) & 511
            assert int(n["code"]) == (_morton(40 & mask, 48 & mask, 56 & mask) | l) and int(n["side"]) == 512 >> l
            assert (n["value"]["x"] == 3.0).all() and (n["value"]["y"] == 4.0).all()
        b = blocks[0]
        assert int(b["code"]) == (_morton(40, 48, 56) | 6) and b["coords"].tolist() == [40, 48, 56]
        assert (b["voxels"]["x"] == 5.0).all() and (b["voxels"]["y"] == 2.0).all()
        # record sizes of the reference's build: 8 + 4 + 8 * sizeof(value) per node, 8 + 12 + 512 * sizeof(value) per block
        vs = 4 + (4 if field == "sdf" else 4 + 8)
        assert os.path.getsize(path) == 4 + 4 + 8 + 6 * (12 + 8 * vs) + 8 + (20 + 512 * vs)


def test_host_octree_interp_and_grad_equal_the_oracle(tmp_path):
    """include/se/octree.hpp interp / grad (octree.hpp:541-563, 565-737) against the oracle's restatement on the same map: a 64^3
    tree with a few blocks (neighbours on every axis, a block at the volume's far corner, gaps), sampled inside blocks, across
    block faces / edges / corners, next to missing blocks and at the volume boundary -- bit for bit."""
    import ctypes as C
    import numpy as np
    from oracle import binding
    exe = str(tmp_path / "host_octree_interp")
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "host_octree_interp.cpp"), "-o", exe], check=True, capture_output=True)
    N, dim = 64, 1.5
    blocks = [(8, 8, 8), (16, 8, 8), (8, 16, 8), (8, 8, 16), (16, 16, 16), (56, 56, 56), (0, 0, 0), (32, 8, 8)]
    rng = np.random.default_rng(7)
    pos = [rng.uniform(8, 24, 3) for _ in range(300)] + [rng.uniform(0, 64, 3) for _ in range(300)]
    pos += [(15.5, 10.2, 9.9), (15.25, 15.75, 9.0), (15.5, 15.5, 15.5), (7.9, 8.1, 8.1), (23.6, 15.5, 15.5), (63.5, 63.5, 63.5), (62.9, 60.1, 57.3),
            (0.2, 0.3, 0.1), (-0.4, 3.0, 2.0), (31.5, 10.0, 10.0), (39.7, 15.5, 8.2), (16.0, 8.0, 8.0), (15.999, 15.999, 15.999)]
    pos = np.asarray(pos, np.float32)
    txt = "\n".join("%.9g %.9g %.9g" % tuple(p) for p in pos)
    args = [str(c) for b in blocks for c in b]
    out = subprocess.run([exe] + args, input=txt, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = np.array([[int(w, 16) for w in ln.split()] for ln in out.stdout.strip().splitlines()], dtype=np.uint32)
    assert got.shape == (len(pos), 4)
    lib = binding.load()
    t = lib.so_ft_create(N, dim, 1.0, 1.0)      # SDF: empty().x = initValue().x = 1
    code, coords, isb = C.c_uint64(0), np.zeros(3, np.int32), C.c_int(0)
    for bx, by, bz in blocks:
        lib.so_ft_insert(t, bx, by, bz, -1, C.byref(code), coords, C.byref(isb))
        for z in range(8):
            for y in range(8):
                for x in range(8):
                    X, Y, Z = np.uint32(bx + x), np.uint32(by + y), np.uint32(bz + z)
                    with np.errstate(over="ignore"):
                        h = (X * np.uint32(73856093)) ^ (Y * np.uint32(19349663)) ^ (Z * np.uint32(83492791))
                    v = np.float32(np.float32(int(h) & 0xFFFF) / np.float32(65536.0) - np.float32(0.5))
                    lib.so_ft_set(t, bx + x, by + y, bz + z, float(v))
    ref = np.zeros((len(pos), 4), np.float32)
    g = np.zeros(3, np.float32)
    for i, p in enumerate(pos):
        ref[i, 0] = lib.so_ft_interp(t, float(p[0]), float(p[1]), float(p[2]))
        lib.so_ft_grad(t, float(p[0]), float(p[1]), float(p[2]), g)
        ref[i, 1:] = g
    lib.so_ft_destroy(t)
    bad = np.nonzero((ref.view(np.uint32) != got).any(axis=1))[0]
    assert bad.size == 0, (pos[bad[:5]], ref[bad[:5]], got[bad[:5]].view(np.float32))
    assert np.abs(ref[:, 1:]).max() > 0 and len(np.unique(ref[:, 0])) > 150
