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
