"""Pins the CPU oracle against every known-answer scenario the reference's own unit tests
hold for the hot path's substrate.  Each test names the reference
test it restates (paths relative to the reference checkout, se_core/test/...).
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

REF = "/root/reference"


def i3():
    return np.zeros(3, np.int32)


# ---- utils/morton_unittest.cpp:37-70 ---------------------------------------------------------
def test_morton_roundtrip_random(oracle):
    rng = np.random.default_rng(0)
    out = i3()
    for x, y, z in rng.integers(0, 4097, size=(1000, 3)):
        oracle.so_unpack_morton(oracle.so_compute_morton(int(x), int(y), int(z)), out)
        assert tuple(out) == (x, y, z)


def test_morton_roundtrip_exhaustive_slab(oracle):
    # ExhaustiveTest walks z in [2048,4096), y in {2048,2049}, x in [0,4096); sampled stride here
    out = i3()
    for z in range(2048, 4096, 37):
        for y in (2048, 2049):
            for x in range(0, 4096, 41):
                oracle.so_unpack_morton(oracle.so_compute_morton(x, y, z), out)
                assert tuple(out) == (x, y, z)


def test_key_kats_recorded_in_survey(oracle):
    # values obtained from the reference's octant_ops.hpp / morton_utils.hpp
    assert oracle.so_encode(136, 128, 136, 6, 9) == 0x0000000000E00A06
    assert oracle.so_encode(56, 12, 254, 6, 9) == 0x000000000092DE06
    assert oracle.so_encode(511, 511, 511, 6, 9) == 0x0000000007FFFE06
    assert oracle.so_compute_morton(1, 2, 4) == 0x111
    assert oracle.so_compute_morton(511, 0, 0) == 0x1249249
    out = i3()
    oracle.so_decode(0xE00A06, out)
    assert tuple(out) == (136, 128, 136)
    assert oracle.so_child_id(0xE00A06, 6, 9) == 5
    assert oracle.so_parent(0xE00A06, 9) == 0x0000000000E00005


def test_mask_table_matches_reference_literals(oracle):
    # se_core/include/se/octree_defines.h:58-80 (first / last entries as spot values + recipe)
    assert oracle.so_mask(0) == 0x7000000000000000
    assert oracle.so_mask(17) == 0x7FFFFFFFFFFFFE00
    assert oracle.so_mask(20) == 0x7FFFFFFFFFFFFFFF
    path = os.path.join(REF, "se_core/include/se/octree_defines.h")
    if os.path.exists(path):
        src = open(path).read()
        src = src[src.index("constexpr uint64_t MASK[]"):]
        lits = [int(h, 16) for h in re.findall(r"0x[0-9a-f]{16}", src)]
        assert len(lits) == 21
        assert [oracle.so_mask(i) for i in range(21)] == lits


# ---- octree/octree_unittest.cpp:36-217 -------------------------------------------------------
def test_octant_face_neighbours(oracle):
    octant, md, ld, side = (112, 80, 160), 8, 5, 8
    code = oracle.so_encode(*octant, ld, md)
    faces = [(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
    out = i3()
    for i, f in enumerate(faces):
        oracle.so_face_neighbour(code, i, ld, md, out)
        assert tuple(out) == tuple(o + side * d for o, d in zip(octant, f))


def test_octant_descendant(oracle):
    md = 8
    code = oracle.so_encode(110, 80, 159, 5, md)
    assert oracle.so_descendant(code, oracle.so_encode(96, 64, 128, 3, md), md)
    assert not oracle.so_descendant(code, oracle.so_encode(128, 64, 64, 3, md), md)


def test_octant_parent(oracle):
    md = 8
    code = oracle.so_encode(112, 80, 160, 5, md)
    p = oracle.so_parent(code, md)
    assert (code & ~0x1FF) == (p & ~0x1FF) and (p & 0x1FF) == 4
    p = oracle.so_parent(p, md)
    assert (p & 0x1FF) == 3 and p == oracle.so_encode(96, 64, 160, 3, md)
    p = oracle.so_parent(p, md)
    assert (p & 0x1FF) == 2 and p == oracle.so_encode(64, 64, 128, 2, md)


def test_far_corner(oracle):
    md, lvl = 5, 2
    cases = {(16, 16, 16): (16, 16, 16), (24, 16, 16): (32, 16, 16), (16, 24, 16): (16, 32, 16),
             (24, 24, 16): (32, 32, 16), (16, 16, 24): (16, 16, 32), (24, 16, 24): (32, 16, 32),
             (24, 24, 24): (32, 32, 32)}
    out = i3()
    for cell, want in cases.items():
        oracle.so_far_corner(oracle.so_encode(*cell, lvl, md), lvl, md, out)
        assert tuple(out) == want


def test_exterior_neighbours_inner_and_edge(oracle):
    md, lvl = 5, 2
    cell = oracle.so_encode(16, 16, 16, lvl, md)
    N = np.zeros(7, np.uint64)
    oracle.so_exterior_neighbours(N, cell, lvl, md)
    want = [(15, 16, 16), (16, 15, 16), (15, 15, 16), (16, 16, 15), (15, 16, 15), (16, 15, 15), (15, 15, 15)]
    p = oracle.so_parent(cell, md)
    for n, w in zip(N, want):
        assert int(n) == oracle.so_encode(*w, lvl, md)
        assert oracle.so_parent(int(n), md) != p
    cell = oracle.so_encode(0, 16, 16, lvl, md)
    oracle.so_exterior_neighbours(N, cell, lvl, md)
    out = i3()
    for n in N:
        oracle.so_unpack_morton(int(n) & ~0x1FF, out)
        assert (out >= 0).all() and (out <= 31).all()


def test_siblings(oracle):
    md, lvl = 5, 2
    cell = oracle.so_encode(16, 16, 16, lvl, md)
    s = np.zeros(8, np.uint64)
    oracle.so_siblings(s, cell, md)
    assert int(s[oracle.so_child_id(cell, lvl, md)]) == cell
    for k in s:
        assert oracle.so_parent(int(k), md) == oracle.so_parent(cell, md)


# ---- algorithms/unique_unittest.cpp:90-118 ---------------------------------------------------
def test_unique_filter_duplicates(oracle):
    t = oracle.so_ft_create(1 << 10, 10.0, 0.0, 0.0)
    blocks = [(56, 12, 12), (56, 12, 15), (128, 128, 128), (128, 128, 125), (128, 128, 127), (128, 136, 129),
              (128, 136, 127), (136, 128, 136), (128, 240, 136), (128, 241, 136)]
    keys = np.array([oracle.so_ft_hash(t, *b, -1) & 0xFFFFFFFF for b in blocks], np.uint32)  # MortonType = unsigned int
    last = oracle.so_unique_u32(keys, 10)
    assert all(keys[i] != keys[i - 1] for i in range(1, last))
    oracle.so_ft_destroy(t)


def _multiscale_keys(oracle):
    md = 10
    t = oracle.so_ft_create(1 << md, 10.0, 0.0, 0.0)
    root_side = 2 ** (md - 4)
    base = (64, 0, 64)
    keys = [oracle.so_ft_hash(t, *base, 4), oracle.so_ft_hash(t, base[0] + root_side // 2, base[1], base[2], 5),
            oracle.so_ft_hash(t, base[0] + root_side // 4, base[1], base[2], 5), oracle.so_ft_hash(t, 128, 24, 80, 5)]
    oracle.so_ft_destroy(t)
    return np.array(sorted(k & 0xFFFFFFFF for k in keys), np.uint32), md


def test_filter_ancestors(oracle):
    keys, md = _multiscale_keys(oracle)
    last = oracle.so_filter_ancestors_u32(keys, len(keys), md)
    assert all(keys[i] != keys[i - 1] for i in range(1, last))
    assert last == 3


def test_unique_multiscale(oracle):
    keys, md = _multiscale_keys(oracle)
    last = oracle.so_unique_multiscale_u32(keys, len(keys), 4)
    assert all(keys[i] != keys[i - 1] for i in range(1, last))
    assert last == 3


# ---- allocation/alloc_unittest.cpp:41-123 ----------------------------------------------------
def test_alloc_empty_single_voxel(oracle):
    t = oracle.so_ft_create(256, 5.0, 0.0, 0.0)
    assert oracle.so_ft_get(t, 25, 65, 127) == 0.0
    oracle.so_ft_destroy(t)


def test_alloc_set_single_voxel_and_fetch_octant(oracle):
    t = oracle.so_ft_create(256, 5.0, 0.0, 0.0)
    vox = (25, 65, 127)
    keys = np.array([oracle.so_ft_hash(t, *vox, -1)], np.uint64)
    oracle.so_ft_allocate(t, keys, 1)
    code, coords = C.c_uint64(), i3()
    assert oracle.so_ft_fetch(t, *vox, C.byref(code), coords)
    assert tuple(coords) == (24, 64, 120)
    oracle.so_ft_set(t, *vox, 2.0)
    assert oracle.so_ft_get(t, *vox) == 2.0
    side, mask = C.c_uint(), C.c_int()
    assert oracle.so_ft_fetch_octant(t, *vox, 3, C.byref(code), C.byref(side), C.byref(mask))
    assert side.value == 32
    oracle.so_ft_destroy(t)


def test_morton_prefix_mask(oracle):
    max_bits, block_side = 21, 8
    size = 2 ** max_bits
    rng = np.random.default_rng(7)
    vox = rng.integers(0, size, size=(10, 3))
    keys = [oracle.so_compute_morton(int(a), int(b), int(c)) for a, b, c in vox]
    leaf_level = max_bits - 3
    edge = size // 2
    out = i3()
    for level in range(0, leaf_level + 1):
        mask = oracle.so_mask(level + 0)  # shift = max_bits - max_level = 0
        for k in keys:
            oracle.so_unpack_morton(k & mask, out)
            assert (out % edge == 0).all()
        edge //= 2


# ---- multiscale/multiscale_unittest.cpp:58-185 -----------------------------------------------
TEN_BLOCKS = [(56, 12, 254), (87, 32, 423), (128, 128, 128), (136, 128, 128), (128, 136, 128), (136, 136, 128),
              (128, 128, 136), (136, 128, 136), (128, 136, 136), (136, 136, 136)]


def test_multiscale_init(oracle):
    t = oracle.so_ft_create(512, 5.0, 0.0, 1.0)
    assert oracle.so_ft_get(t, 137, 138, 130) == 1.0
    oracle.so_ft_destroy(t)


def test_multiscale_plain_alloc(oracle):
    t = oracle.so_ft_create(512, 5.0, 0.0, 1.0)
    keys = np.array([oracle.so_ft_hash(t, *b, -1) for b in TEN_BLOCKS[:2]], np.uint64)
    oracle.so_ft_allocate(t, keys, 2)
    oracle.so_ft_set(t, 56, 12, 254, 3.0)
    assert oracle.so_ft_get(t, 56, 12, 254) == 3.0
    assert oracle.so_ft_get(t, 106, 12, 254) == 1.0
    oracle.so_ft_destroy(t)


def test_multiscale_scaled_alloc_node_value_fallback(oracle):
    t = oracle.so_ft_create(512, 5.0, 0.0, 1.0)
    keys = np.array([oracle.so_ft_hash(t, 200, 12, 25, 5), oracle.so_ft_hash(t, 87, 32, 423, 5)], np.uint64)
    oracle.so_ft_allocate(t, keys, 2)
    assert oracle.so_ft_set_octant_value(t, 87, 32, 420, 5, 0, 10.0)
    assert oracle.so_ft_get(t, 87, 32, 420) == 10.0
    oracle.so_ft_destroy(t)


def test_multiscale_iterator_sides(oracle):
    t = oracle.so_ft_create(512, 5.0, 0.0, 1.0)
    keys = np.array([oracle.so_ft_hash(t, 56, 12, 254, -1)], np.uint64)
    oracle.so_ft_allocate(t, keys, 1)
    nb, nn = C.c_int(), C.c_int()
    oracle.so_ft_counts(t, C.byref(nb), C.byref(nn))
    assert (nb.value, nn.value) == (1, 6)
    sides, codes = np.zeros(nn.value, np.uint32), np.zeros(nn.value, np.uint64)
    oracle.so_ft_node_sides(t, sides, codes)
    assert list(sides) == [512, 256, 128, 64, 32, 16]
    oracle.so_ft_destroy(t)


def test_multiscale_children_mask(oracle):
    t = oracle.so_ft_create(512, 5.0, 0.0, 1.0)
    keys = np.array([oracle.so_ft_hash(t, *b, 5) for b in TEN_BLOCKS], np.uint64)
    oracle.so_ft_allocate(t, keys, 10)
    assert oracle.so_ft_check_children_mask(t) == 0
    oracle.so_ft_destroy(t)


def test_multiscale_octant_alloc(oracle):
    t = oracle.so_ft_create(512, 5.0, 0.0, 1.0)
    keys = np.array([oracle.so_ft_hash(t, *b, -1) for b in TEN_BLOCKS], np.uint64)
    keys[2] = keys[2] | np.uint64(3)
    keys[9] = keys[2] | np.uint64(5)
    oracle.so_ft_allocate(t, keys, 10)
    code, side, mask = C.c_uint64(), C.c_uint(), C.c_int()
    assert oracle.so_ft_fetch_octant(t, *TEN_BLOCKS[4], 3, C.byref(code), C.byref(side), C.byref(mask))
    assert not oracle.so_ft_fetch_octant(t, *TEN_BLOCKS[9], 6, C.byref(code), C.byref(side), C.byref(mask))
    oracle.so_ft_destroy(t)


def test_multiscale_single_and_multiple_insert(oracle):
    t = oracle.so_ft_create(512, 5.0, 0.0, 1.0)
    code, coords, isb = C.c_uint64(), i3(), C.c_int()
    oracle.so_ft_insert(t, 32, 208, 44, -1, C.byref(code), coords, C.byref(isb))
    assert isb.value == 1 and tuple(coords) == (32, 208, 40)
    oracle.so_ft_destroy(t)
    t = oracle.so_ft_create(1024, 10.0, 0.0, 1.0)
    rng = np.random.default_rng(1)
    leaves_level, edge = 7, 512
    side, mask, out = C.c_uint(), C.c_int(), i3()
    for lvl in range(1, leaves_level + 1):
        for _ in range(20):
            v = [int(a) for a in rng.integers(0, 1024, 3)]
            oracle.so_ft_insert(t, *v, lvl, C.byref(code), coords, C.byref(isb))
            assert oracle.so_ft_fetch_octant(t, *v, lvl, C.byref(code), C.byref(side), C.byref(mask))
            oracle.so_decode(code.value, out)
            assert tuple(out) == tuple(edge * (a // edge) for a in v)
            assert mask.value == 0
        edge //= 2
    oracle.so_ft_destroy(t)


# ---- interp/gather_unittest.cpp:63-187 -------------------------------------------------------
@pytest.mark.parametrize("base,mask", [((136, 128, 136), 0), ((132, 128, 135), 1), ((132, 135, 132), 2),
                                       ((135, 132, 132), 4), ((129, 135, 135), 3), ((135, 131, 135), 5),
                                       ((135, 135, 138), 6), ((135, 135, 135), 7)])
def test_gather_all_crossmask_cases(oracle, base, mask):
    t = oracle.so_ft_create(512, 5.0, 0.0, 1.0)
    keys = np.array([oracle.so_ft_hash(t, *b, -1) for b in TEN_BLOCKS], np.uint64)
    oracle.so_ft_allocate(t, keys, 10)
    assert oracle.so_ft_get(t, 137, 138, 130) == 1.0          # GatherTest.Init
    cm = ((base[0] % 8 == 7) << 2) | ((base[1] % 8 == 7) << 1) | (base[2] % 8 == 7)
    assert cm == mask
    pts = np.zeros(8, np.float32)
    oracle.so_ft_gather(t, *base, pts)
    assert (pts == 1.0).all()
    oracle.so_ft_destroy(t)


def test_gather_missing_block_gives_empty_and_allcross_gives_init(oracle):
    # interp_gather.hpp:44-58 (NULL block -> empty()) and :218-234 (case 7 -> get_fine -> initValue);
    # the all-cross observation is recorded in SURVEY.md section 8c(ii).
    t = oracle.so_ft_create(512, 5.0, 0.0, 1.0)
    pts = np.zeros(8, np.float32)
    oracle.so_ft_gather(t, 300, 300, 300, pts)
    assert (pts == 0.0).all()
    oracle.so_ft_gather(t, 303, 303, 303, pts)
    assert (pts == 1.0).all()
    oracle.so_ft_destroy(t)


# ---- octree/ray_iterator_unittest.cpp:46-87 --------------------------------------------------
def test_ray_iterator_fetch_along_ray(oracle):
    t = oracle.so_ft_create(512, 5.0, 0.0, 1.0)
    p = np.array([1.5, 1.5, 1.5], np.float32)
    d = np.array([0.5, 0.5, 0.5], np.float32)
    d = d / np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1]) + np.float32(d[2] * d[2]))
    voxelsize = np.float32(5.0) / np.float32(512)
    stepsize = np.float32(2) * (voxelsize * np.float32(8))
    tt = np.float32(0.6)
    keys = []
    for _ in range(4):
        vox = ((p + tt * d) / voxelsize).astype(np.int32)
        keys.append(oracle.so_ft_hash(t, int(vox[0]), int(vox[1]), int(vox[2]), -1))
        tt = np.float32(tt + stepsize)
    alloc = np.array(keys, np.uint64)
    oracle.so_ft_allocate(t, alloc.copy(), 4)
    codes, tcmin, tcmax, tmm = np.zeros(16, np.uint64), np.zeros(16, np.float32), np.zeros(16, np.float32), np.zeros(2, np.float32)
    n = oracle.so_ft_ray_blocks(t, p, d, 0.4, 4.0, codes, tcmin, tcmax, 16, tmm)
    assert n == 4
    assert list(codes[:4]) == keys
    # values recorded in SURVEY.md section 8c(ii) from the reference's own scenario
    assert [int(c) for c in codes[:4]] == [0xE3FE06, 0xFC0006, 0xFC0E06, 0xFC7E06]
    np.testing.assert_allclose(tcmin[:4], [0.514203, 0.649519, 0.784836, 1.055470], atol=2e-6)
    np.testing.assert_allclose(tcmax[3], 1.190785, atol=2e-6)
    np.testing.assert_allclose(tmm, [0.4, 4.0], atol=1e-6)
    oracle.so_ft_destroy(t)


# ---- bfusion/bspline_lookup.cc:36-37 ---------------------------------------------------------
def test_bspline_lookup_regeneration_matches_reference_literals(oracle):
    lut = np.zeros(1000, np.float32)
    oracle.so_bspline_lookup(lut)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "bspline_lookup_f32.npy"))
    assert (lut == gold).all()
    path = os.path.join(REF, "se_denseslam/src/bfusion/bspline_lookup.cc")
    if os.path.exists(path):
        src = open(path).read()
        body = src[src.index("bspline_lookup[1000]"):]
        body = body[body.index("{") + 1: body.index("}")]
        ref = np.array([float(v) for v in body.split(",") if v.strip()], np.float64).astype(np.float32)
        assert len(ref) == 1000 and (lut == ref).all()


def test_cvt_i32_x86_semantics(oracle):
    assert oracle.so_cvt_i32(3.9) == 3 and oracle.so_cvt_i32(-3.9) == -3
    for bad in (float("nan"), float("inf"), -float("inf"), 3e9, -3e9):
        assert oracle.so_cvt_i32(bad) == -2 ** 31
