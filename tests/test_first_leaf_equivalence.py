"""CPU test of the claim behind k_raycast's stack-free first-leaf search (se_first_leaf_lite): for every ray that is
regular at set-up and never descends from a cell it has, by t_corner, already left, the reference iterator's stack and `h`
carry no information -- a model without them (tests/cpp/first_leaf_equiv.cpp) returns the bit-identical t_min and the same
leaf-found decision as the oracle's restatement of se::ray_iterator (se_core/include/se/ray_iterator.hpp:53-226).  The
rays the model hands back (`flagged`, `irregular`) are the ones the kernel re-runs through the full iterator; the test also
bounds how many those are."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import binding
from supereight_amd.synthetic import make_stream, to_colmajor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fl") / "libfl.so")
    subprocess.run(["g++", "-std=c++17", "-O2", "-march=x86-64-v3", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
                    "-Wno-unknown-pragmas", "-o", so, os.path.join(ROOT, "tests", "cpp", "first_leaf_equiv.cpp")], check=True, capture_output=True)
    lib = binding._declare(C.CDLL(so))
    lib.fl_compare.restype = None
    lib.fl_compare.argtypes = [C.c_void_p, binding.c_f32p, binding.c_f32p, np.ctypeslib.ndpointer(np.int64), np.ctypeslib.ndpointer(np.int32), C.c_int]
    return lib


def _run(lib, field, kind, W, H, N, mu, frames, pose_shift=None, beam=0, first_view=3):
    st = make_stream(kind, W, H, 4.8)
    h = lib.so_pipe_create(field, N, 4.8, W, H)
    tot = np.zeros(8, np.int64)
    try:
        for f in range(frames):
            d = np.ascontiguousarray(st.depth(f), np.float32).reshape(-1)
            pose = st.pose(f)
            k = np.asarray(st.k, np.float32)
            lib.so_pipe_integrate(h, d, to_colmajor(pose), k, 1, mu, f)
            if f >= first_view:
                view = pose.copy()
                if pose_shift is not None:
                    view[:3, 3] += np.asarray(pose_shift, np.float32)
                out = np.zeros(8, np.int64)
                bad = np.zeros(2, np.int32)
                lib.fl_compare(h, to_colmajor(view), k, out, bad, beam)
                assert out[3] == 0 and out[7] == 0, f"frame {f}: {out[3]} rays differ from the iterator (e.g. pixel {bad.tolist()}), model errors {out[7]}"
                tot += out
    finally:
        lib.so_pipe_destroy(h)
    return dict(zip(("rays", "irregular", "flagged", "mismatch", "found", "trips_ref", "trips_lite", "model_bug"), tot.tolist()))


@pytest.mark.parametrize("field,kind,N,mu", [(binding.SDF, "room", 512, 0.1), (binding.SDF, "stress", 256, 0.1), (binding.OFUSION, "stress", 512, 0.02),
                                             (binding.SDF, "stress", 1024, 0.1)])
def test_stack_free_search_equals_the_iterator(lib, field, kind, N, mu):
    r = _run(lib, field, kind, 320, 240, N, mu, 7)
    assert r["rays"] == 4 * 320 * 240 and r["found"] > 0
    assert r["irregular"] == 0                      # the camera is inside the volume: every ray is regular at set-up
    assert r["flagged"] <= r["rays"] // 20000       # the edge-grazing descents are a handful per million rays
    assert r["trips_lite"] <= r["trips_ref"]        # (a handed-back ray stops early; the others take the same trips)


def test_rays_that_miss_the_volume_are_handed_back(lib):
    # camera pulled 6 m out of the 4.8 m volume: most rays enter through a face (regular), the rest miss it (irregular set-up);
    # both kinds must agree with the iterator or be handed back
    r = _run(lib, binding.SDF, "room", 160, 120, 256, 0.1, 5, pose_shift=(0.0, 0.0, -6.0))
    assert r["irregular"] > 0 and r["mismatch"] == 0


@pytest.mark.parametrize("field,kind,W,H,N,mu,frames,first_view", [
    (binding.SDF, "room", 640, 480, 512, 0.1, 6, 3),         # the benchmark's geometry
    (binding.SDF, "stress", 320, 240, 512, 0.1, 40, 30),     # the pan: occluders, depth edges, the volume's faces, rays that leave the cube
    (binding.SDF, "stress", 320, 240, 1024, 0.1, 8, 4),
    (binding.OFUSION, "stress", 320, 240, 512, 0.02, 8, 4),  # coarse childless octants in the tree (not blocks: they never stop the search)
    (binding.SDF, "room", 160, 120, 128, 0.1, 6, 3),         # leaf level 4 < 5: the coarse grid IS the block grid
])
def test_beam_start_returns_the_iterators_leaf(lib, field, kind, W, H, N, mu, frames, first_view):
    """r05: with every 8x8 tile's rays entering the tree at the tile's t_safe (se_beam_start, restated in tests/cpp/first_leaf_equiv.cpp), the
    first leaf and its entry time are bit for bit those of the reference iterator started at the near plane, for every ray that is not handed
    back -- and the search takes fewer than half the trips."""
    base = _run(lib, field, kind, W, H, N, mu, frames, first_view=first_view)
    r = _run(lib, field, kind, W, H, N, mu, frames, beam=1, first_view=first_view)
    r2 = _run(lib, field, kind, W, H, N, mu, frames, beam=2, first_view=first_view)
    print("two-stage:", round(r2["trips_lite"] / r2["rays"], 2), "mismatch", r2["mismatch"], "flagged", r2["flagged"])
    print(W, H, N, "trips per ray:", round(base["trips_lite"] / base["rays"], 2), "->", round(r["trips_lite"] / r["rays"], 2), "handed back:", r["flagged"], "of", r["rays"])
    assert r["rays"] == base["rays"] and r["mismatch"] == 0 and r["model_bug"] == 0
    assert r["found"] + r["flagged"] >= base["found"]              # (a handed-back ray is not counted as found)
    assert r["flagged"] <= base["flagged"] + r["rays"] // 5000
    if N >= 512:
        assert r["trips_lite"] < 0.85 * base["trips_lite"]


def test_beam_start_from_outside_the_volume(lib):
    r = _run(lib, binding.SDF, "room", 160, 120, 256, 0.1, 5, pose_shift=(0.0, 0.0, -6.0), beam=1)
    assert r["irregular"] > 0 and r["mismatch"] == 0
