import os
import zlib

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def crc_rows(a):
    return np.array([zlib.crc32(np.ascontiguousarray(r).tobytes()) for r in a], np.uint32)


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def check_against_golden(g, coords, x, y, active, node_code, nx, ny, vertex, normal):
    assert coords.shape == g["coords"].shape and (coords == g["coords"]).all()
    assert (active == g["active"]).all()
    assert (crc_rows(x) == g["crc_x"]).all() and (crc_rows(y) == g["crc_y"]).all()
    assert np.array_equal(x.astype(np.float64).sum(1), g["sum_x"])
    assert (node_code == g["node_code"]).all()
    assert (crc_rows(nx) == g["node_crc_x"]).all() and (crc_rows(ny) == g["node_crc_y"]).all()
    assert (vertex.view(np.uint32) == g["vertex"].view(np.uint32)).all()
    assert (normal.view(np.uint32) == g["normal"].view(np.uint32)).all()
