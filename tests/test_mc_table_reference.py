"""Build-container pin of include/se_mc_table.h: case by case the same triangles, in the same order and orientation, as
the reference's triTable (se_core/include/se/algorithms/edge_tables.h:66).  /root/reference does not exist on the GPU
box, so the test skips there (it is not a gpu test)."""
import os
import re

import numpy as np
import pytest

from tests.test_mesh_export import load_table

REF = "/root/reference/se_core/include/se/algorithms/edge_tables.h"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present (GPU box)")
def test_triangle_table_equals_the_reference_literals():
    src = open(REF).read()
    body = src[src.index("int triTable[256][16]"):]
    rows = re.findall(r"\{([^{}]*)\}", body)[:256]
    ref = np.array([[int(v) for v in r.split(",") if v.strip()] for r in rows], np.int64)
    assert ref.shape == (256, 16)
    assert (load_table() == ref).all()
