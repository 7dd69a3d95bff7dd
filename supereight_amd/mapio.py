"""Reader of the reference's octree dump (se::Octree<T>::save, se_core/include/se/octree.hpp:898-914).

Layout: int32 size, float dim, uint64 n_nodes, nodes {uint64 code, int32 side, value_[8]}, uint64 n_blocks,
blocks {uint64 code, int32 coords[3], voxel_block_[512]}; value_type is {float x, float y} for SDF and
{float x, <4 pad bytes>, double y} for OFusion.  (The reference's own Octree::load mis-reads `dim` as an
int and copies a single voxel per block, octree.hpp:921-947; this reader does neither.)"""
from __future__ import annotations

import numpy as np

SDF_VALUE = np.dtype([("x", "<f4"), ("y", "<f4")])
OFUSION_VALUE = np.dtype([("x", "<f4"), ("_pad", "<u4"), ("y", "<f8")])


def load_octree(path: str, field: str):
    value = SDF_VALUE if field == "sdf" else OFUSION_VALUE
    node = np.dtype([("code", "<u8"), ("side", "<i4"), ("value", value, 8)])
    block = np.dtype([("code", "<u8"), ("coords", "<i4", 3), ("voxels", value, 512)])
    with open(path, "rb") as fh:
        size = int(np.fromfile(fh, "<i4", 1)[0])
        dim = float(np.fromfile(fh, "<f4", 1)[0])
        nn = int(np.fromfile(fh, "<u8", 1)[0])
        nodes = np.fromfile(fh, node, nn)
        nb = int(np.fromfile(fh, "<u8", 1)[0])
        blocks = np.fromfile(fh, block, nb)
        rest = fh.read()
    if len(nodes) != nn or len(blocks) != nb or rest:
        raise ValueError("truncated or oversized octree file")
    return {"size": size, "dim": dim, "nodes": nodes, "blocks": blocks}
