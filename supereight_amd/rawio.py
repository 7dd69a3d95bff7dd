"""SLAMBench ".raw" depth streams: the on-disk input format of the reference's apps.

Layout per frame (se_tools/scene2raw.cpp:170-176, reader se_apps/include/interface.h:384-426):
    uint32 w, h ; uint16 depth_mm[w*h] ; uint32 w, h ; uint8 rgb[w*h*3]
Frame stride = 16 + 2*w*h + 3*w*h bytes (1 536 016 at 640x480).  The depth frames go straight into
``DenseSLAMPipeline.set_depth_mm`` (mm2metersKernel fused into the upload)."""
from __future__ import annotations

import numpy as np


def write_raw(path: str, depth_mm_frames, rgb_frames=None) -> None:
    with open(path, "wb") as fh:
        for i, d in enumerate(depth_mm_frames):
            d = np.ascontiguousarray(d, dtype=np.uint16)
            h, w = d.shape
            np.array([w, h], np.uint32).tofile(fh)
            d.tofile(fh)
            np.array([w, h], np.uint32).tofile(fh)
            rgb = np.zeros((h, w, 3), np.uint8) if rgb_frames is None else np.ascontiguousarray(rgb_frames[i], np.uint8)
            rgb.tofile(fh)


def read_raw(path: str, with_rgb: bool = False):
    """Yields depth_mm (h, w) uint16 arrays (and rgb if asked) until the file ends."""
    with open(path, "rb") as fh:
        while True:
            hdr = np.fromfile(fh, np.uint32, 2)
            if hdr.size < 2:
                return
            w, h = int(hdr[0]), int(hdr[1])
            d = np.fromfile(fh, np.uint16, w * h)
            hdr2 = np.fromfile(fh, np.uint32, 2)
            rgb = np.fromfile(fh, np.uint8, w * h * 3)
            if d.size < w * h or hdr2.size < 2 or rgb.size < w * h * 3:
                return
            yield (d.reshape(h, w), rgb.reshape(h, w, 3)) if with_rgb else d.reshape(h, w)


def frame_stride(w: int, h: int) -> int:
    return 16 + 2 * w * h + 3 * w * h
