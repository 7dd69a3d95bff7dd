#!/usr/bin/env python3
"""ICL-NUIM scene directory (scene_00_NNNN.depth ray-length files) -> SLAMBench .raw, as the reference's se_tools/scene2raw
does; then e.g.  python bench.py --raw scene.raw --traj livingRoom2.gt.freiburg   (BASELINE.json configs 1 / 3)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supereight_amd.rawio import scene2raw

if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit("usage: scene2raw.py <scene directory> <output.raw>")
    print(scene2raw(sys.argv[1], sys.argv[2]), "frames")
