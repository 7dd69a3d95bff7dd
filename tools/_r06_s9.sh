#!/bin/bash
# r06 session 9: dense brick grid in tiles of 8^3 blocks (vs rows), dense default at 2048^3; whole suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r06i smoke tests
python tools/lib_ab.py --cfgs sdf512,sdf1024,stress1024,sdf2048,of512 r06i_rows default r06i_rows default 2>&1 | tee gpurun_out/r06i_tiled_ab.log | cut -c1-420
