#!/bin/bash
# r03 run 7: raycast register budget forced to 6 / 8 waves per SIMD (80 / 64 VGPRs + scratch), 8 waves also with 4 LDS-staged levels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/lib_ab.py --cfgs sdf2048,sdf1024 default occ6 occ8 2>&1 | tee gpurun_out/r03_ab7.log
SE_HIP_RAY_CACHE_LEVELS=4 python tools/lib_ab.py --cfgs sdf2048,sdf1024 default occ8 2>&1 | sed 's/default:/default(4 LDS levels):/; s/occ8:/occ8(4 LDS levels):/' | tee -a gpurun_out/r03_ab7.log
