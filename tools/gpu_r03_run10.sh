#!/bin/bash
# r03 run 10: closed-loop schedule (scan beside the first pass of the sweep): full GPU suite + closed-loop A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q --durations=5) > gpurun_out/r03j_pytest_gpu.log 2>&1; tail -8 gpurun_out/r03j_pytest_gpu.log
python tools/lib_ab.py --cfgs sdf512,sdf1024,of512,stress512,pooled512 default 2>&1 | tee gpurun_out/r03_ab10.log
SE_HIP_CLOSED_OVERLAP=0 python tools/lib_ab.py --cfgs sdf512,sdf1024,of512,stress512,pooled512 default 2>&1 | sed 's/default:/default(serial scan -> sweep):/' | tee -a gpurun_out/r03_ab10.log
