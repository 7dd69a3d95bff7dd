#!/bin/bash
# r06 session 4: depth-hint prefetch (on / off), beam shell A/B, timelines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r06d smoke tests
python tools/lib_ab.py --cfgs sdf1024,stress1024,sdf512,of512,sdf2048 r06b default default@SE_HIP_PREFETCH=0 r06d_noshell 2>&1 | tee gpurun_out/r06d_prefetch_ab.log | cut -c1-420
SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py sdf1024 --closed > gpurun_out/r06d_wave_timeline_sdf1024_closed.txt 2>&1; head -12 gpurun_out/r06d_wave_timeline_sdf1024_closed.txt | cut -c1-400
