#!/bin/bash
# r06 session 6: the sibling-byte fetch of the > 512^3 search as LDS / global loads instead of one FLAT load
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r06f smoke tests:"stress or parity or fused or schedule or viewpoints"
python tools/lib_ab.py --cfgs sdf1024,stress1024,sdf2048,pooled1024,sdf512 r06e_extra0 default r06e_extra0 default 2>&1 | tee gpurun_out/r06f_flatfix_ab.log | cut -c1-420
SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py sdf1024 --closed > gpurun_out/r06f_wave_timeline_sdf1024_closed.txt 2>&1; head -12 gpurun_out/r06f_wave_timeline_sdf1024_closed.txt | cut -c1-400
SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py sdf2048 --closed > gpurun_out/r06f_wave_timeline_sdf2048_closed.txt 2>&1; head -12 gpurun_out/r06f_wave_timeline_sdf2048_closed.txt | cut -c1-400
