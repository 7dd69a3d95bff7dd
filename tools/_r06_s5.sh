#!/bin/bash
# r06 session 5: geometric extrapolation of the march's speculative sample (0 = off, 1 = on, 2 = on + next z slice)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r06e smoke tests:"stress or parity or fused or schedule"
python tools/lib_ab.py --cfgs sdf1024,stress1024,sdf512,sdf2048,pooled1024 r06e_extra0 default r06e_extra2 r06e_extra0 default 2>&1 | tee gpurun_out/r06e_extrapolate_ab.log | cut -c1-420
SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py sdf1024 --closed > gpurun_out/r06e_wave_timeline_sdf1024_closed.txt 2>&1; head -12 gpurun_out/r06e_wave_timeline_sdf1024_closed.txt | cut -c1-400
SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py sdf512 > gpurun_out/r06e_wave_timeline_sdf512_fused.txt 2>&1; head -12 gpurun_out/r06e_wave_timeline_sdf512_fused.txt | cut -c1-400
