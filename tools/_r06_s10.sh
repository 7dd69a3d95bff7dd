#!/bin/bash
# r06 session 10: in-band batches of one sample taken from its interpolation cell
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r06j smoke tests
python tools/lib_ab.py --cfgs sdf512,stress512,sdf1024,stress1024,sdf2048 r06j_nosolo default r06j_nosolo default 2>&1 | tee gpurun_out/r06j_band_solo_ab.log | cut -c1-420
SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py sdf512 > gpurun_out/r06j_wave_timeline_sdf512_fused.txt 2>&1; head -12 gpurun_out/r06j_wave_timeline_sdf512_fused.txt | cut -c1-400
SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py sdf1024 --closed > gpurun_out/r06j_wave_timeline_sdf1024_closed.txt 2>&1; head -12 gpurun_out/r06j_wave_timeline_sdf1024_closed.txt | cut -c1-400
