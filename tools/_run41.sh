cd $GRAFT_REPO_ROOT
export SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin
timeout 300 python tools/_waveprobe.py room ofusion 512 0.008 2>&1 | grep -v "amdgpu.ids\|SE_HIP_LIB"
timeout 300 python tools/_waveprobe.py room sdf 512 0.1 2>&1 | grep -v "amdgpu.ids\|SE_HIP_LIB"
