#!/bin/bash
# compiles the current source tree into gpurun_ab/<name>.so (extra -D flags allowed) for same-box A/B runs via SE_HIP_LIB
name=$1; shift
mkdir -p "$(dirname "$0")/../gpurun_ab"
cd "$(dirname "$0")/../supereight_amd" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value "$@" -I../include -o ../gpurun_ab/$name.so csrc/se_hip_api.hip 2>&1 | grep -v hip-link; true
