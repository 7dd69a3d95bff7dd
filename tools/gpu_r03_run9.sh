#!/bin/bash
# r03 run 9: packed (two slices per instruction) SDF sweep vs the scalar one
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/lib_ab.py --cfgs sdf512,sdf1024,sdf2048,stress512 sweep_scalar default packed_4waves 2>&1 | tee gpurun_out/r03_ab9.log
(time python -m pytest tests/test_gpu_parity.py tests/test_golden_fixtures.py tests/test_gpu_properties.py "tests/test_gpu_stress_parity.py::test_stress_stream_parity[stress_sdf_640x480_512]" "tests/test_gpu_stress_parity.py::test_stress_stream_parity[stress_sdf_320x240_512_long_pooled]" -m gpu -x -q) > gpurun_out/r03h_pytest_gpu.log 2>&1; tail -5 gpurun_out/r03h_pytest_gpu.log
