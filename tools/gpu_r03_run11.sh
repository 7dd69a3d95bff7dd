#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_sharded.py tests/test_gpu_stress_parity.py -m gpu -x -q --durations=6) > gpurun_out/r03k_pytest_gpu.log 2>&1; tail -12 gpurun_out/r03k_pytest_gpu.log
python tools/soak.py 1080 512 320 240 > gpurun_out/r03_soak_512.json 2> gpurun_out/r03_soak.err; cat gpurun_out/r03_soak_512.json | cut -c1-1500; tail -2 gpurun_out/r03_soak.err
