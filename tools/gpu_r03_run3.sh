#!/bin/bash
# r03 run 3: device-resident ICP + integer scan: targeted tests, A/B, tracking / closed-loop legs, scale prediction
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_tracking.py tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_render.py tests/test_golden_fixtures.py "tests/test_gpu_stress_parity.py::test_stress_stream_parity[stress_sdf_640x480_512]" "tests/test_gpu_stress_parity.py::test_stress_stream_parity[stress_sdf_640x480_1024]" -m gpu -x -q --durations=5) > gpurun_out/r03c_pytest_gpu.log 2>&1; tail -12 gpurun_out/r03c_pytest_gpu.log
python tools/lib_ab.py --cfgs sdf512,sdf1024,sdf2048 scanfloat default 2>&1 | tee gpurun_out/r03_ab3.log
SE_HIP_SYNC_SPIN=0 python tools/lib_ab.py --cfgs sdf512 default 2>&1 | sed 's/default/default(blocking sync)/' | tee -a gpurun_out/r03_ab3.log
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03c_bench.json')); print('value', d['value'], {k:{kk:vv for kk,vv in v.items() if kk in ('fps','closed_loop_fps','tracked_frames','final_position_error_m')} for k,v in d['modes'].items()})"; tail -3 gpurun_out/r03c_bench.err
python tools/scale_predict.py --cfg 512 --out gpurun_out/r03_scale_prediction_512.json 2>&1 | tail -6
python tools/scale_predict.py --cfg 2048 --out gpurun_out/r03_scale_prediction_2048.json 2>&1 | tail -6
