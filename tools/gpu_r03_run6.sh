#!/bin/bash
# r03 run 6: ICP with tracking_result_ written once per frame: tests + tracking leg; late host gate A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_tracking.py tests/test_gpu_render.py "tests/test_gpu_sharded.py::test_sharded_tracking_sees_the_full_images" tests/test_gpu_cpp_mirror.py -m gpu -x -q --durations=5) > gpurun_out/r03f_pytest_gpu.log 2>&1; tail -8 gpurun_out/r03f_pytest_gpu.log
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03f_bench.json')); print('value', d['value'], {k:{kk:vv for kk,vv in v.items() if kk in ('fps','closed_loop_fps','tracked_frames','final_position_error_m')} for k,v in d['modes'].items()})"; tail -3 gpurun_out/r03f_bench.err
python tools/lib_ab.py --cfgs sdf512,sdf1024,sdf2048 default 2>&1 | tee gpurun_out/r03_ab6.log
for g in 1000 600 300; do SE_HIP_GATE_LATE=$g python tools/lib_ab.py --cfgs sdf512,sdf1024,sdf2048 default 2>&1 | sed "s/default/default(gate late $g)/" | tee -a gpurun_out/r03_ab6.log; done
