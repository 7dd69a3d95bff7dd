#!/bin/bash
# r03 run 4: device-resident ICP (two launches per iteration): tests + tracking leg; raycast priority base A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_tracking.py tests/test_gpu_render.py "tests/test_gpu_asbuilt_tolerance.py::test_tracked_pose_stays_near_the_as_built_reference" "tests/test_gpu_sharded.py::test_sharded_tracking_sees_the_full_images" -m gpu -x -q --durations=5) > gpurun_out/r03d_pytest_gpu.log 2>&1; tail -12 gpurun_out/r03d_pytest_gpu.log
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 > gpurun_out/r03d_bench.json 2> gpurun_out/r03d_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03d_bench.json')); print('value', d['value'], {k:{kk:vv for kk,vv in v.items() if kk in ('fps','closed_loop_fps','tracked_frames','final_position_error_m')} for k,v in d['modes'].items()})"; tail -3 gpurun_out/r03d_bench.err
python tools/lib_ab.py --cfgs sdf512,sdf1024 default 2>&1 | tee gpurun_out/r03_ab4.log
SE_HIP_PRIO_BASE=1 python tools/lib_ab.py --cfgs sdf512,sdf1024 default 2>&1 | sed 's/default/default(prio base 1)/' | tee -a gpurun_out/r03_ab4.log
