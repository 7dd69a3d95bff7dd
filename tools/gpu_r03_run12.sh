#!/bin/bash
# r03 run 12: ICP with the wave-shuffle reduction order and the fused vertex+normal pyramid kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_tracking.py tests/test_gpu_render.py tests/test_gpu_cpp_mirror.py "tests/test_gpu_asbuilt_tolerance.py::test_tracked_pose_stays_near_the_as_built_reference" "tests/test_gpu_sharded.py::test_sharded_tracking_sees_the_full_images" -m gpu -x -q --durations=5) > gpurun_out/r03l_pytest_gpu.log 2>&1; tail -8 gpurun_out/r03l_pytest_gpu.log
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 > gpurun_out/r03l_bench.json 2> gpurun_out/r03l_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03l_bench.json')); print('value', d['value'], {k:{kk:vv for kk,vv in v.items() if kk in ('fps','closed_loop_fps','tracked_frames','final_position_error_m')} for k,v in d['modes'].items()})"; tail -3 gpurun_out/r03l_bench.err
OUT=gpurun_out/prof_r03l_track; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r03l_track -- python bench.py --steps 20 --warmup 10 --no-events --no-cpu-baseline --sustain 0 --mode-frames 30 > $OUT/bench.json 2> $OUT/trace.err
SE_PROF_LAST=30 python tools/summarize_prof.py $OUT r03l_track > $OUT/summary.md 2> $OUT/summary.err
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -size +4M -delete
grep -E "k_icp|k_vertex_normal|k_half" $OUT/summary.md | head -12
