#!/bin/bash
# r06 session 2: zero-copy host input + always-deferred marks: whole GPU suite, driver bench (cpp_mirror passes), wave timelines at 512 / 1024 / 2048
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r06b smoke tests
python bench.py --detail gpurun_out/r06b_bench_detail.json > gpurun_out/r06b_bench.json 2> gpurun_out/r06b_bench.err; tail -c 1500 gpurun_out/r06b_bench.json; tail -3 gpurun_out/r06b_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r06b_bench_detail.json')); print(json.dumps(d.get('cpp_mirror'))); print(d['value'], d.get('value_closed_loop'), d['roofline']['frac'])"
for c in sdf512 sdf1024 pooled1024 sdf2048; do
  SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py $c --closed > gpurun_out/r06b_wave_timeline_${c}_closed.txt 2>&1; cat gpurun_out/r06b_wave_timeline_${c}_closed.txt | cut -c1-400
done
SE_HIP_LIB=$PWD/gpurun_ab/wlog.so SE_HIP_WLOG=/tmp/wlog.bin python tools/wave_timeline.py sdf1024 > gpurun_out/r06b_wave_timeline_sdf1024_fused.txt 2>&1; cat gpurun_out/r06b_wave_timeline_sdf1024_fused.txt | cut -c1-400
python tools/lib_ab.py --cfgs sdf1024,stress1024,sdf512 r05 default 2>&1 | tee gpurun_out/r06b_ab.log | cut -c1-420
