#!/bin/bash
# first measurement pass on the GPU box: bench (with and without per-kernel events) + rocprofv3 kernel trace
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 100 --warmup 10 --detail gpurun_out/bench_detail.json > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.err
cat gpurun_out/bench.json
python bench.py --steps 100 --warmup 10 --no-events --no-cpu-baseline > gpurun_out/bench_noevents.json 2>> gpurun_out/bench.err
cat gpurun_out/bench_noevents.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python bench.py --steps 100 --warmup 10 --no-events --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err
tail -5 gpurun_out/prof.err
find gpurun_out/prof -name '*stats*' | head; 
for f in $(find gpurun_out/prof -name '*kernel_stats*csv'); do head -20 $f; done
