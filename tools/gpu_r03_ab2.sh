#!/bin/bash
# r03 run 2: GPU tests on the new code, scan variants + pooled overlap A/B, bench with the new legs, N = 2 dry run over gloo
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(time python -m pytest tests -m gpu -x -q --durations=8) > gpurun_out/r03b_pytest_gpu.log 2>&1; tail -15 gpurun_out/r03b_pytest_gpu.log
python tools/lib_ab.py --cfgs sdf512,sdf1024 base default scan2 scan4 2>&1 | tee gpurun_out/r03_ab2.log
python tools/lib_ab.py --cfgs sdf2048 default scan2 scan4 2>&1 | tee -a gpurun_out/r03_ab2.log
python tools/lib_ab.py --cfgs pooled512,pooled1024,pooled2048,stress512 default 2>&1 | tee -a gpurun_out/r03_ab2.log
SE_HIP_POOLED_OVERLAP=0 python tools/lib_ab.py --cfgs pooled512,pooled1024 default 2>&1 | sed 's/default/default(no pooled overlap)/' | tee -a gpurun_out/r03_ab2.log
python bench.py --detail gpurun_out/r03b_bench_detail.json > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err; tail -c 3000 gpurun_out/r03b_bench.json; tail -5 gpurun_out/r03b_bench.err
python bench.py --steps 20 --warmup 6 --config4 --no-modes --no-cpu-baseline --sustain 0 > gpurun_out/r03b_bench_config4.json 2> gpurun_out/r03b_bench_config4.err; tail -c 1200 gpurun_out/r03b_bench_config4.json; tail -5 gpurun_out/r03b_bench_config4.err
SE_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 5 --config4-steps 3 > gpurun_out/r03b_bench_gloo2.json 2> gpurun_out/r03b_bench_gloo2.err; tail -c 2500 gpurun_out/r03b_bench_gloo2.json; tail -8 gpurun_out/r03b_bench_gloo2.err
