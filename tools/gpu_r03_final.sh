#!/bin/bash
# r03 final validation: build check, smoke, full GPU suite, default bench (the driver's command), the BASELINE configs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03g_smoke.log 2>&1; tail -2 gpurun_out/r03g_smoke.log
(time python -m pytest tests -m gpu -x -q --durations=10) > gpurun_out/r03g_pytest_gpu.log 2>&1; tail -16 gpurun_out/r03g_pytest_gpu.log
(time python bench.py --detail gpurun_out/r03g_bench_detail.json) > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err; tail -c 600 gpurun_out/r03g_bench.json; tail -4 gpurun_out/r03g_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03g_bench_driver.json 2> gpurun_out/r03g_bench_driver.err; python -c "
import json; d=json.load(open('gpurun_out/r03g_bench_driver.json')); print('driver-style', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('hbm_frac'), {k:round(v['frac'],3) for k,v in d['roofline_replay'].items() if isinstance(v,dict)})"
SE_CFG_SKIP_MU01=1 bash tools/gpu_configs.sh 2>&1 | tee gpurun_out/r03g_configs.log | cut -c1-400
python bench.py --stream stress --steps 100 --warmup 10 --no-cpu-baseline --no-modes --detail gpurun_out/cfg_stress512.json > /dev/null 2> gpurun_out/r03g_stress.err; python -c "
import json; d=json.load(open('gpurun_out/cfg_stress512.json')); print('stress512', round(d['value']), (d.get('sustained') or {}).get('fps'), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()}, round(d['roofline']['frac'],3))"
python bench.py --stream stress --res 1024 --steps 60 --warmup 10 --no-cpu-baseline --no-modes --sustain 100 --detail gpurun_out/cfg_stress1024.json > /dev/null 2>> gpurun_out/r03g_stress.err; python -c "
import json; d=json.load(open('gpurun_out/cfg_stress1024.json')); print('stress1024', round(d['value']), (d.get('sustained') or {}).get('fps'), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()}, round(d['roofline']['frac'],3))"
