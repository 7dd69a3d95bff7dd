#!/bin/bash
# last check of the committed tree: smoke, full GPU suite (bounded oracle threads), the driver's bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time python -m pytest tests -m gpu -x -q --durations=5) > gpurun_out/r03i_pytest_gpu.log 2>&1; tail -12 gpurun_out/r03i_pytest_gpu.log
(time python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/r03i_bench_driver.json 2> gpurun_out/r03i_bench_driver.err; python -c "
import json; d=json.loads(open('gpurun_out/r03i_bench_driver.json').read().splitlines()[0]); print('driver-style', round(d['value']), d['ms_per_step'], round(d['roofline']['frac'],3), {k:round(v['fps']) for k,v in d['modes'].items()}, d['cpu_baseline']['value'])"; tail -4 gpurun_out/r03i_bench_driver.err
