cd $GRAFT_REPO_ROOT
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/drv_pytest.log 2>&1; grep -E "passed|failed|real" gpurun_out/drv_pytest.log | tail -3
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | grep -E "smoke|real"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/drv_bench.log 2> gpurun_out/drv_bench.err; tail -c 600 gpurun_out/drv_bench.log | head -c 300; echo; grep real gpurun_out/drv_bench.err
( time python bench.py ) > gpurun_out/drv_bench_default.log 2> gpurun_out/drv_bench_default.err; python -c "
import json; d=json.loads(open('gpurun_out/drv_bench_default.log').read().strip().splitlines()[-1]); print(d['metric'], round(d['value']), d['steps'], d['roofline']['frac'], d['cpu_baseline']['value'])"; grep real gpurun_out/drv_bench_default.err
