#!/bin/bash
# r06 session 8: dense vs pooled bricks at 2048^3 in the pipeline (64 GiB grid on a 288 GB part)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/lib_ab.py --cfgs sdf2048 default default@SE_HIP_DENSE_MAX_GIB=64 default default@SE_HIP_DENSE_MAX_GIB=64 2>&1 | tee gpurun_out/r06h_dense2048_ab.log | cut -c1-420
python bench.py --width 1280 --height 960 --res 2048 --steps 30 --warmup 8 --no-cpu-baseline --sustain 0 --no-modes --detail gpurun_out/r06h_cfg2048_pooled.json > /dev/null 2> gpurun_out/r06h_2048.err
SE_HIP_DENSE_MAX_GIB=64 python bench.py --width 1280 --height 960 --res 2048 --steps 30 --warmup 8 --no-cpu-baseline --sustain 0 --no-modes --detail gpurun_out/r06h_cfg2048_dense.json > /dev/null 2>> gpurun_out/r06h_2048.err
python -c "
import json
for n in ('pooled','dense'):
    d=json.load(open('gpurun_out/r06h_cfg2048_%s.json' % n)); print(n, round(d['value'],1), round(d.get('value_closed_loop',0),1), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()}, round(d['roofline']['frac'],3), d['roofline']['kernel'])"
tail -3 gpurun_out/r06h_2048.err
