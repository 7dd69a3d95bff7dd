#!/bin/bash
# The round's last evidence run (the map kernels are those of profiles/r04r_*; what changed since is the tracked loop and the Python host):
# smoke, every GPU test, the driver's two bench commands, the tracked loop's probe + kernel trace, BASELINE configs 2..5 + the stress stream.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r04u}
bash tools/gpu_run.sh $T smoke tests bench driver
bash tools/gpu_track.sh $T trace
SE_CFG_SKIP_MU01=1 bash tools/gpu_configs.sh 2>&1 | tee gpurun_out/${T}_configs.log | cut -c1-320
for t in sdf512 sdf512_icl sdf1024 sdf2048 ofusion512; do cp gpurun_out/cfg_$t.json gpurun_out/${T}_cfg_$t.json; done
python bench.py --stream stress --steps 100 --warmup 10 --no-cpu-baseline --no-modes --detail gpurun_out/${T}_cfg_stress512.json > /dev/null 2> gpurun_out/${T}_stress.err
python bench.py --stream stress --res 1024 --steps 60 --warmup 10 --no-cpu-baseline --no-modes --sustain 100 --detail gpurun_out/${T}_cfg_stress1024.json > /dev/null 2>> gpurun_out/${T}_stress.err
python -c "
import json
for n in ('stress512','stress1024'):
    d=json.load(open('gpurun_out/${T}_cfg_%s.json' % n)); print(n, round(d['value']), round(d.get('value_closed_loop',0)), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()}, round(d['roofline']['frac'],3))"
