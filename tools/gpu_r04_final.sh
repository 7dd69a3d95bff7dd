#!/bin/bash
# r04 evidence run: full rocprofv3 passes (kernel trace + FETCH / WRITE / SQ / cache / TLB / latency counters) of the four BASELINE
# workloads that fit one GPU, the tracking loop's kernel trace, then the BASELINE configs + stress legs through bench.py.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r04m}
SE_PROF_LAST=50 bash tools/gpu_profile.sh ${T} full > gpurun_out/${T}_prof512.txt 2>&1
SE_PROF_LAST=40 SE_PROF_RES=1024 bash tools/gpu_profile.sh ${T}_1024 full --res 1024 > gpurun_out/${T}_prof1024.txt 2>&1
SE_PROF_LAST=20 SE_PROF_RES=2048 SE_PROF_W=1280 SE_PROF_H=960 bash tools/gpu_profile.sh ${T}_2048 full --width 1280 --height 960 --res 2048 > gpurun_out/${T}_prof2048.txt 2>&1
SE_PROF_LAST=40 SE_PROF_FIELD=ofusion SE_PROF_MU=0.008 bash tools/gpu_profile.sh ${T}_of full --field ofusion --mu 0.008 > gpurun_out/${T}_profof.txt 2>&1
for t in ${T} ${T}_1024 ${T}_2048 ${T}_of; do cp gpurun_out/prof_$t/summary.md gpurun_out/${t}_rocprofv3_summary.md; cp gpurun_out/prof_$t/pmc_traffic.json gpurun_out/${t}_pmc_traffic.json 2>/dev/null; done
SE_CFG_SKIP_MU01=1 bash tools/gpu_configs.sh 2>&1 | tee gpurun_out/${T}_configs.log | cut -c1-300
for t in sdf512 sdf512_icl sdf1024 sdf2048 ofusion512; do cp gpurun_out/cfg_$t.json gpurun_out/${T}_cfg_$t.json; done
python bench.py --stream stress --steps 100 --warmup 10 --no-cpu-baseline --no-modes --detail gpurun_out/${T}_cfg_stress512.json > /dev/null 2> gpurun_out/${T}_stress.err
python bench.py --stream stress --res 1024 --steps 60 --warmup 10 --no-cpu-baseline --no-modes --sustain 100 --detail gpurun_out/${T}_cfg_stress1024.json > /dev/null 2>> gpurun_out/${T}_stress.err
python -c "
import json
for n in ('stress512','stress1024'):
    d=json.load(open('gpurun_out/${T}_cfg_%s.json' % n)); print(n, round(d['value']), round(d.get('value_closed_loop',0)), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()}, round(d['roofline']['frac'],3))"
