#!/bin/bash
# The evidence runs of a round (one gpurun call each; r04: tags r04m, r04r, r04u, r04y -- r05: r05k, r05z -- r06: r06k (final + last + gloo2)):
#   tools/gpu_evidence.sh final [tag]   full rocprofv3 passes (kernel trace + FETCH / WRITE / SQ / cache / TLB / latency counters, separate runs) of the
#                                           four BASELINE workloads that fit one GPU, then configs 2..5 + the stress stream through bench.py   (r04m, r04r)
#   tools/gpu_evidence.sh last [tag]    smoke, every GPU test, the driver's two bench commands, the tracked loop's probe + kernel trace, configs + stress (r04u)
#   tools/gpu_evidence.sh close [tag]   the full rocprofv3 passes of the headline workload, every GPU test, the driver's two bench commands        (r04y)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
MODE=${1:-close}
T=${2:-r06}
configs_and_stress() {
  SE_CFG_SKIP_MU01=1 bash tools/gpu_configs.sh 2>&1 | tee gpurun_out/${T}_configs.log | cut -c1-320
  for t in sdf512 sdf512_icl sdf1024 sdf2048 ofusion512; do cp gpurun_out/cfg_$t.json gpurun_out/${T}_cfg_$t.json; done
  python bench.py --stream stress --steps 100 --warmup 10 --no-cpu-baseline --no-modes --detail gpurun_out/${T}_cfg_stress512.json > /dev/null 2> gpurun_out/${T}_stress.err
  python bench.py --stream stress --res 1024 --steps 60 --warmup 10 --no-cpu-baseline --no-modes --sustain 100 --detail gpurun_out/${T}_cfg_stress1024.json > /dev/null 2>> gpurun_out/${T}_stress.err
  python -c "
import json
for n in ('stress512','stress1024'):
    d=json.load(open('gpurun_out/${T}_cfg_%s.json' % n)); print(n, round(d['value']), round(d.get('value_closed_loop',0)), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()}, round(d['roofline']['frac'],3))"
}
keep_profile() { cp gpurun_out/prof_$1/summary.md gpurun_out/${1}_rocprofv3_summary.md; cp gpurun_out/prof_$1/pmc_traffic.json gpurun_out/${1}_pmc_traffic.json 2>/dev/null; }
case $MODE in
  final)
    SE_PROF_LAST=50 bash tools/gpu_profile.sh ${T} full > gpurun_out/${T}_prof512.txt 2>&1
    SE_PROF_LAST=40 SE_PROF_RES=1024 bash tools/gpu_profile.sh ${T}_1024 full --res 1024 > gpurun_out/${T}_prof1024.txt 2>&1
    SE_PROF_LAST=20 SE_PROF_RES=2048 SE_PROF_W=1280 SE_PROF_H=960 bash tools/gpu_profile.sh ${T}_2048 full --width 1280 --height 960 --res 2048 > gpurun_out/${T}_prof2048.txt 2>&1
    SE_PROF_LAST=40 SE_PROF_FIELD=ofusion SE_PROF_MU=0.008 bash tools/gpu_profile.sh ${T}_of full --field ofusion --mu 0.008 > gpurun_out/${T}_profof.txt 2>&1
    for t in ${T} ${T}_1024 ${T}_2048 ${T}_of; do keep_profile $t; done
    configs_and_stress ;;
  last)
    bash tools/gpu_run.sh $T smoke tests bench driver
    bash tools/gpu_track.sh $T trace
    configs_and_stress ;;
  close)
    SE_PROF_LAST=50 bash tools/gpu_profile.sh ${T} full > gpurun_out/${T}_prof512.txt 2>&1
    keep_profile $T
    grep -E "k_raycast|k_integrate|k_alloc_scan" gpurun_out/${T}_rocprofv3_summary.md | head -12 | cut -c1-200
    bash tools/gpu_run.sh $T smoke tests bench driver ;;
  gloo2)
    # code-path dry run of the N > 1 bench on ONE GPU: two ranks over gloo, both schedules; the line is not a performance figure
    for mode in two_queue one_queue; do
      extra=""; [ $mode = one_queue ] && extra="--sharded-streaming"
      SE_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline $extra > gpurun_out/${T}_bench_gloo2_dryrun_$mode.json 2> gpurun_out/${T}_gloo2_$mode.err
      python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_gloo2_dryrun_$mode.json')); print('gloo2 $mode', round(d['value']), d['n_gpus'], d['config'].get('blocks_allocated'), d['config']['schedule'][:60])" || tail -5 gpurun_out/${T}_gloo2_$mode.err
    done ;;
  *) echo "usage: $0 final|last|close|gloo2 [tag]" ;;
esac
