#!/bin/bash
# One runner for every gpurun session (replaces the per-session scripts of r01-r03).  Stages run in the order given:
#   tools/gpu_run.sh <tag> stage[:arg[,arg...]] ...
#   smoke                 __graft_entry__.smoke()
#   tests[:expr]          pytest -m gpu (optionally -k expr)
#   bench                 the driver's command (python bench.py) -> gpurun_out/<tag>_bench.json
#   driver                python bench.py --gpus 1 --steps 20 --warmup 5
#   configs               BASELINE configs 2..5 + the stress stream (tools/gpu_configs.sh)
#   ab:cfgs:lib1,lib2..   tools/lib_ab.py over gpurun_ab/<lib>.so ("default" = in-tree), cfgs joined by '+'
#   prof:<name>[:args]    tools/gpu_profile.sh <tag>_<name> full [bench args, '+'-separated]
#   sq:<name>[:args]      kernel trace + SQ counter pass only
#   sh:<file>             any other script of tools/
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for st in "$@"; do
  IFS=: read -r what a1 a2 <<< "$st"
  echo "=== [$TAG] $st"
  case $what in
    smoke) python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log ;;
    tests) (time python -m pytest tests -m gpu -x -q --durations=8 ${a1:+-k "$a1"}) > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -14 gpurun_out/${TAG}_pytest_gpu.log ;;
    bench) (time python bench.py --detail gpurun_out/${TAG}_bench_detail.json) > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 900 gpurun_out/${TAG}_bench.json; tail -4 gpurun_out/${TAG}_bench.err ;;
    driver) python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
            python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_driver.json')); print('driver-style', d['value'], d.get('value_closed_loop'), d['ms_per_step'], d['roofline']['frac'], {k:round(v['frac'],3) for k,v in d.get('roofline_replay',{}).items() if isinstance(v,dict)})" ;;
    configs) bash tools/gpu_configs.sh 2>&1 | tee gpurun_out/${TAG}_configs.log | cut -c1-420 ;;
    ab) python tools/lib_ab.py --cfgs "${a1//+/,}" ${a2//,/ } 2>&1 | tee gpurun_out/${TAG}_ab_${a1//+/_}.log | cut -c1-400 ;;
    prof) bash tools/gpu_profile.sh ${TAG}_$a1 full ${a2//+/ } ;;
    sq) bash tools/gpu_profile.sh ${TAG}_$a1 sq ${a2//+/ } ;;
    sh) bash tools/$a1 ;;
    *) echo "unknown stage $what" ;;
  esac
done
