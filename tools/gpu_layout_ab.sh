#!/bin/bash
# dense grid vs pooled bricks (default capacity) at one resolution, same box: pipelined, closed loop, stress stream, tracking on
# usage: tools/gpu_layout_ab.sh [res] [tag]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
RES=${1:-1024}; T=${2:-r04v}
for rep in 1 2; do
for d in 1 0; do
  for st in room stress; do
    extra=""; [ $st = stress ] && extra="--stream stress"
    SE_HIP_DENSE=$d python bench.py --res $RES --steps 60 --warmup 10 --no-cpu-baseline --sustain 0 --mode-frames 40 $extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d.get('modes',{})
print('res $RES dense=$d $st rep $rep: pipelined', round(d['value']), 'closed', round(d.get('value_closed_loop',0)), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()}, 'tracking_on', round(m.get('tracking_on',{}).get('fps',0)), 'blocks', d['config'].get('blocks_allocated'))"
  done
done
done 2>&1 | tee gpurun_out/${T}_layout_ab_$RES.log
