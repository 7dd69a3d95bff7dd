#!/bin/bash
# BASELINE.json configs 2..5 on one GPU (no CPU baseline for the big ones): prints one summary line each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { name=$1; shift; echo "== $name: $*"; python bench.py "$@" --detail gpurun_out/cfg_$name.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print(round(d['value'],1),'fps sustained', round((d.get('sustained') or {}).get('fps',0)), {k:round(v['avg_us'],1) for k,v in d.get('kernels',{}).items()}, 'roofline', r.get('kernel'), round(r.get('achieved',0)), 'GB/s', round(r.get('frac',0),3), 'traffic', r.get('traffic'), 'blocks', d['config']['blocks_allocated'], 'modes', {k:round(v['fps']) for k,v in d.get('modes',{}).items()}, 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))"; }
run sdf512 --steps 100 --warmup 10 --cpu-frames 20 --cpu-reps ${SE_CFG_CPU_REPS:-4}
run sdf512_icl --steps 100 --warmup 10 --icl-like --no-cpu-baseline --no-modes
run sdf1024 --res 1024 --steps 60 --warmup 10 --cpu-frames 8 --cpu-reps 2 --sustain 100
run sdf2048 --width 1280 --height 960 --res 2048 --steps 30 --warmup 8 --no-cpu-baseline --sustain 0 --mode-frames 20
run ofusion512 --field ofusion --mu 0.008 --steps 60 --warmup 10 --cpu-frames 10 --cpu-reps 2 --sustain 100
[ -n "$SE_CFG_SKIP_MU01" ] || run ofusion512_mu01 --field ofusion --mu 0.1 --steps 40 --warmup 10 --no-cpu-baseline --sustain 0 --no-modes
