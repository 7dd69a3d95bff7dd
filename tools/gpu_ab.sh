#!/bin/bash
# A/B of raycast variants on the GPU box: prints fps + per-kernel avg us for each environment setting
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --event-stride 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]), {k:round(v[\"avg_us\"],1) for k,v in d[\"kernels\"].items()}, round(d[\"roofline\"][\"frac\"],3))"; }
run SE_HIP_DENSE=0 SE_HIP_NO_XCD_SWIZZLE=1
run SE_HIP_DENSE=0
run SE_HIP_DENSE=1 SE_HIP_NO_XCD_SWIZZLE=1
run SE_HIP_DENSE=1
