#!/bin/bash
# A/B of variants on the GPU box: prints fps + per-kernel avg us for each environment setting
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --event-stride 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]), {k:round(v[\"avg_us\"],1) for k,v in d[\"kernels\"].items()}, round(d[\"roofline\"][\"frac\"],3))"; }
for v in "$@"; do run $v; done
