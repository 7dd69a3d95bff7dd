#!/bin/bash
# sweep of the integration kernel's grid size (SE_HIP_INTEG_GRID, workgroups of 4 waves) at a given volume size
cd $GRAFT_REPO_ROOT
RES=${1:-1024}; shift
for g in 0 "$@"; do
  if [ "$g" = 0 ]; then unset SE_HIP_INTEG_GRID; else export SE_HIP_INTEG_GRID=$g; fi
  echo -n "grid=$g: "
  python bench.py --res $RES --steps 40 --warmup 8 --no-cpu-baseline $EXTRA | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]), {k:round(v[\"avg_us\"],1) for k,v in d[\"kernels\"].items()})"
done
