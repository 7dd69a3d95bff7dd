#!/bin/bash
# same-box A/B of library variants: usage gpu_ab_libs.sh name1 name2 ...  (gpurun_ab/<name>.so; "default" = the in-tree library)
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no-cpu-baseline --warmup 10 --steps 100 --event-stride 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d[\"value\"]), {k:round(v[\"avg_us\"],1) for k,v in d[\"kernels\"].items()})"; }
for rep in 1 2; do
for name in "$@"; do
  if [ "$name" = default ]; then unset SE_HIP_LIB; else export SE_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_ab/$name.so; fi
  echo "== $name (rep $rep)"
  python tools/ray_ablate.py 2>&1 | head -1
  run
  run --res 1024 --steps 60
done
done
