#!/bin/bash
# r06 session 3: sibling-byte groups for deep levels + fine beam stage on the level-6 grid + NT-store input copy
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r06c smoke tests
python tools/lib_ab.py --cfgs sdf1024,stress1024,sdf2048,pooled1024,sdf512,of512 r06b default default@SE_HIP_BEAM=1 2>&1 | tee gpurun_out/r06c_groups_ab.log | cut -c1-420
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --detail gpurun_out/r06c_bench_detail.json > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err; tail -3 gpurun_out/r06c_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r06c_bench_detail.json')); print(json.dumps(d.get('cpp_mirror'))); print(d['value'], d.get('value_closed_loop'), d['roofline']['frac'])"
