#!/bin/bash
# r06 session 7: tree after the clean-up (prefetch / extrapolation removed), pooled cell fast path on / off, sibling split, driver bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r06g smoke tests
python tools/lib_ab.py --cfgs sdf1024,sdf512,of512,stress1024 r06b default r06g_sibsplit r06b default 2>&1 | tee gpurun_out/r06g_tree_ab.log | cut -c1-420
python tools/lib_ab.py --cfgs pooled512,pooled1024,pooled2048,pooledstress512 r06g_nocell default r06g_nocell default 2>&1 | tee gpurun_out/r06g_pooled_cell_ab.log | cut -c1-420
python bench.py --detail gpurun_out/r06g_bench_detail.json > gpurun_out/r06g_bench.json 2> gpurun_out/r06g_bench.err; tail -3 gpurun_out/r06g_bench.err
python -c "
import json; d=json.load(open('gpurun_out/r06g_bench_detail.json')); print(json.dumps(d.get('cpp_mirror'))); print(d['value'], d.get('value_closed_loop'), json.dumps(d['roofline'])[:600]); print({k: round(v['fps']) for k,v in d['modes'].items()})"
