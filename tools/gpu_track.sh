#!/bin/bash
# The tracked loop on the GPU box: its parity tests, the host-side A/B (tools/track_probe.py), a library A/B over gpurun_ab/*.so and a kernel
# trace of it.   usage: tools/gpu_track.sh [tag] [all|trace]     (trace: the probe's three variants + the kernel trace only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r04t}
WHAT=${2:-all}
mkdir -p gpurun_out
if [ "$WHAT" = all ]; then
(time python -m pytest tests/test_gpu_tracking.py tests/test_gpu_asbuilt_tolerance.py -m gpu -x -q --durations=5) > gpurun_out/${T}_pytest_tracking.log 2>&1; tail -8 gpurun_out/${T}_pytest_tracking.log
fi
python tools/track_probe.py --frames 100 --out gpurun_out/${T}_track_probe.json 2>&1 | cut -c1-260 | tee gpurun_out/${T}_track_probe.log
if [ "$WHAT" = all ]; then
for lib in gpurun_ab/*.so; do
  [ -f "$lib" ] || continue
  for rep in 1 2; do
    echo "default lib:"; python tools/track_probe.py --trace 100 2>/dev/null | cut -c1-200
    echo "$lib:"; SE_HIP_LIB=$lib python tools/track_probe.py --trace 100 2>/dev/null | cut -c1-200
  done
done 2>&1 | tee gpurun_out/${T}_track_lib_ab.log
fi
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_${T}_track -o trk -- python tools/track_probe.py --trace 40 > gpurun_out/${T}_track_trace_run.log 2>&1
python tools/track_trace_summary.py gpurun_out/prof_${T}_track 20 > gpurun_out/${T}_track_trace_summary.md 2>&1; cat gpurun_out/${T}_track_trace_summary.md
find gpurun_out/prof_${T}_track -name '*.db' -delete
