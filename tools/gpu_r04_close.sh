#!/bin/bash
# Closing run of round 4 on the final tree: the full rocprofv3 passes of the headline workload (kernel trace + FETCH / WRITE / SQ / cache / TLB /
# latency counters, separate runs), every GPU test, the driver's two bench commands.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r04y}
SE_PROF_LAST=50 bash tools/gpu_profile.sh ${T} full > gpurun_out/${T}_prof512.txt 2>&1
cp gpurun_out/prof_$T/summary.md gpurun_out/${T}_rocprofv3_summary.md; cp gpurun_out/prof_$T/pmc_traffic.json gpurun_out/${T}_pmc_traffic.json 2>/dev/null
grep -E "k_raycast|k_integrate|k_alloc_scan" gpurun_out/${T}_rocprofv3_summary.md | head -12 | cut -c1-200
bash tools/gpu_run.sh $T smoke tests bench driver
