#!/bin/bash
# r06 session 1: new fused-launch parity tests + whole GPU suite on the aggregated / deferred insertion, scan A/B, PMC evidence at 1024^3 / 2048^3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/gpu_run.sh r06a smoke tests
python tools/lib_ab.py --cfgs of512,sdf1024,stress1024,sdf512,sdf2048 r05 r06a_agg_defer r06a_defer_only r06a_agg_only r06a_nomark@SE_HIP_BEAM=0@SE_HIP_OF_LEAP=0 r05@SE_HIP_BEAM=0@SE_HIP_OF_LEAP=0 2>&1 | tee gpurun_out/r06a_insert_ab.log | cut -c1-420
bash tools/gpu_pmc.sh r06a sdf1024 40 > gpurun_out/r06a_pmc_sdf1024.txt 2>&1; tail -12 gpurun_out/r06a_pmc_sdf1024.txt
bash tools/gpu_pmc.sh r06a pooled1024 40 > gpurun_out/r06a_pmc_pooled1024.txt 2>&1; tail -12 gpurun_out/r06a_pmc_pooled1024.txt
bash tools/gpu_pmc.sh r06a sdf2048 24 > gpurun_out/r06a_pmc_sdf2048.txt 2>&1; tail -12 gpurun_out/r06a_pmc_sdf2048.txt
SE_HIP_DENSE_MAX_GIB=64 bash tools/gpu_pmc.sh r06a_dense sdf2048 24 > gpurun_out/r06a_pmc_dense2048.txt 2>&1; tail -12 gpurun_out/r06a_pmc_dense2048.txt
