#!/bin/bash
# r03 run 5: the evidence passes on the final kernels -- kernel trace + FETCH/WRITE + SQ + TCC + UTCL1 (+ latency) for 512^3, 1024^3, 2048^3,
# traffic for OFusion, and a kernel trace of the tracking-on loop
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
SE_PROF_LAST=50 bash tools/gpu_profile.sh r03e full > gpurun_out/profile_r03e.log 2>&1
SE_PROF_RES=1024 SE_PROF_LAST=40 bash tools/gpu_profile.sh r03e_1024 full --res 1024 > gpurun_out/profile_r03e_1024.log 2>&1
SE_PROF_W=1280 SE_PROF_H=960 SE_PROF_RES=2048 SE_PROF_LAST=16 bash tools/gpu_profile.sh r03e_2048 full --width 1280 --height 960 --res 2048 > gpurun_out/profile_r03e_2048.log 2>&1
SE_PROF_FIELD=ofusion SE_PROF_MU=0.008 SE_PROF_LAST=40 bash tools/gpu_profile.sh r03e_of traffic --field ofusion --mu 0.008 > gpurun_out/profile_r03e_of.log 2>&1
OUT=gpurun_out/prof_r03e_track; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r03e_track -- python bench.py --steps 20 --warmup 10 --no-events --no-cpu-baseline --sustain 0 --mode-frames 30 > $OUT/bench.json 2> $OUT/trace.err
SE_PROF_LAST=30 python tools/summarize_prof.py $OUT r03e_track > $OUT/summary.md 2> $OUT/summary.err
find gpurun_out/prof_r03e_track -name '*.db' -delete; find gpurun_out/prof_r03e_track -name '*kernel_trace.csv' -size +4M -delete
for t in r03e r03e_1024 r03e_2048 r03e_of r03e_track; do echo "=== $t"; head -60 gpurun_out/prof_$t/summary.md; done
