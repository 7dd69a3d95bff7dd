#!/bin/bash
# Profiling passes on the GPU box (run through gpurun).  Kernel trace and PMC counters are collected
# in separate rocprofv3 runs of the same bench command; summaries land in gpurun_out/prof_<tag>/.
# usage: tools/gpu_profile.sh <tag> [full|traffic|sq] [extra bench args]   (sq: kernel trace + the SQ counter pass only)     (SE_PROF_W/H/RES/FIELD describe the workload for pmc_traffic.json)
TAG=${1:-r02}; shift
WHAT=${1:-full}; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export SE_PROF_LAST=${SE_PROF_LAST:-50}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps $SE_PROF_LAST --warmup 10 --no-events --no-cpu-baseline --no-modes --no-closed-loop --sustain 0 $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err
if [ "$WHAT" != sq ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $BENCH > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $BENCH > /dev/null 2> $OUT/pmc_write.err
fi
if [ "$WHAT" = full ] || [ "$WHAT" = sq ]; then
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -o $TAG -- $BENCH > /dev/null 2> $OUT/pmc_sq.err
fi
if [ "$WHAT" = full ]; then
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pmc_cache -o $TAG -- $BENCH > /dev/null 2> $OUT/pmc_cache.err
# address translation + memory latency (r03): dense bricks are scattered over 1 / 8 / 64 GiB
rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/pmc_tlb -o $TAG -- $BENCH > /dev/null 2> $OUT/pmc_tlb.err
rocprofv3 --kernel-trace --pmc TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lat -o $TAG -- $BENCH > /dev/null 2> $OUT/pmc_lat.err
fi
python tools/summarize_prof.py $OUT $TAG > $OUT/summary.md 2> $OUT/summary.err
cat $OUT/summary.md; tail -3 $OUT/summary.err
# keep only the small files for the merge back
find $OUT -name '*.db' -delete
find $OUT -name '*kernel_trace.csv' -size +4M -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
