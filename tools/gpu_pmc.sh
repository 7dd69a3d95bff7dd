#!/bin/bash
# Full counter evidence for stand-alone kernels: kernel trace + SQ / cache / TLB / latency / traffic counter passes, each its own
# rocprofv3 run of tools/ray_probe.py (closed loop: every kernel alone on the chip, in the cache state the pipeline leaves).
# usage: tools/gpu_pmc.sh <tag> <cfg> [frames] [probe flags]     -> gpurun_out/prof_<tag>_<cfg>/summary.md       (SE_PMC_PASSES="sq cache tlb lat fetch write" selects)
TAG=$1; CFG=$2; N=${3:-40}; shift 3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}_${CFG}
mkdir -p $OUT
export SE_PROF_LAST=$((N - 10))
PROBE="python tools/ray_probe.py $CFG $N $@"
PASSES=${SE_PMC_PASSES:-"sq cache tlb lat fetch write"}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $PROBE > $OUT/trace_probe.json 2> $OUT/trace.err
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o $TAG -- $PROBE > /dev/null 2> $OUT/pmc_$name.err; }
for p in $PASSES; do
  case $p in
    sq) pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU ;;
    cache) pass cache TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum ;;
    tlb) pass tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum ;;
    lat) pass lat TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE ;;
    fetch) pass fetch FETCH_SIZE ;;
    write) pass write WRITE_SIZE ;;
  esac
done
python tools/summarize_prof.py $OUT ${TAG}_${CFG} > $OUT/summary.md 2> $OUT/summary.err
cat $OUT/trace_probe.json; grep -E "k_raycast|k_integrate|k_alloc_scan" $OUT/summary.md | cut -c1-220; tail -3 $OUT/summary.err
find $OUT -name '*.db' -delete
find $OUT -name '*kernel_trace.csv' -size +4M -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
cp $OUT/summary.md gpurun_out/${TAG}_${CFG}_pmc_summary.md
