#!/bin/bash
# r03 A/B 1: interleaved bricks x deep speculation in unobserved space, 512^3 / 1024^3 / 2048^3, + translation counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/lib_ab.py --cfgs sdf512,sdf1024 base deep8 il il_deep8 il_deep4 il_deep16 2>&1 | tee gpurun_out/r03_ab1.log
python tools/lib_ab.py --cfgs sdf2048 base il il_deep8 2>&1 | tee -a gpurun_out/r03_ab1.log
for lib in base il; do
  export SE_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_ab/$lib.so
  OUT=gpurun_out/prof_r03_tlb_$lib
  mkdir -p $OUT
  rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/pmc_tlb -o tlb -- python bench.py --res 1024 --steps 30 --warmup 10 --no-events --no-cpu-baseline --no-modes --sustain 0 > /dev/null 2> $OUT/pmc_tlb.err
  SE_PROF_LAST=30 python tools/summarize_prof.py $OUT tlb_$lib > $OUT/summary.md 2> $OUT/summary.err
  cat $OUT/summary.md
  find $OUT -name '*.db' -delete; find $OUT -name '*counter_collection.csv' -size +8M -delete
done
