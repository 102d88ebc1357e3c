#!/bin/bash
# Round evidence on the GPU box:  bash tools/run_profiles.sh <tag>   (run from the repository root through gpurun)
# 1. rocprofv3 --kernel-trace --stats of bench.py --steps 24 = 6 device batches of 5000 images (round 6; 10 of 3000 in rounds 4-5, 12 of 1000 before;
#    condensed by tools/summarize_prof.py: steady state = the last 4 device batches of the timed loop)
# 2. rocprofv3 --pmc passes (one counter group each, no tracing alongside) over tools/conv_traffic_target.py
TAG=${1:-r02}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $REPO/bench.py --steps 24 --warmup 2 --no-host-feed --no-cpu-baseline --no-cross-check --no-kernel-probe --png-images 0 > $OUT/bench_steps24_under_rocprof.json 2> $OUT/bench_rocprof.err
python3 $REPO/tools/summarize_prof.py $(dirname $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)) $OUT/bench_steps24_kernel_stats.md 4 > /dev/null 2> $OUT/summarize.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$i -- python3 $REPO/tools/conv_traffic_target.py > $OUT/pmc_pass_$i.log 2>&1
  mkdir -p $OUT/pmc/pass_$i && cp $(find /tmp/pmc_$i -name "*counter_collection.csv") $OUT/pmc/pass_$i/ 2>/dev/null
done
python3 $REPO/tools/conv_pmc_summary.py $OUT/pmc $REPO/gpurun_out/conv_traffic_layers.json $TAG > $OUT/pmc_summary.log 2>&1
cp $REPO/profiles/${TAG}_conv_* $REPO/profiles/pmc_summary.json $OUT/ 2>/dev/null
tail -30 $OUT/pmc_summary.log
