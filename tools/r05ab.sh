#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r05ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"; do   # TA_* / TCC_* derived groups were refused by the hardware ("exceeds the capabilities") and rocprofv3 then HANGS: not collected
  i=$((i+1))
  rm -rf /tmp/vpmc_$i
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/vpmc_$i -- python3 $REPO/tools/conv_traffic_target.py > $O/pass_$i.log 2>&1
  mkdir -p $O/pass_$i && cp $(find /tmp/vpmc_$i -name "*counter_collection.csv") $O/pass_$i/ 2>/dev/null
  tail -2 $O/pass_$i.log | cut -c1-200
done
python3 $REPO/tools/vmem_pmc_summary.py $O $O/vmem_counters.md | cut -c1-250
