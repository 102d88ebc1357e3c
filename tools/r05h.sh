#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$(pwd); O=$REPO/gpurun_out/r05h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in 4 8; do
  rm -rf /tmp/tr_$w
  TISE_SYTRD_ROWS=$w rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$w -- python3 $REPO/tools/frechet_probe.py > $O/probe_$w.log 2>&1
  python3 $REPO/tools/sytrd_trace.py $(find /tmp/tr_$w -name "*kernel_trace.csv" | head -1) > $O/sytrd_per_column_rows$w.txt 2>&1
  echo "== rows per workgroup $w"; cat $O/sytrd_per_column_rows$w.txt
done
