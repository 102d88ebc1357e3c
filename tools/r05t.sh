#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05t; mkdir -p $O
for rep in 1 2; do for v in 0 1; do
  TISE_CONV_SHRB=$v timeout 300 python tools/split_layer_probe.py 3000 > $O/layers3000_shrb${v}_$rep.txt 2>&1
done; done
