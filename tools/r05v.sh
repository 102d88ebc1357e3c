#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05v; mkdir -p $O
for b in 32 64 96 128 256 1000 3000; do
  timeout 300 python tools/split_layer_probe.py $b > $O/layers_b$b.txt 2>&1
  echo "== batch $b"; grep -v amdgpu $O/layers_b$b.txt | grep "conv launches\|149x149\|73x73\|71x35" | cut -c1-100
done
