#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05x; mkdir -p $O
timeout 600 python tools/conv_ablate2.py > $O/ablate_stores.txt 2>&1; grep -v amdgpu $O/ablate_stores.txt
