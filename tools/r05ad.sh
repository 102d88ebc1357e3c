#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05ad; mkdir -p $O
timeout 600 python tools/tn_occupancy_probe.py 3000 > $O/tn_occupancy.txt 2>&1; grep -v amdgpu $O/tn_occupancy.txt
