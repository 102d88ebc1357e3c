#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05p; mkdir -p $O
python -m pytest tests/test_gpu_clip.py tests/test_gpu_retrieval.py tests/test_gpu_kernels.py -x -q -m gpu -k "preprocess or ring_feed or rp_cli or pa_cli or resize" > $O/pytest_sel.txt 2>&1; tail -4 $O/pytest_sel.txt
python tools/rp_cli_probe.py 30000 > $O/rp_cli_probe.txt 2>&1; cat $O/rp_cli_probe.txt | grep -v amdgpu
