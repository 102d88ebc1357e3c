#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
for w in 4 8 4 8; do echo "== TISE_SYTRD_ROWS=$w"; TISE_SYTRD_ROWS=$w python tools/frechet_probe.py 2>&1 | grep -v amdgpu.ids; done > $O/frechet_waves.txt 2>&1
cat $O/frechet_waves.txt
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "frechet or eigvalsh or cholesky" 2>&1 | tail -3
