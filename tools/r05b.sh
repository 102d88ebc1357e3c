#!/bin/bash
# round 5, second call: PNG ring feed -- parity test and the host-inclusive CLI probe on 30 000 PNGs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
nproc > $O/nproc.txt; free -g >> $O/nproc.txt
python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "png_ring or device_batch or cli_end_to_end" > $O/pytest_sel.txt 2>&1
tail -3 $O/pytest_sel.txt
python tools/cli_probe.py 30000 > $O/cli_host_inclusive.txt 2>&1
cat $O/cli_host_inclusive.txt
