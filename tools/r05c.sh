#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
python tools/host_decode_probe.py 12000 > $O/host_decode_probe.txt 2>&1
TISE_PNG_DECODER=pillow python tools/host_decode_probe.py 12000 > $O/host_decode_probe_pillow.txt 2>&1
python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "png_ring" > $O/pytest_sel.txt 2>&1
tail -3 $O/pytest_sel.txt
python tools/cli_probe.py 30000 > $O/cli_host_inclusive.txt 2>&1
cat $O/host_decode_probe.txt | tail -8; cat $O/host_decode_probe_pillow.txt | tail -7; cat $O/cli_host_inclusive.txt
