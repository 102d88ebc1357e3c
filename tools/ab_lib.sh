#!/bin/bash
# A/B of two builds of libtise_hip.so on the driver's bench command, alternating runs on ONE box (the box-to-box spread is larger
# than most kernel changes): tools/ab_lib.sh OLD.so [ROUNDS] [extra bench flags]
#   the new library is the in-tree one; OLD.so is passed through TISE_LIB_PATH (tise_toolbox_amd/_lib.py)
OLD=$1; ROUNDS=${2:-3}; shift; shift
for i in $(seq 1 $ROUNDS); do
  for which in new old; do
    if [ $which = old ]; then export TISE_LIB_PATH=$OLD; else unset TISE_LIB_PATH; fi
    python3 bench.py --no-cpu-baseline --no-cross-check --no-host-feed --png-images 0 --no-kernel-probe "$@" 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('$which', round(j['value'],1), 'img/s  conv frac', round(r['frac'],4), 'conv ms/batch', round(r['kernels']['conv_split_fast_kernel']['avg_ms'],2), 'fid', j['scores']['fid'])"
  done
done
