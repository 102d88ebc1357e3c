#!/bin/bash
# A/B of the pixel-operand-in-registers form of the default convolution kernel (conv_split_areg_kernel, TISE_CONV_AREG = bit mask
# over the tile widths TN) against the default kernel on the driver's bench command, alternating runs on ONE box:
#   tools/areg_ab.sh [ROUNDS] [MASKS...]      e.g. tools/areg_ab.sh 2 0x3c 0x10 0x28
ROUNDS=${1:-2}; shift
MASKS=${@:-0x3c}
for i in $(seq 1 $ROUNDS); do
  for m in 0 $MASKS; do
    TISE_CONV_AREG=$m python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-cross-check --no-host-feed --png-images 0 --no-kernel-probe 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('areg mask $m', round(j['value'],1), 'img/s  conv frac', round(r['frac'],4), 'conv ms/batch', round(r['kernels']['conv_split_fast_kernel']['avg_ms'],2), 'fid', j['scores']['fid'])"
  done
done
