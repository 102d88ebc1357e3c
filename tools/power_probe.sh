#!/bin/bash
# Power draw and shader clock while the bench's image loop runs:  bash tools/power_probe.sh   (through gpurun)
# rocm-smi is polled every ~0.3 s next to `bench.py --steps 240` (120 k images, ~6 s of image loop).
python bench.py --steps 240 --warmup 4 --no-cpu-baseline --no-cross-check > /tmp/bench_power.json 2> /tmp/bench_power.err &
BP=$!
sleep 9         # imports, weights, resident image set
for i in $(seq 1 40); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk|mclk" | tr '\n' ' ' | sed 's/  */ /g'
  echo
  sleep 0.3
  kill -0 $BP 2>/dev/null || break
done
wait $BP
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo " (idle, after the run)"
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
python -c "
import json; d=json.loads(open('/tmp/bench_power.json').read().strip().split('\n')[-1]); print('bench:', round(d['value']), 'images/s', d['ms_per_step'], 'ms/step')"
