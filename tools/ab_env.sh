#!/bin/bash
# A/B of an environment switch in bench.py:  bash tools/ab_env.sh VAR   (alternates VAR=0 / VAR=1 three times)
V=$1
for i in 1 2 3; do for a in 0 1; do
  env $V=$a python bench.py --no-cpu-baseline --no-cross-check 2>/dev/null | A=$a V=$V python -c "
import json, os, sys
d = json.loads(sys.stdin.read())
print(os.environ['V'] + '=' + os.environ['A'], round(d['value']), round(d['ms_per_step'], 3), round(d['stage_ms_per_device_batch']['trunk'], 3), round(d['roofline']['frac'], 4), d['scores']['fid'], d['scores']['is_mean'])"
done; done
