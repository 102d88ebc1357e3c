#!/bin/bash
# device-batch sweep of bench.py (same 30 000-image job): images/s, ms per 500 images of trunk time, conv roofline fraction
for b in 500 750 1000 1500 500 1000; do
  python bench.py --no-cpu-baseline --no-cross-check --device-batch $b 2>/dev/null | B=$b python -c "
import json, os, sys
d = json.loads(sys.stdin.read()); b = int(os.environ['B'])
print(b, round(d['value']), round(d['ms_per_step'], 3), round(d['stage_ms_per_device_batch']['trunk'] * 500 / b, 3), round(d['roofline']['frac'], 4), d['scores']['fid'])"
done
