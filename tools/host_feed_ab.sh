#!/bin/bash
# host_feed leg of bench.py under the feed switches (ramp schedule, stream priority): which one costs what
for cfg in "" "TISE_FEED_PRIORITY=high" "TISE_FEED_PRIORITY=normal" "TISE_RAMP=0"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-cross-check --png-images 0 --no-kernel-probe 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); h=j['host_feed']; print('[$cfg]', 'value', round(j['value']), 'host_feed', round(h['images_per_s']), 'ratio', round(h['ratio_to_resident'],4), 'loop', round(h['host_loop_seconds'],3))"
done
