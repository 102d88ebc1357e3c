#!/bin/bash
# Alternating A/B of the driver's bench command inside ONE gpurun call (boxes of the pool differ by +-3 %):
#   bash tools/ab_bench.sh <tag> <pairs> "<env of A>" "<env of B>"
# e.g. bash tools/ab_bench.sh r04b 3 "TISE_POOL_PRODUCER=0" "TISE_POOL_PRODUCER=1"
TAG=$1; PAIRS=${2:-3}; A=$3; B=$4
OUT=gpurun_out/$TAG
mkdir -p $OUT
for i in $(seq 1 $PAIRS); do
  for side in A B; do
    if [ $side = A ]; then E="$A"; else E="$B"; fi
    env $E python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --no-kernel-probe > $OUT/ab_${side}_$i.json 2> $OUT/ab_${side}_$i.err
    python - <<PY
import json
d = json.load(open("$OUT/ab_${side}_$i.json"))
print("$side $i [$E]", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms/step", "FID", d.get("scores", {}).get("fid", d.get("fid")))
PY
  done
done 2>&1 | tee $OUT/ab_summary.txt
