mkdir -p gpurun_out/r04q
for i in 1 2; do
  for db in 1000 1500 2000 3000; do
    python bench.py --gpus 1 --steps 20 --warmup 5 --device-batch $db --no-cpu-baseline --no-cross-check --no-kernel-probe --no-host-feed > gpurun_out/r04q/db_${db}_$i.json 2> gpurun_out/r04q/db_${db}_$i.err
    python -c "
import json; d=json.load(open('gpurun_out/r04q/db_${db}_$i.json')); print('device batch $db run $i:', round(d['value'],1), 'img/s', d['scores']['fid'])"
  done
done
