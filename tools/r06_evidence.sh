#!/bin/bash
# Round-6 evidence in ONE gpurun call: rocprofv3 summaries (tools/run_profiles.sh), the default bench line, the driver's command,
# the feed timeline, the start-up profile, the README recipe as a process.   bash tools/r06_evidence.sh r06z
TAG=${1:-r06z}
REPO=$(pwd); O=$REPO/gpurun_out/$TAG; mkdir -p $O
bash tools/run_profiles.sh $TAG > $O/run_profiles.log 2>&1
T0=$(date +%s.%N); python3 bench.py > $O/bench_default_1gpu.json 2> $O/bench_default.err; T1=$(date +%s.%N); python3 -c "print('python bench.py (all legs): %.1f s wall' % ($T1 - $T0))" > $O/bench_default_wall.txt    # (no bc on the boxes)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check > $O/bench_driver_cmd_steps20.json 2> $O/bench_driver.err
python3 tools/png_feed_probe.py 30000 14,16,12,14 2>&1 | grep "workers\|resident" > $O/png_feed_timeline_30000.txt
python3 tools/png_feed_probe.py 12000 14,12,10,8 2>&1 | grep "workers\|resident" > $O/png_feed_timeline_12000.txt
TISE_PNG_UNFILTER=host python3 tools/png_feed_probe.py 12000 14 2>&1 | grep "workers\|resident" > $O/png_feed_timeline_12000_host_unfilter.txt
TISE_PNG_WORKER=python python3 tools/png_feed_probe.py 12000 14 2>&1 | grep "workers\|resident" > $O/png_feed_timeline_12000_python_workers.txt
python3 tools/startup_probe.py > $O/startup_probe.txt 2>&1
python3 tools/ring_setup_probe.py > $O/ring_setup_probe.txt 2>&1
python3 tools/cli_probe.py 30000 quick > $O/cli_probe_quick.txt 2>&1
bash tools/host_feed_ab.sh > $O/host_feed_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_feed && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_feed -- python3 $REPO/tools/png_feed_probe.py 12000 14 > /dev/null 2>&1
python3 - <<PY > $O/png_unfilter_kernel_stats.txt 2>&1
import csv, glob
f = glob.glob("/tmp/prof_feed/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0]))] if f else []
u = [r for r in rows if "png_unfilter" in r.get("Kernel_Name", "")]
print("png_unfilter_kernel launches:", len(u))
for r in u[-12:]:
    print(int(r["Grid_Size"]) // 64 if r.get("Grid_Size") else "?", "images", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
ls $O
