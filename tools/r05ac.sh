#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05ac; mkdir -p $O
cd tools/probes
hipcc -O3 --offload-arch=gfx950 -DORDER=0 -o /tmp/kp0 kstep_phase_probe.hip && hipcc -O3 --offload-arch=gfx950 -DORDER=1 -o /tmp/kp1 kstep_phase_probe.hip || exit 1
for r in 1 2; do for o in 0 1; do echo "== ORDER=$o run $r"; timeout 120 /tmp/kp$o; done; done > $O/kstep_order.txt 2>&1
cat $O/kstep_order.txt | grep -v "^C \|^D "
