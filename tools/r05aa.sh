#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05aa; mkdir -p $O
cd /tmp
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
grep -c "" $O/list_avail.txt
