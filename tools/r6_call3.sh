#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6c3
mkdir -p $O
(cd /tmp; rm -rf /tmp/prof_mbr; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_mbr -o mbr -- python $GRAFT_REPO_ROOT/bench.py --workload mbr_step --batch 8 --beam 4 --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_mbr.log 2>&1)
db=$(find /tmp/prof_mbr -name '*_results.db' | head -1)
python tools/mbr_anatomy.py $db | tee $O/mbr_anatomy.txt

