#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/c3
mkdir -p $O

timeout 600 python -m pytest tests/test_decode_step_gpu.py tests/test_decode_full.py tests/test_fst.py tests/test_mbr.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python bench.py --workload decode --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > $O/dec.json 2> $O/dec.err
python - <<PY
import json
try:
    d = json.loads(open("$O/dec.json").read().strip().splitlines()[-1])
    t = d["config"]["timing"]
    print("ms/batch %.1f search_s %.4f steps %s us/step %.1f launches %s" % (d["ms_per_step"], t["search_s"], t["steps"], 1e6 * t["search_s"] / t["steps"], t["launches_per_step"]))
except Exception as e:
    print("failed", e); print(open("$O/dec.err").read()[-2000:])
PY
cd /tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o dec -- python $GRAFT_REPO_ROOT/bench.py --workload decode --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT; python tools/step_chain.py $(find $O/prof -name "*.db" | head -1) | tee $O/step_chain.txt
rm -rf $O/prof
