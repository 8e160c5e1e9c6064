#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6c4
mkdir -p $O
timeout 900 python -m pytest tests/test_decode_full.py tests/test_mbr.py tests/test_mixed_gpu.py tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 600 python bench.py --workload mbr_step --batch 8 --beam 4 --steps 6 --warmup 3 --no-cpu-baseline > $O/mbr.json 2> $O/mbr.err; python -c "
import json; d=json.loads(open('$O/mbr.json').read().strip().splitlines()[-1]); print('mbr ms/step', d['ms_per_step'], 'search', d['config']['nbest_search_ms'], d['config']['train_half'])"
(cd /tmp; rm -rf /tmp/prof_mbr; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_mbr -o mbr -- python $GRAFT_REPO_ROOT/bench.py --workload mbr_step --batch 8 --beam 4 --steps 3 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_mbr.log 2>&1)
db=$(find /tmp/prof_mbr -name '*_results.db' | head -1)
python tools/mbr_anatomy.py $db > $O/mbr_anatomy.txt; grep -n "^(\|^one\|host time" $O/mbr_anatomy.txt
timeout 300 python bench.py --workload decode --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/dec8.json 2>$O/dec8.err; python -c "
import json; d=json.loads(open('$O/dec8.json').read().strip().splitlines()[-1]); print('decode B=8 ms/batch', d['ms_per_step'], d['config'].get('timing'))"
