#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6c17; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_decode_step_gpu.py tests/test_decode_full.py tests/test_decode.py tests/test_fst.py tests/test_mbr.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --workload decode --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > $O/dec64.json 2>$O/dec64.err; python -c "
import json; d=json.loads(open('$O/dec64.json').read().strip().splitlines()[-1]); t=d['config']['timing']; print('decode B=64 ms/batch', d['ms_per_step'], 'search_s', t['search_s'], 'us/step', 1e6*t['search_s']/t['steps'])"
timeout 600 python bench.py --workload mbr_step --batch 8 --beam 4 --steps 6 --warmup 3 --no-cpu-baseline > $O/mbr.json 2> $O/mbr.err; python -c "
import json; d=json.loads(open('$O/mbr.json').read().strip().splitlines()[-1]); print('mbr ms/step', d['ms_per_step'], 'search', d['config']['nbest_search_ms'])"
(cd /tmp; rm -rf /tmp/prof_dec; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_dec -o dec -- python $GRAFT_REPO_ROOT/bench.py --workload decode --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/step_chain.py $(find /tmp/prof_dec -name "*.db" | head -1) | tee $O/step_chain.txt | tail -14
