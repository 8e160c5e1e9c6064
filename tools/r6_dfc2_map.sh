#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6dfc2; mkdir -p $O
PIKA_HIPCC_EXTRA="-DPIKA_TUNING_KNOBS" python -m pika_amd.build --force > $O/build_t.log 2>&1 || tail -5 $O/build_t.log
python tools/dfc2_bench.py PIKA_DFC2_MAP rows cols 2>&1 | grep -v amdgpu.ids | tee $O/dfc2_map.txt
PIKA_HIPCC_EXTRA="" python -m pika_amd.build --force > $O/build.log 2>&1 || tail -5 $O/build.log
timeout 900 python -m pytest tests/test_decode_step_gpu.py tests/test_decode_full.py tests/test_fst.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --workload decode --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > $O/dec64.json 2>$O/dec64.err; python -c "
import json; d=json.loads(open('$O/dec64.json').read().strip().splitlines()[-1]); t=d['config']['timing']; print('decode B=64 ms/batch', d['ms_per_step'], 'search_s', t['search_s'], 'us/step', 1e6*t['search_s']/t['steps'])"
timeout 600 python bench.py --workload mbr_step --batch 8 --beam 4 --steps 6 --warmup 3 --no-cpu-baseline > $O/mbr.json 2> $O/mbr.err; python -c "
import json; d=json.loads(open('$O/mbr.json').read().strip().splitlines()[-1]); print('mbr ms/step', d['ms_per_step'], 'search', d['config']['nbest_search_ms'])"
