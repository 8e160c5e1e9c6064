#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do timeout 600 python -m pytest tests/test_decode_full.py -x -q -m gpu -k "fst_fused" 2>&1 | grep -E "^E|passed|failed" | head -12; done
PIKA_HIPCC_EXTRA="-DPIKA_TUNING_KNOBS" python -m pika_amd.build --force > /dev/null 2>&1
echo "--- generic prep kernel"
for i in 1 2; do PIKA_DSTEP_PREP_GENERIC=1 timeout 600 python -m pytest tests/test_decode_full.py -x -q -m gpu -k "fst_fused" 2>&1 | grep -E "^E|passed|failed" | head -6; done
