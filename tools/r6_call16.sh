#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_decode_full.py -q -m gpu -s -k "benchmarked_row_layout" 2>&1 | tail -12
