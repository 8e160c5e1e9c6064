#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
python tools/las_host_profile.py > gpurun_out/las_host_profile.txt 2>&1; grep -v "amdgpu.ids" gpurun_out/las_host_profile.txt | head -100 | cut -c1-180
