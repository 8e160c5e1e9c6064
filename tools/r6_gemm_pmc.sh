#!/bin/bash
# PMC view of gemm_pp on two shapes (8192^3; the joint's dh product): MFMA-busy, wave-parked / issue-stall split, LDS-issue
# stalls, L2 hit rate -- separate --pmc passes with --kernel-trace only.  Beside profiles/r6_gemm_pp_ablation.txt.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cat > /tmp/pp_two.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pika_amd import gemm as G
dev = torch.device("cuda:0")
SHAPES = {"square": (8192, 8192, 8192, False), "joint_dh": (391680, 1024, 5056, True)}
for name, (M, N, K, o16) in ((os.environ["PP_SHAPE"], SHAPES[os.environ["PP_SHAPE"]]),):
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if o16 else torch.float32)
    for _ in range(4):
        G.launch(G.matrix(a)[0], G.matrix(b)[0], out, N, M, N, K)
    torch.cuda.synchronize()
PY
for shape in square joint_dh; do export PP_SHAPE=$shape; echo "== $shape"
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/pmc_g
  (cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_g -- python /tmp/pp_two.py > /dev/null 2>&1)
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_g/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_pp" in r["Kernel_Name"]:
            agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    print("  " + "  ".join("%s=%.4g" % (k, v) for k, v in sorted(m.items())))
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        print("    of wave cycles: parked (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %% (LDS-issue %.1f %%), issuing %.1f %%;  MFMA pipe busy = "
              "MFMA_BUSY / (GUI_ACTIVE / 8 XCDs x 1024 SIMDs) = %.1f %%" % (100 * m["SQ_WAIT_ANY"] / w, 100 * m["SQ_WAIT_INST_ANY"] / w,
              100 * m["SQ_WAIT_INST_LDS"] / w, 100 * m["SQ_ACTIVE_INST_ANY"] / w, 100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)))
    if "TCC_HIT_sum" in m:
        print("    L2 hit rate %.1f %% (%.3g x 128-B requests)" % (100 * m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), m["TCC_HIT_sum"] + m["TCC_MISS_sum"]))
PY
done; done 2>&1 | tee $O/r6_gemm_pmc.txt
