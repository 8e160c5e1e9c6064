#!/bin/bash
# Is the joint's weight-gradient product (dW2 = dY^T h, 5056 x 1024 over 391680 rows) bound by fabric traffic?  FETCH_SIZE and
# the L2 hit rate of gemm_pp_tn on that shape alone, per split-K setting.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6dw; mkdir -p $O; export TMPDIR=/tmp
export DW_BENCH_ONLY=joint_dW2 DW_BENCH_VAR=PIKA_GEMM_TN_SPLIT
for sp in 0 2 3 6; do
  echo "== PIKA_GEMM_TN_SPLIT=$sp"
  python tools/dw_bench.py $sp 2>&1 | grep -v amdgpu.ids
  for ctr in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    rm -rf /tmp/pmc_dw
    (cd /tmp; PIKA_GEMM_TN_SPLIT=$sp PIKA_GEMM_SPLIT_TARGET=$sp timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_dw -- python $GRAFT_REPO_ROOT/tools/dw_bench.py worker > /dev/null 2>&1)
    python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_dw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_pp_tn" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in agg.items():
    print("   %-24s n=%d mean=%.5g" % (c, len(v), sum(v) / len(v)))
PY
  done
done 2>&1 | tee $O/dw_pmc.txt
