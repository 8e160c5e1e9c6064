#!/bin/bash
# round 3: kernel trace + MFMA-busy counters of the train step (mixed arithmetic, one hipGraph per step)
mkdir -p gpurun_out/k && cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/prof_ts /tmp/pmc_g
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_ts -o ts -- python $R/bench.py --workload train_step --precision mixed --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/k/prof_ts.log 2>&1
cd $R
db=$(find /tmp/prof_ts -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db --top 110 > gpurun_out/k/train_step_mixed_graph_kernel_stats.csv
python tools/rocpd_gaps.py $db --skip-first 3000 > gpurun_out/k/train_step_mixed_graph_gaps.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_g -- python $R/bench.py --workload train_step --precision mixed --steps 2 --warmup 3 --no-cpu-baseline > $R/gpurun_out/k/pmc_g.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
fs = glob.glob("/tmp/pmc_g/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_pp" in n or "attn_" in n or "rnnt_" in n or "gate_" in n:
            key = n.replace("(anonymous namespace)::", "").replace("void ", "")[:44] + " grid=" + r.get("Grid_Size", "?")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    calls = len(next(iter(c.values())))
    gui = m.get("GRBM_GUI_ACTIVE", 0)
    mfma = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    util = 100.0 * mfma / (gui * 256 * 4) if gui else float("nan")
    rows.append((gui * calls, k, calls, gui, util, m))
out = open("gpurun_out/k/train_step_mixed_pmc_mfma.txt", "w")
out.write("# MFMA-busy of the matrix kernels inside the mixed-arithmetic train step (one --pmc pass, kernel-trace only); MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs)\n")
for _, k, calls, gui, util, m in sorted(rows, reverse=True)[:30]:
    w = m.get("SQ_WAVE_CYCLES", 0) or 1
    line = "%-66s calls=%3d GUI_ACTIVE=%.3e MfmaUtil=%5.1f%% wait_any/wave=%.2f wait_inst/wave=%.2f active/wave=%.2f" % (
        k, calls, gui, util, m.get("SQ_WAIT_ANY", 0) / w, m.get("SQ_WAIT_INST_ANY", 0) / w, m.get("SQ_ACTIVE_INST_ANY", 0) / w)
    out.write(line + "\n")
out.close()
print(open("gpurun_out/k/train_step_mixed_pmc_mfma.txt").read()[:3000])
PY
head -4 gpurun_out/k/train_step_mixed_graph_gaps.txt; tail -1 gpurun_out/k/train_step_mixed_graph_kernel_stats.csv; tail -c 300 gpurun_out/k/prof_ts.log
