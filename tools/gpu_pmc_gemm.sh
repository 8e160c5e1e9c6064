#!/bin/bash
# MFMA-busy counters of the ping-pong GEMM kernels inside the train step (one --pmc pass, kernel-trace only)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_g
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_g -- python $R/bench.py --workload train_step --steps 2 --warmup 1 --batch 32 --no-cpu-baseline --no-fp32-leg > $R/gpurun_out/pmc_g.log 2>&1
cd $R
tail -2 gpurun_out/pmc_g.log | cut -c1-300
python - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/pmc_g/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_pp" in n or "attn_" in n or "log_softmax" in n or "rnnt_" in n:
            key = n.replace("(anonymous namespace)::", "")[:40] + " grid=" + r.get("Grid_Size", "?")
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    calls = len(next(iter(c.values())))
    gui = m.get("GRBM_GUI_ACTIVE", 0)
    mfma = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    # gfx94x MfmaUtil formula: busy cycles summed over SIMDs / (active cycles x 256 CUs x 4 SIMDs)
    util = 100.0 * mfma / (gui * 256 * 4) if gui else float("nan")
    rows.append((gui * calls, k, calls, gui, util, m))
out = open("gpurun_out/pmc_gemm_summary.txt", "w")
for _, k, calls, gui, util, m in sorted(rows, reverse=True)[:24]:
    line = "%-64s calls=%3d GUI_ACTIVE=%.3e MfmaUtil=%5.1f%% wait_any/wave=%.2f wait_inst/wave=%.2f active/wave=%.2f" % (
        k, calls, gui, util, m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
        m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
        m.get("SQ_ACTIVE_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1))
    print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/pmc_g
