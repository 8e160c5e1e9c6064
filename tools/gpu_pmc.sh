#!/bin/bash
# HBM traffic counters of the gradient writer: separate --pmc passes, kernel-trace only
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_w $R/gpurun_out/pmc_r
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-step > $R/gpurun_out/pmc_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-step > $R/gpurun_out/pmc_r.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for tag in ("pmc_w", "pmc_r"):
    fs = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True)
    print(tag, fs)
    for f in fs:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
            print("  %-60s %-12s n=%3d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
PY
