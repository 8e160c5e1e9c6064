#!/bin/bash
# HBM traffic of the train step (separate --pmc passes, kernel-trace only, as MI355X_MICROARCH.md prescribes):
# per step, whole step and the joint-lattice kernels -> gpurun_out/r4_train_step_pmc_hbm.json
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_ts_w $R/gpurun_out/pmc_ts_r
STEPS=4
for c in WRITE_SIZE:pmc_ts_w FETCH_SIZE:pmc_ts_r; do
  PIKA_TRAIN_GRAPH=0 PIKA_BENCH_WATCHDOG=500 timeout 600 rocprofv3 --kernel-trace --pmc ${c%%:*} --output-format csv -d $R/gpurun_out/${c##*:} -- \
    python $R/bench.py --workload train_step --steps $STEPS --warmup 2 --no-cpu-baseline > $R/gpurun_out/${c##*:}.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections, json
steps = $STEPS + 2
out = {"steps_in_trace": steps, "method": "rocprofv3 --kernel-trace --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes of "
       "bench.py --workload train_step --steps $STEPS --warmup 2 (eager launch sequence: PIKA_TRAIN_GRAPH=0, same kernels); units "
       "KiB; FETCH doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)"}
tot = {}
for tag, name in (("pmc_ts_w", "WRITE_SIZE"), ("pmc_ts_r", "FETCH_SIZE")):
    fs = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True)
    agg = collections.defaultdict(float)
    n = collections.defaultdict(int)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"]
            agg[k] += float(r["Counter_Value"])
            n[k] += 1
    scale = 1024.0 * (2.0 if name == "FETCH_SIZE" else 1.0)
    tot[name] = {k: (v * scale / steps, n[k] / steps) for k, v in agg.items()}
    out[name + "_bytes_per_step_total"] = sum(v[0] for v in tot[name].values())
keys = sorted(set(tot["WRITE_SIZE"]) | set(tot["FETCH_SIZE"]),
              key=lambda k: -(tot["WRITE_SIZE"].get(k, (0, 0))[0] + tot["FETCH_SIZE"].get(k, (0, 0))[0]))
out["kernels"] = [{"kernel": k[:110], "launches_per_step": tot["WRITE_SIZE"].get(k, tot["FETCH_SIZE"].get(k))[1],
                   "write_bytes_per_step": tot["WRITE_SIZE"].get(k, (0, 0))[0],
                   "fetch_bytes_per_step": tot["FETCH_SIZE"].get(k, (0, 0))[0]} for k in keys[:24]]
out["hbm_bytes_per_step"] = out["WRITE_SIZE_bytes_per_step_total"] + out["FETCH_SIZE_bytes_per_step_total"]
json.dump(out, open("gpurun_out/r4_train_step_pmc_hbm.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
for r in out["kernels"][:12]:
    print("%-90s x%.1f  W %.2f GB  R %.2f GB" % (r["kernel"][:90], r["launches_per_step"], r["write_bytes_per_step"] / 1e9, r["fetch_bytes_per_step"] / 1e9))
PY
rm -rf gpurun_out/pmc_ts_w gpurun_out/pmc_ts_r
