#!/bin/bash
# Traffic beyond L2 per step of a bench.py workload: separate --pmc WRITE_SIZE / --pmc FETCH_SIZE passes with --kernel-trace
# only (MI355X_MICROARCH.md), eager launches of the same kernels; FETCH doubled (gfx950 tallies 128-B requests at 64 B).
#   bash tools/r6_pmc.sh NAME STEPS_IN_TRACE  <bench.py arguments...>      -> gpurun_out/r6_NAME_pmc_hbm.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
NAME=$1; STEPS=$2; shift 2
rm -rf /tmp/pmc_w /tmp/pmc_r
for c in WRITE_SIZE:pmc_w FETCH_SIZE:pmc_r; do
  (cd /tmp; PIKA_TRAIN_GRAPH=0 PIKA_BENCH_WATCHDOG=500 timeout 600 rocprofv3 --kernel-trace --pmc ${c%%:*} --output-format csv -d /tmp/${c##*:} -- \
    python $R/bench.py "$@" > $R/gpurun_out/${NAME}_${c##*:}.log 2>&1)
done
python - "$NAME" "$STEPS" "$*" <<'PY'
import csv, glob, collections, json, sys
name, steps, cmd = sys.argv[1], float(sys.argv[2]), sys.argv[3]
out = {"steps_in_trace": steps, "command": "bench.py " + cmd,
       "method": "rocprofv3 --kernel-trace --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes, eager launches (PIKA_TRAIN_GRAPH=0 / "
                 "--decode-eager: the same kernels as the graph replays); units KiB; FETCH doubled per MI355X_MICROARCH.md (gfx950 "
                 "tallies 128-B requests at 64 B); every launch of the process divided by steps_in_trace (warm-up and set-up included)"}
tot = {}
for tag, ctr in (("pmc_w", "WRITE_SIZE"), ("pmc_r", "FETCH_SIZE")):
    agg, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob("/tmp/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                agg[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    scale = 1024.0 * (2.0 if ctr == "FETCH_SIZE" else 1.0)
    tot[ctr] = {k: (v * scale / steps, n[k] / steps) for k, v in agg.items()}
    out[ctr + "_bytes_per_step_total"] = sum(v[0] for v in tot[ctr].values())
keys = sorted(set(tot["WRITE_SIZE"]) | set(tot["FETCH_SIZE"]),
              key=lambda k: -(tot["WRITE_SIZE"].get(k, (0, 0))[0] + tot["FETCH_SIZE"].get(k, (0, 0))[0]))
out["kernels"] = [{"kernel": k[:110], "launches_per_step": tot["WRITE_SIZE"].get(k, tot["FETCH_SIZE"].get(k))[1],
                   "write_bytes_per_step": tot["WRITE_SIZE"].get(k, (0, 0))[0],
                   "fetch_bytes_per_step": tot["FETCH_SIZE"].get(k, (0, 0))[0]} for k in keys[:28]]
out["hbm_bytes_per_step"] = out["WRITE_SIZE_bytes_per_step_total"] + out["FETCH_SIZE_bytes_per_step_total"]
json.dump(out, open("gpurun_out/r6_%s_pmc_hbm.json" % name, "w"), indent=1)
print(name, json.dumps({k: v for k, v in out.items() if k not in ("kernels", "method")}))
for r in out["kernels"][:8]:
    print("   %-80s x%.1f  W %.3f GB  R %.3f GB" % (r["kernel"][:80], r["launches_per_step"], r["write_bytes_per_step"] / 1e9, r["fetch_bytes_per_step"] / 1e9))
PY
