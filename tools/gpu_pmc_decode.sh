#!/bin/bash
# HBM-side traffic of ONE beam-search step (separate --pmc passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes;
# eager launches: bench.py --decode-eager, same kernels) -> gpurun_out/r6_decode_step_pmc.json, and the launch-chain anatomy of the
# graphed search -> gpurun_out/r6_decode_step_chain_b64.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/pmc_dec_w $R/gpurun_out/pmc_dec_r /tmp/prof_dec
for c in WRITE_SIZE:pmc_dec_w FETCH_SIZE:pmc_dec_r; do
  PIKA_BENCH_WATCHDOG=500 timeout 600 rocprofv3 --kernel-trace --pmc ${c%%:*} --output-format csv -d $R/gpurun_out/${c##*:} -- \
    python $R/bench.py --workload decode --batch 64 --steps 1 --warmup 0 --no-cpu-baseline --decode-eager > $R/gpurun_out/${c##*:}.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_dec -o dec -- python $R/bench.py --workload decode --batch 64 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof_dec.err
cd $R
python tools/step_chain.py $(find /tmp/prof_dec -name "*.db" | head -1) > gpurun_out/r6_decode_step_chain_b64.txt
cat gpurun_out/r6_decode_step_chain_b64.txt
python - <<'PY'
import csv, glob, collections, json
STEP = ("dgemm_kernel", "dgemm_sk_kernel", "dstep_", "dfc2_logits", "beam_partials", "fst_advance", "lstm_cell")
out = {"method": "rocprofv3 --kernel-trace --pmc WRITE_SIZE / --pmc FETCH_SIZE in separate passes of bench.py --workload decode "
                 "--batch 64 --steps 1 --warmup 0 with --decode-eager (eager launches of the same kernels); units KiB; "
                 "FETCH doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); per search step = totals of "
                 "the step-chain kernels / launches of beam_partials_kernel"}
tot, steps = {}, None
for tag, name in (("pmc_dec_w", "WRITE_SIZE"), ("pmc_dec_r", "FETCH_SIZE")):
    agg, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name or not any(k in r["Kernel_Name"] for k in STEP):
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
            agg[k] += float(r["Counter_Value"])
            n[k] += 1
    st = max((c for k, c in n.items() if "beam_partials" in k), default=1)
    steps = st if steps is None else steps
    scale = 1024.0 * (2.0 if name == "FETCH_SIZE" else 1.0)
    tot[name] = {k: (v * scale / st, n[k] / st) for k, v in agg.items()}
    out[name + "_bytes_per_step"] = sum(v[0] for v in tot[name].values())
out["search_steps_in_trace"] = steps
keys = sorted(set(tot["WRITE_SIZE"]) | set(tot["FETCH_SIZE"]),
              key=lambda k: -(tot["WRITE_SIZE"].get(k, (0, 0))[0] + tot["FETCH_SIZE"].get(k, (0, 0))[0]))
out["kernels"] = [{"kernel": k, "launches_per_step": tot["WRITE_SIZE"].get(k, tot["FETCH_SIZE"].get(k))[1],
                   "write_bytes_per_step": tot["WRITE_SIZE"].get(k, (0, 0))[0],
                   "fetch_bytes_per_step": tot["FETCH_SIZE"].get(k, (0, 0))[0]} for k in keys]
out["hbm_bytes_per_step"] = out["WRITE_SIZE_bytes_per_step"] + out["FETCH_SIZE_bytes_per_step"]
json.dump(out, open("gpurun_out/r6_decode_step_pmc.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
for r in out["kernels"]:
    print("%-62s x%.1f  W %.2f MB  R %.2f MB" % (r["kernel"], r["launches_per_step"], r["write_bytes_per_step"] / 1e6, r["fetch_bytes_per_step"] / 1e6))
PY
rm -rf gpurun_out/pmc_dec_w gpurun_out/pmc_dec_r gpurun_out/pmc_dec_w.log gpurun_out/pmc_dec_r.log
