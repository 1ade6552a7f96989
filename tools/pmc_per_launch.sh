#!/bin/bash
# Per-DISPATCH HBM read/write traffic of the Winograd kernels in one bench step, joined with the layer tags of the launch table
# (run on the GPU box).  usage: tools/pmc_per_launch.sh <out.csv>
set -e
out=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_pl
python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --prof-dump $R/gpurun_out/pmc_pl/launches.csv > $R/gpurun_out/pmc_pl/bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $R/gpurun_out/pmc_pl/$c -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_pl/$c.log 2>&1
done
python3 - "$R/gpurun_out/pmc_pl" "$R/$out" <<'PY'
import csv, glob, sys
root, out = sys.argv[1], sys.argv[2]
names = {"wino_kernel": "4064", "wino22_kernel": "4022", "wino_wgrad_kernel": "4164"}
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for k in names:
        per[(c, k)] = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("wino_pipe_kernel", "wino_kernel") == k]
launches = list(csv.DictReader(open(f"{root}/launches.csv")))
w = csv.writer(open(out, "w"))
w.writerow(["kernel", "tag", "M", "N", "K", "C", "splits", "ms", "alg_mb", "read_mb", "write_mb", "hbm_over_alg"])
for k, cfg in names.items():
    seq = [r for r in launches if r["cfg"] == cfg]
    if not seq: continue
    # the launch table holds the instrumented pass (2 steps); the PMC run holds warmup + steps + ...: align on the LAST len(seq) dispatches of a whole number of steps
    per_step = len(seq) // 2
    fs, ws = per[("FETCH_SIZE", k)], per[("WRITE_SIZE", k)]
    n = min(len(fs), len(ws))
    fs, ws = fs[n - per_step:n], ws[n - per_step:n]
    for r, f_, w_ in zip(seq[:per_step], fs, ws):
        rd, wr = 2 * f_ * 1024 / 1e6, w_ * 1024 / 1e6
        w.writerow([k, r["tag"], r["M"], r["N"], r["K"], r["C"], r["splits"], r["ms"], r["alg_mb"], f"{rd:.1f}", f"{wr:.1f}",
                    f"{(rd + wr) / float(r['alg_mb']):.2f}"])
PY
