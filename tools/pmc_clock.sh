#!/bin/bash
# Average shader clock per kernel under the bench's load: GRBM_GUI_ACTIVE cycles / dispatch duration (run on the GPU box).
# usage: tools/pmc_clock.sh <out.csv>
set -e
out=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_clk
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_clk -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_clk/run.log 2>&1
python3 - "$R/gpurun_out/pmc_clk" "$R/$out" <<'PY'
import csv, glob, sys, collections
root, out = sys.argv[1], sys.argv[2]
cc = glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = dur.get(r["Dispatch_Id"])
    if not d: continue
    k = d[1].split("(")[0].replace("void ", "")
    a = acc[k]; a[0] += float(r["Counter_Value"]); a[1] += d[0]; a[2] += 1
w = csv.writer(open(out, "w"))
w.writerow(["kernel", "launches", "total_ms", "gui_active_cycles", "mhz"])
for k, a in sorted(acc.items(), key=lambda x: -x[1][1])[:30]:
    w.writerow([k, a[2], f"{a[1]/1e6:.3f}", f"{a[0]:.0f}", f"{a[0]/a[1]*1e3:.0f}"])
PY
head -12 $R/$out
