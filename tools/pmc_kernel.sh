#!/bin/bash
# usage: tools/pmc_kernel.sh <outdir> <kernel-substring> <conv_bench args...>   (run on the GPU box)
# Collects SQ counters for one kernel in separate rocprofv3 --pmc passes and prints per-counter sums.
out=$1; shift; kern=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$out
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" \
           "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/$out/p$i -o p --output-format csv -- python3 $R/tools/conv_bench.py "$@" > $R/gpurun_out/$out/p$i.log 2>&1 || exit 1
done
python3 - "$R/gpurun_out/$out" "$kern" <<'PY'
import csv, glob, sys, collections
root, kern = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(root + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(tot): print(f"{k:32s} {tot[k]/max(n[k],1):16.0f}  (avg over {n[k]} dispatches)")
PY
