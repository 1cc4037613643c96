#!/bin/bash
# PMC passes over one shape of tools/l3_check.py (development tool).  usage: tools/pmc_l3.sh <shape name> <outdir-tag>
# Counters are collected in separate passes with --kernel-trace only (see MI355X guide).
shape=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/$tag/p$i -o p -- python $R/tools/l3_check.py --iters 2 $PMC_ARGS $shape > $R/gpurun_out/$tag.p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("$R/gpurun_out/$tag/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "conv_l3_fwd" not in k and "conv_fwd_x3" not in k and "wgrad_kernel" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        print("   %-28s %.4g (per launch, %d launches)" % (c, agg[k][c] / cnt[(k, c)], cnt[(k, c)]))
PY
