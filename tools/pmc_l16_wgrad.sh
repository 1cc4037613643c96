#!/bin/bash
# as tools/pmc_l16.sh for the weight-gradient kernel (l16_check.py --wgrad)
shape=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/$tag/p$i -o p -- python $R/tools/l16_check.py --wgrad --iters 2 $shape > $R/gpurun_out/$tag.p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("$R/gpurun_out/$tag/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "conv_l16_wgrad" not in k and "conv_wgrad_x3" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        print("   %-28s %.4g (per launch, %d launches)" % (c, agg[k][c] / cnt[(k, c)], cnt[(k, c)]))
PY
