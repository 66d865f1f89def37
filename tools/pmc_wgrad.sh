#!/bin/bash
# LDS / MFMA utilisation counters of the 3x3 conv kernel on the layer shapes of tools/kbench.py (separate passes)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_wgrad
mkdir -p $OUT
FILTER=${1:-layer}
i=0
for pmc in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/p$i -o run -- python3 $R/tools/kbench.py --only wgrad --filter "$FILTER" --reps 3 > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        if 'conv_wgrad' not in n: continue
        key=(n.split('<')[1].split('>')[0], r['Grid_Size'])
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    print(k)
    for c,vals in sorted(v.items()): print('   %-34s %.4g' % (c, sum(vals)/len(vals)))
PY
