#!/bin/bash
# GPU-box helper: rocprofv3 PMC passes (own runs, --kernel-trace only) for one probe case.
# usage: tools/pmc_run.sh <case> <outdir-under-gpurun_out>
CASE=${1:-c3x3}; OUT=$GRAFT_REPO_ROOT/gpurun_out/${2:-pmc_$CASE}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM"
P3="FETCH_SIZE GRBM_GUI_ACTIVE"
P4="WRITE_SIZE TCC_HIT TCC_MISS"
P5="SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL TA_TA_BUSY GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace -d $OUT/p$i -o r -- python $GRAFT_REPO_ROOT/tools/conv_probe.py $CASE 3 > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for i in range(1,6):
    fs = glob.glob('$OUT/p%d/**/*counter_collection.csv' % i, recursive=True)
    for f in fs:
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
        for k, d in acc.items():
            if 'conv_kernel' in k or 'warp' in k or 'cfr' in k:
                print('pass%d %s' % (i, k), {c: v for c, v in d.items()})
PY
