#!/bin/bash
# SQ counter passes for one Winograd layer:  bash tools/wino_pmc.sh base.19 tag [tile code] [kernel-name filter]
#   -> gpurun_out/pmc_<tag>/summary.txt     (tile 4 / wino_f4x4_3x3_f32 by default; 23 / f2x2_3x3_x3; 43 / f4x4_3x3_x3)
L=${1:-base.19}; TAG=${2:-w4}; TILE=${3:-4}; FILT=${4:-wino_f4x4_3x3_f32}
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM" \
         "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32" ; do
  i=$((i+1))
  CHECK=0 TILES=$TILE ITERS=6 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -o p -- python $R/tools/wino_one.py $L > $O/p$i.log 2>&1
done
python - > $O/summary.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$O/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if '$FILT' not in k: continue
        agg[k[:60]][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k[:60], r['Counter_Name'])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-30s %16.0f per launch' % (c, v / cnt[(k, c)]))
PY
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
cat $O/summary.txt
