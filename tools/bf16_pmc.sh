#!/bin/bash
# SQ counter passes for one bf16 layer:  bash tools/bf16_pmc.sh tag [cin cout hw k dil batch]
TAG=${1:-bf16}; shift
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -o p -- python $R/tools/bf16_one.py "$@" > $O/p$i.log 2>&1
done
python - > $O/summary.txt <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$O/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_bf16_nhwc' not in k: continue
        agg[k[:70]][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k[:70], r['Counter_Name'])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-30s %16.0f per launch' % (c, v / cnt[(k, c)]))
PY
find $O -name '*counter_collection.csv' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
tail -3 $O/p1.log; cat $O/summary.txt
