#!/bin/bash
# The three-kernel F(4x4,3x3) / bf16x3 form against the fused kernels, layer by layer (bs 32), and its per-kernel split:
#   bash tools/wino4s_probe.sh [layer ...]   -> gpurun_out/wino4s_probe.txt
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out; mkdir -p $O
L=${@:-base.19 base.17b head.0 base.24 head.1 base.12}
cd /tmp && export TMPDIR=/tmp
{
  TILES=${TILES:-4,23,44,47} ITERS=${ITERS:-20} python $R/tools/wino_one.py $L
  for l in $L; do
    rm -rf $O/w4s_stats
    CHECK=0 TILES=${PTILE:-44} ITERS=20 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/w4s_stats -o s -- python $R/tools/wino_one.py $l > /dev/null 2>&1
    echo "--- $l (tile ${PTILE:-44}): kernel, calls, average us"
    python - <<PY
import csv, glob
for f in glob.glob('$O/w4s_stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'wino4s' in r['Name']:
            print('   %-40s %5s %9.1f' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  done
  rm -rf $O/w4s_stats
  echo "--- power experiment: the same launches on random and on all-zero activations (identical instruction stream)"
  CHECK=0 TILES=4,44 ITERS=20 python $R/tools/wino_one.py base.19
  ZERO=1 CHECK=0 TILES=4,44 ITERS=20 python $R/tools/wino_one.py base.19
} 2>&1 | tee $O/wino4s_probe.txt
