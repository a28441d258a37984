#!/bin/bash
# Copy the summaries of a finished tools/collect_profiles.sh run from the scratch directory into the tracked profiles/:
#   bash tools/publish_profiles.sh r04
R=${1:?round tag}; S=gpurun_out/prof_$R; D=profiles
cd "$(dirname "$0")/.."
cp $S/condensed/${R}_* $D/
for f in bench_full.json.log rfb300_bench.json.log rfb512_bench.json.log rfb300ctx_bench.json.log bf16_512b16_bench.json.log \
         bench_train.json.log bench_2rank_rehearsal.json.log bench_train_2rank_rehearsal.json.log layer_report.txt layer_report_512.txt \
         ctx_policy.txt wino_variants.txt wino_accuracy.txt train_configs.txt nms_probe.txt attn_probe.txt ctx_attn_time.txt \
         x3_probe.txt bf16_pmc.txt wino4_pmc.txt wino_x3_pmc.txt wino4s_probe.txt wino4s_pmc.txt mfma_power.txt \
         wino4f_pmc.txt wino4f_pmc_conv2_2.txt wino4f_x3_pmc.txt wino4s_x3_pmc.txt f16x2_probe.txt wino_accuracy_shipped.txt lds_dma12.txt wino4s_stage_split.txt res_probe.txt; do
  [ -s $S/$f ] && grep -v "amdgpu.ids" $S/$f > $D/${R}_$f
done
[ -s $S/ctx_parity.txt ] && grep -v "amdgpu.ids" $S/ctx_parity.txt > $D/${R}_ctx_parity.txt
# the training step's kernel table (rocprofv3 --stats of tools/train_bench.py --batch 32 --steps 10)
python tools/prof_summary.py --stats "$(ls -t $S/train_stats/*kernel_stats.csv | head -1)" --tag "${R}_train" --workload 300,32,1,20 \
    --cmd "python tools/train_bench.py --batch 32 --steps 10" --out $D > /dev/null
# ... and of configs[3]'s per-GPU shape, when the collection made it; the suite's tail
if ls $S/train512ctx_stats/*kernel_stats.csv > /dev/null 2>&1; then
  python tools/prof_summary.py --stats "$(ls -t $S/train512ctx_stats/*kernel_stats.csv | head -1)" --tag "${R}_train512ctx" --workload 512,8,2,60 \
      --cmd "python tools/train_bench.py --size 512 --batch 8 --phase 2 --classes 60 --steps 10" --out $D > /dev/null
fi
[ -s $S/pytest_gpu.txt ] && grep -v "amdgpu.ids" $S/pytest_gpu.txt > $D/${R}_pytest_gpu.txt
# a published summary that is empty or a Python traceback is not evidence: refuse it
bad=0
for f in $D/${R}_*; do
  if [ ! -s "$f" ]; then echo "EMPTY: $f" >&2; rm -f "$f"; bad=1; fi
  if grep -q "^Traceback (most recent call last)" "$f" 2>/dev/null; then echo "TRACEBACK: $f" >&2; bad=1; fi
done
ls $D | grep "^${R}_" | wc -l
exit $bad
