#!/bin/bash
# Collect the measurement artefacts behind profiles/ on a GPU box (run through gpurun from the repo root; QUICK=1 skips the
# probes of kernels a change did not touch -- their files from the earlier pass of the round stay in gpurun_out/prof_<tag>/):
#   bash tools/collect_profiles.sh [tag]        -> gpurun_out/prof_<tag>/...  and the condensed files in profiles/ layout
# The PMC passes run on their own, with --kernel-trace only (never together with --stats or the sys/runtime trace
# domains).  Workloads: the headline (RFBNet-300 bs 32), 300 + Context-Transformer, RFBNet-512, the bf16 mode at the
# per-GPU shape of BASELINE configs[4] (512, bs 16), the training step.
set -u
TAG=${1:-final}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run_set() {   # name, workload (size,batch,phase,classes), extra bench args...
  local name=$1 wl=$2; shift 2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${name}_stats" -o bench -- \
      python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs "$@" > "$O/${name}_bench.json.log" 2> "$O/${name}_bench.err"
  for C in FETCH_SIZE WRITE_SIZE; do
    CTDET_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$O/${name}_pmc_$C" -o p -- \
        python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline "$@" > /dev/null 2>&1
  done
  python "$R/tools/prof_summary.py" --stats "$(ls $O/${name}_stats/*kernel_stats.csv | head -1)" \
      --fetch "$(ls $O/${name}_pmc_FETCH_SIZE/*counter_collection.csv | head -1)" \
      --write "$(ls $O/${name}_pmc_WRITE_SIZE/*counter_collection.csv | head -1)" \
      --tag "${TAG}_${name}" --workload "$wl" --cmd "python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $*" \
      --out "$O/condensed" > /dev/null
}
run_set rfb300 300,32,1,20
run_set rfb300ctx 300,32,2,60 --phase 2 --classes 60
run_set rfb512 512,32,1,20 --size 512
[ -n "${QUICK:-}" ] || { run_set bf16_512b16 512,16,1,20 --size 512 --batch 16 --dtype bf16; }
cd "$R"
python bench.py --steps 20 --warmup 5 > "$O/bench_full.json.log" 2> "$O/bench_full.err"
CTDET_STREAMS=1 timeout 600 python tools/layer_report.py > "$O/layer_report.txt" 2>&1
CTDET_STREAMS=1 timeout 600 python tools/layer_report.py --size 512 > "$O/layer_report_512.txt" 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train_stats" -o tr -- \
    python "$R/tools/train_bench.py" --batch 32 --steps 10 > "$O/train_bench.log" 2>&1
cd "$R"
rm -f "$O/train_configs.txt"
for cfg in "--size 300 --batch 32" "--size 300 --batch 32 --phase 2 --classes 60" "--size 512 --batch 8 --phase 2 --classes 60"; do
  timeout 300 python tools/train_bench.py $cfg --steps 6 2>&1 | tail -1 >> "$O/train_configs.txt"
done
(timeout 300 python tools/nms_probe.py | head -4; SIZE=512 timeout 300 python tools/nms_probe.py | head -4) > "$O/nms_probe.txt" 2>&1
STAGES=1 TILES=44,47 timeout 300 python tools/wino_one.py base.17 base.19 base.24 head.0 head.1 base.12 > "$O/wino4s_stage_split.txt" 2>&1
timeout 200 python tools/res_probe.py > "$O/res_probe.txt" 2>&1
[ -n "${QUICK:-}" ] || { timeout 300 python tools/attn_probe.py > "$O/attn_probe.txt" 2>&1; }
[ -n "${QUICK:-}" ] || { timeout 400 python tools/x3_probe.py > "$O/x3_probe.txt" 2>&1; }
[ -n "${QUICK:-}" ] || { timeout 300 python tools/wino_accuracy.py > "$O/wino_accuracy.txt" 2>&1; }
# the shipped Context-Transformer policy: error budget + sweep against the fp32 CPU path at 8 and 128 reference threads (the policy
# comparison itself is tools/ctx_policy2.sh / ctx_policy3.sh: its own GPU call, published as profiles/<tag>_ctx_policy.txt)
[ -n "${QUICK:-}${SKIP_CTX_PARITY:-}" ] || { timeout 1500 python tools/ctx_parity.py --budget --sweep --policies h2 --also-threads 128 > "$O/ctx_parity.txt" 2>&1; }
timeout 300 python bench.py --train --steps 5 --warmup 2 > "$O/bench_train.json.log" 2> "$O/bench_train.err"
[ -n "${QUICK:-}" ] || { timeout 300 python bench.py --gpus 2 --share-devices --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > "$O/bench_2rank_rehearsal.json.log" 2> "$O/bench_2rank_rehearsal.err"; }
[ -n "${QUICK:-}" ] || { timeout 300 python bench.py --train --gpus 2 --share-devices --steps 3 --warmup 1 > "$O/bench_train_2rank_rehearsal.json.log" 2> "$O/bench_train_2rank_rehearsal.err"; }
[ -n "${QUICK:-}" ] || { timeout 300 python tools/ctx_attn_time.py > "$O/ctx_attn_time.txt" 2>&1; }
[ -n "${QUICK:-}" ] || { bash tools/wino_pmc.sh base.19 ${TAG}_w4 > /dev/null 2>&1; cp "$R/gpurun_out/pmc_${TAG}_w4/summary.txt" "$O/wino4_pmc.txt"; }
# the three Winograd kernels on the same layer: SQ counters behind "SIMD time = MFMA cycles + 4 cycles per VALU instruction"
[ -n "${QUICK:-}" ] || { bash tools/wino_pmc.sh base.19 ${TAG}_x3d 23 "wino_f2x2_3x3_x3<" > /dev/null 2>&1; cp "$R/gpurun_out/pmc_${TAG}_x3d/summary.txt" "$O/wino_x3_pmc.txt"; }
CTDET_WINO_TILES=2,4,23,44,46,47,48 TILES=2,4,23,44,46,47,48 timeout 600 python tools/wino_one.py base.2 base.5 base.7 base.10 base.12 base.17 base.19 base.24 head.0 > "$O/wino_variants.txt" 2>&1
# the fused F(4x4,3x3) / bf16x3 kernel (tile 46) on conv1_2 and conv2_2: SQ counters, the LDS-DMA probe behind its patch staging
# the fused F(4x4,3x3) kernel on the f16x2 operand form (tile 48; bf16x3 = 46 beside it) on conv1_2 and conv2_2: SQ counters
[ -n "${QUICK:-}" ] || { bash tools/wino_pmc.sh base.2 ${TAG}_w4f 48 wino_f4x4_3x3_x3 > /dev/null 2>&1; cp "$R/gpurun_out/pmc_${TAG}_w4f/summary.txt" "$O/wino4f_pmc.txt"; }
[ -n "${QUICK:-}" ] || { bash tools/wino_pmc.sh base.7 ${TAG}_w4f7 48 wino_f4x4_3x3_x3 > /dev/null 2>&1; cp "$R/gpurun_out/pmc_${TAG}_w4f7/summary.txt" "$O/wino4f_pmc_conv2_2.txt"; }
[ -n "${QUICK:-}" ] || { bash tools/wino_pmc.sh base.2 ${TAG}_w4fx 46 wino_f4x4_3x3_x3 > /dev/null 2>&1; cp "$R/gpurun_out/pmc_${TAG}_w4fx/summary.txt" "$O/wino4f_x3_pmc.txt"; }
[ -n "${QUICK:-}" ] || { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$R/tools/ubench/lds_dma12.hip" -o /tmp/lds_dma12 2>/dev/null && /tmp/lds_dma12 > "$O/lds_dma12.txt" 2>&1; }
# the three-kernel F(4x4,3x3) form: layer by layer against the fused kernels with its per-kernel split, SQ counters of its GEMM
# kernel, and the micro-benchmark of what the bf16 matrix pipe sustains on real data
[ -n "${QUICK:-}" ] || { timeout 900 bash tools/wino4s_probe.sh base.19 base.17b head.0 base.24 head.1 base.12 > /dev/null 2>&1; cp "$R/gpurun_out/wino4s_probe.txt" "$O/wino4s_probe.txt"; }
bash tools/wino_pmc.sh base.19 ${TAG}_w4s 47 wino4h_gemm > /dev/null 2>&1; cp "$R/gpurun_out/pmc_${TAG}_w4s/summary.txt" "$O/wino4s_pmc.txt"
[ -n "${QUICK:-}" ] || { bash tools/wino_pmc.sh base.19 ${TAG}_w4sx 44 wino4s_gemm > /dev/null 2>&1; cp "$R/gpurun_out/pmc_${TAG}_w4sx/summary.txt" "$O/wino4s_x3_pmc.txt"; }
[ -n "${QUICK:-}" ] || { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result "$R/tools/ubench/f16x2_probe.hip" -o /tmp/f16x2_probe 2>/dev/null && /tmp/f16x2_probe > "$O/f16x2_probe.txt" 2>&1; }
for h in 1 0; do CTDET_H2=$h timeout 400 python tools/wino_accuracy.py 2>&1 | grep "shipped\|itself"; done > "$O/wino_accuracy_shipped.txt" 2>&1
[ -n "${QUICK:-}" ] || { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$R/tools/ubench/mfma_power.hip" -o /tmp/mfma_power 2>/dev/null && /tmp/mfma_power > "$O/mfma_power.txt" 2>&1; }
[ -n "${QUICK:-}" ] || { bash tools/bf16_pmc.sh ${TAG}_bf16 > /dev/null 2>&1; cp "$R/gpurun_out/pmc_${TAG}_bf16/summary.txt" "$O/bf16_pmc.txt"; }
cd "$R"
# keep what prof_summary.py needs, drop the bulky traces
find "$O" -name '*kernel_trace.csv' -delete
find "$O" -name '*agent_info.csv' -delete
find "$O" -name '*counter_collection.csv' -delete
ls -la "$O" "$O/condensed" | head -60
tail -c 600 "$O/bench_full.json.log"
