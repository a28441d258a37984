#!/bin/bash
# Collect the measurement artefacts behind profiles/ on a GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh [tag]        -> gpurun_out/prof_<tag>/...
# then condense with tools/prof_summary.py (see profiles/README.md).  The PMC passes run on their own, with
# --kernel-trace only (never together with --stats or the sys/runtime trace domains).
set -u
TAG=${1:-final}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o bench -- \
    python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench.json.log" 2> "$O/bench.err"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o f -- \
    python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o w -- \
    python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cd "$R"
CTDET_STREAMS=1 timeout 600 python tools/layer_report.py > "$O/layer_report.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train" -o tr -- \
    python tools/train_bench.py --batch 32 --steps 10 > "$O/train_bench.log" 2>&1
timeout 300 python tools/wgrad_probe.py > "$O/wgrad_probe.txt" 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bf16" -o bench -- \
    python "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 > "$O/bench_bf16.json.log" 2> "$O/bench_bf16.err"
cd "$R"
CTDET_DTYPE=bf16 CTDET_STREAMS=1 timeout 600 python tools/layer_report.py > "$O/layer_report_bf16.txt" 2>&1
timeout 300 python tools/bf16_probe.py > "$O/bf16_probe.txt" 2>&1
# keep what prof_summary.py needs, drop the bulky traces
find "$O" -name '*kernel_trace.csv' -delete
find "$O" -name '*agent_info.csv' -delete
ls -laR "$O" | head -40
tail -1 "$O/bench.json.log"
