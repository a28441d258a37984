#!/bin/bash
# Context-Transformer tile policy: throughput and parity sweep (9 randn cases, CPU reference at 8 threads like the tests) for
# pairs of CTDET_CTX_W4S_MIN_CIN (three-kernel F(4x4,3x3) / bf16x3 from that many input channels up, 0 = never) and
# CTDET_CTX_F4_MAX_CIN (fused F(4x4,3x3) / fp32 kept up to that many input channels); everything else runs F(2x2,3x3) /
# bf16x3 with two accumulators (engine.ctx_tile_set, ctx_w4s_min_cin, ctx_f4_max_cin).  First pair = shipped.
for pair in "128 128" "256 128" "64 0" "0 128" "0 64" "0 0" "0 256"; do
  set -- $pair
  echo "=== CTDET_CTX_W4S_MIN_CIN=$1 CTDET_CTX_F4_MAX_CIN=$2"
  CTDET_CTX_W4S_MIN_CIN=$1 CTDET_CTX_F4_MAX_CIN=$2 python bench.py --phase 2 --classes 60 --steps 30 --warmup 8 --no-other-configs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ctx300 bs32', d['value'], d['ms_per_step'])"
  CTDET_CTX_W4S_MIN_CIN=$1 CTDET_CTX_F4_MAX_CIN=$2 python tools/ctx_parity.py --sweep --kinds randn 2>&1 | grep -v amdgpu | tail -12
done
echo "=== the shipped pair against the CPU reference at 128 threads (CTDET_REF_THREADS=128): same device output, other reference"
CTDET_REF_THREADS=128 python tools/ctx_parity.py --sweep --kinds randn 2>&1 | grep -v amdgpu | tail -12
