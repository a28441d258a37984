#!/bin/bash
# Round 5, second sweep: the shipped Context-Transformer tile policy on the committed table that holds the fused F(4x4,3x3) /
# bf16x3 kernel (tile 46) for conv1_2 .. conv3_3: (CTDET_CTX_F4_TILE, CTDET_CTX_F4_MAX_CIN) variants, three-kernel form off
# (CTDET_CTX_W4S_MIN_CIN=0), every case against the fp32 CPU reference at 8 and at 128 threads.
for v in "4 256" "46 256" "46 128" "4 128"; do
  set -- $v
  echo "=== CTDET_CTX_F4_TILE=$1 CTDET_CTX_F4_MAX_CIN=$2 CTDET_CTX_W4S_MIN_CIN=0"
  env CTDET_CTX_F4_TILE=$1 CTDET_CTX_F4_MAX_CIN=$2 CTDET_CTX_W4S_MIN_CIN=0 python bench.py --phase 2 --classes 60 --steps 30 --warmup 8 --no-other-configs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ctx300 bs32', d['value'], d['ms_per_step'])"
  env CTDET_CTX_F4_TILE=$1 CTDET_CTX_F4_MAX_CIN=$2 CTDET_CTX_W4S_MIN_CIN=0 python tools/ctx_parity.py --sweep --kinds randn --also-threads 128 2>&1 | grep -v amdgpu | tail -14
done
