#!/bin/bash
# Context-Transformer tile policy: throughput and parity sweep for CTDET_CTX_F4_MAX_CIN = 0 / 64 / 128 / 256 (engine.ctx_f4_max_cin)
for cap in 0 64 128 256; do
  echo "=== CTDET_CTX_F4_MAX_CIN=$cap"
  CTDET_CTX_F4_MAX_CIN=$cap python bench.py --phase 2 --classes 60 --steps 30 --warmup 8 --no-other-configs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ctx300 bs32', d['value'], d['ms_per_step'])"
  python tools/ctx_parity.py --sweep --kinds randn --f4-max-cin $cap 2>&1 | grep -v amdgpu | tail -12
done
