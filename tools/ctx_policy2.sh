#!/bin/bash
# Context-Transformer tile policy, round 5: the candidate (CTDET_CTX_W4S_MIN_CIN, CTDET_CTX_F4_MAX_CIN) pairs of
# tools/ctx_policy.sh, every sweep case judged against the fp32 CPU reference at 8 AND at 128 threads (same device output;
# tools/ctx_parity.py --also-threads), plus throughput of RFBNet-300 + Context-Transformer at bs 32.
#   bash tools/ctx_policy2.sh "0 0" "0 64" ...      (default: the four pairs VERDICT r04 names)
pairs=("$@")
[ ${#pairs[@]} -eq 0 ] && pairs=("0 0" "0 64" "0 256" "128 128")
for pair in "${pairs[@]}"; do
  set -- $pair
  echo "=== CTDET_CTX_W4S_MIN_CIN=$1 CTDET_CTX_F4_MAX_CIN=$2 ${EXTRA_ENV}"
  env $EXTRA_ENV CTDET_CTX_W4S_MIN_CIN=$1 CTDET_CTX_F4_MAX_CIN=$2 python bench.py --phase 2 --classes 60 --steps 30 --warmup 8 --no-other-configs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('ctx300 bs32', d['value'], d['ms_per_step'])"
  env $EXTRA_ENV CTDET_CTX_W4S_MIN_CIN=$1 CTDET_CTX_F4_MAX_CIN=$2 python tools/ctx_parity.py --sweep --kinds randn --also-threads 128 2>&1 | grep -v amdgpu | tail -14
done
