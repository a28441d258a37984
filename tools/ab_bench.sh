#!/bin/bash
# Same-box A/B of bench.py under environment variants:  tools/ab_bench.sh OUTDIR "VAR=a VAR2=b" "VAR=c" ...   (each variant twice, alternating)
out=$1; shift
mkdir -p "$out"
i=0
for rep in 1 2; do
  for v in "$@"; do
    tag=$(echo "$v" | tr ' =/' '___')
    env $v python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > "$out/ab_${tag}_$rep.json" 2> "$out/ab_${tag}_$rep.err"
    python - "$out/ab_${tag}_$rep.json" "$v" <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], 'FAILED', e); sys.exit(0)
oc = d.get('other_configs') or {}
print('%-40s %8.1f img/s %7.3f ms  %s' % (sys.argv[2], d['value'], d['ms_per_step'], ' '.join('%s=%s' % (k.replace('rfb', '').replace('_bs', 'b'), v.get('images_per_s')) for k, v in oc.items())))
r = d.get('roofline') or {}
bk = r.get('by_kernel') or {}
print('    by_kernel: ' + '  '.join('%s %.3f ms (%.3f)' % (k[:28], v['ms_per_step'], v['executed_frac']) for k, v in bk.items()))
stg = r.get('stages') or {}
print('    stages:    ' + '  '.join('%s %.1f us x%.0f' % (k, v['avg_launch_us'], v['launches_per_step']) for k, v in stg.items() if isinstance(v, dict) and 'avg_launch_us' in v))
print('    other:     ' + json.dumps(stg.get('other_kernels_us_per_step')))
PY
  done
done
