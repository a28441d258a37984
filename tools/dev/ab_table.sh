for rep in 1 2; do
for tb in tools/dev/r03_conv_tune.json ""; do
  CTDET_TUNE_TABLE=$tb python bench.py --no-other-configs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('table=${tb:-new}', d['value'], d['ms_per_step'])"
done; done
