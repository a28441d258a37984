#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE PMC passes) into the
small tracked files under profiles/.

    python tools/prof_summary.py --stats gpurun_out/prof_r1b/bench_kernel_stats.csv \
        --fetch gpurun_out/pmc_fetch/f_counter_collection.csv \
        --write gpurun_out/pmc_write/w_counter_collection.csv --tag r01
"""
import argparse, csv, collections, json, os, re

ap = argparse.ArgumentParser()
ap.add_argument('--stats', required=True)
ap.add_argument('--fetch')
ap.add_argument('--write')
ap.add_argument('--tag', default='r01')
ap.add_argument('--cmd', default='python bench.py --steps 20 --warmup 5 --no-cpu-baseline')
ap.add_argument('--workload', default='300,32,1,20', help='size,batch,phase,classes of the profiled bench run')
ap.add_argument('--out', default=None, help='output directory (default: profiles/)')
a = ap.parse_args()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = a.out or os.path.join(REPO, 'profiles')
os.makedirs(out_dir, exist_ok=True)


def short(n):
    n = n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*', '', n)


rows = list(csv.DictReader(open(a.stats)))
pmc = {}
for kind, path in (('FETCH_SIZE', a.fetch), ('WRITE_SIZE', a.write)):
    if not path:
        continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != kind:
            continue
        per_dispatch[r['Dispatch_Id']] += float(r['Counter_Value'])
        names[r['Dispatch_Id']] = short(r['Kernel_Name'])
    for d, v in per_dispatch.items():
        acc[names[d]][0] += v
        acc[names[d]][1] += 1
    pmc[kind] = {k: v[0] / v[1] for k, v in acc.items()}

lines = ['# rocprofv3 summary %s' % a.tag, '',
         'Command: `rocprofv3 --kernel-trace --stats --output-format csv -- %s` on one MI355X' % a.cmd,
         '(PMC columns: separate passes `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` with `--kernel-trace`, 3 eager steps;',
         'values are the average per launch in KiB as reported; MI355X_MICROARCH.md: FETCH_SIZE under-reports wide',
         'coalesced reads by 2x on gfx950, so the HBM column doubles it before adding WRITE_SIZE.)', '',
         '| kernel | calls | total ms | avg us | % | FETCH KiB/launch | WRITE KiB/launch | HBM MB/launch (2*F+W) |',
         '|---|---|---|---|---|---|---|---|']
traffic = {}
for r in rows:
    n = short(r['Name'])
    f = pmc.get('FETCH_SIZE', {}).get(n)
    w = pmc.get('WRITE_SIZE', {}).get(n)
    hbm = (2 * f + w) * 1024 / 1e6 if f is not None and w is not None else None
    if hbm is not None:
        traffic[n] = {'fetch_kib': round(f, 1), 'write_kib': round(w, 1), 'hbm_bytes': int((2 * f + w) * 1024)}
    lines.append('| `%s` | %s | %.3f | %.1f | %s | %s | %s | %s |' % (
        n, r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, r['Percentage'],
        '%.0f' % f if f is not None else '-', '%.0f' % w if w is not None else '-',
        '%.1f' % hbm if hbm is not None else '-'))
open(os.path.join(out_dir, '%s_kernel_stats.md' % a.tag), 'w').write('\n'.join(lines) + '\n')
traffic['__workload__'] = dict(zip(('size', 'batch', 'phase', 'classes'), map(int, a.workload.split(','))))
if len(traffic) > 1:      # no PMC passes (the training profile): no table
    json.dump(traffic, open(os.path.join(out_dir, '%s_pmc_traffic.json' % a.tag), 'w'), indent=1, sort_keys=True)
open(os.path.join(out_dir, '%s_kernel_stats.csv' % a.tag), 'w').write(open(a.stats).read())
print('\n'.join(lines[:22]))
