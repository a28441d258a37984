#!/usr/bin/env python3
"""Measure every tile config for every conv launch of the given networks on the MI355X and
merge the winners into context-transformer_amd/ctdet/conv_tune_gfx950.json (committed).

    python tools/tune_convs.py [--out gpurun_out/conv_tune_gfx950.json] [--cases 300:32:20:1 ...]
case = size:batch:classes:phase[:setting]
"""
import argparse, json, os, sys, types
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'context-transformer_amd')); sys.path.insert(0, REPO)
os.environ['CTDET_TUNE'] = '2'
os.environ['CTDET_CTX_TILES'] = 'any'       # the table holds the unconstrained choice; policies map it at plan time
# Winograd variants timed: the two fp32-MFMA kernels (default); CTDET_WINO_TILES=2,4,23,24 adds the F(2x2) bf16x3 forms, which
# win per layer alone but not in the two-stream pipeline
from ctdet import engine, synth  # noqa: E402
from models.RFB_Net_vgg import build_net  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--out', default=engine.TUNE_TABLE)
ap.add_argument('--w4s', type=float, default=0.0, metavar='MARGIN',
                help='targeted pass instead of a full re-tune: keep the table, time every layer that has the three-kernel '
                     'F(4x4,3x3) form (tile 44) against its current choice and move it when 44 is faster by MARGIN (e.g. 0.03)')
ap.add_argument('--w4f', type=float, default=0.0, metavar='MARGIN',
                help='the same targeted pass for the fused F(4x4,3x3) / bf16x3 kernel (tile 46, csrc/ct_wino4f.hip): every 3x3 / '
                     'stride 1 / dilation 1 layer with 16-channel chunks, with its pooling fusion as the runtime runs it')
ap.add_argument('--cases', nargs='*', default=['300:32:20:1', '300:32:60:2:transfer', '300:4:20:1', '300:1:20:1',
                                                 '300:2:20:1', '512:32:20:1', '512:1:20:1', '300:2:60:2:transfer',
                                                 '300:2:15:2:incre', '512:1:60:2:transfer', '300:8:20:1', '300:16:20:1',
                                                 '512:16:20:1', '512:8:20:1', '512:4:20:1'])
a = ap.parse_args()
table = {}
if os.path.exists(engine.TUNE_TABLE):
    table.update(json.load(open(engine.TUNE_TABLE)))
for case in a.cases:
    f = case.split(':')
    size, batch, C, phase = int(f[0]), int(f[1]), int(f[2]), int(f[3])
    setting = f[4] if len(f) > 4 else 'transfer'
    net = build_net(types.SimpleNamespace(method='ours', phase=phase, setting=setting), size, C)
    net.load_state_dict(synth.fill_state_dict(net.state_dict()))
    net = net.eval().cuda(); net.device = 'cuda'
    if a.w4s or a.w4f:
        os.environ['CTDET_TUNE'] = '0'
        rt = net.runtime(batch)
        be = rt.backend
        for tile, margin, ok_key, group in ((44, a.w4s, 'wino4s_ok', engine.WINO4S_TILES), (46, a.w4f, 'wino4f_ok', engine.WINO4F_TILES)):
            if not margin:
                continue
            moved = []
            for st in rt.conv_steps():
                if not st.rt.get(ok_key):
                    continue
                key = st.tune_key(batch)
                cur_tile, cur_x3 = st.rt.get('wino'), st.rt.get('x3')
                if cur_tile in group:
                    continue
                t_cur = min(be._time_conv(st), be._time_conv(st))
                be.enable_wino(st, tile=tile)
                t_new = min(be._time_conv(st), be._time_conv(st))
                if t_new < (1.0 - margin) * t_cur:
                    if not cur_tile and isinstance(table.get(key), str) and table[key] not in engine.WINO_NAME.values():
                        # non-Winograd layer (dilated): what to run where tile 44 is not allowed (the first case that moves a shape
                        # records it; a later case sees 'wino4s' in the table and leaves the record alone)
                        table[key + '|alt'] = table[key]
                    table[key] = engine.WINO_NAME[tile]
                    moved.append('%s %.0f->%.0f us' % (st.name, t_cur * 1e3, t_new * 1e3))
                else:                               # back to what the table says
                    if cur_tile:
                        be.enable_wino(st, tile=cur_tile)
                    else:
                        be.enable_wino(st, False)
                        if cur_x3 is not None:
                            be.enable_x3(st, cur_x3)
            print('%s: moved to %s: %s' % (case, engine.WINO_NAME[tile], '; '.join(moved) or 'none'), flush=True)
        del rt, net
        torch.cuda.empty_cache()
        continue
    rt = net.runtime(batch)
    # second tuning pass on warmed-up state: keeps the better of two measurements
    t = rt.tuned_configs()
    table.update(t)
    print('%s: %d conv shapes tuned' % (case, len(t)), flush=True)
    del rt, net
    torch.cuda.empty_cache()
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
json.dump(table, open(a.out, 'w'), indent=0, sort_keys=True)
print('wrote %d entries to %s' % (len(table), a.out))
