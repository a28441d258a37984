import os, sys, torch, ctypes as C
sys.path.insert(0,'context-transformer_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from ctdet import engine, _lib
def run(Cin, Cout, H, B=32):
    be = engine.HipBackend('cuda:0')
    w = torch.nn.Parameter(torch.randn(Cout, Cin, 3, 3, device='cuda') * 0.05)
    st = engine.ConvStep('t', [engine.ConvPart(w, None, None, True)], Cin, 3, 3, 1, 1, 1, 1, 'x', 0, H, H, 'y', 0)
    bufs = {'x': torch.randn(B, Cin, H, H, device='cuda'), 'y': torch.empty(B, Cout, H, H, device='cuda')}
    st.rt['config'] = engine.WINO
    be.prepare_conv(st, bufs, B)
    for _ in range(3): be.run_conv(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): be.run_conv(st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
print(os.environ.get('CTDET_WINO_EXP'), ' 512@38: %.0f us   256@75: %.0f us   64@300: %.0f us' % (run(512,512,38), run(256,256,75), run(64,64,300)))
