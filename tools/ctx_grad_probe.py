"""Worst parameter-gradient errors of the RFBNet-512 + Context-Transformer bs-8 step against float64 autograd
(tests/test_gpu_ctx_train.py::_ctx_step_on_device_pattern), for the environment it is started in.  Used to see which
tile choice a parameter over the test's 2e-4 comes from:  CTDET_TRAIN_W4F=0 python tools/ctx_grad_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'context-transformer_amd'))
import test_gpu_ctx_train as T

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
net, errs, refs, fwd = T._ctx_step_on_device_pattern(size, 'transfer', 60, 8, False, seed, cpu32=True)
c32 = fwd.pop('grad cpu32')
print({k: '%.1e' % v for k, v in fwd.items()})
top = sorted(errs.items(), key=lambda kv: -kv[1])[:10]
print('device (torch-CPU float32): ' + ' '.join('%s:%.2e (%.2e)' % (n, e, c32.get(n, -1)) for n, e in top))
top32 = sorted(((n, e) for n, e in c32.items() if n != 'phi.bias'), key=lambda kv: -kv[1])[:6]
print('worst of torch-CPU float32: ' + ' '.join('%s:%.2e' % kv for kv in top32))
