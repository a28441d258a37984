#!/usr/bin/env python3
"""One bf16 NHWC convolution, N launches (for rocprofv3 --pmc passes): python tools/bf16_one.py [cin cout hw k dil batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import bf16_probe as P
from ctdet import _lib
a = [int(v) for v in sys.argv[1:]] + [512, 512, 38, 3, 1, 32][len(sys.argv) - 1:]
cin, cout, hw, k, dl, B = a
x = torch.randn(B, cin, hw, hw, device=P.DEV)
w = torch.randn(cout, cin, k, k, device=P.DEV) * 0.05
b = torch.zeros(cout, device=P.DEV)
_, ms = P.run(_lib.lib(), x, w, b, 1, dl * (k - 1) // 2, dl, True, iters=int(os.environ.get('ITERS', 6)))
print('%d -> %d @%d^2 %dx%d d%d bs %d: %.1f us  %.1f TFLOP/s' % (cin, cout, hw, k, k, dl, B, ms * 1e3, 2.0 * B * hw * hw * cin * cout * k * k / ms / 1e9))
