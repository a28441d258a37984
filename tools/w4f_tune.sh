#!/bin/bash
# Targeted tuner pass for the fused F(4x4,3x3) / bf16x3 kernel (tile 46): every case of the committed table, 3 % margin; the
# merged table lands in gpurun_out/conv_tune_w4f.json (copy it over context-transformer_amd/ctdet/conv_tune_gfx950.json).
python tools/tune_convs.py --w4f 0.03 --out gpurun_out/conv_tune_w4f.json "$@" 2>&1 | grep -v amdgpu.ids
