// libctdet: the fused epilogue of one (cout, 4x4 output tile) of the F(4x4,3x3) kernels -- ct_wino4.hip (fused form)
// and ct_wino4s.hip (three-kernel bf16x3 form) apply the same arithmetic to their output-transformed sums.  Internal header.
#pragma once
#include "ct_common.h"
#include "ct_f16x2.h"

namespace ctdet {
namespace w4 {

typedef int emit_i32x4 __attribute__((ext_vector_type(4)));
constexpr int kEmitInvalidOff = 0x7FFFFFF0;

// y = A^T M A of one tile.  *scale + shift, residual, ReLU / per-channel floor (NaN propagates), the four 2x2 pooling
// windows a 4x4 tile holds, NCHW or head-scatter stores.  Args: any record with the epilogue fields of Wino4Args.
// ymul: a factor folded into the per-channel scale (the f16x2 kernels' 2^-(eU + eV): exact).  track / amax_run: the thread's
// running maximum of |v| over everything this call stores (ct_conv_desc.out_absmax; the caller folds it into the slot once, at
// the end of the kernel) -- a reference and a flag, not a nullable pointer, so that the value stays in a register.
template <class Args>
__device__ __forceinline__ void emit_tile4(const Args& a, const __amdgpu_buffer_rsrc_t rout,
                                           const __amdgpu_buffer_rsrc_t rres, const int n, const int ty, const int tx,
                                           const int co, const float (&y)[4][4], const float ymul, const bool track, float& amax_run)
{
    const int OH = a.H, OW = a.W;                      // pad 1, stride 1: same spatial size
    const int oy = 4 * ty, ox = 4 * tx;
    const bool c1 = ox + 1 < OW, c2 = ox + 2 < OW, c3 = ox + 3 < OW;
    float sc = a.scale[co] * ymul, sh = a.shift[co];
    float lo = a.lo ? a.lo[co] : (a.relu ? 0.f : -INFINITY);
    // The three per-channel values are needed by every row below, and every row sits behind its own `yy >= OH` test: left
    // alone, the compiler waits for these loads at the first use in EACH row block with s_waitcnt vmcnt(0) -- which from the
    // second row on also waits for the previous rows' STORES to drain (loads and stores share the counter).  Making the
    // values opaque here forces the one wait to this point, in front of all stores of the tile.
    asm volatile("" : "+v"(sc), "+v"(sh), "+v"(lo));
    float pl[2][2] = {{-INFINITY, -INFINITY}, {-INFINITY, -INFINITY}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = oy + i;
        if (yy >= OH) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = y[i][j] * sc + sh;
        if (a.res) {
            const unsigned ro = (unsigned)(((((size_t)n * a.res_ctot + a.res_coff + co) * OH + yy) * OW + ox) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = j == 0 || (j == 1 ? c1 : j == 2 ? c2 : c3);
                const float r = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(rres, ok ? ro + 4 * j : (unsigned)kEmitInvalidOff, 0, 0));
                v[j] = v[j] * a.res_scale + r;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] < lo ? lo : v[j];      // NaN propagates
        if (track) {
            ctdet::h2::track_absmax(amax_run, v[0]);
            if (c1) ctdet::h2::track_absmax(amax_run, v[1]);
            if (c2) ctdet::h2::track_absmax(amax_run, v[2]);
            if (c3) ctdet::h2::track_absmax(amax_run, v[3]);
        }
        pl[i >> 1][0] = fmaxf(pl[i >> 1][0], c1 ? fmaxf(v[0], v[1]) : v[0]);
        if (c2) pl[i >> 1][1] = fmaxf(pl[i >> 1][1], c3 ? fmaxf(v[2], v[3]) : v[2]);
        if (!a.write_full) continue;
        if (a.nseg > 0) {          // heads: permute(0,2,3,1) + view + cat of models/RFB_Net_vgg.py:239-248
#pragma unroll
            for (int g = 0; g < 3; ++g)
                if (g < a.nseg && co >= a.seg[g].co_begin && co < a.seg[g].co_end) {
                    float* dst = a.seg[g].ptr + (size_t)n * a.seg[g].img_stride + a.seg[g].base +
                                 (size_t)(yy * OW + ox) * a.seg[g].pix_stride + (co - a.seg[g].co_begin);
                    dst[0] = v[0];
                    if (c1) dst[a.seg[g].pix_stride] = v[1];
                    if (c2) dst[2 * a.seg[g].pix_stride] = v[2];
                    if (c3) dst[3 * a.seg[g].pix_stride] = v[3];
                }
            continue;
        }
        const unsigned oo = (unsigned)(((((size_t)n * a.out_ctot + a.out_coff + co) * OH + yy) * OW + ox) * 4);
        if (c3) {
            emit_i32x4 pk;
            pk.x = __builtin_bit_cast(int, v[0]);
            pk.y = __builtin_bit_cast(int, v[1]);
            pk.z = __builtin_bit_cast(int, v[2]);
            pk.w = __builtin_bit_cast(int, v[3]);
            __builtin_amdgcn_raw_buffer_store_b128(pk, rout, oo, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[0]), rout, oo, 0, 0);
            if (c1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[1]), rout, oo + 4, 0, 0);
            if (c2) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[2]), rout, oo + 8, 0, 0);
        }
    }
    // a 4x4 output tile holds the four windows (2ty + pi, 2tx + pj) of MaxPool2d(2, 2[, ceil_mode])
    // (models/RFB_Net_vgg.py:328-330)
    if (a.pool_out) {
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int pj = 0; pj < 2; ++pj) {
                const int py = 2 * ty + pi, px = 2 * tx + pj;
                if (py < a.pool_oh && px < a.pool_ow && oy + 2 * pi < OH && ox + 2 * pj < OW)
                    a.pool_out[(((size_t)n * a.pool_ctot + a.pool_coff + co) * a.pool_oh + py) * a.pool_ow + px] = pl[pi][pj];
            }
    }
}

template <class Args>
__device__ __forceinline__ void emit_tile4(const Args& a, const __amdgpu_buffer_rsrc_t rout,
                                           const __amdgpu_buffer_rsrc_t rres, const int n, const int ty, const int tx,
                                           const int co, const float (&y)[4][4])
{
    float unused = 0.f;
    emit_tile4(a, rout, rres, n, ty, tx, co, y, 1.f, false, unused);
}

}  // namespace w4
}  // namespace ctdet
