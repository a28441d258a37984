// Weight pre-transform U = G g G^T for the two Winograd kernels (ct_wino.hip F(2x2,3x3), ct_wino4.hip F(4x4,3x3)), in
// the register order their MFMA A fragments are loaded in.  One argument record for both so that the training
// engine's recorded, batched re-pack (ct_pack_run) replays a mixed list with one launch.  Internal header.
#pragma once
#include "ct_common.h"
#include "ct_wino4_points.h"

namespace ctdet {

struct WinoPackArgs {
    const float* w[6];
    int mbeg[7];
    int nparts, cin, cout, chunks, kblocks;
    int dgrad;               // 1: weights of the data-gradient convolution (channels swapped, taps flipped)
    int cin_fwd;
    int tile;                // 2: F(2x2,3x3) layout, 4: F(4x4,3x3) layout, 23: F(2x2,3x3) split into bf16x3 pieces,
                             // 44: F(4x4,3x3) bf16x3 pieces as the GEMM operand of ct_wino4s.hip,
                             // 46: F(4x4,3x3) bf16x3 pieces in the per-wave unit order of ct_wino4f.hip
    float* U;
};

constexpr int kWinoCC = 8;                          // input channels per chunk (both kernels)
constexpr int kWinoKB = 64;                         // output channels per workgroup (both kernels)
constexpr int kWino2ChunkFloats = 16 * 4 * 64 * 2;  // F(2x2): [wave 8][piece 4][lane 64][4]
constexpr int kWino4ChunkFloats = 8 * 9 * 64 * 4;   // F(4x4): [wave 8][point 9][lane 64][4]
constexpr int kWinoX3CC = 16;                       // ct_wino_x3.hip: input channels per chunk (one bf16 MFMA k-group)
constexpr int kWinoX3ChunkBytes = 8 * 2 * 2 * 3 * 64 * 16;   // [wave 8][point 2][cout half 2][piece 3][lane 64][8 bf16]

constexpr int kWino4sBM = 128;                      // ct_wino4s.hip: output channels per GEMM workgroup (four 32-row fragments)
constexpr int kWino4sFragBytes = 1024;              // one MFMA operand fragment: [k half 2][row 32][8 bf16]
constexpr int kWino4sChunkBytes = 4 * 3 * kWino4sFragBytes;   // [sub 4][piece 3] of one (128-row block, 16-channel chunk)

// ct_wino4f.hip (fused F(4x4,3x3) on bf16x3): per (cout block of 64, 16-channel chunk) eight wave regions of nine 3 KB
// "units" = (transform point, cout half), each [piece 3][lane 64][8 bf16]
constexpr int kWino4fUnitBytes = 3 * 1024;
constexpr int kWino4fWaveBytes = 9 * kWino4fUnitBytes;
constexpr int kWino4fChunkBytes = 8 * kWino4fWaveBytes;       // = 36 points x 64 couts x 16 channels x 3 pieces x 2 bytes

// forward: g = w[co][ci];  data gradient: this conv's (co, ci) = forward (ci, co), taps rotated 180 degrees
__device__ __forceinline__ const float* wino_taps(const WinoPackArgs& p, int co, int ci)
{
    const int fco = p.dgrad ? ci : co, fci = p.dgrad ? co : ci;
    int part = 0;
    while (part + 1 < p.nparts && fco >= p.mbeg[part + 1]) ++part;
    return p.w[part] + ((size_t)(fco - p.mbeg[part]) * p.cin_fwd + fci) * 9;
}

__device__ __forceinline__ float wino_ggt(const WinoPackArgs& p, const float* g, const float (&Ga)[3], const float (&Gb)[3])
{
    float val = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            val += Ga[i] * (p.dgrad ? g[(2 - i) * 3 + (2 - j)] : g[i * 3 + j]) * Gb[j];
    return val;
}

// U[kb][chunk][wave][piece][lane][4]: piece = (x, s-pair), element = (s parity, cout half); wave w / lane (l31, hh)
// gets exactly the A fragments it feeds to its MFMAs, as four float4
__device__ __forceinline__ void wino2_pack_body(const WinoPackArgs& p, long first, long stride)
{
    const long total = (long)p.kblocks * p.chunks * kWino2ChunkFloats;
    for (long idx = first; idx < total; idx += stride) {
        const int e = (int)(idx & 3), ln = (int)((idx >> 2) & 63), pc = (int)((idx >> 8) & 3), wv = (int)((idx >> 10) & 7);
        const int hh = ln >> 5;
        const int k = (ln & 31) + 32 * (e & 1);
        const int s = 2 * (pc & 1) + (e >> 1);
        const int xi = 2 * wv + (pc >> 1);
        const long rest = idx >> 13;
        const int chunk = (int)(rest % p.chunks);
        const int kb = (int)(rest / p.chunks);
        const int co = kb * kWinoKB + k, ci = chunk * kWinoCC + 2 * s + hh;
        float val = 0.f;
        if (co < p.cout) {
            // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
            float Ga[3], Gb[3];
            auto grow = [](int r, float (&o)[3]) {
                if (r == 0) { o[0] = 1.f; o[1] = 0.f; o[2] = 0.f; }
                else if (r == 1) { o[0] = .5f; o[1] = .5f; o[2] = .5f; }
                else if (r == 2) { o[0] = .5f; o[1] = -.5f; o[2] = .5f; }
                else { o[0] = 0.f; o[1] = 0.f; o[2] = 1.f; }
            };
            grow(xi >> 2, Ga);
            grow(xi & 3, Gb);
            val = wino_ggt(p, wino_taps(p, co, ci), Ga, Gb);
        }
        p.U[idx] = val;
    }
}

// U[kb][chunk][wave 8][point 9][lane 64][4]: wave = (point group, cout half), element = channel pair s.
// One thread = one (cout, cin) filter: G g G^T once (6x3 then 6x6), 36 stores; the 64 threads of a wave cover
// (16 lanes x 4 channel pairs) = 256 contiguous bytes of every point's plane.
__device__ __forceinline__ void wino4_pack_body(const WinoPackArgs& p, long first, long stride)
{
    const long total = (long)p.kblocks * p.chunks * 512;          // [kb][chunk][cout half 2][lane 64][s 4]
    for (long idx = first; idx < total; idx += stride) {
        const int s = (int)(idx & 3), ln = (int)((idx >> 2) & 63), half = (int)((idx >> 8) & 1);
        const long rest = idx >> 9;
        const int chunk = (int)(rest % p.chunks);
        const int kb = (int)(rest / p.chunks);
        const int hh = ln >> 5;
        const int co = kb * kWinoKB + half * 32 + (ln & 31), ci = chunk * kWinoCC + 2 * s + hh;
        float g[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) g[i][j] = 0.f;
        if (co < p.cout) {
            const float* w = wino_taps(p, co, ci);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) g[i][j] = p.dgrad ? w[(2 - i) * 3 + (2 - j)] : w[i * 3 + j];
        }
        // U = G g G^T in double, rounded once (ct_wino4_points.h)
        double t[6][3];                                           // G g
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double o[6];
            w4::gmul6(g[0][j], g[1][j], g[2][j], o);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][j] = o[i];
        }
        float* base = p.U + ((size_t)kb * p.chunks + chunk) * kWino4ChunkFloats + ln * 4 + s;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double o[6];                                          // (G g) G^T, row i
            w4::gmul6(t[i][0], t[i][1], t[i][2], o);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int xi = i * 6 + j;
                base[((2 * (xi / 9) + half) * 9 + xi % 9) * 256] = (float)o[j];
            }
        }
    }
}

// ct_wino_x3.hip: U[kb][chunk 16 ch][wave 8][x 2][cout half 2][piece 3][lane 64][8 bf16] -- wave w owns the points 2w + x;
// lane (l31, hh) holds, for cout = kb*64 + half*32 + l31, the channels chunk*16 + 8*hh .. +7 of one bf16 piece: the A
// operand of v_mfma_f32_32x32x16_bf16.  One thread = one (point, cout, 8 channels): G g G^T in fp32 (the F(2x2) G has
// entries 0, 1, +-1/2: the same values ct_wino.hip multiplies with), then the exact three-piece split.
__device__ __forceinline__ void winox3_pack_body(const WinoPackArgs& p, long first, long stride)
{
    const long total = (long)p.kblocks * p.chunks * (8 * 2 * 2 * 64);
    unsigned short* const out = reinterpret_cast<unsigned short*>(p.U);
    for (long idx = first; idx < total; idx += stride) {
        const int ln = (int)(idx & 63), half = (int)((idx >> 6) & 1), x = (int)((idx >> 7) & 1), wv = (int)((idx >> 8) & 7);
        const long rest = idx >> 11;
        const int chunk = (int)(rest % p.chunks);
        const int kb = (int)(rest / p.chunks);
        const int hh = ln >> 5;
        const int co = kb * kWinoKB + half * 32 + (ln & 31);
        const int xi = 2 * wv + x;
        float Ga[3], Gb[3];
        auto grow = [](int r, float (&o)[3]) {          // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
            if (r == 0) { o[0] = 1.f; o[1] = 0.f; o[2] = 0.f; }
            else if (r == 1) { o[0] = .5f; o[1] = .5f; o[2] = .5f; }
            else if (r == 2) { o[0] = .5f; o[1] = -.5f; o[2] = .5f; }
            else { o[0] = 0.f; o[1] = 0.f; o[2] = 1.f; }
        };
        grow(xi >> 2, Ga);
        grow(xi & 3, Gb);
        unsigned short* base = out + ((((((size_t)kb * p.chunks + chunk) * 8 + wv) * 2 + x) * 2 + half) * 3 * 64 + ln) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = chunk * kWinoX3CC + 8 * hh + e;
            float val = 0.f;
            if (co < p.cout) val = wino_ggt(p, wino_taps(p, co, ci), Ga, Gb);
            const unsigned hb = __builtin_bit_cast(unsigned, val) & 0xFFFF0000u;
            const float r1 = val - __builtin_bit_cast(float, hb);
            const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
            const unsigned lb = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, mb));
            base[e] = (unsigned short)(hb >> 16);
            base[64 * 8 + e] = (unsigned short)(mb >> 16);
            base[2 * 64 * 8 + e] = (unsigned short)(lb >> 16);
        }
    }
}

// ct_wino4s.hip: U[point 36][cout block of 128][chunk 16 ch][sub 4][piece 3][k half 2][row 32][8 bf16] -- per transform
// point a [cout] x [cin] GEMM operand whose (128 couts x 16 channels) blocks are 12 KB of ready-made
// v_mfma_f32_32x32x16_bf16 fragments, copied to LDS by DMA without touching a register.  One thread = one (cout, cin)
// filter: G g G^T in double (as wino4_pack_body), rounded to fp32 once, then the exact three-piece split.
__device__ __forceinline__ void wino4s_pack_body(const WinoPackArgs& p, long first, long stride)
{
    const long total = (long)p.kblocks * kWino4sBM * p.cin;
    const size_t plane = (size_t)p.kblocks * p.chunks * kWino4sChunkBytes / 2;       // bf16 elements per point
    unsigned short* const out = reinterpret_cast<unsigned short*>(p.U);
    for (long idx = first; idx < total; idx += stride) {
        const int ci = (int)(idx % p.cin);
        const int co = (int)(idx / p.cin);
        float g[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) g[i][j] = 0.f;
        if (co < p.cout) {
            const float* w = wino_taps(p, co, ci);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) g[i][j] = p.dgrad ? w[(2 - i) * 3 + (2 - j)] : w[i * 3 + j];
        }
        double t[6][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double o[6];
            w4::gmul6(g[0][j], g[1][j], g[2][j], o);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][j] = o[i];
        }
        const int cb = co / kWino4sBM, sub = (co % kWino4sBM) / 32, chunk = ci / 16, kh = (ci % 16) / 8;
        unsigned short* base = out + ((((size_t)cb * p.chunks + chunk) * 4 + sub) * 3) * (kWino4sFragBytes / 2) +
                               (kh * 32 + co % 32) * 8 + ci % 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double o[6];
            w4::gmul6(t[i][0], t[i][1], t[i][2], o);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float val = (float)o[j];
                const unsigned hb = __builtin_bit_cast(unsigned, val) & 0xFFFF0000u;
                const float r1 = val - __builtin_bit_cast(float, hb);
                const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
                const unsigned lb = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, mb));
                unsigned short* q = base + (size_t)(i * 6 + j) * plane;
                q[0] = (unsigned short)(hb >> 16);
                q[kWino4sFragBytes / 2] = (unsigned short)(mb >> 16);
                q[2 * (kWino4sFragBytes / 2)] = (unsigned short)(lb >> 16);
            }
        }
    }
}

// ct_wino4f.hip: U[cout block of 64][chunk 16 ch][wave 8][unit 9][piece 3][lane 64][8 bf16].  Wave w multiplies the
// transform points 4w .. 4w+3 for both cout halves (units 2 (xi & 3) + half) and point 32 + (w >> 1) for the cout half w & 1
// (unit 8); lane (l31, hh) of a unit holds cout = 64 kb + 32 half + l31, channels 16 chunk + 8 hh .. + 7 of one bf16 piece:
// the A operand of v_mfma_f32_32x32x16_bf16.  One thread = one (cout, cin) filter: G g G^T in double, rounded once, then
// the exact three-piece split (as wino4s_pack_body).
__device__ __forceinline__ void wino4f_pack_body(const WinoPackArgs& p, long first, long stride)
{
    const long total = (long)p.kblocks * kWinoKB * p.cin;
    unsigned short* const out = reinterpret_cast<unsigned short*>(p.U);
    for (long idx = first; idx < total; idx += stride) {
        const int ci = (int)(idx % p.cin);
        const int co = (int)(idx / p.cin);
        float g[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) g[i][j] = 0.f;
        if (co < p.cout) {
            const float* w = wino_taps(p, co, ci);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) g[i][j] = p.dgrad ? w[(2 - i) * 3 + (2 - j)] : w[i * 3 + j];
        }
        double t[6][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double o[6];
            w4::gmul6(g[0][j], g[1][j], g[2][j], o);
#pragma unroll
            for (int i = 0; i < 6; ++i) t[i][j] = o[i];
        }
        const int kb = co / kWinoKB, half = (co % kWinoKB) / 32, chunk = ci / 16;
        const int ln = (co % 32) + 32 * ((ci % 16) / 8);
        unsigned short* base = out + ((size_t)kb * p.chunks + chunk) * (kWino4fChunkBytes / 2) + ln * 8 + ci % 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double o[6];
            w4::gmul6(t[i][0], t[i][1], t[i][2], o);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int xi = i * 6 + j;
                const int wv = xi < 32 ? xi >> 2 : 2 * (xi - 32) + half;
                const int unit = xi < 32 ? 2 * (xi & 3) + half : 8;
                const float val = (float)o[j];
                const unsigned hb = __builtin_bit_cast(unsigned, val) & 0xFFFF0000u;
                const float r1 = val - __builtin_bit_cast(float, hb);
                const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
                const unsigned lb = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, mb));
                unsigned short* q = base + (size_t)wv * (kWino4fWaveBytes / 2) + unit * (kWino4fUnitBytes / 2);
                q[0] = (unsigned short)(hb >> 16);
                q[512] = (unsigned short)(mb >> 16);
                q[1024] = (unsigned short)(lb >> 16);
            }
        }
    }
}

__device__ __forceinline__ void wino_pack_any(const WinoPackArgs& p, long first, long stride)
{
    if (p.tile == 46) wino4f_pack_body(p, first, stride);
    else if (p.tile == 44) wino4s_pack_body(p, first, stride);
    else if (p.tile == 23) winox3_pack_body(p, first, stride);
    else if (p.tile == 4) wino4_pack_body(p, first, stride);
    else wino2_pack_body(p, first, stride);
}

// fills and validates a record; launches it, or appends it to the open recording (ct_pack_record_begin)
int pack_wino_any(const float* const* w, const int* cout, int nparts, int cin, int dgrad, int tile, float* upacked,
                  ct_stream_t stream, const char* who);

}  // namespace ctdet
