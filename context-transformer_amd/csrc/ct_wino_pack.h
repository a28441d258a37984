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

// the f16x2 operand form of the same kernels (ct_f16x2.h; variant 3 of ct_conv2d_wino4s_fwd): two binary16 pieces, and a 256-byte
// trailer behind the packed weights with { bit pattern of max |g|, exponent eU } (the weights are stored as U 2^eU)
constexpr int kWino4hChunkBytes = 4 * 2 * kWino4sFragBytes;   // [sub 4][piece 2] of one (128-row block, 16-channel chunk)
constexpr int kWino4hTrailerBytes = 256;
constexpr int kWino4hAmaxSlots = 1024;                        // workspace header: partial maxima of |input|, one per absmax workgroup
constexpr int kWino4hHeaderBytes = 2 * kWino4hAmaxSlots * 4;  // ... followed by the exponent eV the input transform chose

// ct_wino4f.hip on f16x2 (tile 48): units of [piece 2][lane 64][8 f16] = 2 KB, same wave / unit numbering as tile 46, same trailer
constexpr int kWino4fhUnitBytes = 2 * 1024;
constexpr int kWino4fhWaveBytes = 9 * kWino4fhUnitBytes;
constexpr int kWino4fhChunkBytes = 8 * kWino4fhWaveBytes;

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

// The two F(4x4,3x3) / bf16x3 layouts (ct_wino4s.hip: tile 44, ct_wino4f.hip: tile 46).  One thread = one (cout, 8 consecutive
// input channels, transform row i): G g G^T in double (as wino4_pack_body), rounded to fp32 once, the exact three-piece
// split, and per (point (i, j), piece) ONE 16-byte store -- the 8 channels are exactly the 8 bf16 a lane of
// v_mfma_f32_32x32x16_bf16 holds, and threads with consecutive couts write consecutive 16-byte rows of a fragment.  (Until
// round 5 a thread owned one filter and scattered 108 two-byte stores: 1.4 ms per training step, where every weight is
// re-packed.)
//   tile 44: U[point 36][cout block of 128][chunk 16 ch][sub 4][piece 3][k half 2][row 32][8 bf16] -- per transform point a
//            [cout] x [cin] GEMM operand whose (128 couts x 16 channels) blocks are 12 KB of ready-made fragments (LDS-DMA)
//   tile 46: U[cout block of 64][chunk 16 ch][wave 8][unit 9][piece 3][lane 64][8 bf16] -- wave w multiplies the points
//            4w .. 4w+3 for both cout halves (units 2 (xi & 3) + half) and point 32 + (w >> 1) for the cout half w & 1 (unit 8)
__device__ __forceinline__ double wino_pick6(const double (&o)[6], int i)
{
    return i == 0 ? o[0] : i == 1 ? o[1] : i == 2 ? o[2] : i == 3 ? o[3] : i == 4 ? o[4] : o[5];
}

template <int TILE>
__device__ __forceinline__ void wino4x3_pack_body(const WinoPackArgs& p, long first, long stride)
{
    constexpr int BMROWS = TILE == 44 ? kWino4sBM : kWinoKB;
    const int rows = p.kblocks * BMROWS;                 // couts incl. the zero rows of the last block
    const int groups = p.cin / 8;
    const long total = (long)rows * groups * 6;
    unsigned char* const out = reinterpret_cast<unsigned char*>(p.U);
    const size_t plane44 = (size_t)p.kblocks * p.chunks * kWino4sChunkBytes;       // bytes per point (tile 44)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    for (long idx = first; idx < total; idx += stride) {
        const int co = (int)(idx % rows);
        const long rest = idx / rows;
        const int ci8 = (int)(rest % groups);
        const int i = (int)(rest / groups);
        u32x4 v[6][3];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) v[j][pc] = u32x4{0u, 0u, 0u, 0u};
        if (co < p.cout) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float* w = wino_taps(p, co, ci8 * 8 + e);
                double t[3];                                      // row i of G g
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    double o[6];
                    if (p.dgrad) w4::gmul6(w[(2 - 0) * 3 + (2 - c)], w[(2 - 1) * 3 + (2 - c)], w[(2 - 2) * 3 + (2 - c)], o);
                    else w4::gmul6(w[0 * 3 + c], w[1 * 3 + c], w[2 * 3 + c], o);
                    t[c] = wino_pick6(o, i);
                }
                double o[6];                                      // (G g) G^T, row i
                w4::gmul6(t[0], t[1], t[2], o);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float val = (float)o[j];
                    const unsigned hb = __builtin_bit_cast(unsigned, val) & 0xFFFF0000u;
                    const float r1 = val - __builtin_bit_cast(float, hb);
                    const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
                    const unsigned lb = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, mb)) & 0xFFFF0000u;
                    const int sh = 16 * (1 - (e & 1));            // even channel: low half-word
                    v[j][0][e >> 1] |= hb >> sh;
                    v[j][1][e >> 1] |= mb >> sh;
                    v[j][2][e >> 1] |= lb >> sh;
                }
            }
        }
        const int chunk = ci8 >> 1, kh = ci8 & 1;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int xi = i * 6 + j;
            unsigned char* q;
            if (TILE == 44) {
                const int cb = co / kWino4sBM, sub = (co % kWino4sBM) / 32;
                q = out + (size_t)xi * plane44 + ((((size_t)cb * p.chunks + chunk) * 4 + sub) * 3) * kWino4sFragBytes +
                    (kh * 32 + co % 32) * 16;
            } else {
                const int kb = co / kWinoKB, half = (co % kWinoKB) / 32;
                const int wv = xi < 32 ? xi >> 2 : 2 * (xi - 32) + half;
                const int unit = xi < 32 ? 2 * (xi & 3) + half : 8;
                q = out + ((size_t)kb * p.chunks + chunk) * kWino4fChunkBytes + (size_t)wv * kWino4fWaveBytes +
                    unit * kWino4fUnitBytes + (co % 32 + 32 * kh) * 16;
            }
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<u32x4*>(q + pc * 1024) = v[j][pc];
        }
    }
}

__device__ __forceinline__ void wino_pack_any(const WinoPackArgs& p, long first, long stride)
{
    if (p.tile == 46) wino4x3_pack_body<46>(p, first, stride);
    else if (p.tile == 44) wino4x3_pack_body<44>(p, first, stride);
    else if (p.tile == 23) winox3_pack_body(p, first, stride);
    else if (p.tile == 4) wino4_pack_body(p, first, stride);
    else wino2_pack_body(p, first, stride);
}

// the f16x2 layouts (tile 47: GEMM operand of ct_wino4s.hip variant 3; 48: per-wave units of ct_wino4f.hip variant 2): max |g| pass,
// then the packing proper (csrc/ct_wino4s.hip); not recordable
int pack_wino_h2(const float* const* w, const int* cout, int nparts, int cin, int dgrad, int tile, void* upacked, ct_stream_t stream,
                 const char* who);
size_t wino_h2_trailer_offset(int cin, int cout, int tile);

// fills and validates a record; launches it, or appends it to the open recording (ct_pack_record_begin)
int pack_wino_any(const float* const* w, const int* cout, int nparts, int cin, int dgrad, int tile, float* upacked,
                  ct_stream_t stream, const char* who);

}  // namespace ctdet
