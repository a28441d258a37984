// libctdet: backward pass of the Context-Transformer block (models/RFB_Net_vgg.py:253-271; the
// reference gets it from autograd over matmul/softmax/linear).  With
//     Q = theta(X)+X   K = phi(Pl)+Pl   V = g(Pl)+Pl   A = softmax(Q K^T)   D = A V
//     Y = X + D*Wz     N = Y/|Y|        out = scale * N OBJ^T              (X = conf, Pl = pooled conf)
// the gradient of `out` flows back as
//   ctx_out_bwd_kernel   per query row: dN -> dY -> (dX direct, dD = dY*Wz, delta = dD.D), dWz, dOBJ
//   ctx_rows_to_x3_kernel  Q, K, V, dD rows split once into bf16x3 fragments (three exact bfloat16 pieces per fp32 value,
//                        csrc/ct_conv_x3.hip has the arithmetic) in the orders the two kernels below read
//   ctx_attn_bwd_q       flash-style, one wave = 32 queries, loops over key tiles (bf16x3 on v_mfma_f32_32x32x16_bf16):
//                          S^T = K Q^T, dA^T = V dD^T, dS = A*(dA - delta), dQ^T += K^T dS^T
//   ctx_attn_bwd_kv      one wave = 32 keys, loops over query tiles:
//                          S = Q K^T, dA = dD V^T, dV^T += dD^T A, dK^T += Q^T dS
//                        (the [P,M] affinity matrix is recomputed from the saved row log-sum-exp with the forward
//                         kernel's products in the forward kernel's order, never stored; the query range is split
//                         over blockIdx.z, partial sums meet in dK/dV through float atomics)
//   ctx_linear_bwd_kernel  y = Lin(x)+x: dx (+)= dy + dy W, dW += dy^T x, db += sum dy
//   ctx_pool_bwd_kernel    max-pool backward of the context pooling (first maximum per window)
#include "ct_common.h"
#include "ct_attn_common.h"
#include <mutex>

namespace {

constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ------------------------------------------------------------------------------------------------
// per-row part: everything after the aggregation
// ------------------------------------------------------------------------------------------------
struct OutBwdArgs {
    const float* conf;      // [B][P][d]
    const float* D;         // saved [B][P_pad][64]
    const float* dout;      // [B][P][ostride]
    const float* wz;
    const float* obj_w;     // [T][d]
    float* dconf;           // [B][P][d]   (written)
    float* dDs;             // [B][P_pad][64] swizzled
    float* delta;           // [B][P_pad]
    float* dwz;             // [d]    (atomics)
    float* dobj;            // [T][d] (atomics)
    int P, P_pad, d, T, ostride, ooff, chunks;
    float scale;
};

// 256 threads = 64 rows x 4 feature quarters (16 features each); a block walks `chunks` 64-row chunks.
__global__ __launch_bounds__(256) void ctx_out_bwd_kernel(const OutBwdArgs a)
{
    __shared__ float objw[32 * DP];          // [t][i]
    __shared__ float Ns[64 * (DP + 1)];      // normalised rows of the chunk
    __shared__ float dOs[64 * 33];           // dOut rows of the chunk
    __shared__ float wzs[DP];
    __shared__ float red[DP];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int rl = tid >> 2, part = tid & 3, f0 = part * 16;
    for (int e = tid; e < 32 * DP; e += 256) {
        const int t = e / DP, i = e % DP;
        objw[e] = (t < a.T && i < a.d) ? a.obj_w[t * a.d + i] : 0.f;
    }
    if (tid < DP) { wzs[tid] = tid < a.d ? a.wz[tid] : 0.f; red[tid] = 0.f; }
    __syncthreads();

    float dwz_acc[16];
    float dobj_acc[8];                       // outputs e = tid + 256*j of the [32][64] dOBJ tile
#pragma unroll
    for (int j = 0; j < 16; ++j) dwz_acc[j] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dobj_acc[j] = 0.f;

    for (int c = 0; c < a.chunks; ++c) {
        const int row = (blockIdx.x * a.chunks + c) * 64 + rl;
        const bool live = row < a.P;
        float X[16], Dv[16], Y[16];
        float n2 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = f0 + j;
            const bool ok = live && i < a.d;
            X[j] = ok ? a.conf[((size_t)b * a.P + row) * a.d + i] : 0.f;
            Dv[j] = ok ? a.D[((size_t)b * a.P_pad + row) * DP + i] : 0.f;
            Y[j] = X[j] + Dv[j] * wzs[i];
            n2 += Y[j] * Y[j];
        }
        n2 += __shfl_xor(n2, 1);
        n2 += __shfl_xor(n2, 2);
        const float rn = live ? 1.f / sqrtf(n2) : 0.f;
        // stage dOut row + N row
        for (int t = part; t < 32; t += 4)
            dOs[rl * 33 + t] = (live && t < a.T) ? a.dout[((size_t)b * a.P + row) * a.ostride + a.ooff + t] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            Y[j] *= rn;                       // Y now holds N
            Ns[rl * (DP + 1) + f0 + j] = Y[j];
        }
        __syncthreads();
        // dN = scale * dOut . OBJ ;  c = N . dN
        float dN[16];
        float cdot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) dN[j] = 0.f;
        for (int t = 0; t < a.T; ++t) {
            const float g = dOs[rl * 33 + t] * a.scale;
#pragma unroll
            for (int j = 0; j < 16; ++j) dN[j] += g * objw[t * DP + f0 + j];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) cdot += Y[j] * dN[j];
        cdot += __shfl_xor(cdot, 1);
        cdot += __shfl_xor(cdot, 2);
        float dl = 0.f;
        if (row < a.P_pad) {
            float* dsw = a.dDs + ((size_t)b * a.P_pad + row) * DP;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = f0 + j;
                const float dY = rn * (dN[j] - Y[j] * cdot);
                const float dD = dY * wzs[i];
                if (live && i < a.d) a.dconf[((size_t)b * a.P + row) * a.d + i] = dY;
                dsw[(i & 1) * 32 + (i >> 1)] = dD;
                dl += dD * Dv[j];
                dwz_acc[j] += dY * Dv[j];
            }
        }
        dl += __shfl_xor(dl, 1);
        dl += __shfl_xor(dl, 2);
        if (part == 0 && row < a.P_pad) a.delta[(size_t)b * a.P_pad + row] = dl;
        // dOBJ[t][i] += scale * sum_rows dOut[row][t] * N[row][i]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = tid + 256 * j, t = e >> 6, i = e & 63;
            float acc = 0.f;
            for (int r = 0; r < 64; ++r) acc += dOs[r * 33 + t] * Ns[r * (DP + 1) + i];
            dobj_acc[j] += acc;
        }
        __syncthreads();
    }
    // block totals -> global (atomics)
#pragma unroll
    for (int j = 0; j < 16; ++j) atomicAdd(&red[f0 + j], dwz_acc[j]);
    __syncthreads();
    if (tid < a.d) atomic_add_f32(&a.dwz[tid], red[tid]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = tid + 256 * j, t = e >> 6, i = e & 63;
        if (t < a.T && i < a.d) atomic_add_f32(&a.dobj[t * a.d + i], dobj_acc[j] * a.scale);
    }
}

// ------------------------------------------------------------------------------------------------
// query side: dQ
// ------------------------------------------------------------------------------------------------
// Rows of a "swizzled" fp32 operand ([B][rows_pad][2][32]: feature i at (i & 1) * 32 + (i >> 1), what ctx_project_kernel
// and ctx_out_bwd_kernel write) split into bf16x3 fragments in one of the three orders of x3_emit (ct_attn_common.h).
__global__ __launch_bounds__(256) void ctx_rows_to_x3_kernel(const float* __restrict__ src, int rows_pad,
                                                            unsigned short* __restrict__ out, int mode)
{
    const int b = blockIdx.y, tid = threadIdx.x;
    const int o = tid & 63, rg = tid >> 6;
    const float* sb = src + (size_t)b * rows_pad * DP;
    for (int r = rg; r < 64; r += 4) {
        const int row = blockIdx.x * 64 + r;
        if (row >= rows_pad) break;
        x3_emit(sb[(size_t)row * DP + (o & 1) * 32 + (o >> 1)], out, mode, b, row, rows_pad, o);
    }
}

// dQ on bf16x3 (round 3; the arithmetic of ctx_attn_kernel): per 32-key tile and wave 24 + 24 + 24
// v_mfma_f32_32x32x16_bf16 instead of 32 + 32 + 32 fp32 MFMAs at twice the issue time.  S^T is recomputed with exactly the
// forward kernel's products and summation order (hi.hi in one accumulator, the five small products in a second one), so
// exp2(S log2 e - lse) reproduces the forward probabilities against the saved log-sum-exp.
struct BwdQx3Args {
    const unsigned char *Qx, *dDx;      // mode 0 rows of Q and dD
    const unsigned char *Kx, *Vk, *Kv;  // 32-key tiles: K mode 1, V mode 1, K mode 2
    const float *lse, *delta;
    float* dQ;                          // [B][P_pad][64] natural feature order
    int P_pad, M, M_pad;
};

__global__ __launch_bounds__(256, 2) void ctx_attn_bwd_q(const BwdQx3Args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kv_raw[];       // 2 x 36 KB
    unsigned char (*kv)[3 * XT_BYTES] = reinterpret_cast<unsigned char (*)[3 * XT_BYTES]>(kv_raw);   // per buffer: K (S), V (dA), K (dQ) tiles

    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x * QB + wave * QW + l31;

    i32x4 qf[4][3], df[4][3];
    {
        const unsigned char* qp = a.Qx + ((size_t)b * a.P_pad + q) * XQ_BYTES;
        const unsigned char* dp = a.dDx + ((size_t)b * a.P_pad + q) * XQ_BYTES;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                qf[g][p] = *reinterpret_cast<const i32x4*>(qp + (p * 8 + 2 * g + h) * 16);
                df[g][p] = *reinterpret_cast<const i32x4*>(dp + (p * 8 + 2 * g + h) * 16);
            }
    }
    const float lse2 = a.lse[(size_t)b * a.P_pad + q];
    const float dlt = a.delta[(size_t)b * a.P_pad + q];

    const int nt = a.M_pad / KT;
    const unsigned char* Kxb = a.Kx + (size_t)b * nt * XT_BYTES;
    const unsigned char* Vkb = a.Vk + (size_t)b * nt * XT_BYTES;
    const unsigned char* Kvb = a.Kv + (size_t)b * nt * XT_BYTES;

    i32x4 treg[9];                                         // 36 KB per tile / 256 threads
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            treg[i] = *reinterpret_cast<const i32x4*>(Kxb + (size_t)t * XT_BYTES + (tid + 256 * i) * 16);
            treg[3 + i] = *reinterpret_cast<const i32x4*>(Vkb + (size_t)t * XT_BYTES + (tid + 256 * i) * 16);
            treg[6 + i] = *reinterpret_cast<const i32x4*>(Kvb + (size_t)t * XT_BYTES + (tid + 256 * i) * 16);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 9; ++i)
            *reinterpret_cast<i32x4*>(&kv[buf][(i / 3) * XT_BYTES + (tid + 256 * (i % 3)) * 16]) = treg[i];
    };

    f32x16 dq0, dq1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }

    load_tile(0);
    store_tile(0);
    __syncthreads();

    // piece pairs (A piece, B piece) of the six products, smallest first; the last one is hi.hi
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < nt;
        if (more) load_tile(t + 1);

        // ---- S^T = K Q^T (as the forward kernel) and dA^T = V dD^T ----
        f32x16 s, ss, da;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; ss[r] = 0.f; da[r] = 0.f; }
        {
            const unsigned char* kb = &kv[buf][l31 * 16];
            const unsigned char* vb = &kv[buf][XT_BYTES + l31 * 16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                i32x4 kf[3], vf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    kf[p] = *reinterpret_cast<const i32x4*>(kb + ((p * 8 + 2 * g + h) * KT) * 16);
                    vf[p] = *reinterpret_cast<const i32x4*>(vb + ((p * 8 + 2 * g + h) * KT) * 16);
                }
#pragma unroll
                for (int c = 0; c < 5; ++c)
                    ss = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[PA[c]]),
                                                                 __builtin_bit_cast(bf16x8, qf[g][PB[c]]), ss, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[0]),
                                                            __builtin_bit_cast(bf16x8, qf[g][0]), s, 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[PA[c]]),
                                                                 __builtin_bit_cast(bf16x8, df[g][PB[c]]), da, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += ss[r];
        }
        // dS^T = A^T * (dA^T - delta), A from the saved log-sum-exp
        const bool edge = (t + 1) * KT > a.M;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = __builtin_amdgcn_exp2f(s[r] * kLog2e - lse2);
            if (edge && t * KT + acc_row(r, h) >= a.M) p = 0.f;
            s[r] = p * (da[r] - dlt);
        }
        // ---- dQ^T += K^T dS^T ----  B operand: this lane's own dS (registers 8 kg .. 8 kg + 7), split in place
        {
            i32x4 pf[2][3];
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) {
                unsigned ph[8], pm[8], pl[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) split3(s[8 * kg + j], ph[j], pm[j], pl[j]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    pf[kg][0][w] = pack_hi(ph[2 * w], ph[2 * w + 1]);
                    pf[kg][1][w] = pack_hi(pm[2 * w], pm[2 * w + 1]);
                    pf[kg][2][w] = pack_hi(pl[2 * w], pl[2 * w + 1]);
                }
            }
            const unsigned char* rb = &kv[buf][2 * XT_BYTES + l31 * 16];
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) {
                i32x4 rf[3][2];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        rf[p][db] = *reinterpret_cast<const i32x4*>(rb + ((((p * 2 + kg) * 2 + h) * DP) + 32 * db) * 16);
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rf[PA[c]][0]),
                                                                  __builtin_bit_cast(bf16x8, pf[kg][PB[c]]), dq0, 0, 0, 0);
                    dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rf[PA[c]][1]),
                                                                  __builtin_bit_cast(bf16x8, pf[kg][PB[c]]), dq1, 0, 0, 0);
                }
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }
    // accumulator rows 4g .. 4g+3 are features 8g + 4h .. + 3 (dq0) and 32 + those (dq1)
    float4* orow = reinterpret_cast<float4*>(a.dQ + ((size_t)b * a.P_pad + q) * DP);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        orow[2 * g + h] = make_float4(dq0[4 * g], dq0[4 * g + 1], dq0[4 * g + 2], dq0[4 * g + 3]);
        orow[8 + 2 * g + h] = make_float4(dq1[4 * g], dq1[4 * g + 1], dq1[4 * g + 2], dq1[4 * g + 3]);
    }
}

// ------------------------------------------------------------------------------------------------
// key side: dK, dV
// ------------------------------------------------------------------------------------------------
// dK, dV on bf16x3: per 32-query tile and wave 4 x 24 v_mfma_f32_32x32x16_bf16 (S, dA, dV, dK) instead of 4 x 32 fp32
// MFMAs at twice the issue time.  S = Q K^T uses the forward kernel's products in the forward kernel's order (operand
// roles swapped: the query tile is the LDS operand here, the wave's own keys sit in registers), so the probabilities
// recomputed against the saved log-sum-exp are the forward's.  A and dS stay in the accumulator registers and go to the
// two gradient contractions as the register operand (split in place); Q and dD tiles are staged in both fragment orders.
struct BwdKVx3Args {
    const unsigned char *Kx0, *Vx0;         // mode 0 rows of K and V (this wave's keys, register operands)
    const unsigned char *Qx1, *dDx1;        // 32-query tiles, mode 1 (contraction over features: S, dA)
    const unsigned char *Qx2, *dDx2;        // 32-query tiles, mode 2 (contraction over queries: dK, dV)
    const float *lse, *delta;
    float *dK, *dV;                         // [B][M_pad][64] natural feature order, accumulated with atomics
    int P_pad, M, M_pad, split;
};
// One kernel for both gradients needs 351 registers and 96 KB of LDS (one wave per SIMD: 4.4 ms at 512 bs 8); as two
// launches -- WANT_V: S and dV (query tiles Q1, dD2), WANT_K: S, dA and dK (Q1, dD1, Q2) -- each fits two waves per SIMD and two
// workgroups per CU at the price of recomputing S (120 instead of 96 MFMAs per tile).
template <bool WANT_K>
struct KvCfg {
    static constexpr int NTILE = WANT_K ? 3 : 2;                         // staged tiles per query tile
    static constexpr int BUF_BYTES = NTILE * XT_BYTES + 2 * KT * 4;      // + lse, delta of the tile's queries
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
};

template <bool WANT_K>
__global__ __launch_bounds__(256, 2) void ctx_attn_bwd_kv(const BwdKVx3Args a)
{
    constexpr int NTILE = KvCfg<WANT_K>::NTILE, BUF_BYTES = KvCfg<WANT_K>::BUF_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int key = blockIdx.x * QB + wave * QW + l31;           // < M_pad (M_pad % 128 == 0)

    i32x4 kf[4][3], vf[WANT_K ? 4 : 1][3];
    {
        const unsigned char* kp = a.Kx0 + ((size_t)b * a.M_pad + key) * XQ_BYTES;
        const unsigned char* vp = a.Vx0 + ((size_t)b * a.M_pad + key) * XQ_BYTES;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                kf[g][p] = *reinterpret_cast<const i32x4*>(kp + (p * 8 + 2 * g + h) * 16);
                if (WANT_K) vf[g][p] = *reinterpret_cast<const i32x4*>(vp + (p * 8 + 2 * g + h) * 16);
            }
    }
    const bool key_live = key < a.M;

    const int nt_all = a.P_pad / KT;
    const size_t boff = (size_t)b * nt_all * XT_BYTES;
    // staged tiles: slot 0 = Q mode 1 (S); WANT_K: slot 1 = dD mode 1 (dA), slot 2 = Q mode 2 (dK); else slot 1 = dD mode 2 (dV)
    const unsigned char* src[3] = {a.Qx1 + boff, (WANT_K ? a.dDx1 : a.dDx2) + boff, a.Qx2 + boff};
    const float* lseb = a.lse + (size_t)b * a.P_pad;
    const float* delb = a.delta + (size_t)b * a.P_pad;
    const int t_begin = (int)((long)nt_all * blockIdx.z / a.split);
    const int t_end = (int)((long)nt_all * (blockIdx.z + 1) / a.split);

    i32x4 treg[3 * NTILE];
    float pl = 0.f;
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < 3 * NTILE; ++i)
            treg[i] = *reinterpret_cast<const i32x4*>(src[i / 3] + (size_t)t * XT_BYTES + (tid + 256 * (i % 3)) * 16);
        if (tid < 32) pl = lseb[t * KT + tid];
        else if (tid < 64) pl = delb[t * KT + tid - 32];
    };
    auto store_tile = [&](int buf) {
        unsigned char* base = lds + buf * BUF_BYTES;
#pragma unroll
        for (int i = 0; i < 3 * NTILE; ++i)
            *reinterpret_cast<i32x4*>(base + (i / 3) * XT_BYTES + (tid + 256 * (i % 3)) * 16) = treg[i];
        if (tid < 64) reinterpret_cast<float*>(base + NTILE * XT_BYTES)[tid] = pl;
    };

    f32x16 g0, g1;                                         // dK or dV: features acc_row (g0) and 32 + acc_row (g1)
#pragma unroll
    for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }

    if (t_begin < t_end) {
        load_tile(t_begin);
        store_tile(0);
    }
    __syncthreads();

    // piece pairs (LDS-operand piece, register-operand piece) of the six products, smallest first; the last one is hi.hi.
    // For S the forward kernel pairs (K piece PA, Q piece PB): here the query tile is the LDS operand, so (Q PB, K PA).
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        const bool more = t + 1 < t_end;
        if (more) load_tile(t + 1);
        const unsigned char* base = lds + buf * BUF_BYTES;
        const float* ls = reinterpret_cast<const float*>(base + NTILE * XT_BYTES);

        // ---- S = Q K^T (and dA = dD V^T): rows = the tile's queries, column = this lane's key ----
        f32x16 s, ss, da;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; ss[r] = 0.f; da[r] = 0.f; }
        {
            const unsigned char* qb = base + l31 * 16;
            const unsigned char* db = base + XT_BYTES + l31 * 16;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                i32x4 qa[3], dd[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    qa[p] = *reinterpret_cast<const i32x4*>(qb + ((p * 8 + 2 * g + h) * KT) * 16);
                    if (WANT_K) dd[p] = *reinterpret_cast<const i32x4*>(db + ((p * 8 + 2 * g + h) * KT) * 16);
                }
#pragma unroll
                for (int c = 0; c < 5; ++c)
                    ss = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[PB[c]]),
                                                                 __builtin_bit_cast(bf16x8, kf[g][PA[c]]), ss, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qa[0]),
                                                            __builtin_bit_cast(bf16x8, kf[g][0]), s, 0, 0, 0);
                if (WANT_K) {
#pragma unroll
                    for (int c = 0; c < 6; ++c)
                        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, dd[PA[c]]),
                                                                     __builtin_bit_cast(bf16x8, vf[g][PB[c]]), da, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += ss[r];
        }
        // A[q][key] (WANT_K: dS[q][key]) for this lane's key; rows are the tile's queries acc_row(r, h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = acc_row(r, h);
            float p = __builtin_amdgcn_exp2f(s[r] * kLog2e - ls[qq]);
            if (!key_live) p = 0.f;
            s[r] = WANT_K ? p * (da[r] - ls[32 + qq]) : p;
        }
        // ---- dV^T += dD^T A  /  dK^T += Q^T dS: register operand = this lane's A / dS (registers 8 kg .. 8 kg + 7) ----
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            i32x4 pp[3];
            {
                unsigned ph[8], pm[8], pl8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) split3(s[8 * kg + j], ph[j], pm[j], pl8[j]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    pp[0][w] = pack_hi(ph[2 * w], ph[2 * w + 1]);
                    pp[1][w] = pack_hi(pm[2 * w], pm[2 * w + 1]);
                    pp[2][w] = pack_hi(pl8[2 * w], pl8[2 * w + 1]);
                }
            }
            const unsigned char* rr = base + (WANT_K ? 2 : 1) * XT_BYTES + l31 * 16;      // Q mode 2 / dD mode 2
            i32x4 rf[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
                    rf[p][hb] = *reinterpret_cast<const i32x4*>(rr + ((((p * 2 + kg) * 2 + h) * DP) + 32 * hb) * 16);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rf[PA[c]][0]),
                                                             __builtin_bit_cast(bf16x8, pp[PB[c]]), g0, 0, 0, 0);
                g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rf[PA[c]][1]),
                                                             __builtin_bit_cast(bf16x8, pp[PB[c]]), g1, 0, 0, 0);
            }
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }
    // accumulator row r = feature acc_row(r, h) (g0) and 32 + that (g1)
    float* orow = (WANT_K ? a.dK : a.dV) + ((size_t)b * a.M_pad + key) * DP;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = acc_row(r, h);
        atomic_add_f32(&orow[i], g0[r]);
        atomic_add_f32(&orow[32 + i], g1[r]);
    }
}

// ------------------------------------------------------------------------------------------------
// y = Linear(x) + x backward.  dy rows have stride `dy_stride` (64 for the padded gradient rows, the
// output stride for the fc_base half); x is [B][rows][d].
// ------------------------------------------------------------------------------------------------
struct LinBwdArgs {
    const float* dy;
    const float* x;
    const float* W;          // [d][d] (out, in)
    float* dx;               // [B][rows][d]
    float* dW;               // [d][d] atomics
    float* db;               // [d]    atomics
    long long dy_batch_stride;
    int rows, d, dy_stride, accumulate, chunks;
};

__global__ __launch_bounds__(256) void ctx_linear_bwd_kernel(const LinBwdArgs a)
{
    __shared__ float Ws[DP * (DP + 1)];      // W[o][i]
    __shared__ float dYs[64 * (DP + 1)];
    __shared__ float Xs[64 * (DP + 1)];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int rl = tid >> 2, part = tid & 3, f0 = part * 16;
    for (int e = tid; e < DP * DP; e += 256) {
        const int o = e / DP, i = e % DP;
        Ws[o * (DP + 1) + i] = (o < a.d && i < a.d) ? a.W[o * a.d + i] : 0.f;
    }
    float dw_acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) dw_acc[j] = 0.f;
    float db_acc = 0.f;
    const float* dyb = a.dy + (size_t)b * a.dy_batch_stride;

    for (int c = 0; c < a.chunks; ++c) {
        const int r0 = (blockIdx.x * a.chunks + c) * 64;
        if (r0 >= a.rows) break;
        __syncthreads();
        for (int e = tid; e < 64 * DP; e += 256) {
            const int r = e / DP, i = e % DP;
            const int row = r0 + r;
            const bool ok = row < a.rows && i < a.d;
            dYs[r * (DP + 1) + i] = ok ? dyb[(size_t)row * a.dy_stride + i] : 0.f;
            Xs[r * (DP + 1) + i] = ok ? a.x[((size_t)b * a.rows + row) * a.d + i] : 0.f;
        }
        __syncthreads();
        // dx[row][i] = dy[row][i] + sum_o dy[row][o] W[o][i]
        {
            float acc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = dYs[rl * (DP + 1) + f0 + j];
            for (int o = 0; o < a.d; ++o) {
                const float g = dYs[rl * (DP + 1) + o];
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] += g * Ws[o * (DP + 1) + f0 + j];
            }
            const int row = r0 + rl;
            if (row < a.rows) {
                float* out = a.dx + ((size_t)b * a.rows + row) * a.d;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int i = f0 + j;
                    if (i < a.d) out[i] = a.accumulate ? out[i] + acc[j] : acc[j];
                }
            }
        }
        // dW[o][i] += sum_r dy[r][o] x[r][i]   (thread: o = tid>>2, i = f0..f0+15)
        {
            const int o = rl;
            for (int r = 0; r < 64; ++r) {
                const float g = dYs[r * (DP + 1) + o];
#pragma unroll
                for (int j = 0; j < 16; ++j) dw_acc[j] += g * Xs[r * (DP + 1) + f0 + j];
            }
            if (tid < DP)
                for (int r = 0; r < 64; ++r) db_acc += dYs[r * (DP + 1) + tid];
        }
    }
    const int o = rl;
    if (o < a.d) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (f0 + j < a.d) atomic_add_f32(&a.dW[o * a.d + f0 + j], dw_acc[j]);
    }
    if (tid < a.d) atomic_add_f32(&a.db[tid], db_acc);
}

// channels-last max-pool backward, kernel = stride = k, ceil_mode: every input cell belongs to one
// window; it receives the window's gradient iff it is the first maximum in (h, w) scan order.
__global__ __launch_bounds__(256) void ctx_pool_bwd_kernel(const float* __restrict__ in, long long in_img,
                                                           const float* __restrict__ dpool,
                                                           long long pool_img, float* __restrict__ din,
                                                           long long din_img, int batch, int H, int W,
                                                           int OH, int OW, int ch, int k)
{
    const long total = (long)batch * H * W * ch;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % ch);
        long t = idx / ch;
        const int w = (int)(t % W);
        t /= W;
        const int hh = (int)(t % H);
        const int n = (int)(t / H);
        const int oh = hh / k, ow = w / k;
        const float* p = in + (long long)n * in_img;
        const int h1 = min(oh * k + k, H), w1 = min(ow * k + k, W);
        float m = -INFINITY;
        int am = -1;
        for (int y = oh * k; y < h1; ++y)
            for (int x = ow * k; x < w1; ++x) {
                const float v = p[((long)y * W + x) * ch + c];
                if (v > m || am < 0) { m = v; am = y * W + x; }
            }
        if (am == hh * W + w)
            din[(long long)n * din_img + ((long)hh * W + w) * ch + c] +=
                dpool[(long long)n * pool_img + ((long)oh * OW + ow) * ch + c];
    }
}

struct BwdWs {
    float *Qs, *Ksw, *Vsw, *dDs, *delta, *dQ, *dK, *dV;
    unsigned char *Qx0, *dDx0, *Kx1, *Vx1, *Kx2;      // bf16x3 fragments for ctx_attn_bwd_q (x3_emit modes 0 / 1 / 2)
    unsigned char *Kx0, *Vx0, *Qx1, *dDx1, *Qx2, *dDx2;   // ... and for ctx_attn_bwd_kv
    int P_pad, M_pad;
    size_t total;
};

BwdWs carve_bwd(char* base, int batch, int P, int M)
{
    BwdWs w{};
    w.P_pad = (P + QB - 1) / QB * QB;
    w.M_pad = (M + QB - 1) / QB * QB;
    size_t off = 0;
    auto take = [&](size_t floats) {
        char* p = base ? base + off : nullptr;
        off += ctdet::align_up(floats * 4, 256);
        return (float*)p;
    };
    const size_t pq = (size_t)batch * w.P_pad * DP, mk = (size_t)batch * w.M_pad * DP;
    w.Qs = take(pq); w.dDs = take(pq); w.dQ = take(pq);
    w.delta = take((size_t)batch * w.P_pad);
    w.Ksw = take(mk); w.Vsw = take(mk);
    w.dK = take(mk); w.dV = take(mk);          // adjacent: zeroed with one memset
    const size_t px = (size_t)batch * w.P_pad * XQ_BYTES / 4, mx = (size_t)batch * (w.M_pad / KT) * XT_BYTES / 4;
    w.Qx0 = (unsigned char*)take(px); w.dDx0 = (unsigned char*)take(px);
    w.Kx1 = (unsigned char*)take(mx); w.Vx1 = (unsigned char*)take(mx); w.Kx2 = (unsigned char*)take(mx);
    const size_t mx0 = (size_t)batch * w.M_pad * XQ_BYTES / 4, px1 = (size_t)batch * (w.P_pad / KT) * XT_BYTES / 4;
    w.Kx0 = (unsigned char*)take(mx0); w.Vx0 = (unsigned char*)take(mx0);
    w.Qx1 = (unsigned char*)take(px1); w.dDx1 = (unsigned char*)take(px1);
    w.Qx2 = (unsigned char*)take(px1); w.dDx2 = (unsigned char*)take(px1);
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t ct_ctx_attention_bwd_workspace_bytes(int batch, int num_priors, int num_ctx)
{
    return carve_bwd(nullptr, batch, num_priors, num_ctx).total;
}

extern "C" int ct_ctx_attention_bwd(const float* conf, const float* pool, int batch, int num_priors,
                                    int num_ctx, const ct_ctx_params* prm, const void* saved,
                                    const float* dout, float* dconf, float* dpool, const ct_ctx_grads* grads,
                                    void* workspace, size_t workspace_bytes, ct_stream_t stream)
{
    CT_REQUIRE(conf && pool && prm && saved && dout && dconf && dpool && grads && workspace,
               "ct_ctx_attention_bwd: null pointer");
    CT_REQUIRE(batch > 0 && num_priors > 0 && num_ctx > 0, "ct_ctx_attention_bwd: bad sizes");
    CT_REQUIRE(prm->d >= 1 && prm->d <= DP && prm->t >= 1 && prm->t <= 32, "ct_ctx_attention_bwd: d=%d t=%d",
               prm->d, prm->t);
    CT_REQUIRE(grads->theta_w && grads->theta_b && grads->phi_w && grads->phi_b && grads->g_w && grads->g_b &&
                   grads->wz && grads->obj_w, "ct_ctx_attention_bwd: null gradient buffer");
    CT_REQUIRE(!prm->fc_w || (grads->fc_w && grads->fc_b && prm->fc_b), "ct_ctx_attention_bwd: fc gradient missing");
    const size_t need = ct_ctx_attention_bwd_workspace_bytes(batch, num_priors, num_ctx);
    if (workspace_bytes < need)
        return ctdet::fail(CT_ERR_WORKSPACE, "ct_ctx_attention_bwd: workspace %zu < %zu", workspace_bytes, need);
    BwdWs w = carve_bwd((char*)workspace, batch, num_priors, num_ctx);
    hipStream_t st = ctdet::as_stream(stream);
    const int d = prm->d, T = prm->t;
    const int ostride = (prm->fc_w ? d : 0) + T;
    const dim3 blk(256);
    float* none = nullptr;
    const float* save_d = (const float*)saved;
    const float* save_lse = save_d + (size_t)batch * w.P_pad * DP;

    // zero the accumulated outputs
    CT_HIP(hipMemsetAsync(grads->theta_w, 0, (size_t)d * d * 4, st));
    CT_HIP(hipMemsetAsync(grads->phi_w, 0, (size_t)d * d * 4, st));
    CT_HIP(hipMemsetAsync(grads->g_w, 0, (size_t)d * d * 4, st));
    CT_HIP(hipMemsetAsync(grads->theta_b, 0, (size_t)d * 4, st));
    CT_HIP(hipMemsetAsync(grads->phi_b, 0, (size_t)d * 4, st));
    CT_HIP(hipMemsetAsync(grads->g_b, 0, (size_t)d * 4, st));
    CT_HIP(hipMemsetAsync(grads->wz, 0, (size_t)d * 4, st));
    CT_HIP(hipMemsetAsync(grads->obj_w, 0, (size_t)T * d * 4, st));
    if (prm->fc_w) {
        CT_HIP(hipMemsetAsync(grads->fc_w, 0, (size_t)d * d * 4, st));
        CT_HIP(hipMemsetAsync(grads->fc_b, 0, (size_t)d * 4, st));
    }
    CT_HIP(hipMemsetAsync(w.dK, 0, ((char*)w.dV - (char*)w.dK) + (size_t)batch * w.M_pad * DP * 4, st));

    // recompute the projections in the operand layouts of the two MFMA kernels
    hipLaunchKernelGGL(ctx_project_kernel, dim3(w.P_pad / 64, batch), blk, 0, st, conf, num_priors, w.P_pad, d,
                       prm->theta_w, prm->theta_b, w.Qs, none, none, none, 0);
    CT_LAUNCH_CHECK("ctx_project_kernel(theta)");
    hipLaunchKernelGGL(ctx_project_kernel, dim3(w.M_pad / 64, batch), blk, 0, st, pool, num_ctx, w.M_pad, d,
                       prm->phi_w, prm->phi_b, w.Ksw, none, none, none, 0);
    CT_LAUNCH_CHECK("ctx_project_kernel(phi)");
    hipLaunchKernelGGL(ctx_project_kernel, dim3(w.M_pad / 64, batch), blk, 0, st, pool, num_ctx, w.M_pad, d,
                       prm->g_w, prm->g_b, w.Vsw, none, none, none, 0);
    CT_LAUNCH_CHECK("ctx_project_kernel(g)");

    OutBwdArgs oa{};
    oa.conf = conf; oa.D = save_d; oa.dout = dout; oa.wz = prm->wz; oa.obj_w = prm->obj_w;
    oa.dconf = dconf; oa.dDs = w.dDs; oa.delta = w.delta;
    oa.dwz = grads->wz; oa.dobj = grads->obj_w;
    oa.P = num_priors; oa.P_pad = w.P_pad; oa.d = d; oa.T = T; oa.ostride = ostride;
    oa.ooff = prm->fc_w ? d : 0; oa.chunks = 4; oa.scale = prm->scale;
    hipLaunchKernelGGL(ctx_out_bwd_kernel, dim3((w.P_pad / 64 + oa.chunks - 1) / oa.chunks, batch), blk, 0, st, oa);
    CT_LAUNCH_CHECK("ctx_out_bwd_kernel");

    // enough workgroups to fill 256 CUs twice over
    const int kv_blocks = (w.M_pad / QB) * batch;
    const int kv_split = std::max(1, std::min(w.P_pad / (8 * KT), (1024 + kv_blocks - 1) / kv_blocks));
    {
        auto to_x3 = [&](const float* src, int rows_pad, unsigned char* dst, int mode) {
            hipLaunchKernelGGL(ctx_rows_to_x3_kernel, dim3(rows_pad / 64, batch), blk, 0, st, src, rows_pad,
                               (unsigned short*)dst, mode);
        };
        to_x3(w.Qs, w.P_pad, w.Qx0, 0);
        to_x3(w.dDs, w.P_pad, w.dDx0, 0);
        to_x3(w.Ksw, w.M_pad, w.Kx1, 1);
        to_x3(w.Vsw, w.M_pad, w.Vx1, 1);
        to_x3(w.Ksw, w.M_pad, w.Kx2, 2);
        to_x3(w.Ksw, w.M_pad, w.Kx0, 0);
        to_x3(w.Vsw, w.M_pad, w.Vx0, 0);
        to_x3(w.Qs, w.P_pad, w.Qx1, 1);
        to_x3(w.dDs, w.P_pad, w.dDx1, 1);
        to_x3(w.Qs, w.P_pad, w.Qx2, 2);
        to_x3(w.dDs, w.P_pad, w.dDx2, 2);
        CT_LAUNCH_CHECK("ctx_rows_to_x3_kernel");
        BwdQx3Args qa{};
        qa.Qx = w.Qx0; qa.dDx = w.dDx0; qa.Kx = w.Kx1; qa.Vk = w.Vx1; qa.Kv = w.Kx2;
        qa.lse = save_lse; qa.delta = w.delta; qa.dQ = w.dQ;
        qa.P_pad = w.P_pad; qa.M = num_ctx; qa.M_pad = w.M_pad;
        static std::once_flag once_q;
        static hipError_t attr_q = hipSuccess;
        std::call_once(once_q, [] {
            attr_q = hipFuncSetAttribute((const void*)ctx_attn_bwd_q, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         2 * 3 * XT_BYTES);
        });
        CT_HIP(attr_q);
        hipLaunchKernelGGL(ctx_attn_bwd_q, dim3(w.P_pad / QB, batch), blk, 2 * 3 * XT_BYTES, st, qa);
        CT_LAUNCH_CHECK("ctx_attn_bwd_q");
    }
    {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute((const void*)ctx_attn_bwd_kv<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           KvCfg<true>::LDS_BYTES);
            if (attr_err == hipSuccess)
                attr_err = hipFuncSetAttribute((const void*)ctx_attn_bwd_kv<false>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, KvCfg<false>::LDS_BYTES);
        });
        CT_HIP(attr_err);
        BwdKVx3Args ka{};
        ka.Kx0 = w.Kx0; ka.Vx0 = w.Vx0; ka.Qx1 = w.Qx1; ka.dDx1 = w.dDx1; ka.Qx2 = w.Qx2; ka.dDx2 = w.dDx2;
        ka.lse = save_lse; ka.delta = w.delta; ka.dK = w.dK; ka.dV = w.dV;
        ka.P_pad = w.P_pad; ka.M = num_ctx; ka.M_pad = w.M_pad; ka.split = kv_split;
        hipLaunchKernelGGL(ctx_attn_bwd_kv<false>, dim3(w.M_pad / QB, batch, kv_split), blk, KvCfg<false>::LDS_BYTES, st, ka);
        CT_LAUNCH_CHECK("ctx_attn_bwd_kv<dV>");
        hipLaunchKernelGGL(ctx_attn_bwd_kv<true>, dim3(w.M_pad / QB, batch, kv_split), blk, KvCfg<true>::LDS_BYTES, st, ka);
        CT_LAUNCH_CHECK("ctx_attn_bwd_kv<dK>");
    }

    // projections backward
    auto linear = [&](const float* dy, long long dy_bs, int dy_stride, const float* x, int rows, const float* W,
                      float* dx, int accumulate, float* dW, float* db, const char* name) -> int {
        LinBwdArgs la{};
        la.dy = dy; la.x = x; la.W = W; la.dx = dx; la.dW = dW; la.db = db;
        la.dy_batch_stride = dy_bs; la.rows = rows; la.d = d; la.dy_stride = dy_stride;
        la.accumulate = accumulate; la.chunks = 4;
        hipLaunchKernelGGL(ctx_linear_bwd_kernel, dim3((rows + 64 * la.chunks - 1) / (64 * la.chunks), batch), blk,
                           0, st, la);
        CT_LAUNCH_CHECK(name);
        return CT_OK;
    };
    if (int rc = linear(w.dQ, (long long)w.P_pad * DP, DP, conf, num_priors, prm->theta_w, dconf, 1,
                        grads->theta_w, grads->theta_b, "ctx_linear_bwd_kernel(theta)")) return rc;
    if (prm->fc_w)
        if (int rc = linear(dout, (long long)num_priors * ostride, ostride, conf, num_priors, prm->fc_w, dconf, 1,
                            grads->fc_w, grads->fc_b, "ctx_linear_bwd_kernel(fc_base)")) return rc;
    if (int rc = linear(w.dK, (long long)w.M_pad * DP, DP, pool, num_ctx, prm->phi_w, dpool, 0, grads->phi_w,
                        grads->phi_b, "ctx_linear_bwd_kernel(phi)")) return rc;
    if (int rc = linear(w.dV, (long long)w.M_pad * DP, DP, pool, num_ctx, prm->g_w, dpool, 1, grads->g_w,
                        grads->g_b, "ctx_linear_bwd_kernel(g)")) return rc;
    return CT_OK;
}

extern "C" int ct_ctx_pool_bwd(const float* in, long long in_img_stride, const float* dpool,
                               long long dpool_img_stride, float* din, long long din_img_stride, int batch,
                               int h, int w, int ch, int k, ct_stream_t stream)
{
    CT_REQUIRE(in && dpool && din && batch > 0 && h > 0 && w > 0 && ch > 0 && k >= 1, "ct_ctx_pool_bwd: bad argument");
    const int oh = (h + k - 1) / k, ow = (w + k - 1) / k;
    const long total = (long)batch * h * w * ch;
    hipLaunchKernelGGL(ctx_pool_bwd_kernel, dim3((int)std::min<long>((total + 255) / 256, 256 * 16)), dim3(256), 0,
                       ctdet::as_stream(stream), in, in_img_stride, dpool, dpool_img_stride, din, din_img_stride,
                       batch, h, w, oh, ow, ch, k);
    CT_LAUNCH_CHECK("ctx_pool_bwd_kernel");
    return CT_OK;
}
