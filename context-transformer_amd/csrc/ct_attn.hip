// libctdet: the Context-Transformer block (models/RFB_Net_vgg.py:253-271) as two kernels.
//
//   ctx_project_x3_kernel  theta/phi/g = Linear(x) + x, split into three exact bfloat16 pieces and written straight
//                        into the MFMA fragment order the attention kernel reads (zero padded d -> 64; layouts at the
//                        kernel).  ctx_project_kernel (ct_attn_common.h) still writes fc_base for the 'incre' branch.
//   ctx_attn_kernel      flash-style fused  softmax(theta phi^T) g; the [P,M] affinity matrix (86 MB/image in
//                        the reference) never exists.  One wave = 32 queries, workgroup = 128.  Since round 3 both
//                        contractions run as bf16x3 on v_mfma_f32_32x32x16_bf16 (every fp32 operand split exactly
//                        into three bfloat16 pieces, six piece products per multiply, fp32 accumulation; the
//                        theta.phi^T logits in two accumulators -- csrc/ct_conv_x3.hip has the arithmetic): 48 bf16
//                        MFMAs per 32-key tile and wave instead of 64 fp32 ones at twice the issue time; measured on
//                        the 512 train-mode logits (|S| ~ 370, d = 64) the log-sum-exp is 0.75x and the block output
//                        0.56x the error of the fp32 chain against float64.  theta /
//                        phi / g are split ONCE by ctx_project_x3_kernel, straight into the fragment order the MFMAs
//                        read (keys of a V fragment in the order the S^T accumulator holds them), only the
//                        probabilities are split in registers.
//                        Per 32-key tile:
//                          S^T = K Q^T   A = phi fragments from LDS, B = theta fragments in registers
//                                        -> lane (q = l&31, h = l>>5) holds 16 keys of query q,
//                                        so the softmax max / sum are in-lane + one lane^32 swap
//                          O^T += V^T P^T  B operand = the S^T accumulator registers split in place (the key order of
//                                        a g fragment is chosen to match), A = g fragments from LDS
//                        Epilogue in registers: (conf + O/l * Wz) -> L2 normalise -> cosine
//                        classifier OBJ_Target * scale, written to out[B,P,(d)+T].
#include "ct_common.h"
#include "ct_attn_common.h"

namespace {

// y = Linear(x) + x (as ctx_project_kernel), written as bf16x3 pieces in MFMA fragment order:
//   mode 0 (theta -> Q, B operand of S^T = K Q^T): Qx[b][row][piece 3][octet 8][8]      element d at octet d/8, slot d%8
//   mode 1 (phi -> K, A operand of S^T):           Kx[b][tile][piece][octet 8][key 32][8]
//   mode 2 (g -> V, A operand of O^T += V^T P^T):  Vx[b][tile][piece][kg 2][h 2][d 64][8]  slot j of (kg, h) = the key the
//          S^T accumulator register 8 kg + j of lane half h holds (acc_row): P^T goes to the MFMA as it sits in registers
__global__ __launch_bounds__(256) void ctx_project_x3_kernel(const float* __restrict__ x, int rows_valid, int rows_pad,
                                                             int d, const float* __restrict__ W,
                                                             const float* __restrict__ bias,
                                                             unsigned short* __restrict__ out, int mode)
{
    __shared__ float Wt[DP * DP];                                  // Wt[i][o]
    __shared__ __attribute__((aligned(16))) float Xt[DP * 64];     // Xt[i][row]: four rows of a feature are one 16-byte broadcast
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    for (int e = tid; e < DP * DP; e += 256) {
        const int i = e / DP, o = e % DP;
        Wt[e] = (i < d && o < d) ? W[o * d + i] : 0.f;
    }
    for (int e = tid; e < 64 * DP; e += 256) {
        const int r = e / DP, i = e % DP;
        const int row = r0 + r;
        Xt[i * 64 + r] = (row < rows_valid && i < d) ? x[((size_t)b * rows_valid + row) * d + i] : 0.f;
    }
    __syncthreads();
    // thread = output feature o of rows 16 rg .. 16 rg + 15, four rows at a time: one weight read and one 16-byte row read
    // per four FMAs (one row per pass: two LDS reads per FMA, the kernel was LDS-bound); every output still sums its 64
    // products in the order i = 0 .. 63, so the values are those of ctx_project_kernel
    const int o = tid & 63, rg = tid >> 6;
    const float bo = (o < d) ? bias[o] : 0.f;
    for (int rq = 0; rq < 4; ++rq) {
        const int rl = 16 * rg + 4 * rq;
        if (r0 + rl >= rows_pad) break;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int i = 0; i < DP; ++i) {
            const float w = Wt[i * DP + o];
            const float4 xv = *reinterpret_cast<const float4*>(&Xt[i * 64 + rl]);
            acc[0] += xv.x * w; acc[1] += xv.y * w; acc[2] += xv.z * w; acc[3] += xv.w * w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = r0 + rl + k;
            if (row >= rows_pad) break;
            const float y = (row < rows_valid && o < d) ? acc[k] + bo + Xt[o * 64 + rl + k] : 0.f;
            x3_emit(y, out, mode, b, row, rows_pad, o);
        }
    }
}

struct AttnArgs {
    const unsigned char* Qx;
    const unsigned char* Kx;
    const unsigned char* Vx;
    const float* conf;
    const float* wz;
    const float* obj_w;
    float* out;
    float* save_d;      // training: aggregated context rows D = softmax(S) g, [B][P_pad][64]
    float* save_lse;    // training: log2-domain log-sum-exp of each affinity row, [B][P_pad]
    int P, P_pad, M, M_pad, d, T, ostride, ooff;
    float scale;
};

__global__ __launch_bounds__(256) void ctx_attn_kernel(const AttnArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char kv[2][2 * XT_BYTES];   // per buffer: K tile, V tile (fragments)
    __shared__ float objw[32 * DP];
    __shared__ float wzs[DP];

    // (Round 5 measured the XCD-aware numbering -- all query blocks of an image on one XCD, so that its K / V stripe stays in that
    // L2 -- and dropped it: 1.28 ms against 1.06 at P = 11 620 / M = 1 858 / bs 32.  The kernel is bound by its vector + matrix
    // issue time, not by the stripe re-reads, and eight images in flight instead of 32 lose the tail balance.)
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x * QB + wave * QW + l31;       // < P_pad by construction

    for (int e = tid; e < 32 * DP; e += 256) {
        const int t = e / DP, dd = e % DP;
        objw[e] = (t < a.T && dd < a.d) ? a.obj_w[t * a.d + dd] : 0.f;
    }
    if (tid < DP) wzs[tid] = tid < a.d ? a.wz[tid] : 0.f;

    // Q fragments (B operand of S^T): group g = 16 features, lane half h = octet 2g + h, three pieces
    i32x4 qf[4][3];
    {
        const unsigned char* qp = a.Qx + ((size_t)b * a.P_pad + q) * XQ_BYTES;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                qf[g][p] = *reinterpret_cast<const i32x4*>(qp + (p * 8 + 2 * g + h) * 16);
    }

    const int nt = a.M_pad / KT;
    const unsigned char* Kxb = a.Kx + (size_t)b * nt * XT_BYTES;
    const unsigned char* Vxb = a.Vx + (size_t)b * nt * XT_BYTES;

    i32x4 treg[6];                                         // 24 KB per tile / 256 threads
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            treg[i] = *reinterpret_cast<const i32x4*>(Kxb + (size_t)t * XT_BYTES + (tid + 256 * i) * 16);
            treg[3 + i] = *reinterpret_cast<const i32x4*>(Vxb + (size_t)t * XT_BYTES + (tid + 256 * i) * 16);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            *reinterpret_cast<i32x4*>(&kv[buf][(tid + 256 * i) * 16]) = treg[i];
            *reinterpret_cast<i32x4*>(&kv[buf][XT_BYTES + (tid + 256 * i) * 16]) = treg[3 + i];
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, mb_run = -INFINITY, l_run = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    // piece pairs (A piece, B piece) of the six products, smallest first; the last one is hi.hi
    constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < nt;
        if (more) load_tile(t + 1);

        // ---- S^T = K Q^T ----  hi.hi products in `s`, the five small ones in `ss`
        f32x16 s, ss;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; ss[r] = 0.f; }
        {
            const unsigned char* kb = &kv[buf][l31 * 16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                i32x4 kf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) kf[p] = *reinterpret_cast<const i32x4*>(kb + ((p * 8 + 2 * g + h) * KT) * 16);
#pragma unroll
                for (int c = 0; c < 5; ++c)
                    ss = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[PA[c]]),
                                                                 __builtin_bit_cast(bf16x8, qf[g][PB[c]]), ss, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[0]),
                                                            __builtin_bit_cast(bf16x8, qf[g][0]), s, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += ss[r];
        }

        // ---- online softmax over this tile's 32 keys (16 in-lane + partner lane^32) ----
        if ((t + 1) * KT > a.M) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t * KT + acc_row(r, h) >= a.M) s[r] = -INFINITY;
        }
        float mloc = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);
        // exp(x) = 2^(x*log2 e) on the hardware exp2 (v_exp_f32): only keys within a few units of the row maximum carry
        // weight, and there fma(s, log2 e, -mb) is good to ~2 ulp.  `mb` is the ROUNDED product m * log2 e and every use
        // must see that same value: the probabilities of all tiles that share a maximum are then scaled by one common
        // factor 2^(m log2 e - mb), which the normalisation removes, and the rescale between two maxima is exactly
        // 2^(mb_old - mb_new).  Left to -ffp-contract the compiler is free to fuse the multiply into one of the
        // subtractions (it did: alpha = 2^(m_run log2 e - mb) != 1 with an unchanged maximum, 2e-5 per tile, compounding
        // over 156 tiles), hence the opaque copy.
        constexpr float kLog2e = 1.4426950408889634f;
        float mb = m_new * kLog2e;
        asm volatile("" : "+v"(mb));
        const float alpha = __builtin_amdgcn_exp2f(mb_run - mb);
        mb_run = mb;
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r] * kLog2e - mb);
            lsum += s[r];
        }
        lsum += __shfl_xor(lsum, 32);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

        // ---- O^T += V^T P^T ----  B operand: this lane's own probabilities (registers 8 kg .. 8 kg + 7), split in place
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
            const unsigned char* vb = &kv[buf][XT_BYTES + l31 * 16];
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) {
                i32x4 vf[3][2];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int db = 0; db < 2; ++db)
                        vf[p][db] = *reinterpret_cast<const i32x4*>(vb + ((((p * 2 + kg) * 2 + h) * DP) + 32 * db) * 16);
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[PA[c]][0]),
                                                                 __builtin_bit_cast(bf16x8, pf[kg][PB[c]]), o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[PA[c]][1]),
                                                                 __builtin_bit_cast(bf16x8, pf[kg][PB[c]]), o1, 0, 0, 0);
                }
            }
        }

        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: residual, L2 normalise, cosine classifier ----
    if (q >= a.P) {
        if (a.save_d) {                       // padding rows: exp2(s - inf) = 0 in the backward pass
            float4* drow = reinterpret_cast<float4*>(a.save_d + ((size_t)b * a.P_pad + q) * DP);
            for (int g = 0; g < 4; ++g) {
                drow[2 * g + h] = make_float4(0.f, 0.f, 0.f, 0.f);
                drow[8 + 2 * g + h] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (h == 0) a.save_lse[(size_t)b * a.P_pad + q] = INFINITY;
        }
        return;
    }
    const float inv_l = 1.f / l_run;
    if (a.save_d) {
        float4* drow = reinterpret_cast<float4*>(a.save_d + ((size_t)b * a.P_pad + q) * DP);
#pragma unroll
        for (int g = 0; g < 4; ++g) {         // acc rows 4g..4g+3 are features 8g+4h .. +3 (and +32)
            drow[2 * g + h] = make_float4(o0[4 * g] * inv_l, o0[4 * g + 1] * inv_l, o0[4 * g + 2] * inv_l,
                                          o0[4 * g + 3] * inv_l);
            drow[8 + 2 * g + h] = make_float4(o1[4 * g] * inv_l, o1[4 * g + 1] * inv_l, o1[4 * g + 2] * inv_l,
                                              o1[4 * g + 3] * inv_l);
        }
        if (h == 0) a.save_lse[(size_t)b * a.P_pad + q] = mb_run + log2f(l_run);
    }
    const float* crow = a.conf + ((size_t)b * a.P + q) * a.d;
    float x[32];
    float n2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d0 = acc_row(r, h), d1 = 32 + d0;
        const float c0 = d0 < a.d ? crow[d0] : 0.f;
        const float c1 = d1 < a.d ? crow[d1] : 0.f;
        x[r] = c0 + o0[r] * inv_l * wzs[d0];
        x[16 + r] = c1 + o1[r] * inv_l * wzs[d1];
        n2 += x[r] * x[r] + x[16 + r] * x[16 + r];
    }
    n2 += __shfl_xor(n2, 32);
    const float nrm = sqrtf(n2);
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = x[r] / nrm;
    float* orow = a.out + ((size_t)b * a.P + q) * a.ostride + a.ooff;
    for (int t = 0; t < a.T; ++t) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d0 = acc_row(r, h);
            acc += x[r] * objw[t * DP + d0] + x[16 + r] * objw[t * DP + 32 + d0];
        }
        acc += __shfl_xor(acc, 32);
        if (h == 0) orow[t] = acc * a.scale;
    }
}

struct Ws {
    unsigned char *Qx, *Kx, *Vx;      // bf16x3 fragments (ctx_project_x3_kernel)
    int P_pad, M_pad;
    size_t total;
};

Ws carve(char* base, int batch, int P, int M)
{
    Ws w{};
    w.P_pad = (P + QB - 1) / QB * QB;
    w.M_pad = (M + KT - 1) / KT * KT;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += ctdet::align_up(bytes, 256);
        return (unsigned char*)p;
    };
    w.Qx = take((size_t)batch * w.P_pad * XQ_BYTES);
    w.Kx = take((size_t)batch * (w.M_pad / KT) * XT_BYTES);
    w.Vx = take((size_t)batch * (w.M_pad / KT) * XT_BYTES);
    w.total = off;
    return w;
}

int check_params(const ct_ctx_params* prm, const char* who)
{
    CT_REQUIRE(prm->d >= 1 && prm->d <= DP && prm->t >= 1 && prm->t <= 32, "%s: d=%d (<=64) t=%d (<=32)", who,
               prm->d, prm->t);
    CT_REQUIRE(prm->theta_w && prm->theta_b && prm->phi_w && prm->phi_b && prm->g_w && prm->g_b && prm->wz &&
                   prm->obj_w, "%s: null parameter", who);
    CT_REQUIRE(!prm->fc_w || prm->fc_b, "%s: fc_b missing", who);
    return CT_OK;
}

int forward_impl(const float* conf, const float* pool, int batch, int num_priors, int num_ctx,
                 const ct_ctx_params* prm, float* out, float* save_d, float* save_lse, void* workspace,
                 size_t workspace_bytes, hipStream_t st)
{
    const size_t need = carve(nullptr, batch, num_priors, num_ctx).total;
    if (workspace_bytes < need)
        return ctdet::fail(CT_ERR_WORKSPACE, "ct_ctx_attention_fwd: workspace %zu < %zu", workspace_bytes, need);
    Ws w = carve((char*)workspace, batch, num_priors, num_ctx);
    const int d = prm->d;
    const int ostride = (prm->fc_w ? d : 0) + prm->t;
    const dim3 blk(256);
    float* none = nullptr;
    { CT_PROF("ctx_project_kernel", st); hipLaunchKernelGGL(ctx_project_x3_kernel, dim3(w.P_pad / 64, batch), blk, 0, st, conf, num_priors,
                       w.P_pad, d, prm->theta_w, prm->theta_b, (unsigned short*)w.Qx, 0); }
    CT_LAUNCH_CHECK("ctx_project_x3_kernel(theta)");
    { CT_PROF("ctx_project_kernel", st); hipLaunchKernelGGL(ctx_project_x3_kernel, dim3((w.M_pad + 63) / 64, batch), blk, 0, st, pool, num_ctx,
                       w.M_pad, d, prm->phi_w, prm->phi_b, (unsigned short*)w.Kx, 1); }
    CT_LAUNCH_CHECK("ctx_project_x3_kernel(phi)");
    { CT_PROF("ctx_project_kernel", st); hipLaunchKernelGGL(ctx_project_x3_kernel, dim3((w.M_pad + 63) / 64, batch), blk, 0, st, pool, num_ctx,
                       w.M_pad, d, prm->g_w, prm->g_b, (unsigned short*)w.Vx, 2); }
    CT_LAUNCH_CHECK("ctx_project_x3_kernel(g)");
    if (prm->fc_w) {
        { CT_PROF("ctx_project_kernel", st); hipLaunchKernelGGL(ctx_project_kernel, dim3((num_priors + 63) / 64, batch), blk, 0, st, conf,
                           num_priors, num_priors, d, prm->fc_w, prm->fc_b, none, none, none, out, ostride); }
        CT_LAUNCH_CHECK("ctx_project_kernel(fc_base)");
    }
    AttnArgs a{};
    a.Qx = w.Qx; a.Kx = w.Kx; a.Vx = w.Vx;
    a.conf = conf; a.wz = prm->wz; a.obj_w = prm->obj_w; a.out = out;
    a.save_d = save_d; a.save_lse = save_lse;
    a.P = num_priors; a.P_pad = w.P_pad; a.M = num_ctx; a.M_pad = w.M_pad;
    a.d = d; a.T = prm->t; a.ostride = ostride; a.ooff = prm->fc_w ? d : 0;
    a.scale = prm->scale;
    { CT_PROF("ctx_attn_kernel", st); hipLaunchKernelGGL(ctx_attn_kernel, dim3(w.P_pad / QB, batch), blk, 0, st, a); }
    CT_LAUNCH_CHECK("ctx_attn_kernel");
    return CT_OK;
}

}  // namespace

extern "C" size_t ct_ctx_attention_workspace_bytes(int batch, int num_priors, int num_ctx, int)
{
    return carve(nullptr, batch, num_priors, num_ctx).total;
}

extern "C" int ct_ctx_attention_fwd(const float* conf, const float* pool, int batch, int num_priors,
                                    int num_ctx, const ct_ctx_params* prm, float* out, void* workspace,
                                    size_t workspace_bytes, ct_stream_t stream)
{
    CT_REQUIRE(conf && pool && prm && out && workspace, "ct_ctx_attention_fwd: null pointer");
    CT_REQUIRE(batch > 0 && num_priors > 0 && num_ctx > 0, "ct_ctx_attention_fwd: bad sizes");
    if (int rc = check_params(prm, "ct_ctx_attention_fwd")) return rc;
    return forward_impl(conf, pool, batch, num_priors, num_ctx, prm, out, nullptr, nullptr, workspace,
                        workspace_bytes, ctdet::as_stream(stream));
}

extern "C" size_t ct_ctx_attention_saved_bytes(int batch, int num_priors)
{
    const size_t P_pad = (size_t)(num_priors + QB - 1) / QB * QB;
    return (size_t)batch * P_pad * (DP + 1) * sizeof(float);
}

extern "C" int ct_ctx_attention_fwd_train(const float* conf, const float* pool, int batch, int num_priors,
                                          int num_ctx, const ct_ctx_params* prm, float* out, void* saved,
                                          size_t saved_bytes, void* workspace, size_t workspace_bytes,
                                          ct_stream_t stream)
{
    CT_REQUIRE(conf && pool && prm && out && workspace && saved, "ct_ctx_attention_fwd_train: null pointer");
    CT_REQUIRE(batch > 0 && num_priors > 0 && num_ctx > 0, "ct_ctx_attention_fwd_train: bad sizes");
    if (int rc = check_params(prm, "ct_ctx_attention_fwd_train")) return rc;
    if (saved_bytes < ct_ctx_attention_saved_bytes(batch, num_priors))
        return ctdet::fail(CT_ERR_WORKSPACE, "ct_ctx_attention_fwd_train: saved buffer %zu < %zu", saved_bytes,
                           ct_ctx_attention_saved_bytes(batch, num_priors));
    const size_t P_pad = (size_t)(num_priors + QB - 1) / QB * QB;
    float* save_d = (float*)saved;
    float* save_lse = save_d + (size_t)batch * P_pad * DP;
    return forward_impl(conf, pool, batch, num_priors, num_ctx, prm, out, save_d, save_lse, workspace,
                        workspace_bytes, ctdet::as_stream(stream));
}
