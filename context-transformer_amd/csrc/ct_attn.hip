// libctdet: the Context-Transformer block (models/RFB_Net_vgg.py:253-271) as two kernels.
//
//   ctx_project_kernel   theta/phi/g (and fc_base) = Linear(x) + x, written straight into the
//                        operand layouts the attention kernel wants (zero padded d -> 64):
//                          Qs [B][P_pad][2][32]   Qs[.][p][h][s] = theta[p][2s+h]
//                          Kt [B][64][M_pad]      phi transposed (d-major, keys contiguous)
//                          Vs [B][M_pad][64]      g rows
//   ctx_attn_kernel      flash-style fused  softmax(theta phi^T) g  on the fp32 MFMA path
//                        (v_mfma_f32_32x32x2_f32), the [P,M] affinity matrix (86 MB/image in
//                        the reference) never exists.  One wave = 32 queries, workgroup = 128.
//                        Per 32-key tile:
//                          S^T = K Q^T   A = Kt tile from LDS (keys contiguous), B = Q registers
//                                        -> lane (q = l&31, h = l>>5) holds 16 keys of query q,
//                                        so the softmax max / sum are in-lane + one lane^32 swap
//                          O^T += V^T P^T  B operand = the S^T accumulator registers as they
//                                        are (key pairing is free to choose), A = V tile rows
//                        Epilogue in registers: (conf + O/l * Wz) -> L2 normalise -> cosine
//                        classifier OBJ_Target * scale, written to out[B,P,(d)+T].
#include "ct_common.h"
#include "ct_attn_common.h"

namespace {

struct AttnArgs {
    const float* Qs;
    const float* Kt;
    const float* Vs;
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
    __shared__ __attribute__((aligned(16))) float ks[2][DP * KT];   // [d][key]
    __shared__ __attribute__((aligned(16))) float vs[2][KT * DP];   // [key][d]
    __shared__ float objw[32 * DP];
    __shared__ float wzs[DP];

    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x * QB + wave * QW + l31;       // < P_pad by construction

    for (int e = tid; e < 32 * DP; e += 256) {
        const int t = e / DP, dd = e % DP;
        objw[e] = (t < a.T && dd < a.d) ? a.obj_w[t * a.d + dd] : 0.f;
    }
    if (tid < DP) wzs[tid] = tid < a.d ? a.wz[tid] : 0.f;

    // Q fragment: B operand of step s is Qs[q][h][s]
    float qreg[32];
    {
        const float4* qp = reinterpret_cast<const float4*>(a.Qs + ((size_t)b * a.P_pad + q) * DP + h * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 v = qp[i];
            qreg[4 * i + 0] = v.x; qreg[4 * i + 1] = v.y; qreg[4 * i + 2] = v.z; qreg[4 * i + 3] = v.w;
        }
    }

    const float* Ktb = a.Kt + (size_t)b * DP * a.M_pad;
    const float* Vsb = a.Vs + (size_t)b * a.M_pad * DP;
    const int nt = a.M_pad / KT;

    float4 kreg[2], vreg[2];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;                   // 512 float4 per tile
            const int row = f >> 3, c4 = f & 7;
            kreg[i] = *reinterpret_cast<const float4*>(Ktb + (size_t)row * a.M_pad + t * KT + c4 * 4);
            vreg[i] = *reinterpret_cast<const float4*>(Vsb + (size_t)t * KT * DP + f * 4);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            *reinterpret_cast<float4*>(&ks[buf][f * 4]) = kreg[i];   // row*32 + c4*4 == f*4
            *reinterpret_cast<float4*>(&vs[buf][f * 4]) = vreg[i];
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < nt;
        if (more) load_tile(t + 1);

        // ---- S^T = K Q^T ----  (A fragments read from LDS four steps ahead of their MFMAs)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kb = &ks[buf][h * KT + l31];
        {
            float af[2][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) af[0][u] = kb[(2 * u) * KT];
#pragma unroll
            for (int g4 = 0; g4 < 8; ++g4) {
                const int cur = g4 & 1;
                if (g4 + 1 < 8) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) af[cur ^ 1][u] = kb[(2 * (4 * g4 + 4 + u)) * KT];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][u], qreg[4 * g4 + u], s, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
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
        // exp(x) = 2^(x*log2 e) on the hardware exp2 (v_exp_f32): only keys within a few units of the
        // row maximum carry weight, and there (s - m) is exact, so the result is good to ~2 ulp
        constexpr float kLog2e = 1.4426950408889634f;
        const float mb = m_new * kLog2e;
        const float alpha = __builtin_amdgcn_exp2f(m_run * kLog2e - mb);
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

        // ---- O^T += V^T P^T ----  (V fragments two key-steps ahead)
        const float* vb = &vs[buf][l31];
        {
            float vf[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                vf[0][2 * u] = vb[acc_row(u, h) * DP];
                vf[0][2 * u + 1] = vb[acc_row(u, h) * DP + 32];
            }
#pragma unroll
            for (int g2 = 0; g2 < 8; ++g2) {
                const int cur = g2 & 1;
                if (g2 + 1 < 8) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        vf[cur ^ 1][2 * u] = vb[acc_row(2 * g2 + 2 + u, h) * DP];
                        vf[cur ^ 1][2 * u + 1] = vb[acc_row(2 * g2 + 2 + u, h) * DP + 32];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[cur][2 * u], s[2 * g2 + u], o0, 0, 0, 0);
                    o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[cur][2 * u + 1], s[2 * g2 + u], o1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
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
        if (h == 0) a.save_lse[(size_t)b * a.P_pad + q] = m_run * 1.4426950408889634f + log2f(l_run);
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
    float *Qs, *Kt, *Vs;
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
        return (float*)p;
    };
    w.Qs = take((size_t)batch * w.P_pad * DP * 4);
    w.Kt = take((size_t)batch * DP * w.M_pad * 4);
    w.Vs = take((size_t)batch * w.M_pad * DP * 4);
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
    { CT_PROF("ctx_project_kernel", st); hipLaunchKernelGGL(ctx_project_kernel, dim3(w.P_pad / 64, batch), blk, 0, st, conf, num_priors,
                       w.P_pad, d, prm->theta_w, prm->theta_b, w.Qs, none, none, none, 0); }
    CT_LAUNCH_CHECK("ctx_project_kernel(theta)");
    { CT_PROF("ctx_project_kernel", st); hipLaunchKernelGGL(ctx_project_kernel, dim3((w.M_pad + 63) / 64, batch), blk, 0, st, pool, num_ctx,
                       w.M_pad, d, prm->phi_w, prm->phi_b, none, w.Kt, none, none, 0); }
    CT_LAUNCH_CHECK("ctx_project_kernel(phi)");
    { CT_PROF("ctx_project_kernel", st); hipLaunchKernelGGL(ctx_project_kernel, dim3((w.M_pad + 63) / 64, batch), blk, 0, st, pool, num_ctx,
                       w.M_pad, d, prm->g_w, prm->g_b, none, none, w.Vs, none, 0); }
    CT_LAUNCH_CHECK("ctx_project_kernel(g)");
    if (prm->fc_w) {
        { CT_PROF("ctx_project_kernel", st); hipLaunchKernelGGL(ctx_project_kernel, dim3((num_priors + 63) / 64, batch), blk, 0, st, conf,
                           num_priors, num_priors, d, prm->fc_w, prm->fc_b, none, none, none, out, ostride); }
        CT_LAUNCH_CHECK("ctx_project_kernel(fc_base)");
    }
    AttnArgs a{};
    a.Qs = w.Qs; a.Kt = w.Kt; a.Vs = w.Vs;
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
