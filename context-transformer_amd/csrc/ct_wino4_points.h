// libctdet: the 1-D transforms of Winograd F(4,3) / F(3,4) shared by ct_wino4.hip (forward, data gradient),
// ct_wino4_wgrad.hip (weight gradient) and the weight pre-transform (ct_wino_pack.h).
//
// Cook-Toom with the interpolation points 0, +-p, +-q, inf, p = 3/4, q = 3/2 (round 3; rounds 1-2 used the textbook
// 0, +-1, +-2, inf).  Why: the kernel's rounding error is that of the sequential fp32 channel sum IN THE TRANSFORM
// DOMAIN (measured by stage substitution, DESIGN.md: U, V and the output transform together are a fifth of it), and
// it scales with how much larger the transform-domain partial sums are than the outputs A^T recovers from them.
// A simulation of one 512-channel layer with post-ReLU inputs (fp32 emulation, numpy): 0,+-1,+-2: rms 2.7e-6 / max
// 7.6e-6 of the output range; 0,+-3/4,+-3/2: 1.3e-6 / 2.0e-6; 0,+-5/8,+-3/2: 1.2e-6 / 2.3e-6; F(2x2,3x3): 0.3e-6.
// Every constant below is a dyadic rational, exact in fp32; the forms cost one more multiply per bt6 and three per
// at4 than the p = 1, q = 2 forms.
//
//   B^T = rows of prod_{l != j} (x - p_l):            G[j] = [1, p_j, p_j^2] / N_j,  N_j = prod_{l != j} (p_j - p_l)
//     [p2q2   0   -(p2+q2)   0     1  0]                N_0 = p2 q2, N_{+-p} = 2 p2 (p2 - q2), N_{+-q} = 2 q2 (q2 - p2)
//     [0   -p q2   -q2       p     1  0]   (+p)        A^T = [1 1 1 1 1 0; 0 p -p q -q 0; 0 p2 p2 q2 q2 0; 0 p3 -p3 q3 -q3 1]
//     [0    p q2   -q2      -p     1  0]   (-p)
//     [0   -q p2   -p2       q     1  0]   (+q)
//     [0    q p2   -p2      -q     1  0]   (-q)
//     [0   p2q2     0    -(p2+q2)  0  1]   (inf)
#pragma once

namespace ctdet {
namespace w4 {

constexpr float P = 0.75f, Q = 1.5f;
constexpr float P2 = P * P, Q2 = Q * Q, P3 = P2 * P, Q3 = Q2 * Q;
constexpr float P2Q2 = P2 * Q2, SPQ = P2 + Q2;
constexpr double N0 = (double)P2 * Q2, NP = 2.0 * P2 * ((double)P2 - Q2), NQ = 2.0 * Q2 * ((double)Q2 - P2);

// x -> B^T x (also the row pass: V = (B^T d) B means B^T applied along the other index)
__device__ __forceinline__ void bt6(const float (&d)[6], float (&o)[6])
{
    const float a = fmaf(-Q2, d[2], d[4]);
    const float b = P * fmaf(-Q2, d[1], d[3]);
    const float c = fmaf(-P2, d[2], d[4]);
    const float e = Q * fmaf(-P2, d[1], d[3]);
    o[0] = fmaf(P2Q2, d[0], fmaf(-SPQ, d[2], d[4]));
    o[1] = a + b;
    o[2] = a - b;
    o[3] = c + e;
    o[4] = c - e;
    o[5] = fmaf(P2Q2, d[1], fmaf(-SPQ, d[3], d[5]));
}

// m -> A^T m
__device__ __forceinline__ void at4(const float (&m)[6], float (&y)[4])
{
    const float p = m[1] + m[2], n = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
    y[0] = m[0] + p + r;
    y[1] = fmaf(Q, s, P * n);
    y[2] = fmaf(Q2, r, P2 * p);
    y[3] = fmaf(Q3, s, P3 * n) + m[5];
}

// The same two transforms in double (ct_wino4s.hip: its transform kernels are memory-bound, so B^T d B and A^T M A are
// evaluated in double and rounded ONCE -- the fp32 chains above round after every operation).
__device__ __forceinline__ void bt6d(const double (&d)[6], double (&o)[6])
{
    const double a = d[4] - (double)Q2 * d[2];
    const double b = (double)P * (d[3] - (double)Q2 * d[1]);
    const double c = d[4] - (double)P2 * d[2];
    const double e = (double)Q * (d[3] - (double)P2 * d[1]);
    o[0] = (double)P2Q2 * d[0] - (double)SPQ * d[2] + d[4];
    o[1] = a + b;
    o[2] = a - b;
    o[3] = c + e;
    o[4] = c - e;
    o[5] = (double)P2Q2 * d[1] - (double)SPQ * d[3] + d[5];
}

__device__ __forceinline__ void at4d(const double (&m)[6], double (&y)[4])
{
    const double p = m[1] + m[2], n = m[1] - m[2], r = m[3] + m[4], s = m[3] - m[4];
    y[0] = m[0] + p + r;
    y[1] = (double)Q * s + (double)P * n;
    y[2] = (double)Q2 * r + (double)P2 * p;
    y[3] = (double)Q3 * s + (double)P3 * n + m[5];
}

// e -> A e   (A = (A^T)^T: rows [1 0 0 0], [1 +-p p2 +-p3], [1 +-q q2 +-q3], [0 0 0 1])
__device__ __forceinline__ void a6(const float (&e)[4], float (&o)[6])
{
    const float ep = fmaf(P2, e[2], e[0]), op = fmaf(P3, e[3], P * e[1]);
    const float eq = fmaf(Q2, e[2], e[0]), oq = fmaf(Q3, e[3], Q * e[1]);
    o[0] = e[0];
    o[1] = ep + op;
    o[2] = ep - op;
    o[3] = eq + oq;
    o[4] = eq - oq;
    o[5] = e[3];
}

// a -> G a for a filter column / row a = (a0, a1, a2); evaluated in double (the weight pre-transform runs once per
// parameter version, one thread per filter), so U = G g G^T is the correctly rounded transform of the fp32 filter
__device__ __forceinline__ void gmul6(double a0, double a1, double a2, double (&o)[6])
{
    const double ep = (a0 + (double)P2 * a2) * (1.0 / NP), op = ((double)P * a1) * (1.0 / NP);
    const double eq = (a0 + (double)Q2 * a2) * (1.0 / NQ), oq = ((double)Q * a1) * (1.0 / NQ);
    o[0] = a0 * (1.0 / N0);
    o[1] = ep + op;
    o[2] = ep - op;
    o[3] = eq + oq;
    o[4] = eq - oq;
    o[5] = a2;
}

// u -> G^T u for a 6-vector u (weight gradient: dw = G^T dU G), fp32 like the accumulators it reads
__device__ __forceinline__ void gt3(const float (&u)[6], float (&o)[3])
{
    constexpr float iN0 = (float)(1.0 / N0), iNP = (float)(1.0 / NP), iNQ = (float)(1.0 / NQ);
    const float sp = u[1] + u[2], dp = u[1] - u[2], sq = u[3] + u[4], dq = u[3] - u[4];
    o[0] = fmaf(iN0, u[0], fmaf(iNP, sp, iNQ * sq));
    o[1] = fmaf(P * iNP, dp, (Q * iNQ) * dq);
    o[2] = fmaf(P2 * iNP, sp, fmaf(Q2 * iNQ, sq, u[5]));
}

}  // namespace w4
}  // namespace ctdet
