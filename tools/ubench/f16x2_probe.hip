// Micro-benchmark behind the "f16x2" operand form: an fp32 value as TWO binary16 pieces
// (x = hi + lo, hi = rne16(x), lo = rne16(x - hi): 11 + 1 + 11 significant bits) and a product as THREE piece products
// (hi.hi, hi.lo, lo.hi) on v_mfma_f32_32x32x16_f16, against the bf16x3 form's six.  Asks the hardware four questions:
//   1. does the f16 MFMA honour SUBNORMAL inputs (the lo pieces of small values are subnormal)?
//   2. what does the pipe sustain on real data with f16 operands (the bf16 figure is power-limited: profiles/r04_mfma_power.txt)?
//   3. is the two-instruction-per-value split (v_cvt_pk_f16_f32 + v_fma_mix_f32) exact, i.e. identical to the host's rne split?
//   4. error against fp64 of a 32 x 32 x K GEMM: f16x2 / three products vs bf16x3 / six products, one and two accumulators.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f16x2_probe.hip -o /tmp/f16x2_probe && /tmp/f16x2_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2_f16(float x0, float x1, unsigned& hi, unsigned& lo)
{
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{x0, x1}, h2));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h), "v"(x1));
    hi = h;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{r0, r1}, h2));
}

__global__ void split_kernel(const float* in, unsigned* out, int n2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    unsigned a, b;
    split2_f16(in[2 * i], in[2 * i + 1], a, b);
    out[2 * i] = a;
    out[2 * i + 1] = b;
}

// one wave: D[32][32] = sum over steps of A-piece . B-piece products; operands pre-split on the host, [step][piece][lane][4 words]
template <int KIND>      // 0: bf16x3 six products, 1: f16x2 three products, 2: f16x2 four products
__global__ __launch_bounds__(64) void gemm_kernel(const i32x4* A, const i32x4* B, float* D, int steps, int dual)
{
    const int lane = threadIdx.x;
    constexpr int NP = KIND == 0 ? 3 : 2;
    f32x16 acc, acs;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acs[r] = 0.f; }
    for (int s = 0; s < steps; ++s) {
        i32x4 a[NP], b[NP];
        for (int p = 0; p < NP; ++p) {
            a[p] = A[(s * NP + p) * 64 + lane];
            b[p] = B[(s * NP + p) * 64 + lane];
        }
        if (KIND == 0) {
            constexpr int PA[6] = {1, 2, 0, 1, 0, 0}, PB[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                f32x16& d = (dual && t < 5) ? acs : acc;
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[PA[t]]), __builtin_bit_cast(bf16x8, b[PB[t]]), d, 0, 0, 0);
            }
        } else {
            // (lo, lo) first when asked for, then (lo, hi), (hi, lo), (hi, hi)
            constexpr int PA[4] = {1, 1, 0, 0}, PB[4] = {1, 0, 1, 0};
#pragma unroll
            for (int t = (KIND == 2 ? 0 : 1); t < 4; ++t) {
                f32x16& d = (dual && t < 3) ? acs : acc;
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[PA[t]]), __builtin_bit_cast(f16x8, b[PB[t]]), d, 0, 0, 0);
            }
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        D[row * 32 + col] = acc[r] + acs[r];
    }
}

template <int KIND>                  // 0: bf16 32x32x16, 1: f16 32x32x16
__global__ __launch_bounds__(256) void rate_kernel(const i32x4* __restrict__ ops, float* out, int iters)
{
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    i32x4 pa[4], pb[4];
    for (int j = 0; j < 4; ++j) {
        pa[j] = ops[(threadIdx.x * 8 + j) % 2048];
        pb[j] = ops[(threadIdx.x * 8 + 4 + j) % 2048];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (KIND == 0)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pa[j]), __builtin_bit_cast(bf16x8, pb[j]), acc[j], 0, 0, 0);
            else
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, pa[j]), __builtin_bit_cast(f16x8, pb[j]), acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double gauss()
{
    const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return std::sqrt(-2 * std::log(u)) * std::cos(6.2831853 * v);
}
static unsigned short f16_bits(float x) { _Float16 h = (_Float16)x; unsigned short b; memcpy(&b, &h, 2); return b; }
static float f16_val(unsigned short b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }
static unsigned f32_bits(float x) { unsigned b; memcpy(&b, &x, 4); return b; }
static float bits_f32(unsigned b) { float x; memcpy(&x, &b, 4); return x; }

int main()
{
    // ---- 1. subnormal inputs
    {
        std::vector<unsigned> ha(64 * 4), hb(64 * 4);
        const unsigned short sub = f16_bits(9.5367431640625e-07f);       // 2^-20: subnormal in binary16
        const unsigned short big = f16_bits(1024.f);
        for (auto& w : ha) w = sub | ((unsigned)sub << 16);
        for (auto& w : hb) w = big | ((unsigned)big << 16);
        i32x4 *A, *B; float* D;
        hipMalloc(&A, 64 * 16); hipMalloc(&B, 64 * 16); hipMalloc(&D, 1024 * 4);
        // pieces: [step 1][piece 2][lane 64]: use the same fragment for hi and lo slots of KIND 1 -> (lo,hi) + (hi,lo) + (hi,hi) = 3 x
        std::vector<unsigned> a2(2 * 64 * 4), b2(2 * 64 * 4);
        for (int p = 0; p < 2; ++p) { memcpy(&a2[p * 256], ha.data(), 1024); memcpy(&b2[p * 256], hb.data(), 1024); }
        hipFree(A); hipFree(B);
        hipMalloc(&A, 2048); hipMalloc(&B, 2048);
        hipMemcpy(A, a2.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(B, b2.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((gemm_kernel<1>), dim3(1), dim3(64), 0, 0, A, B, D, 1, 0);
        float d0;
        hipMemcpy(&d0, D, 4, hipMemcpyDeviceToHost);
        printf("1. subnormal f16 input 2^-20 x 1024, 16 k, three products: expected %.8g, device %.8g  -> %s\n", 3 * 16 * 9.5367431640625e-07 * 1024, d0,
               d0 == 0.f ? "FLUSHED TO ZERO" : "subnormals honoured");
        hipFree(A); hipFree(B); hipFree(D);
    }
    // ---- 3. the split
    {
        const int n = 1 << 20;
        std::vector<float> x(n);
        srand(7);
        for (int i = 0; i < n; ++i) {
            const int kind = i & 7;
            double v = gauss();
            if (kind == 1) v *= 1e-3; else if (kind == 2) v *= 1e-6; else if (kind == 3) v *= 3e4; else if (kind == 4) v *= std::exp(8 * gauss());
            if (std::fabs(v) > 60000) v = 60000;
            x[i] = (float)v;
        }
        x[0] = 0.f; x[1] = -0.f; x[2] = 65504.f; x[3] = 6.1e-5f; x[4] = 5.96e-8f; x[5] = 1e-10f;
        float* dx; unsigned* dout;
        hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(split_kernel, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n / 2);
        std::vector<unsigned> o(n);
        hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
        long bad = 0; double worst = 0;
        for (int i = 0; i < n / 2; ++i)
            for (int e = 0; e < 2; ++e) {
                const float v = x[2 * i + e];
                const unsigned short hi = (unsigned short)(o[2 * i] >> (16 * e)), lo = (unsigned short)(o[2 * i + 1] >> (16 * e));
                const unsigned short hh = f16_bits(v);
                const unsigned short hl = f16_bits(v - f16_val(hh));
                if (hi != hh || lo != hl) { if (bad < 5) printf("   mismatch x=%g: device %04x %04x host %04x %04x\n", v, hi, lo, hh, hl); ++bad; }
                const double rep = (double)f16_val(hi) + (double)f16_val(lo);
                if (v != 0.f && std::fabs(v) > 1e-3) worst = std::fmax(worst, std::fabs(rep - v) / std::fabs(v));
            }
        printf("3. split of %d values (v_cvt_pk_f16_f32 + v_fma_mix_f32): %ld differ from the host's rne split; worst |hi + lo - x| / |x| for |x| > 1e-3: %.3g (2^-22 = %.3g)\n",
               n, bad, worst, std::ldexp(1.0, -22));
        hipFree(dx); hipFree(dout);
    }
    // ---- 4. GEMM error
    {
        const int K = 512, steps = K / 16;
        srand(11);
        std::vector<float> Am(32 * K), Bm(K * 32);
        for (auto& v : Am) v = (float)(gauss() * 0.05);
        for (auto& v : Bm) { const double g = gauss(); v = (float)(g > 0 ? g * std::exp(gauss()) : (rand() % 4 ? 0.0 : g)); }
        std::vector<double> ref(1024, 0.0);
        double range = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0;
                for (int k = 0; k < K; ++k) s += (double)Am[i * K + k] * Bm[k * 32 + j];
                ref[i * 32 + j] = s;
                range = std::fmax(range, std::fabs(s));
            }
        float amax = 0, bmax = 0;
        for (float v : Am) amax = std::fmax(amax, std::fabs(v));
        for (float v : Bm) bmax = std::fmax(bmax, std::fabs(v));
        const float sA = std::ldexp(1.f, 14 - (int)std::ceil(std::log2(amax))), sB = std::ldexp(1.f, 14 - (int)std::ceil(std::log2(bmax)));
        i32x4 *A, *B; float* D;
        hipMalloc(&A, steps * 3 * 64 * 16); hipMalloc(&B, steps * 3 * 64 * 16); hipMalloc(&D, 4096);
        auto run = [&](int kind, int dual, float extra, const char* name) {
            const int np = kind == 0 ? 3 : 2;
            std::vector<unsigned short> ha((size_t)steps * np * 64 * 8), hb(ha.size());
            for (int s = 0; s < steps; ++s)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int k = s * 16 + 8 * (l >> 5) + e, rc = l & 31;
                        float a = Am[rc * K + k], b = Bm[k * 32 + rc];
                        unsigned short pa[3], pb[3];
                        if (kind == 0) {
                            auto sp = [](float v, unsigned short* p) {
                                const unsigned h = f32_bits(v) & 0xFFFF0000u; const float r = v - bits_f32(h);
                                const unsigned m = f32_bits(r) & 0xFFFF0000u; const unsigned lo = f32_bits(r - bits_f32(m));
                                p[0] = h >> 16; p[1] = m >> 16; p[2] = lo >> 16;
                            };
                            sp(a, pa); sp(b, pb);
                        } else {
                            a *= sA; b *= sB * extra;
                            pa[0] = f16_bits(a); pa[1] = f16_bits(a - f16_val(pa[0]));
                            pb[0] = f16_bits(b); pb[1] = f16_bits(b - f16_val(pb[0]));
                        }
                        for (int p = 0; p < np; ++p) {
                            ha[(((size_t)s * np + p) * 64 + l) * 8 + e] = pa[p];
                            hb[(((size_t)s * np + p) * 64 + l) * 8 + e] = pb[p];
                        }
                    }
            hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
            hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
            if (kind == 0) hipLaunchKernelGGL((gemm_kernel<0>), dim3(1), dim3(64), 0, 0, A, B, D, steps, dual);
            else if (kind == 1) hipLaunchKernelGGL((gemm_kernel<1>), dim3(1), dim3(64), 0, 0, A, B, D, steps, dual);
            else hipLaunchKernelGGL((gemm_kernel<2>), dim3(1), dim3(64), 0, 0, A, B, D, steps, dual);
            std::vector<float> d(1024);
            hipMemcpy(d.data(), D, 4096, hipMemcpyDeviceToHost);
            const double inv = kind == 0 ? 1.0 : 1.0 / ((double)sA * sB * extra);
            double se = 0, mx = 0;
            for (int i = 0; i < 1024; ++i) { const double e = d[i] * inv - ref[i]; se += e * e; mx = std::fmax(mx, std::fabs(e)); }
            printf("   %-44s rms %.3e  max %.3e of the output range\n", name, std::sqrt(se / 1024) / range, mx / range);
        };
        printf("4. 32 x 32 x %d GEMM on the device against fp64 (weights ~ N(0, 0.05), activations post-ReLU log-normal; f16 scales 2^%d, 2^%d)\n", K,
               (int)std::log2(sA), (int)std::log2(sB));
        // fp32 reference chain on the host for scale
        {
            double se = 0, mx = 0;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    float s = 0.f;
                    for (int k = 0; k < K; ++k) s = std::fmaf(Am[i * K + k], Bm[k * 32 + j], s);
                    const double e = s - ref[i * 32 + j]; se += e * e; mx = std::fmax(mx, std::fabs(e));
                }
            printf("   %-44s rms %.3e  max %.3e of the output range\n", "host fp32 fma chain (sequential k)", std::sqrt(se / 1024) / range, mx / range);
        }
        run(0, 0, 1.f, "bf16x3 six products, one accumulator");
        run(0, 1, 1.f, "bf16x3 six products, two accumulators");
        run(1, 0, 1.f, "f16x2 three products, one accumulator");
        run(1, 1, 1.f, "f16x2 three products, two accumulators");
        run(2, 1, 1.f, "f16x2 four products, two accumulators");
        run(1, 0, 1.f / 64, "f16x2 three products, B scaled 64x too small");
        run(1, 0, 1.f / 1024, "f16x2 three products, B scaled 1024x too small");
        hipFree(A); hipFree(B); hipFree(D);
    }
    // ---- 2. sustained rate
    {
        float* out; i32x4* ops;
        hipMalloc(&out, 512 * 256 * sizeof(float));
        hipMalloc(&ops, 2048 * sizeof(i32x4));
        std::vector<unsigned> h(2048 * 4);
        srand(1);
        for (int kind = 0; kind < 2; ++kind)
            for (int data = 0; data < 3; ++data) {           // 0: zeros, 1: gaussian "hi" pieces, 2: f16 only: gaussian lo pieces (2^-11 of the hi scale)
                if (kind == 0 && data == 2) continue;
                for (auto& w : h) {
                    if (!data) { w = 0; continue; }
                    const float sc = kind == 1 ? (data == 1 ? 256.f : 0.125f) : 1.f;
                    const float f0 = (float)gauss() * sc, f1 = (float)gauss() * sc;
                    if (kind == 0) w = (f32_bits(f0) >> 16) | (f32_bits(f1) & 0xFFFF0000u);
                    else w = f16_bits(f0) | ((unsigned)f16_bits(f1) << 16);
                }
                hipMemcpy(ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
                for (int iters : {20000, 200000}) {
                    hipEvent_t e0, e1;
                    hipEventCreate(&e0); hipEventCreate(&e1);
                    if (kind == 0) hipLaunchKernelGGL((rate_kernel<0>), dim3(512), dim3(256), 0, 0, ops, out, 100);
                    else hipLaunchKernelGGL((rate_kernel<1>), dim3(512), dim3(256), 0, 0, ops, out, 100);
                    hipDeviceSynchronize();
                    hipEventRecord(e0);
                    if (kind == 0) hipLaunchKernelGGL((rate_kernel<0>), dim3(512), dim3(256), 0, 0, ops, out, iters);
                    else hipLaunchKernelGGL((rate_kernel<1>), dim3(512), dim3(256), 0, 0, ops, out, iters);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    const double flop = 512.0 * 4 * iters * 4.0 * 32.0 * 32 * 16 * 2;
                    printf("2. %s 32x32x16, %s operands, %6d x 4 MFMAs per wave: %8.2f ms = %7.1f TFLOP/s\n", kind ? "f16 " : "bf16",
                           data == 0 ? "zero      " : data == 1 ? "gaussian  " : "small (lo)", iters, ms, flop / ms / 1e9);
                }
            }
    }
    return 0;
}
