// mfma_probe.hip -- diagnostic (not product): pins the gfx950 facts the f16+mxfp4 convolution arithmetic relies on.
//   1. operand lane/nibble map and E8M0 scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (e2m1) operands
//   2. semantics of v_cvt_scalef32_pk_fp4_f32 (direction of the scale, rounding, saturation)
//   3. issue rates: f16 32x32x16, scaled fp4 32x32x64, and the 2:1 mix the convolution issues
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o gpurun_out/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static const float FP4[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static float fp4_dec(int c) { float v = FP4[c & 7]; return (c & 8) ? -v : v; }

// ---- 1. layout ------------------------------------------------------------------------------
__global__ void mfma_fp4_kernel(const uint4* a, const uint4* b, const unsigned* sa, const unsigned* sb, float* out) {
    const int l = threadIdx.x;
    i32x8 av = {}, bv = {};
    const uint4 aq = a[l], bq = b[l];
    av[0] = aq.x; av[1] = aq.y; av[2] = aq.z; av[3] = aq.w;
    bv[0] = bq.x; bv[1] = bq.y; bv[2] = bq.z; bv[3] = bq.w;
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, (int)sa[l], 0, (int)sb[l]);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
    f32x16 d = {};
    d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, d, 4, 4, 2, (int)sa[l], 1, (int)sb[l]);
    for (int i = 0; i < 16; ++i) out[1024 + l * 16 + i] = d[i];
}

// ---- 2. conversion --------------------------------------------------------------------------
__global__ void cvt_kernel(const float* v, const float* s, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        unsigned u = 0xdead0000u;
        u = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(u, v[2 * i], v[2 * i + 1], s[i], 0);
        unsigned w = 0x12345678u;
        w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, v[2 * i], v[2 * i + 1], s[i], 2);
        out[2 * i] = u; out[2 * i + 1] = w;
    }
}

// ---- 3. rates ---------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(const uint4* src, float* out, int iters, unsigned long long* cyc) {
    const int l = threadIdx.x & 63;
    uint4 q0 = src[l], q1 = src[64 + l], q2 = src[128 + l], q3 = src[192 + l];
    const f16x8 ah = __builtin_bit_cast(f16x8, q0), bh = __builtin_bit_cast(f16x8, q1);
    i32x8 a4 = {}, b4 = {};
    a4[0] = q2.x; a4[1] = q2.y; a4[2] = q2.z; a4[3] = q2.w;
    b4[0] = q3.x; b4[1] = q3.y; b4[2] = q3.z; b4[3] = q3.w;
    const int sc = 127 | (126 << 8);
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4 e0 = {}, e1 = {}, e2 = {}, e3 = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // 12 x f16 32x32x16
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
            }
        } else if (MODE == 1) {     // 12 x scaled fp4 32x32x64
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, c0, 4, 4, 0, sc, 0, sc);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, c1, 4, 4, 0, sc, 0, sc);
                c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, c2, 4, 4, 0, sc, 0, sc);
                c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, c3, 4, 4, 0, sc, 0, sc);
            }
        } else if (MODE == 2) {     // the convolution's mix: (f16, f16, fp4) x 4 accumulators
#pragma unroll
            for (int j = 0; j < 1; ++j) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, c0, 4, 4, 0, sc, 0, sc);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, c1, 4, 4, 0, sc, 0, sc);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, c2, 4, 4, 0, sc, 0, sc);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, c3, 4, 4, 0, sc, 0, sc);
            }
        } else if (MODE == 3) {     // 12 x f16 16x16x32
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, e1, 0, 0, 0);
                e2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, e2, 0, 0, 0);
                e3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, e3, 0, 0, 0);
            }
        } else if (MODE == 4) {     // 12 x scaled fp4 16x16x128
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                e0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a4, b4, e0, 4, 4, 0, sc, 0, sc);
                e1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a4, b4, e1, 4, 4, 0, sc, 0, sc);
                e2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a4, b4, e2, 4, 4, 0, sc, 0, sc);
                e3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a4, b4, e3, 4, 4, 0, sc, 0, sc);
            }
        } else if (MODE == 5) {     // 12 x bf16x3-style f16 (reference: what the current kernel issues per 16 channels x3)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
            }
        } else if (MODE == 6) {     // fp8 scaled 32x32x64 (e4m3), for the rate table
            i32x8 a8 = {}, b8 = {};
            a8[0] = q2.x; a8[1] = q2.y; a8[2] = q2.z; a8[3] = q2.w; a8[4] = q0.x; a8[5] = q0.y; a8[6] = q0.z; a8[7] = q0.w;
            b8[0] = q3.x; b8[1] = q3.y; b8[2] = q3.z; b8[3] = q3.w; b8[4] = q1.x; b8[5] = q1.y; b8[6] = q1.z; b8[7] = q1.w;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c0, 0, 0, 0, sc, 0, sc);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c1, 0, 0, 0, sc, 0, sc);
                c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c2, 0, 0, 0, sc, 0, sc);
                c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c3, 0, 0, 0, sc, 0, sc);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; ++i) s += e0[i] + e1[i] + e2[i] + e3[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}

template <int MODE>
static void run_rate(const char* name, const uint4* src, float* out, unsigned long long* cyc, double flop_per_iter, int waves_per_simd) {
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int threads = 256 * waves_per_simd, blocks = 256;
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, src, out, 200, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, src, out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc + MODE, 8, hipMemcpyDeviceToHost));
    const double waves = (double)blocks * threads / 64;
    printf("rate %-34s waves/SIMD %d: %8.3f ms  %8.1f TFLOP/s(eq)  %.1f cyc/iter(12 mfma) -> clock %.2f GHz\n", name, waves_per_simd, ms,
           flop_per_iter * iters * waves / (ms * 1e-3) / 1e12, (double)c / iters, (double)c / (ms * 1e-3) / 1e9);
}

int main() {
    srand(1);
    // ---- layout test ----
    std::vector<float> A(32 * 64), B(64 * 32);
    std::vector<int> ac(32 * 64), bc(64 * 32);
    for (int i = 0; i < 32 * 64; ++i) { ac[i] = rand() & 15; A[i] = fp4_dec(ac[i]); bc[i] = rand() & 15; B[i] = fp4_dec(bc[i]); }
    // hypothesis: lane l holds row/col (l&31), k = 32*(l>>5) + n, nibble n of the 128-bit operand (low bits first)
    std::vector<uint32_t> ha(64 * 4, 0), hb(64 * 4, 0), hsa(64), hsb(64);
    std::vector<int> ea(64), eb(64), ea2(64), eb1(64);
    for (int l = 0; l < 64; ++l) {
        const int r = l & 31, hh = l >> 5;
        for (int n = 0; n < 32; ++n) {
            const int k = 32 * hh + n;
            ha[l * 4 + n / 8] |= (uint32_t)ac[r * 64 + k] << (4 * (n & 7));        // A[r][k]
            hb[l * 4 + n / 8] |= (uint32_t)bc[k * 32 + r] << (4 * (n & 7));        // B[k][r]
        }
        ea[l] = 120 + rand() % 14; eb[l] = 120 + rand() % 14; ea2[l] = 120 + rand() % 14; eb1[l] = 120 + rand() % 14;
        hsa[l] = ea[l] | (0x55u << 8) | ((uint32_t)ea2[l] << 16) | (0x33u << 24);
        hsb[l] = eb[l] | ((uint32_t)eb1[l] << 8) | (0x77u << 16) | (0x11u << 24);
    }
    uint4 *da, *db; unsigned *dsa, *dsb; float* dout;
    CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dout, 8192));
    CK(hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_fp4_kernel, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
    std::vector<float> out(2048);
    CK(hipMemcpy(out.data(), dout, 8192, hipMemcpyDeviceToHost));
    for (int variant = 0; variant < 2; ++variant) {
        int bad = 0; double maxd = 0;
        for (int l = 0; l < 64; ++l)
            for (int reg = 0; reg < 16; ++reg) {
                const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);      // C/D map (guide section 3)
                double ref = 0;
                for (int k = 0; k < 64; ++k) {
                    const int kb = k >> 5;
                    const int sa_e = variant == 0 ? ea[row + 32 * kb] : ea2[row + 32 * kb];
                    const int sb_e = variant == 0 ? eb[col + 32 * kb] : eb1[col + 32 * kb];
                    ref += (double)A[row * 64 + k] * B[k * 32 + col] * ldexp(1.0, sa_e - 127) * ldexp(1.0, sb_e - 127);
                }
                const double d = fabs(ref - out[variant * 1024 + l * 16 + reg]);
                if (d > 1e-6 * fabs(ref) + 1e-12) ++bad;
                if (d > maxd) maxd = d;
            }
        printf("layout fp4 32x32x64 (opsel variant %d): %d / 1024 mismatches, max |diff| %.3g  [%s]\n", variant, bad, maxd, bad ? "FAIL" : "OK");
    }
    printf("  sample out[0..3] = %g %g %g %g\n", out[0], out[1], out[2], out[3]);

    // ---- conversion test ----
    {
        const float vals[] = {0.f, 0.2f, 0.25f, 0.3f, 0.5f, 0.74f, 0.75f, 0.76f, 1.f, 1.25f, 1.26f, 1.75f, 2.5f, 3.5f, 5.f, 5.1f, 6.f, 7.f, 100.f, -1.f, -0.25f, -7.f, 1e-3f, -3.f};
        const float scs[] = {1.f, 2.f, 0.5f, 0.25f, 3.f, 1024.f};
        std::vector<float> v, s;
        const int nv = sizeof(vals) / 4, ns = sizeof(scs) / 4;
        for (int j = 0; j < ns; ++j) for (int i = 0; i < nv / 2; ++i) { v.push_back(vals[2 * i]); v.push_back(vals[2 * i + 1]); s.push_back(scs[j]); }
        const int n = (int)s.size();
        float *dv, *ds; unsigned* du;
        CK(hipMalloc(&dv, v.size() * 4)); CK(hipMalloc(&ds, s.size() * 4)); CK(hipMalloc(&du, n * 8));
        CK(hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, s.data(), s.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(256), 0, 0, dv, ds, du, n);
        std::vector<unsigned> u(2 * n);
        CK(hipMemcpy(u.data(), du, n * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            printf("cvt_pk_fp4 v=(%g,%g) scale=%g -> sel0 %08x (lo %g hi %g)  sel2 %08x\n", v[2 * i], v[2 * i + 1], s[i], u[2 * i],
                   fp4_dec(u[2 * i] & 15), fp4_dec((u[2 * i] >> 4) & 15), u[2 * i + 1]);
    }

    // ---- rates ----
    {
        std::vector<uint32_t> h(256 * 4);
        for (auto& x : h) {
            // random f16 pairs in [-1,1) (also a valid random fp4/fp8 bit soup)
            _Float16 a = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1), b = (_Float16)((rand() / (float)RAND_MAX) * 2 - 1);
            uint16_t ua, ub; memcpy(&ua, &a, 2); memcpy(&ub, &b, 2);
            x = ua | ((uint32_t)ub << 16);
        }
        uint4* src; float* o; unsigned long long* cyc;
        CK(hipMalloc(&src, 4096)); CK(hipMalloc(&o, 256 * 512 * 4)); CK(hipMalloc(&cyc, 64));
        CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
        const double f16 = 2.0 * 32 * 32 * 16, f4 = 2.0 * 32 * 32 * 64;
        for (int w = 1; w <= 2; ++w) {
            run_rate<0>("f16 32x32x16", src, o, cyc, 12 * f16, w);
            run_rate<1>("mxfp4 32x32x64", src, o, cyc, 12 * f4, w);
            run_rate<2>("conv mix 2 f16 + 1 mxfp4 (=32ch)", src, o, cyc, 4 * 2 * f16, w);   // algorithmic: 32 channels per triple
            run_rate<3>("f16 16x16x32", src, o, cyc, 12 * 2.0 * 16 * 16 * 32, w);
            run_rate<4>("mxfp4 16x16x128", src, o, cyc, 12 * 2.0 * 16 * 16 * 128, w);
            run_rate<6>("mxfp8 32x32x64", src, o, cyc, 12 * f4, w);
        }
    }
    return 0;
}
