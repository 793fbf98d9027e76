// pack.hip -- one-time (per weight update) repacking of reference-layout fp32 parameters into the
// layouts the kernels consume.  Reference layouts: nn.Conv2d OIHW, nn.Linear (out,in)
// (models/unet_model.py state_dict).
#include "common.h"
#include "../../include/tedm_hip.h"

namespace {

// Conv weights -> MFMA B-fragment order for v_mfma_f32_32x32x16_bf16, bf16 hi/lo planes:
//   wpk[ct][k16][tap][plane][lane][j],  lane = (h<<5)|r :  W[cout = ct*32+r][cin = k16*16+8h+j][tap]
// (B[k = 8h+j][col = r], cdna guide section 3).  One 16-byte store per (fragment, lane, plane).
__global__ void conv_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int T, uint4* __restrict__ wpk,
                                 long long total, int fp16) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    long long f = i >> 6;
    const int tap = (int)(f % T); f /= T;
    const int nk16 = Cin / 16;
    const int k16 = (int)(f % nk16);
    const int ct = (int)(f / nk16);
    const int r = lane & 31, h = lane >> 5;
    const int cout = ct * 32 + r;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cin = k16 * 16 + 8 * h + j;
        v[j] = cout < Cout ? w[((size_t)cout * Cin + cin) * T + tap] : 0.f;
    }
    uint4 hi, lo;
    if (fp16) split8_f16(v, hi, lo); else split8(v, hi, lo);
    const size_t o = (((size_t)ct * nk16 + k16) * T + tap) * 2;
    wpk[(o + 0) * 64 + lane] = hi;
    wpk[(o + 1) * 64 + lane] = lo;
}

// Same fragment order for the DGRAD convolution: dX = conv(dY, W') with
//   W'[o = ci - ci0][i = co][ky'][kx'] = W[co][ci][KH-1-ky'][KW-1-kx'],  ci in [ci0, ci0+nci)
// (transposed channels, spatially flipped taps).  `Cout` of the original weight is the dgrad's Cin.
__global__ void conv_pack_dgrad_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int KW, int ci0, int nci,
                                       uint4* __restrict__ wpk, long long total, int fp16) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int T = KH * KW;
    const int lane = (int)(i & 63);
    long long f = i >> 6;
    const int tap = (int)(f % T); f /= T;
    const int nk16 = Cout / 16;                       // dgrad reduces over the original output channels
    const int k16 = (int)(f % nk16);
    const int ct = (int)(f / nk16);
    const int r = lane & 31, h = lane >> 5;
    const int o = ct * 32 + r;                        // dgrad output channel = original input channel - ci0
    const int ky = tap / KW, kx = tap - ky * KW;
    const int otap = (KH - 1 - ky) * KW + (KW - 1 - kx);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int co = k16 * 16 + 8 * h + j;
        v[j] = o < nci ? w[((size_t)co * Cin + ci0 + o) * T + otap] : 0.f;
    }
    uint4 hi, lo;
    if (fp16) split8_f16(v, hi, lo); else split8(v, hi, lo);
    const size_t q = (((size_t)ct * nk16 + k16) * T + tap) * 2;
    wpk[(q + 0) * 64 + lane] = hi;
    wpk[(q + 1) * 64 + lane] = lo;
}

// TEDM_PREC_F16F4 (common.h): per (cout tile, 32-channel chunk, tap) three 1-KB fragments
//   f = 0,1: fp16 B fragments of v_mfma_f32_32x32x16_f16 for channels [0,16) / [16,32) of the chunk
//   f = 2  : fp4 B fragment of v_mfma_scale_f32_32x32x64_f8f6f4: lanes 0-31 (k-slots 0-31, multiplied with the
//            activations' fp4 RESIDUAL block) hold fp4(W), lanes 32-63 (k-slots 32-63, multiplied with the activations'
//            fp4 COPY block) hold fp4(W - fp16(W))
// followed, after all fragments, by the E8M0 scale bytes: one dword per (cout tile, chunk, kx, lane), byte ky = scale of
// this lane's block for tap (ky, kx) -- the MFMA's op_sel picks the byte, so one register serves a whole tap column.
template <typename F>
__device__ void pack_f16f4_lane(F fetch, int lane, int KH, int KW, int tap, size_t frag0, size_t sdw, uint4* wpk, unsigned char* sc) {
    const int h = lane >> 5;
    float v[32];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { v[i] = fetch(i); amax = fmaxf(amax, fabsf(v[i])); }
    unsigned e_v4, e_lo;
    f4_scales(amax, e_v4, e_lo);
    uint4 q;
    unsigned* qw = (unsigned*)&q;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float x[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = v[g * 8 + j];
        const uint4 hi = split8_f16_res(x, lo);
        // fp16 fragment element j of lane half hh of k-step ks is channel ks*16 + 8*hh + j: group g = 2*ks + hh
        if ((g & 1) == h) wpk[(frag0 + (g >> 1)) * 64 + lane] = hi;
        qw[g] = h == 0 ? cvt8_fp4(x, e_v4) : cvt8_fp4(lo, e_lo);
    }
    wpk[(frag0 + 2) * 64 + lane] = q;
    const int ky = tap / KW;
    sc[(sdw * 64 + lane) * 4 + ky] = (unsigned char)(h == 0 ? e_v4 : e_lo);
    (void)KH;
}

__global__ void conv_pack_f16f4_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int KW, uint4* __restrict__ wpk,
                                       long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int T = KH * KW;
    const int lane = (int)(i & 63);
    long long f = i >> 6;
    const int tap = (int)(f % T); f /= T;
    const int NC = Cin / 32;
    const int c = (int)(f % NC);
    const int ct = (int)(f / NC);
    const int cout = ct * 32 + (lane & 31);
    const long long CT = (Cout + 31) / 32;
    unsigned char* sc = (unsigned char*)(wpk + (size_t)CT * NC * T * 3 * 64);
    const int kx = tap % KW;
    pack_f16f4_lane([&](int k) { return cout < Cout ? w[((size_t)cout * Cin + c * 32 + k) * T + tap] : 0.f; },
                    lane, KH, KW, tap, (((size_t)ct * NC + c) * T + tap) * 3, ((size_t)ct * NC + c) * KW + kx, wpk, sc);
}

__global__ void conv_pack_dgrad_f16f4_kernel(const float* __restrict__ w, int Cout, int Cin, int KH, int KW, int ci0, int nci,
                                             uint4* __restrict__ wpk, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int T = KH * KW;
    const int lane = (int)(i & 63);
    long long f = i >> 6;
    const int tap = (int)(f % T); f /= T;
    const int NC = Cout / 32;                          // dgrad reduces over the original output channels
    const int c = (int)(f % NC);
    const int ct = (int)(f / NC);
    const int o = ct * 32 + (lane & 31);
    const int ky = tap / KW, kx = tap - ky * KW;
    const int otap = (KH - 1 - ky) * KW + (KW - 1 - kx);
    const long long CT = (nci + 31) / 32;
    unsigned char* sc = (unsigned char*)(wpk + (size_t)CT * NC * T * 3 * 64);
    pack_f16f4_lane([&](int k) { return o < nci ? w[((size_t)(c * 32 + k) * Cin + ci0 + o) * T + otap] : 0.f; },
                    lane, KH, KW, tap, (((size_t)ct * NC + c) * T + tap) * 3, ((size_t)ct * NC + c) * KW + kx, wpk, sc);
}

// W'[a][b][co][ci][dy][dx] = sum of the 3x3 taps (ky, kx) whose source pixel under nearest-x2 upsampling is (dy, dx) of the
// parity's 2x2 footprint: rows a = 0: {ky 0} | {ky 1, 2};  a = 1: {ky 0, 1} | {ky 2}  (same for columns).
__global__ void up2_weights_kernel(const float* __restrict__ w3, float* __restrict__ w2, long long n /* Cout*Cin */) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float w[3][3];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k / 3][k % 3] = w3[i * 9 + k];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    float s = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int gy = (a + ky - 1 + 2) / 2 - 1 + (1 - a);    // source row offset (0 or 1) inside the footprint
                            const int gx = (b + kx - 1 + 2) / 2 - 1 + (1 - b);
                            if (gy == dy && gx == dx) s += w[ky][kx];
                        }
                    w2[(((size_t)(a * 2 + b) * n + i) * 2 + dy) * 2 + dx] = s;
                }
}

__global__ void copy_kernel(const float* __restrict__ s, float* __restrict__ d, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = s[i];
}

}  // namespace

int pack_conv_weights(const float* w, int Cout, int Cin, int KH, int KW, void* wpk, int enc, hipStream_t st) {
    if (Cin % 16) return TEDM_EINVAL;
    const int T = KH * KW;
    const long long CT = (Cout + 31) / 32;
    if (enc == ENC_F16F4) {
        if (Cin % 32 || KH > 4) return TEDM_EINVAL;
        const long long total = CT * (Cin / 32) * T * 64;
        hipLaunchKernelGGL(conv_pack_f16f4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           w, Cout, Cin, KH, KW, (uint4*)wpk, total);
        return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
    }
    const int fp16 = enc == ENC_F16;
    const long long total = CT * (Cin / 16) * T * 64;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       w, Cout, Cin, T, (uint4*)wpk, total, fp16);
    return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
}

int pack_conv_weights_dgrad(const float* w, int Cout, int Cin, int KH, int KW, int ci0, int nci, void* wpk, int enc,
                            hipStream_t st) {
    if (Cout % 16 || ci0 < 0 || ci0 + nci > Cin) return TEDM_EINVAL;
    const long long CT = (nci + 31) / 32;
    if (enc == ENC_F16F4) {
        if (Cout % 32 || KH > 4) return TEDM_EINVAL;
        const long long total = CT * (Cout / 32) * KH * KW * 64;
        hipLaunchKernelGGL(conv_pack_dgrad_f16f4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           w, Cout, Cin, KH, KW, ci0, nci, (uint4*)wpk, total);
        return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
    }
    const int fp16 = enc == ENC_F16;
    const long long total = CT * (Cout / 16) * KH * KW * 64;
    hipLaunchKernelGGL(conv_pack_dgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       w, Cout, Cin, KH, KW, ci0, nci, (uint4*)wpk, total, fp16);
    return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
}

int pack_conv_weights_up2(const float* w3, int Cout, int Cin, float* w2x2, void* wpk, size_t pack_stride, int enc, hipStream_t st) {
    const long long n = (long long)Cout * Cin;
    hipLaunchKernelGGL(up2_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w3, w2x2, n);
    if (hipGetLastError() != hipSuccess) return TEDM_EHIP;
    for (int pq = 0; pq < 4; ++pq) {
        const int rc = pack_conv_weights(w2x2 + (size_t)pq * n * 4, Cout, Cin, 2, 2, (char*)wpk + pq * pack_stride, enc, st);
        if (rc != TEDM_OK) return rc;
    }
    return TEDM_OK;
}

int copy_rows(const float* src, float* dst, long long n, hipStream_t st) {
    if (n <= 0) return TEDM_OK;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
    return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
}
