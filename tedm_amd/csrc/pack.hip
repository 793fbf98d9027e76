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

__global__ void copy_kernel(const float* __restrict__ s, float* __restrict__ d, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = s[i];
}

}  // namespace

int pack_conv_weights(const float* w, int Cout, int Cin, int KH, int KW, void* wpk, int fp16, hipStream_t st) {
    if (Cin % 16) return TEDM_EINVAL;
    const int T = KH * KW;
    const long long CT = (Cout + 31) / 32;
    const long long total = CT * (Cin / 16) * T * 64;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       w, Cout, Cin, T, (uint4*)wpk, total, fp16);
    return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
}

int pack_conv_weights_dgrad(const float* w, int Cout, int Cin, int KH, int KW, int ci0, int nci, void* wpk, int fp16,
                            hipStream_t st) {
    if (Cout % 16 || ci0 < 0 || ci0 + nci > Cin) return TEDM_EINVAL;
    const long long CT = (nci + 31) / 32;
    const long long total = CT * (Cout / 16) * KH * KW * 64;
    hipLaunchKernelGGL(conv_pack_dgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       w, Cout, Cin, KH, KW, ci0, nci, (uint4*)wpk, total, fp16);
    return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
}

int copy_rows(const float* src, float* dst, long long n, hipStream_t st) {
    if (n <= 0) return TEDM_OK;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
    return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
}
