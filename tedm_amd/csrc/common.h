// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libtedm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define TEDM_DEVFN __device__ __forceinline__

// v * sigmoid(v).  v_rcp_f32 (1 ulp) instead of the IEEE divide: `a / b` expands to an 11-instruction div_scale / fma /
// div_fixup sequence, which tripled the VALU cost of every GroupNorm+SiLU prologue.
TEDM_DEVFN float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// Split 8 fp32 values into bf16 hi (round-to-nearest) and bf16 lo = bf16(v - hi): hi + lo carries
// ~16 significand bits, so A_hi*B_hi + A_lo*B_hi + A_hi*B_lo reproduces an fp32 product to ~2^-17.
TEDM_DEVFN void split8(const float (&v)[8], uint4& hi, uint4& lo) {
    bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 hj = (__bf16)v[j];
        h[j] = hj;
        l[j] = (__bf16)(v[j] - (float)hj);
    }
    hi = __builtin_bit_cast(uint4, h);
    lo = __builtin_bit_cast(uint4, l);
}

// fp16 variants: hi = fp16(v) (11 significand bits), lo = fp16(v - hi) (22 bits together)
TEDM_DEVFN void split8_f16(const float (&v)[8], uint4& hi, uint4& lo) {
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 hj = (_Float16)v[j];
        h[j] = hj;
        l[j] = (_Float16)(v[j] - (float)hj);
    }
    hi = __builtin_bit_cast(uint4, h);
    lo = __builtin_bit_cast(uint4, l);
}
TEDM_DEVFN uint4 cvt8_f16(const float (&v)[8]) {
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
    return __builtin_bit_cast(uint4, h);
}

TEDM_DEVFN float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
TEDM_DEVFN float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
// reduce over aligned groups of `width` lanes (power of two <= 64)
TEDM_DEVFN float group_sum(float v, int width) {
    for (int m = width >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// ---- conv_igemm.hip -------------------------------------------------------------------
// Prologue applied to the conv input while it is staged into LDS.
enum { PRO_NONE = 0, PRO_GN = 1, PRO_LN = 2 };
// Per-output-channel statistics the epilogue can emit (deterministic slabs, no atomics).
enum { STATS_NONE = 0, STATS_SUMSQ = 1, STATS_MAX = 2 };

struct ConvArgs {
    const float* x1; int C1;       // source 1, NHWC, C1 channels
    const float* x2; int C2;       // optional source 2 (virtual channel concat after x1)
    int B, H, W;                   // source spatial size (before the optional x2 nearest upsample)
    int Ho, Wo, Cout;
    const void* wpk;               // packed weights (pack.hip: conv_pack_kernel)
    const float* bias;             // [Cout] or null
    const float* pro_a;            // PRO_GN: a[B][Cin]           PRO_LN: g[Cin]
    const float* pro_b;            // PRO_GN: b[B][Cin]           PRO_LN: stats[B*H*W][2] = (rstd, mean*rstd)
    float* out;                    // NHWC (B,Ho,Wo,Cout)
    const float* add;              // optional addend, NHWC (B,Ho,Wo,Cout)
    const float* add_a;            // if non-null the addend is silu(add_a[b][c]*add + add_b[b][c])
    const float* add_b;
    float* stats;                  // STATS_SUMSQ: [B][E][Cout][2]   STATS_MAX: [B][E][Cout]
    int stats_mode;
    int tiles_x, tiles_y;          // spatial tiles per image
    int ny;                        // cout-tile groups (filled by the launcher)
    int xcd_order;                 // block-order policy (filled by the launcher)
    int tpw;                       // spatial tiles per workgroup (filled by the launcher)
    int nk16;                      // (C1+C2)/16
    int prec;                      // TEDM_PREC_*
    int prio_mode;                 // 0 none, 1: waves in even SIMD slots run at s_setprio 1 (filled by the launcher)
    unsigned long long* trace;     // diagnostic builds (-DTEDM_DIAG_TRACE) only: per-wave phase timestamps
};

int conv_launch(const ConvArgs& a, int KH, int stride, int ups, int pro, hipStream_t st);
int conv_stats_entries(int KH, int stride, int Cout, int Ho, int Wo);   // E of the stats slab
size_t conv_packed_bytes(int Cout, int Cin, int taps);
void conv_kernel_label(const ConvArgs& a, int KH, int stride, int ups, int pro, char* buf, int cap);

// ---- pack.hip -------------------------------------------------------------------------
int pack_conv_weights(const float* w, int Cout, int Cin, int KH, int KW, void* wpk, int fp16, hipStream_t st);
int pack_conv_weights_dgrad(const float* w, int Cout, int Cin, int KH, int KW, int ci0, int nci, void* wpk, int fp16,
                            hipStream_t st);
int copy_rows(const float* src, float* dst, long long n, hipStream_t st);
