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
// The four operations are pinned (no contraction, no re-association): two inlined copies of a staging loop must round every
// element identically, or a tile staged by the loop prologue and the same tile staged inside the loop differ in the last bit
// -- and with 11-bit operands (TEDM_PREC_FP16) such a bit flips output roundings, i.e. results would depend on how tiles
// are grouped into workgroups (on the batch size).
TEDM_DEVFN float silu_f(float v) {
    float t, e, d, rcp, y;
    asm("v_mul_f32 %0, 0xbfb8aa3b, %1" : "=v"(t) : "v"(v));           // -log2(e) * v
    asm("v_exp_f32 %0, %1" : "=v"(e) : "v"(t));
    asm("v_add_f32 %0, 1.0, %1" : "=v"(d) : "v"(e));
    asm("v_rcp_f32 %0, %1" : "=v"(rcp) : "v"(d));
    asm("v_mul_f32 %0, %1, %2" : "=v"(y) : "v"(v), "v"(rcp));
    return y;
}

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

// ---- TEDM_PREC_F16F4 operand encoding --------------------------------------------------------------------
// A 32-channel block of an operand is carried as  fp16 hi (11 significand bits)  +  two MX-fp4 (e2m1, E8M0 block scale)
// blocks: `lo4` = fp4(v - hi) and `v4` = fp4(v).  One v_mfma_scale_f32_32x32x64_f8f6f4 then adds BOTH cross terms of
//   A*B ~= Ah*Bh + Al*B4 + A4*Bl
// in a single 32-cycle issue: its K = 64 is the concatenation [Al | A4] x [B4 ; Bl] (lanes 0-31 hold k-slots 0-31, lanes
// 32-63 k-slots 32-63, one scale byte per lane = per 32-slot block; layout pinned on hardware by tools/mfma_probe.hip).
// Per 32 channels: 2 f16 MFMAs + 1 fp4 MFMA = 96 matrix-pipe cycles instead of the 192 of bf16x3; eps error 2-2.6e-4
// against the fp32 reference (tests/precision_study.py), gate 1e-3.
typedef __attribute__((ext_vector_type(8))) int i32x8;

// max over the aligned group of 4 lanes of a NON-NEGATIVE value (DPP quad_perm, no LDS).  Compared as unsigned integers:
// the ordering of non-negative floats, without the canonicalising v_max x,x that fmaxf puts in front of every compare.
TEDM_DEVFN float quad_max(float v) {
    unsigned x = __builtin_bit_cast(unsigned, v);
    unsigned y = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false);   // [1,0,3,2]
    x = x > y ? x : y;
    y = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false);            // [2,3,0,1]
    x = x > y ? x : y;
    return __builtin_bit_cast(float, x);
}
// E8M0 scale bytes of a block whose largest magnitude is `amax`: the fp4 copy is scaled by 2^(floor(log2 amax) - 2)
// (OCP MX rule, e2m1 emax = 2, saturating), the fp4 residual by the fp16 half-ulp bound 2^(E-11) (2^-25 in fp16's
// subnormal range) with the same emax.
TEDM_DEVFN void f4_scales(float amax, unsigned& e_v4, unsigned& e_lo) {
    const int E = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);
    int a = E - 2, l = (E - 11 > 102 ? E - 11 : 102) - 2;
    e_v4 = (unsigned)(a < 1 ? 1 : a);
    e_lo = (unsigned)(l < 1 ? 1 : l);
}
TEDM_DEVFN unsigned cvt8_fp4(const float (&v)[8], unsigned e8m0) {
    const float sc = __builtin_bit_cast(float, e8m0 << 23);
    unsigned u = 0;
    u = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(u, v[0], v[1], sc, 0);
    u = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(u, v[2], v[3], sc, 1);
    u = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(u, v[4], v[5], sc, 2);
    u = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(u, v[6], v[7], sc, 3);
    return u;
}
// fp16 hi (finite: clamped to +-65504) and the fp32 residual of 8 values.  One packed convert per pair and one
// mixed-precision FMA per value (v_fma_mix_f32: f16 half * -1 + v); written as single-instruction asm because hipcc
// otherwise converts every value twice (scalar convert for the residual, packed convert for the store) and back.
TEDM_DEVFN uint4 split8_f16_res(const float (&v)[8], float (&lo)[8]) {
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = __builtin_amdgcn_fmed3f(v[2 * j], -65504.f, 65504.f), b = __builtin_amdgcn_fmed3f(v[2 * j + 1], -65504.f, 65504.f);
        unsigned hp;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(a), "v"(b));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo[2 * j]) : "v"(hp), "v"(v[2 * j]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[2 * j + 1]) : "v"(hp), "v"(v[2 * j + 1]));
        w[j] = hp;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// MODE.FP16_OVFL = 1 for the rest of this wave's kernel: float -> fp16 conversions saturate at +-65504 instead of producing
// inf (probed on gfx950 with tools/ovfl_probe.hip: v_cvt_pk_f16_f32 honours it, the residual of a saturated value stays finite).
TEDM_DEVFN void fp16_saturate_on() { __builtin_amdgcn_s_setreg(1 | (23 << 6), 1); }
// split8_f16_res without the eight explicit clamps; the wave must have called fp16_saturate_on()
TEDM_DEVFN uint4 split8_f16_res_sat(const float (&v)[8], float (&lo)[8]) {
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = v[2 * j], b = v[2 * j + 1];
        unsigned hp;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(a), "v"(b));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo[2 * j]) : "v"(hp), "v"(v[2 * j]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lo[2 * j + 1]) : "v"(hp), "v"(v[2 * j + 1]));
        w[j] = hp;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- Q16: 16-bit block-integer activation storage -----------------------------------------------------------------------------
// The two conv outputs inside a ResnetBlock are written once and read once (by the GroupNorm prologue of the next kernel).  On
// the HBM-bound levels of the fast inference mode they are stored as int16 with one power-of-two scale per BLOCK, where a block
// is what ONE LANE of the producing conv's MFMA accumulator holds: 16 pixels of one channel --
//     block (b, y, xt, hh, c) = pixels x in [32 xt, 32 xt + 32) of row y with ((x >> 2) & 1) == hh, channel c
// -- so the producer finds the block maximum and rounds without any cross-lane operation, and the res_conv epilogue that adds
// such a tensor decodes lane-locally too.  Buffer of a (B, H, W, C) tensor (C % 8 == 0):
//     int16  q[B][H][W][C]                  value = q * 2^(Eb - 127),  q in [-32767, 32767]
//     uint8  Eb[B][H][TX][2][C]              TX = ceil(W / 32); Eb = 0 <=> the block is all zero (or below 2^-112)
// (scales right behind the data: q16_bytes).  Relative to the block's maximum the rounding step is 2^-15..2^-14: 16 significant
// bits where fp16 has 11.  Measured cost on eps (tests/precision_study.py branch): <= 1e-4 in quadrature at 256x256 (fp16 storage:
// 5e-4), DESIGN.md section 2.  Decoding is EXACT arithmetic (integer * power of two), so a consumer of Q16 data computes bit
// for bit what it would compute from the decoded fp32 tensor.
TEDM_DEVFN size_t q16_scale_offset(int B, int H, int W, int C) { return (size_t)2 * B * H * W * C; }      // bytes
TEDM_DEVFN size_t q16_scale_index(long long row /* b*H + y */, int x, int W, int C) {
    return (((size_t)row * ((W + 31) >> 5) + (x >> 5)) * 2 + ((x >> 2) & 1)) * C;
}
// `amax`: maximum |v| over the block.  Returns the scale byte Eb and the encode scale 2^(14-E) (0 when flushed).
TEDM_DEVFN unsigned q16_exponent(float amax, float& enc) {
    const unsigned eb32 = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;      // biased fp32 exponent of amax
    const bool flush = eb32 < 15u;
    const unsigned ebc = eb32 > 254u ? 254u : eb32;                                // inf / nan saturate
    enc = flush ? 0.f : __builtin_bit_cast(float, (268u - ebc) << 23);
    return flush ? 0u : ebc - 14u;
}
// round to nearest even, saturate at +-32767; the int16 is the LOW HALF of the result
TEDM_DEVFN unsigned q16_word(float v, float enc) {
    const float t = __builtin_amdgcn_fmed3f(v * enc, -32767.f, 32767.f);
    return __builtin_bit_cast(unsigned, t + 12582912.f);                           // 1.5 * 2^23: the integer sits in the low mantissa bits
}
TEDM_DEVFN float q16_scale(unsigned eb) { return __builtin_bit_cast(float, eb << 23); }
// 8 consecutive channels of one pixel: data dwords d, scale bytes sc (channel k = byte k)
TEDM_DEVFN void q16_decode8(const uint4 d, const uint2 sc, float (&v)[8]) {
    const unsigned w[4] = {d.x, d.y, d.z, d.w};
    const unsigned s2[2] = {sc.x, sc.y};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = (int)w[i];
        v[2 * i] = (float)((m << 16) >> 16) * q16_scale((s2[i >> 1] >> (16 * (i & 1))) & 0xffu);
        v[2 * i + 1] = (float)(m >> 16) * q16_scale((s2[i >> 1] >> (16 * (i & 1) + 8)) & 0xffu);
    }
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
enum { PRO_NONE = 0, PRO_GN = 1, PRO_LN = 2, PRO_GNQ = 3 };   // PRO_GNQ: PRO_GN on a Q16-stored source (one source, C % 8 == 0)
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
    int out_q16;                   // `out` is a Q16 buffer (int16 data + block scales, see q16_decode8): Cout % 32 == 0, no addend, no parity form
    int add_q16;                   // `add` is a Q16 buffer (only with add_a / add_b: the silu(a*add+b) resnet addend)
    unsigned long long* trace;     // diagnostic builds (-DTEDM_DIAG_TRACE) only: per-wave phase timestamps
    // ups == 3 (KH = 2, template PAD < 0): the four parity convolutions of a nearest-x2 upsample + 3x3 conv.  (H, W) =
    // (Ho, Wo) is the low-resolution grid; "tile" index t = 4 * spatial tile + parity (a, b); parity weights are wpk_pstride
    // bytes apart; output pixel (2y + a, 2x + b) of a (2 Ho, 2 Wo) tensor; top / left zero padding (1 - a, 1 - b).
    size_t wpk_pstride;
};

int conv_launch(const ConvArgs& a, int KH, int stride, int ups, int pro, hipStream_t st);
int conv_stats_entries(int KH, int stride, int Cout, int Ho, int Wo);   // E of the stats slab
size_t conv_packed_bytes(int Cout, int Cin, int taps);
void conv_kernel_label(const ConvArgs& a, int KH, int stride, int ups, int pro, char* buf, int cap);

// ---- pack.hip -------------------------------------------------------------------------
// Weight encodings (which packed layout a TEDM_PREC_* mode consumes): bf16 hi/lo planes, fp16 hi/lo planes, fp16 + fp4.
enum { ENC_BF16 = 0, ENC_F16 = 1, ENC_F16F4 = 2 };
inline int prec_encoding(int prec) { return (prec == 2 /*FP16X2*/ || prec == 4 /*FP16*/) ? ENC_F16 : (prec == 3 /*F16F4*/ ? ENC_F16F4 : ENC_BF16); }
int pack_conv_weights(const float* w, int Cout, int Cin, int KH, int KW, void* wpk, int enc, hipStream_t st);
int pack_conv_weights_dgrad(const float* w, int Cout, int Cin, int KH, int KW, int ci0, int nci, void* wpk, int enc,
                            hipStream_t st);
// nearest-x2 upsample + 3x3 conv == four 2x2 convs on the low-resolution image, one per output parity (a, b): the taps that
// read the same source pixel are pre-summed.  w3: (Cout, Cin, 3, 3); w2x2: scratch (4, Cout, Cin, 2, 2) fp32; wpk: 4 packs.
int pack_conv_weights_up2(const float* w3, int Cout, int Cin, float* w2x2, void* wpk, size_t pack_stride, int enc, hipStream_t st);
int copy_rows(const float* src, float* dst, long long n, hipStream_t st);
