// conv_slide128.hip -- sliding-window 3x3 convolution, 128 input channels (two 64-channel sources: the decoder's channel concat)
// to 64 output channels at full resolution (TEDM_PREC_F16F4).
//
// Replaces, for block1.proj of ups.3's ResnetBlocks and of final_res_block (models/unet_model.py:122,127 under :355-366: the
// input is torch.cat((x, h.pop()), dim = 1)), the tiled implicit GEMM of conv_igemm.inc.  Same pipeline as conv_slide.hip (one
// 4-wave workgroup per CU, one wave per SIMD, resident weights, LDS ring of operand records, everything between the MFMAs) with
// the reduction split over the waves, because 128 input channels are twice the weights and twice the ring bytes per row:
//   * wave (ct, kh) owns output-channel tile ct and the input-channel half kh (= one SOURCE tensor): its 54 weight fragments
//     (2 chunks x 9 taps x 3) are resident as in the 64-channel kernel;
//   * bands of 2 output rows; ring of 6 rows (4 feed the band, 2 are encoded for the next one), 4 chunks per row;
//   * at the end of a band wave (ct, 1) hands its two partial accumulators to wave (ct, 0) through LDS; wave (ct, 0) adds them to
//     its own under the MFMAs of the NEXT band: + bias, stores, GroupNorm sums.  The waves with kh = 1 carry most of the
//     encoder instead (6 of the 9 items of a band).
// Result = (sum over chunks 0, 1) + (sum over chunks 2, 3) + bias: deterministic, every output accumulates in the same order
// whatever the batch or the row split, but NOT the tiled kernel's order (chunk 0..3 in sequence): the two agree to rounding
// (tests/test_gpu_conv_slide.py), and conv_igemm.hip sends every launch of this shape here (no per-batch kernel choice).
#include "conv_slide_common.h"
#include <stdio.h>
#include "../../include/tedm_hip.h"

extern int g_conv_slide, g_conv_slide_rows;

namespace {

constexpr int S2_NR = 6;                     // ring rows
constexpr int S2_NC = 4;                     // 32-channel chunks (Cin = 128)
constexpr int S2_ROWB = S2_NC * SL_SEG;      // bytes of a ring row
constexpr int S2_ZERO = S2_NR * S2_ROWB;     // all-zero row
constexpr int S2_XCH = S2_ZERO + S2_ROWB;    // partial accumulators: [band parity][ct][8 x 1 KB]
constexpr int S2_LDS = S2_XCH + 2 * 2 * 8192;   // 142 528 B

// vector-memory operations of one band by MFMA gap t (0..107) for the two wave roles (see the band loops):
//   kh = 1: item j's two loads behind MFMA 16 j + 13 (6 items);  kh = 0: the same for 3 items, one output store behind MFMA
//   3 v + 1 (v < 32), the statistics store behind MFMA 97
constexpr int s2_vm_in_gaps(int kh, int lo, int hi, bool st) {
    int n = 0;
    for (int t = lo; t < hi; ++t) {
        if (t % 16 == 13 && t / 16 < (kh ? 6 : 3)) n += 2;
        if (!kh && t % 3 == 1 && t / 3 < 32) n += 1;
        if (!kh && st && t == 97) n += 1;
    }
    return n;
}
// ... issued between item k's loads (previous band; a store of the same gap comes after them) and its first stage (gap 16 k)
constexpr int s2_vm_after(int kh, int k, bool st) {
    const int t = 16 * k + 13;
    return (!kh && t % 3 == 1 && t / 3 < 32 ? 1 : 0) + s2_vm_in_gaps(kh, t + 1, 108, st) + s2_vm_in_gaps(kh, 0, 16 * k, st);
}

template <bool ST>
__global__ __launch_bounds__(256, 1) void conv3x3_slide128_kernel(ConvArgs a, int rows_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fp16_saturate_on();

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ct = wave & 1, kh = wave >> 1;                   // output-channel tile, input-channel half (source tensor)
    const int W = a.W, H = a.H;

    int id = blockIdx.x;
    const int tx = id % a.tiles_x; id /= a.tiles_x;
    const int nyc = H / rows_wg;
    const int yc = id % nyc, n = id / nyc;
    const int y0 = yc * rows_wg, nb = rows_wg >> 1;

    // ---- resident weights: chunks 2 kh, 2 kh + 1 of output tile ct (register plan: conv_slide.hip) --------------------------
    u32x4 wf[2][3][3][3];
    unsigned wsc[2][3];
    {
        const u32x4* wq = (const u32x4*)a.wpk;
        const unsigned* ws = (const unsigned*)(wq + (size_t)2 * S2_NC * 9 * 3 * 64);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        wf[c][ky][kx][p] = wq[((((size_t)ct * S2_NC + 2 * kh + c) * 9 + (ky * 3 + kx)) * 3 + p) * 64 + lane];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wsc[c][kx] = ws[((size_t)(ct * S2_NC + 2 * kh + c) * 3 + kx) * 64 + lane];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        if (((c * 3 + ky) * 3 + kx) * 3 + p < SL_WAGPR) asm volatile("" : "+a"(wf[c][ky][kx][p]));
                        else asm volatile("" : "+v"(wf[c][ky][kx][p]));
                    }
    }

    // ---- LDS: 6-row ring | all-zero row | exchange buffers; zero padding as in conv_slide.hip ---------------------------------
    for (int o = tid * 16; o < S2_XCH; o += 256 * 16) *(uint4*)(smem + o) = make_uint4(0, 0, 0, 0);

    // ---- encoder -------------------------------------------------------------------------------------------------------------
    // A batch is 2 input rows x 4 chunks.  Row ct of the batch belongs to the waves with this ct.  Wave (ct, 1) encodes chunks
    // 0, 1, 2 of it (6 items: pixel columns q / q + 16 per chunk), wave (ct, 0) chunk 3 (2 items) plus, for SOURCE ct, the two
    // columns past 32 of both rows (1 item on lanes 0-31: row, chunk of the source, column, piece; lanes 32-63 mirror them).
    const int piece = lane & 3, q = lane >> 2;
    const __amdgpu_buffer_rsrc_t src0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x1 + (size_t)n * H * W * 64), 0, H * W * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t src1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x2 + (size_t)n * H * W * 64), 0, H * W * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t srcr = ct ? src1 : src0;      // source of this wave's remainder item
    const int ix0 = tx * 32 - 1;
    unsigned coff[2];                                           // main items: byte offset of (column, piece) inside an image row, chunk 0 of a source
    int ldst[2];                                                // ... LDS offset inside a ring row, chunk 0
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int hx = q + 16 * j, ix = ix0 + hx;
        const bool ok = ix >= 0 && ix < W;
        const int xc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        coff[j] = (unsigned)(xc * 256 + piece * 32);
        ldst[j] = (ok ? hx : SL_HC) * SL_PST + piece * 4;
    }
    // remainder item (kh = 0 waves): lane -> (row rj, chunk rc of source ct, column)
    const int rl = lane & 31, rj = rl >> 4, rc = (rl >> 3) & 1;
    unsigned rem_coff;
    int rem_ldst;
    {
        const int hx = 32 + ((rl >> 2) & 1), ix = ix0 + hx;
        const bool ok = ix >= 0 && ix < W;
        const int xc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        rem_coff = (unsigned)(xc * 256 + rc * 128 + piece * 32);
        rem_ldst = (2 * ct + rc) * SL_SEG + (ok ? hx : SL_HC) * SL_PST + piece * 4;
    }

    float raw[6][8];
    int erow = 0, srow = 0;                                     // this wave's row of the batch: LDS offset of its ring row; source byte offset of the row to load
    int rem_dst = 0;
    unsigned rem_src = 0;
    auto wrap = [](int sl) { return sl >= S2_NR ? sl - S2_NR : sl; };
    // slot_e: ring slot of the batch's row 0 to ENCODE; y_ld: image row of row 0 of the batch to LOAD
    auto batch_rows = [&](int slot_e, int y_ld) __attribute__((always_inline)) {
        erow = wrap(slot_e + ct) * S2_ROWB;
        const int y = y_ld + ct;
        srow = (y < 0 ? 0 : (y >= H ? H - 1 : y)) * (W * 256);
        rem_dst = wrap(slot_e + rj) * S2_ROWB + rem_ldst;
        const int yr = y_ld + rj;
        rem_src = (unsigned)((yr < 0 ? 0 : (yr >= H ? H - 1 : yr)) * (W * 256)) + rem_coff;
    };
    // item k of role KH: chunk and column set (main items), or the remainder item
    auto load_item = [&](auto KH, auto K) __attribute__((always_inline)) {
        constexpr int khc = decltype(KH)::value, k = decltype(K)::value;
        constexpr bool rem = khc == 0 && k == 2;
        constexpr int c = khc ? k >> 1 : 3;                     // chunk of a main item
        f32x4 v0, v1;
        if constexpr (rem) {
            v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srcr, (int)rem_src, 0, 0));
            v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srcr, (int)rem_src + 16, 0, 0));
        } else if constexpr (c < 2) {
            v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src0, (int)coff[k & 1] + (c & 1) * 128, srow, 0));
            v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src0, (int)coff[k & 1] + (c & 1) * 128 + 16, srow, 0));
        } else {
            v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src1, (int)coff[k & 1] + (c & 1) * 128, srow, 0));
            v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src1, (int)coff[k & 1] + (c & 1) * 128 + 16, srow, 0));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { raw[k][e] = v0[e]; raw[k][4 + e] = v1[e]; }
    };
    // The encoding of one item as 16 micro-stages (conv_slide.hip's stages 8..23: no prologue here); stage 0 of an item first
    // waits for its loads (`vm`: vector-memory operations issued since, counted by hand -- see conv_slide.hip).
    unsigned e_v4 = 0, e_lo = 0, f4lo = 0, f4v = 0, em = 0;
    float esc_lo = 0.f, esc_v4 = 0.f, elo0 = 0.f, elo1 = 0.f;
    u32x4 ehi = {0, 0, 0, 0};
    auto stage = [&](auto KH, auto T, auto VM) __attribute__((always_inline)) {
        constexpr int khc = decltype(KH)::value, t = decltype(T)::value, vm = decltype(VM)::value;
        constexpr int k = t / 16, sp = t % 16;
        float (&x)[8] = raw[k];
        if constexpr (sp == 0 && vm >= 0) __builtin_amdgcn_s_waitcnt(sl_vmcnt_imm(vm > 63 ? 63 : vm));
        if constexpr (sp == 0 || sp == 1) {
            constexpr int e = sp * 4;
            float m = sp == 0 ? 0.f : __builtin_bit_cast(float, em);
            m = fmaxf(fmaxf(m, fabsf(x[e])), fabsf(x[e + 1]));
            m = fmaxf(fmaxf(m, fabsf(x[e + 2])), fabsf(x[e + 3]));
            em = __builtin_bit_cast(unsigned, m);
        } else if constexpr (sp == 2) {
            asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(em) : "v"(em));
        } else if constexpr (sp == 3) {
            asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(em) : "v"(em));
        } else if constexpr (sp == 4) {
            f4_scales(__builtin_bit_cast(float, em), e_v4, e_lo);
        } else if constexpr (sp == 5) {
            esc_lo = __builtin_bit_cast(float, e_lo << 23);
            esc_v4 = __builtin_bit_cast(float, e_v4 << 23);
        } else if constexpr (sp >= 6 && sp < 10) {
            constexpr int j = sp - 6;
            unsigned hp;
            asm("v_cvt_pk_f16_f32 %0, %3, %4\n\tv_fma_mix_f32 %1, %0, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                : "=&v"(hp), "=&v"(elo0), "=&v"(elo1) : "v"(x[2 * j]), "v"(x[2 * j + 1]));
            ehi[j] = hp;
            f4lo = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(j == 0 ? 0u : f4lo, elo0, elo1, esc_lo, j);
        } else if constexpr (sp == 10 || sp == 11) {
            constexpr int j = (sp - 10) * 2;
            f4v = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(j == 0 ? 0u : f4v, x[2 * j], x[2 * j + 1], esc_v4, j);
            f4v = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(f4v, x[2 * j + 2], x[2 * j + 3], esc_v4, j + 1);
        } else if constexpr (sp == 12) {
            constexpr bool rem = khc == 0 && k == 2;
            constexpr int c = khc ? k >> 1 : 3;
            const int dst = rem ? rem_dst : erow + c * SL_SEG + ldst[k & 1];
            *(u32x4*)(smem + dst + piece * 12) = ehi;
            *(unsigned*)(smem + dst + 64) = f4lo;
            *(unsigned*)(smem + dst + 80) = f4v;
            *(unsigned*)(smem + dst + 96) = (piece & 1) ? e_v4 : e_lo;
        } else if constexpr (sp == 13) {
            load_item(KH, std::integral_constant<int, k>{});     // the same item of the batch after next
        }
    };

    const int lane_a = r * SL_PST + h * 16 + 2 * kh * SL_SEG;   // fp16 / fp4 operand reads, this wave's chunks
    const int lane_s = r * SL_PST + 96 + 4 * h + 2 * kh * SL_SEG;
    const int cout = ct * 32 + r;
    const float bias = a.bias ? a.bias[cout] : 0.f;
    const unsigned lane_off = (unsigned)(4 * h * 64 + cout);
    const int E = (H >> 2) * a.tiles_x;
    const int xlane = ct * 8192 + lane * 16;                    // this lane's slot in an exchange buffer
#define S2_SCHED() __builtin_amdgcn_sched_barrier(0)

    // Operand fragments of a group (chunk of the half, tap column, halo row), read SL_PF groups ahead of their MFMAs.  The first
    // SL_PF (= 2) groups of a band read halo rows 0 and 1, which the PREVIOUS band already used as its rows 2 and 3: they are
    // issued in front of the band's barrier (`band_prefetch`), so the first MFMAs behind it find their operands landed.
    f32x16 acc[2];
    constexpr int NB = SL_PF + 1;
    static_assert(SL_PF <= 2, "groups prefetched in front of the barrier must read halo rows 0 and 1 only");
    u32x4 fa0[NB], fa1[NB], faq[NB];
    int fas[NB];
    int abase[4], sbase[4];
    auto read_group = [&](auto G) __attribute__((always_inline)) {
        constexpr int g = decltype(G)::value;                    // 0..23
        constexpr int c = g / 12, kx = (g / 4) % 3, p = g % 4;
        constexpr int off = c * SL_SEG + kx * SL_PST;
        fas[g % NB] = *(const int*)(smem + sbase[p] + off);
        S2_SCHED();
        faq[g % NB] = *(const u32x4*)(smem + abase[p] + off + 64);
        S2_SCHED();
        fa1[g % NB] = *(const u32x4*)(smem + abase[p] + off + 32);
        S2_SCHED();
        fa0[g % NB] = *(const u32x4*)(smem + abase[p] + off);
        S2_SCHED();
    };
    // `slot_r`: ring slot of the band's first halo row, `y_first` its image row
    auto band_prefetch = [&](int slot_r, int y_first) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int y = y_first + p;
            const int off = (y >= 0 && y < H) ? wrap(slot_r + p) * S2_ROWB : S2_ZERO;
            abase[p] = lane_a + off;
            sbase[p] = lane_s + off;
        }
        sl_for<SL_PF>([&](auto G) __attribute__((always_inline)) { read_group(G); });
    };
    // one band's MFMAs (108) with `between(MFMA ordinal u)` called behind each
    auto band_mfmas = [&](auto&& between) __attribute__((always_inline)) {
        S2_SCHED();
        sl_for<24>([&](auto G) __attribute__((always_inline)) {
            constexpr int g = decltype(G)::value;
            constexpr int c = g / 12, kx = (g / 4) % 3, p = g % 4;
            if constexpr (g + SL_PF < 24) read_group(std::integral_constant<int, g + SL_PF>{});
            const f16x8 a0 = __builtin_bit_cast(f16x8, fa0[g % NB]), a1 = __builtin_bit_cast(f16x8, fa1[g % NB]);
            i32x8 a4 = {};
            a4[0] = faq[g % NB][0]; a4[1] = faq[g % NB][1]; a4[2] = faq[g % NB][2]; a4[3] = faq[g % NB][3];
            const int as = fas[g % NB];
            constexpr int cum = p == 0 ? 0 : (p == 1 ? 1 : (p == 2 ? 3 : 5));
            sl_for<3>([&](auto KY) __attribute__((always_inline)) {
                constexpr int ky = decltype(KY)::value;
                constexpr int ml = p - ky;
                if constexpr (ml >= 0 && ml < 2) {
                    constexpr int tapi = ky - (p > 1 ? p - 1 : 0);
                    constexpr int u0 = ((c * 3 + kx) * 6 + cum + tapi) * 3;
                    i32x8 b4 = {};
                    b4[0] = wf[c][ky][kx][2][0]; b4[1] = wf[c][ky][kx][2][1]; b4[2] = wf[c][ky][kx][2][2]; b4[3] = wf[c][ky][kx][2][3];
                    constexpr bool first = c == 0 && kx == 0 && ky == 0;
                    sl_for<3>([&](auto M) __attribute__((always_inline)) {
                        constexpr int m = decltype(M)::value;
                        f32x16 cin;
                        if constexpr (first && m == 0) { cin = f32x16{}; } else { cin = acc[ml]; }
                        if constexpr (m == 0)      acc[ml] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(f16x8, wf[c][ky][kx][0]), cin, 0, 0, 0);
                        else if constexpr (m == 1) acc[ml] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, __builtin_bit_cast(f16x8, wf[c][ky][kx][1]), cin, 0, 0, 0);
                        else acc[ml] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, cin, 4, 4, 0, as, ky, (int)wsc[c][kx]);
                        between(std::integral_constant<int, u0 + m>{});
                        S2_SCHED();
                    });
                }
            });
        });
    };

    // ring row i holds image row y0 - 1 + i in slot i mod 6; band b: output rows y0 + 2 b, + 1; halo rows i = 2 b .. 2 b + 3;
    // it encodes rows i = 2 b + 4, + 5 and loads rows i = 2 b + 6, + 7
    if (kh) {
        // ================= waves (ct, 1): source 1 half of the reduction, 6 encoder items per band ==========================
        using KH = std::integral_constant<int, 1>;
        __syncthreads();                                         // the zero fill
        batch_rows(0, y0 - 1);
        sl_for<6>([&](auto K) __attribute__((always_inline)) { load_item(KH{}, K); });
#pragma unroll 1
        for (int pb_ = 0; pb_ < 2; ++pb_) {                      // prime: ring rows 0..3, then the loads of rows 4, 5
            batch_rows(2 * pb_, y0 - 1 + 2 * pb_ + 2);
            sl_for<6 * 16>([&](auto T) __attribute__((always_inline)) { stage(KH{}, T, std::integral_constant<int, (decltype(T)::value % 16 == 0 ? 0 : -1)>{}); });
        }
        __builtin_amdgcn_s_waitcnt(sl_vmcnt_imm(0));
        __syncthreads();
        int slot_e = 4, slot_r = 0;
        band_prefetch(0, y0 - 1);
#pragma unroll 1
        for (int b = 0; b < nb; ++b) {
            batch_rows(slot_e, y0 - 1 + 2 * b + 6);
            band_mfmas([&](auto U) __attribute__((always_inline)) {
                constexpr int u = decltype(U)::value;
                if constexpr (u < 6 * 16) stage(KH{}, U, std::integral_constant<int, s2_vm_after(1, u / 16, ST)>{});
            });
            // hand the two partial accumulators to wave (ct, 0): 8 x 16-byte stores per lane
            char* xb = smem + S2_XCH + (b & 1) * 16384 + xlane;
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v = {acc[ml][4 * q4], acc[ml][4 * q4 + 1], acc[ml][4 * q4 + 2], acc[ml][4 * q4 + 3]};
                    *(f32x4*)(xb + (ml * 4 + q4) * 1024) = v;
                }
            slot_e = wrap(slot_e + 2); slot_r = wrap(slot_r + 2);
            band_prefetch(slot_r, y0 - 1 + 2 * b + 2);
            __syncthreads();
        }
        __syncthreads();                                         // (pairs with the barrier in front of the other role's tail)
    } else {
        // ================= waves (ct, 0): source 0 half, 3 encoder items, the epilogue of the previous band ====================
        using KH = std::integral_constant<int, 0>;
        __syncthreads();
        batch_rows(0, y0 - 1);
        sl_for<3>([&](auto K) __attribute__((always_inline)) { load_item(KH{}, K); });
#pragma unroll 1
        for (int pb_ = 0; pb_ < 2; ++pb_) {
            batch_rows(2 * pb_, y0 - 1 + 2 * pb_ + 2);
            sl_for<3 * 16>([&](auto T) __attribute__((always_inline)) { stage(KH{}, T, std::integral_constant<int, (decltype(T)::value % 16 == 0 ? 0 : -1)>{}); });
        }
        __builtin_amdgcn_s_waitcnt(sl_vmcnt_imm(0));
        __syncthreads();
        int slot_e = 4, slot_r = 0;
        band_prefetch(0, y0 - 1);
        float pend[2][16];                                       // this wave's own partial sums of the previous band
#pragma unroll
        for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int i = 0; i < 16; ++i) pend[ml][i] = 0.f;
        float s1 = 0.f, s2 = 0.f;
        float* orow = nullptr;
        const char* xb = smem + S2_XCH + xlane;
        f32x4 part = {0.f, 0.f, 0.f, 0.f};
        // one output register of the previous band: value v = 16 ml + i  (partner's partial + own partial + bias)
        auto epi_value = [&](auto V) __attribute__((always_inline)) {
            constexpr int v = decltype(V)::value, ml = v >> 4, i = v & 15;
            if constexpr ((i & 3) == 0) part = *(const f32x4*)(xb + (ml * 4 + (i >> 2)) * 1024);
            const float x = (pend[ml][i] + part[i & 3]) + bias;
            (orow + (size_t)ml * W * 64 + ((i & 3) + 8 * (i >> 2)) * 64)[lane_off] = x;
            if (ST) { s1 += x; s2 = fmaf(x, x, s2); }
        };
        auto epi_stats = [&](int oy) __attribute__((always_inline)) {      // rows oy, oy + 1 done: entry of their 4-row group (partial after its
            if (!ST) return;                                                // first two rows, overwritten with the full sums after the last two)
            const float t1 = s1 + __shfl_xor(s1, 32), t2 = s2 + __shfl_xor(s2, 32);
            const int e = (oy >> 2) * a.tiles_x + tx;
            *((float2*)a.stats + ((size_t)n * E + e) * 64 + cout) = make_float2(t1, t2);
        };
#pragma unroll 1
        for (int b = 0; b < nb; ++b) {
            batch_rows(slot_e, y0 - 1 + 2 * b + 6);
            // rows of the deferred epilogue: the previous band's (band 0: its own rows -- garbage it overwrites one band later)
            const int oyp = y0 + 2 * (b > 0 ? b - 1 : 0);
            orow = a.out + ((size_t)(n * H + oyp) * W + tx * 32) * 64;
            xb = smem + S2_XCH + ((b + 1) & 1) * 16384 + xlane;
            if ((oyp & 2) == 0) { s1 = 0.f; s2 = 0.f; }
            band_mfmas([&](auto U) __attribute__((always_inline)) {
                constexpr int u = decltype(U)::value;
                if constexpr (u < 3 * 16) stage(KH{}, U, std::integral_constant<int, s2_vm_after(0, u / 16, ST)>{});
                if constexpr (u % 3 == 1 && u / 3 < 32) epi_value(std::integral_constant<int, u / 3>{});
                if constexpr (u == 97) epi_stats(oyp);
            });
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int i = 0; i < 16; ++i) pend[ml][i] = acc[ml][i];
            slot_e = wrap(slot_e + 2); slot_r = wrap(slot_r + 2);
            band_prefetch(slot_r, y0 - 1 + 2 * b + 2);
            __syncthreads();
        }
        // the last band
        {
            const int oy = y0 + 2 * (nb - 1);
            orow = a.out + ((size_t)(n * H + oy) * W + tx * 32) * 64;
            xb = smem + S2_XCH + ((nb + 1) & 1) * 16384 + xlane;
            if ((oy & 2) == 0) { s1 = 0.f; s2 = 0.f; }
            sl_for<32>([&](auto V) __attribute__((always_inline)) { epi_value(V); });
            epi_stats(oy);
        }
        __syncthreads();
    }
}

template <bool ST>
int launch_slide128(const ConvArgs& a, int rows_wg, hipStream_t st) {
    auto kern = conv3x3_slide128_kernel<ST>;
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return TEDM_EHIP;
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return TEDM_EHIP;
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    const int grid = a.B * (a.H / rows_wg) * a.tiles_x;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)S2_LDS, st, a, rows_wg);
    return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
}

}  // namespace

bool conv_slide128_supported(const ConvArgs& a, int KH, int stride, int ups, int pro) {
    if (!g_conv_slide || a.prec != TEDM_PREC_F16F4 || KH != 3 || stride != 1 || ups || pro != PRO_NONE) return false;
    if (a.Cout != 64 || a.C1 != 64 || !a.x2 || a.C2 != 64) return false;
    if (a.add || a.out_q16 || a.add_q16 || a.stats_mode == STATS_MAX) return false;
    if (a.H != a.Ho || a.W != a.Wo || a.W % 32 || a.H % 8 || a.H < 16) return false;
    return true;
}

void conv_slide128_label(const ConvArgs& a, char* buf, int cap) {
    snprintf(buf, cap, "conv3x3_slide128_kernel<%s>", a.stats_mode == STATS_SUMSQ ? "true" : "false");
}

int conv_slide128_launch(const ConvArgs& a_in, hipStream_t st) {
    ConvArgs a = a_in;
    a.tiles_x = a.W / 32;
    int rows = g_conv_slide_rows;
    if (rows <= 0 || rows % 8 || a.H % rows) {
        rows = sl_pick_rows(a.B, a.H, a.tiles_x);
    }
    return a.stats_mode == STATS_SUMSQ ? launch_slide128<true>(a, rows, st) : launch_slide128<false>(a, rows, st);
}
