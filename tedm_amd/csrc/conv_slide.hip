// conv_slide.hip -- sliding-window 3x3 convolution for the full-resolution 64-output-channel layers (TEDM_PREC_F16F4).
//
// Replaces, for Block.proj at the 64-channel levels (models/unet_model.py:122,127; ResnetBlock :161-175, the last-level 3x3
// :323-324), the tiled implicit GEMM of conv_igemm.inc.  Same arithmetic, same operand encoding, same accumulation order per
// output element -- the results are bit-identical to conv_igemm_kernel<3,3,1,1,0,...,TEDM_PREC_F16F4> -- but a different
// pipeline, built for depth instead of width:
//   * ONE 4-wave workgroup per CU, one wave per SIMD with the whole 512-register budget.  A wave owns 32 output channels and
//     keeps ALL their weight fragments (2 chunks x 9 taps x 3 fragments = 216 registers + 6 scale dwords) resident for the whole
//     kernel: no weight traffic in the loop (the tiled kernel re-fetched 442 KB of fragments from L2 per 512 output pixels,
//     1.7x its activation traffic).
//   * A workgroup walks DOWN a 32-column strip of one image in bands of 8 output rows.  The MFMA operand records of the input
//     rows live in an 18-row LDS ring: 10 rows feed the current band while the 8 new rows of the next band are encoded, so every
//     input row is loaded and encoded ONCE per strip (the 16-row tile re-encoded 18 rows per 16) and there is no halo in y.
//   * The fp32 activations of the band after next are in flight in registers (72 VGPRs per wave = 72 KB per CU) for a whole band
//     period (~3 us) before they are encoded: the loads never wait, and nothing waits for them.
//   * Encoding (prologue, fp16 / fp4 split, LDS stores), the loads and the MFMAs of a band are one straight-line block: the
//     wave's own VALU work issues in the shadow of its MFMAs; one barrier per band.
#include "conv_slide_common.h"
#include <stdio.h>
#include "../../include/tedm_hip.h"

int g_conv_slide = 1;            // tedm_set_tuning("conv_slide", 0|1)
int g_conv_slide_rows = 0;       // tedm_set_tuning("conv_slide_rows", rows per workgroup; 0 = automatic)

namespace {

constexpr int SL_NR = 18;                    // ring rows
constexpr int SL_NC = 2;                     // 32-channel chunks (Cin = 64)
constexpr int SL_ROWB = SL_NC * SL_SEG;      // bytes of a ring row
constexpr int SL_ZERO = SL_NR * SL_ROWB;     // byte offset of the all-zero row
constexpr int SL_LDS = SL_ZERO + SL_ROWB;    // 148 960 B

// silu(a x + b) of two channels with the packed-fp32 forms (v_pk_fma / v_pk_mul / v_pk_add: one instruction per pair) around the
// two transcendentals per channel: the same five operations per channel as silu_f (common.h) behind the same fma, hence the same
// bits -- every instruction between two MFMAs of this single-wave kernel costs ~3 cycles whatever it is.
typedef float f32x2 __attribute__((ext_vector_type(2)));
TEDM_DEVFN void silu2_gn(float& x0, float& x1, float a0, float a1, float b0, float b1) {
#pragma clang fp contract(off)
    const f32x2 v = __builtin_elementwise_fma((f32x2){a0, a1}, (f32x2){x0, x1}, (f32x2){b0, b1});
    const f32x2 t = v * (f32x2){-1.44269504088896340736f, -1.44269504088896340736f};       // 0xbfb8aa3b
    const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    const f32x2 d = e + (f32x2){1.f, 1.f};
    const f32x2 rc = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2 y = v * rc;
    x0 = y[0]; x1 = y[1];
}

template <int PRO, bool ST>
__global__ __launch_bounds__(256, 1) void conv3x3_slide64_kernel(ConvArgs a, int rows_wg) {
    constexpr int NC = SL_NC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    fp16_saturate_on();

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int ct = wave & 1, rh = wave >> 1;                   // output-channel tile, row half of a band
    const int W = a.W, H = a.H;

    // block -> (image, row range, strip); strips of one row range are neighbours in the grid
    int id = blockIdx.x;
    const int tx = id % a.tiles_x; id /= a.tiles_x;
    const int nyc = H / rows_wg;
    const int yc = id % nyc, n = id / nyc;
    const int y0 = yc * rows_wg, nb = rows_wg >> 3;

    // ---- resident weights -------------------------------------------------------------------------------------------------
    u32x4 wf[NC][3][3][3];                                     // [chunk][ky][kx][fp16 lo-half channels | fp16 hi-half | fp4]
    unsigned wsc[NC][3];                                       // E8M0 scales of a tap column, byte ky
    {
        const u32x4* wq = (const u32x4*)a.wpk;
        const unsigned* ws = (const unsigned*)(wq + (size_t)2 * NC * 9 * 3 * 64);
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        wf[c][ky][kx][p] = wq[((((size_t)ct * NC + c) * 9 + (ky * 3 + kx)) * 3 + p) * 64 + lane];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) wsc[c][kx] = ws[((size_t)(ct * NC + c) * 3 + kx) * 64 + lane];
        // Register file plan (one wave per SIMD: 256 VGPRs + 256 AGPRs).  MFMA-only values go to the accumulator file: the 64
        // accumulators and SL_WAGPR of the 54 weight fragments; the rest of the fragments and everything the VALU touches stay in
        // VGPRs.  Without the pins the allocator parks the fragments in AGPRs as SPILL slots and copies each one into an operand
        // register in front of its MFMA (224 v_accvgpr moves per band).
#pragma unroll
        for (int c = 0; c < NC; ++c)
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

    // ---- LDS: 18-row ring | one all-zero row ----------------------------------------------------------------------------------
    // Zero padding costs nothing per element: rows outside the image are never READ (the fragment reads of such a halo row go to
    // the zero row: a wave-uniform address select per band), columns outside the image are never WRITTEN (the lanes that would
    // stage them store into the spare slot of their row; the ring starts zeroed, so those records stay zero while it rotates).
    for (int o = tid * 16; o < SL_LDS; o += 256 * 16) *(uint4*)(smem + o) = make_uint4(0, 0, 0, 0);

    // ---- encoder state ----------------------------------------------------------------------------------------------------
    // A batch is 8 consecutive input rows.  Wave w encodes chunk w & 1 of rows (w >> 1) + {0, 2, 4, 6} of the batch: per row two
    // items per lane (pixel columns q and q + 16 of the halo, 8 channels = one `piece` of the pixel's 32-channel block) and, for
    // the two columns past 32, one more item on lanes 0-31 (lanes 32-63 mirror them: same address, same value).
    const int piece = lane & 3, q = lane >> 2;
    const int cm = wave & 1, jw = wave >> 1;
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x1 + (size_t)n * H * W * 64), 0, H * W * 256, 0x00020000);   // this image
    const int ix0 = tx * 32 - 1;
    unsigned coff[3];                                          // source byte offset inside an image row (scalar row pointer + 32-bit lane offset loads)
    int ldst[3];                                               // LDS offset inside a ring row
    const int rk = (lane >> 3) & 3;                            // remainder item: row jw + 2 rk of the batch
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int hx = j < 2 ? q + 16 * j : 32 + ((lane >> 2) & 1);
        const int ix = ix0 + hx;
        const bool ok = ix >= 0 && ix < W;
        const int xc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        coff[j] = (unsigned)(xc * 256 + cm * 128 + piece * 32);
        ldst[j] = cm * SL_SEG + (ok ? hx : SL_HC) * SL_PST + piece * 4;
    }
    float pa[8], pb[8];
    if (PRO == PRO_GN) {
        const float4* qa = (const float4*)(a.pro_a + (size_t)n * 64 + cm * 32 + piece * 8);
        const float4* qb = (const float4*)(a.pro_b + (size_t)n * 64 + cm * 32 + piece * 8);
        const float4 a0 = qa[0], a1 = qa[1], b0 = qb[0], b1 = qb[1];
        pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w; pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
        pb[0] = b0.x; pb[1] = b0.y; pb[2] = b0.z; pb[3] = b0.w; pb[4] = b1.x; pb[5] = b1.y; pb[6] = b1.z; pb[7] = b1.w;
    }

#ifdef SL_ABL        // tools/slide_lab.hip ablations (results wrong, timing only): 1 no encoder stages / loads in the band, 2 no output stores,
                     // 4 no epilogue in the band at all.  (A switch "no fragment reads" existed in round 4's first lab runs and was
                     // removed: with loop-invariant operands the compiler merges the two phases' identical MFMA chains, so it timed
                     // half the MFMAs; every variant kept here has its 216 MFMAs per band in the ISA.)
    constexpr int ABL = SL_ABL;
#else
    constexpr int ABL = 0;
#endif
    float raw[9][8];
    // Per batch (scalars, set up once per band): LDS offset of the wave's four ring rows, source pointer of its four rows of the
    // batch to LOAD; the remainder item's row is per lane (rk).
    int erow[4], srow[4];
    int rem_dst = 0;
    unsigned rem_src = 0;
    auto wrap = [](int sl) { return sl >= SL_NR ? sl - SL_NR : sl; };     // ring slot arithmetic without divisions: sl < 2 NR
    // slot_e: ring slot of the batch's row jw (this wave's first row) to ENCODE; y_ld: image row of row jw of the batch to LOAD
    auto batch_rows = [&](int slot_e, int y_ld) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            erow[kk] = wrap(slot_e + 2 * kk) * SL_ROWB;
            const int y = y_ld + 2 * kk;
            srow[kk] = (y < 0 ? 0 : (y >= H ? H - 1 : y)) * (W * 256);
        }
        rem_dst = wrap(slot_e + 2 * rk) * SL_ROWB + ldst[2];
        const int y = y_ld + 2 * rk;
        rem_src = (unsigned)((y < 0 ? 0 : (y >= H ? H - 1 : y)) * (W * 256)) + coff[2];
    };
    // Loads are unconditional (clamped address), see conv_igemm.inc.  Item k < 8: row jw + 2 (k >> 1) of the batch, column set k & 1.
    auto load_item = [&](auto K) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        const int vo = (int)(k < 8 ? coff[k & 1] : rem_src), so = k < 8 ? srow[(k >> 1) & 3] : 0;        // buffer_load v, voff, rsrc, soff
        // (bit_cast of the WHOLE vector: element-wise casts of the builtin's result are folded to one dword by this compiler)
        const f32x4 v0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, vo, so, 0));
        const f32x4 v1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, vo + 16, so, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) { raw[k][e] = v0[e]; raw[k][4 + e] = v1[e]; }
    };
    // The encoding of one item (one pixel's 8-channel piece; bit for bit conv_igemm.inc write_item, F16F4) as 24 micro-stages of
    // at most ~6 VALU instructions: in the band loop ONE stage stands behind every MFMA, i.e. runs in its shadow.  The item is
    // encoded in place (raw[k] becomes the prologue's output); state between stages: em, e_v4, e_lo, esc*, ehi, elo, f4*.
    unsigned e_v4 = 0, e_lo = 0, f4lo = 0, f4v = 0, em = 0;
    float esc_lo = 0.f, esc_v4 = 0.f, elo0 = 0.f, elo1 = 0.f;
    u32x4 ehi = {0, 0, 0, 0};
    auto stage = [&](auto T) __attribute__((always_inline)) {
        constexpr int t = decltype(T)::value;
        constexpr int k = t / 24, sp = t % 24;
        float (&x)[8] = raw[k];
        if constexpr (sp == (PRO == PRO_GN ? 0 : 8)) {
            // The item's loads were issued one band ago (stage 21).  The wait is explicit and counted by hand: the compiler's own
            // pass loses loop-carried loads here (no wait at all, or vmcnt(0) at the first inline asm).  N = vector-memory operations
            // issued after them: the later items' loads, the stores of the epilogue, this band's earlier loads (see sl_vm_after).
            constexpr int n = sl_vm_after(k, PRO == PRO_GN ? 0 : 8, ST, ABL);
            __builtin_amdgcn_s_waitcnt(sl_vmcnt_imm(n > 63 ? 63 : n));
        }
        if constexpr (sp < 8) {                                  // prologue: silu(a x + b), two channels per even stage
            if constexpr (PRO == PRO_GN && (sp & 1) == 0) silu2_gn(x[sp], x[sp + 1], pa[sp], pa[sp + 1], pb[sp], pb[sp + 1]);
        } else if constexpr (sp == 8 || sp == 9) {               // block maximum: |.| of four channels per stage
            constexpr int e = (sp - 8) * 4;
            float m = sp == 8 ? 0.f : __builtin_bit_cast(float, em);
            m = fmaxf(fmaxf(m, fabsf(x[e])), fabsf(x[e + 1]));
            m = fmaxf(fmaxf(m, fabsf(x[e + 2])), fabsf(x[e + 3]));
            em = __builtin_bit_cast(unsigned, m);
        } else if constexpr (sp == 10) {                         // ... over the quad = the pixel's 32-channel block (quad_max, common.h: the
            // same integer maxima; the DPP operand sits on the max itself.  s_nop: a DPP read needs two wait states after the VALU write)
            asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(em) : "v"(em));
        } else if constexpr (sp == 11) {
            asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(em) : "v"(em));
        } else if constexpr (sp == 12) {
            f4_scales(__builtin_bit_cast(float, em), e_v4, e_lo);
        } else if constexpr (sp == 13) {
            esc_lo = __builtin_bit_cast(float, e_lo << 23);
            esc_v4 = __builtin_bit_cast(float, e_v4 << 23);
        } else if constexpr (sp >= 14 && sp < 18) {              // fp16 hi + fp4 residual of one channel pair (split8_f16_res_sat, cvt8_fp4)
            constexpr int j = sp - 14;
            unsigned hp;                                         // (one statement: between separate ones the compiler pads with s_nop)
            asm("v_cvt_pk_f16_f32 %0, %3, %4\n\tv_fma_mix_f32 %1, %0, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                : "=&v"(hp), "=&v"(elo0), "=&v"(elo1) : "v"(x[2 * j]), "v"(x[2 * j + 1]));
            ehi[j] = hp;
            f4lo = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(j == 0 ? 0u : f4lo, elo0, elo1, esc_lo, j);
        } else if constexpr (sp == 18 || sp == 19) {             // fp4 copy, two pairs per stage
            constexpr int j = (sp - 18) * 2;
            f4v = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(j == 0 ? 0u : f4v, x[2 * j], x[2 * j + 1], esc_v4, j);
            f4v = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(f4v, x[2 * j + 2], x[2 * j + 3], esc_v4, j + 1);
        } else if constexpr (sp == 20) {                         // the record: [fp16 x32 | fp4 residual | fp4 copy | scales]
            const int dst = k < 8 ? erow[k >> 1] + ldst[k & 1] : rem_dst;
            *(u32x4*)(smem + dst + piece * 12) = ehi;
            *(unsigned*)(smem + dst + 64) = f4lo;
            *(unsigned*)(smem + dst + 80) = f4v;
            // scales: byte 0 of the dword at +96 (residual block) / +100 (copy block).  Every lane of the quad holds both; each stores
            // ONE dword at +96 + 4 piece (pieces 2, 3 fill the pad with copies): no exec mask, no branch in the band
            *(unsigned*)(smem + dst + 96) = (piece & 1) ? e_v4 : e_lo;
        } else if constexpr (sp == 21) {
            load_item(std::integral_constant<int, k>{});         // the same item of the batch after next
        }
    };

    // ---- prime: rows 0..9 of the ring (input rows y0-1 .. y0+8), then the loads of the batch band 0 encodes ----------------
    // Two standard batches: rows -6..1 (rows < 0 land in slots 12..17, overwritten by the batch band 0 encodes) and rows 2..9.
    __syncthreads();                                             // the zero fill
    // (ring row i holds image row y0 - 1 + i in slot i mod 18; this wave's first row of a batch starting at i0 is i0 + jw)
    batch_rows(0, y0 - 1 - 6 + jw);
    sl_for<9>([&](auto K) __attribute__((always_inline)) { load_item(K); });
#pragma unroll 1
    for (int pb_ = 0; pb_ < 2; ++pb_) {
        const int i0 = pb_ ? 2 : -6;
        batch_rows(wrap(i0 + jw + SL_NR), y0 - 1 + i0 + 8 + jw);
        sl_for<9 * 24>([&](auto T) __attribute__((always_inline)) { stage(T); });       // (stage 21 of each item loads the next batch)
    }
    __builtin_amdgcn_s_waitcnt(sl_vmcnt_imm(0));                 // band 0 counts from an empty queue
    __syncthreads();
    int slot_e = 10 + jw, slot_r = 4 * rh;                      // ring slots of: the wave's first row of the batch band 0 encodes; its first halo row

    // ---- bands ------------------------------------------------------------------------------------------------------------
    f32x16 acc[4];
    const int lane_a = r * SL_PST + h * 16;                     // fp16 / fp4 operand reads
    const int lane_s = r * SL_PST + 96 + 4 * h;                 // scale dword of this lane's k-half (byte 0)
    const int cout = ct * 32 + r;
    const float bias = a.bias ? a.bias[cout] : 0.f;
    const unsigned lane_off = (unsigned)(4 * h * 64 + cout);
    const int E = (H >> 2) * a.tiles_x;
    float s1 = 0.f, s2 = 0.f;

    // The band body is ONE straight-line block, written in the order it should issue: a wave alone on its SIMD overlaps nothing
    // by itself -- whatever VALU / LDS / VMEM work stands BETWEEN two MFMAs in program order runs in the first one's shadow
    // (an MFMA holds the issue port 8 of its 32 cycles; ~5 more instructions fit), whatever stands behind a run of MFMAs waits
    // for all of them (measured with the stamps of tools/slide_lab.hip: the MFMA loop alone runs at 32.4 cycles per MFMA, and
    // every instruction placed behind it added its full issue time).  So, per band and wave (216 MFMAs):
    //   * two PHASES of two output rows each (108 MFMAs): the epilogue of a phase -- bias, stores, GroupNorm sums -- is spread
    //     over the MFMAs of the NEXT phase, one output register behind every third MFMA (the last phase of a band under the first
    //     phase of the next band; for band 0 that slot stores garbage into rows this wave overwrites one band later);
    //   * operand fragments are read one group (halo row, tap column) ahead of the MFMAs that consume them;
    //   * encoder stage t stands behind MFMA t (9 items x 24 stages = 216);
    //   * SL_SCHED pins that order (the machine scheduler otherwise re-clusters the MFMAs).
#ifdef CONV_SLIDE_NO_SCHED
#define SL_SCHED() do {} while (0)
#else
#define SL_SCHED() __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef SL_TRACE      // tools/slide_lab.hip: cycle stamps at the phase boundaries of the first 12 bands, per wave
#define SL_STAMP(ev) do { if (a.trace && lane == 0 && b < 12) a.trace[((size_t)blockIdx.x * 4 + wave) * 64 + b * 4 + (ev)] = (ev) == 2 ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter(); } while (0)
#else
#define SL_STAMP(ev) do {} while (0)
#endif
    // one output register of the deferred epilogue: rows 2 PH + {0, 1} of the wave's four, value v = 16 m + i
    float* orow[2] = {nullptr, nullptr};
    auto epi_value = [&](auto PH, auto V) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value, v = decltype(V)::value, m = v >> 4, i = v & 15;
        if constexpr (ph == 0 && v == 0) { s1 = 0.f; s2 = 0.f; }
        // (no inline asm with operands in the band: the compiler answers one with `s_waitcnt vmcnt(0)`, i.e. a wait for the loads
        //  and stores issued a moment ago.  The price: the GroupNorm sums, which nothing anchors, collect at the end of the phase.)
        const float x = acc[2 * ph + m][i] + bias;
        if (!(ABL & 2)) (orow[m] + ((i & 3) + 8 * (i >> 2)) * 64)[lane_off] = x;
        if (ST) { s1 += x; s2 = fmaf(x, x, s2); }
    };
    auto epi_stats = [&](int oy) __attribute__((always_inline)) {        // after rows 2, 3: the entry of the 4-row group starting at oy - 2
        if (!ST) return;
        const float t1 = s1 + __shfl_xor(s1, 32), t2 = s2 + __shfl_xor(s2, 32);
        const int e = ((oy - 2) >> 2) * a.tiles_x + tx;
        // (both lane halves hold the sums and store them to the same address: an `if (h == 0)` becomes a branch that cuts the band)
        *((float2*)a.stats + ((size_t)n * E + e) * 64 + cout) = make_float2(t1, t2);
    };

#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        SL_STAMP(0);
        batch_rows(slot_e, y0 - 1 + 8 * b + 18 + jw);
        int abase[6], sbase[6];                                 // LDS address of the wave's 6 halo rows (+ lane part): fragments / scale byte
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
            const int y = y0 - 1 + 8 * b + 4 * rh + pr;
            const int off = (y >= 0 && y < H) ? wrap(slot_r + pr) * SL_ROWB : SL_ZERO;
            abase[pr] = lane_a + off;
            sbase[pr] = lane_s + off;
        }
        const int oy0 = y0 + 8 * b + 4 * rh, oyp = y0 + 8 * (b > 0 ? b - 1 : 0) + 4 * rh + 2;   // this band's rows 0, 1; the previous band's rows 2, 3

        // operand fragments of a group, read SL_PF groups ahead of their MFMAs: a group is 3 or 6 MFMAs (96 / 192 cycles) and an
        // LDS read under this load takes 200-300, so one group ahead stalls every short group (measured: the band's phases then ADD)
        constexpr int NB = SL_PF + 1;
        u32x4 fa0[NB], fa1[NB], faq[NB];
        int fas[NB];
        auto read_group = [&](auto G) __attribute__((always_inline)) {
            constexpr int g = decltype(G)::value;                // 0..47: phase, chunk, tap column, halo row of the phase
            constexpr int ph = g / 24, c = (g % 24) / 12, kx = ((g % 24) / 4) % 3, pr = 2 * ph + g % 4;
            constexpr int off = c * SL_SEG + kx * SL_PST;
            // (issued in the REVERSE of the order the MFMAs consume them: LDS returns in order, so the wait of the group's first
            //  MFMA covers all four and the compiler emits one s_waitcnt per group instead of three)
            fas[g % NB] = *(const int*)(smem + sbase[pr] + off);
            SL_SCHED();
            faq[g % NB] = *(const u32x4*)(smem + abase[pr] + off + 64);
            SL_SCHED();
            fa1[g % NB] = *(const u32x4*)(smem + abase[pr] + off + 32);
            SL_SCHED();
            fa0[g % NB] = *(const u32x4*)(smem + abase[pr] + off);
            SL_SCHED();
        };
        sl_for<SL_PF>([&](auto G) __attribute__((always_inline)) { read_group(G); });
        SL_SCHED();
        sl_for<48>([&](auto G) __attribute__((always_inline)) {
            constexpr int g = decltype(G)::value;
            constexpr int ph = g / 24, c = (g % 24) / 12, kx = ((g % 24) / 4) % 3, p = g % 4;
            if constexpr (g % 24 == 0) {                         // the rows the deferred epilogue of this phase writes
                const int oy = ph == 0 ? oyp : oy0;
                orow[0] = a.out + ((size_t)(n * H + oy) * W + tx * 32) * 64;
                orow[1] = orow[0] + (size_t)W * 64;
            }
            if constexpr (g + SL_PF < 48) read_group(std::integral_constant<int, g + SL_PF>{});
            const f16x8 a0 = __builtin_bit_cast(f16x8, fa0[g % NB]), a1 = __builtin_bit_cast(f16x8, fa1[g % NB]);
            i32x8 a4 = {};
            a4[0] = faq[g % NB][0]; a4[1] = faq[g % NB][1]; a4[2] = faq[g % NB][2]; a4[3] = faq[g % NB][3];
            const int as = fas[g % NB];
            constexpr int cum = p == 0 ? 0 : (p == 1 ? 1 : (p == 2 ? 3 : 5));      // taps of the column in front of halo row p
            sl_for<3>([&](auto KY) __attribute__((always_inline)) {
                constexpr int ky = decltype(KY)::value;
                constexpr int ml = p - ky;                       // row of the phase this tap feeds
                if constexpr (ml >= 0 && ml < 2) {
                    constexpr int mt = 2 * ph + ml;
                    constexpr int tapi = ky - (p > 1 ? p - 1 : 0);
                    constexpr int u0 = ((c * 3 + kx) * 6 + cum + tapi) * 3;        // ordinal of this tap's first MFMA in the phase
                    i32x8 b4 = {};
                    b4[0] = wf[c][ky][kx][2][0]; b4[1] = wf[c][ky][kx][2][1]; b4[2] = wf[c][ky][kx][2][2]; b4[3] = wf[c][ky][kx][2][3];
                    constexpr bool first = c == 0 && kx == 0 && ky == 0;           // first product of this accumulator: C = 0
                    sl_for<3>([&](auto M) __attribute__((always_inline)) {
                        constexpr int m = decltype(M)::value;
                        f32x16 cin;
                        if constexpr (first && m == 0) { cin = f32x16{}; } else { cin = acc[mt]; }
                        if constexpr (m == 0)      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(f16x8, wf[c][ky][kx][0]), cin, 0, 0, 0);
                        else if constexpr (m == 1) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, __builtin_bit_cast(f16x8, wf[c][ky][kx][1]), cin, 0, 0, 0);
                        else acc[mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a4, b4, cin, 4, 4, 0, as, ky, (int)wsc[c][kx]);
                        constexpr int u = u0 + m, t = ph * 108 + u;
#if !defined(SL_GAPMODE) || SL_GAPMODE == 0
                        if constexpr (!(ABL & 1)) stage(std::integral_constant<int, t>{});
#else       // tools/slide_lab.hip: all three stages of a tap behind ONE of its MFMAs (1: the first fp16, 2: the second, 3: the fp4 one)
                        if constexpr (!(ABL & 1) && u % 3 == SL_GAPMODE - 1) {
                            stage(std::integral_constant<int, t - u % 3>{});
                            stage(std::integral_constant<int, t - u % 3 + 1>{});
                            stage(std::integral_constant<int, t - u % 3 + 2>{});
                        }
#endif
                        if constexpr (u % 3 == 1 && u / 3 < 32 && !(ABL & 4))
                            epi_value(std::integral_constant<int, ph ^ 1>{}, std::integral_constant<int, u / 3>{});
                        if constexpr (ph == 0 && u == 97) epi_stats(oyp);
                        SL_SCHED();
                    });
                }
            });
        });
#ifdef SL_ABL        // lab builds without the epilogue: keep every accumulator live (round 4's first lab runs lost the MFMAs of phase 0 to
                     // dead-code elimination -- the "MFMA only" figures of those runs are for half the MFMAs)
        if constexpr ((ABL & 4) != 0) asm volatile("" :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));
#endif
        slot_e = wrap(slot_e + 8); slot_r = wrap(slot_r + 8);
        SL_STAMP(1);
        SL_STAMP(2);
        __syncthreads();
        SL_STAMP(3);
    }
    // rows 2, 3 of the last band
    {
        const int oy = y0 + 8 * (nb - 1) + 4 * rh + 2;
        orow[0] = a.out + ((size_t)(n * H + oy) * W + tx * 32) * 64;
        orow[1] = orow[0] + (size_t)W * 64;
        sl_for<32>([&](auto V) __attribute__((always_inline)) { epi_value(std::integral_constant<int, 1>{}, V); });
        epi_stats(oy);
    }
}

template <int PRO, bool ST>
int launch_slide(const ConvArgs& a, int rows_wg, hipStream_t st) {
    auto kern = conv3x3_slide64_kernel<PRO, ST>;
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return TEDM_EHIP;
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return TEDM_EHIP;
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    const int grid = a.B * (a.H / rows_wg) * a.tiles_x;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)SL_LDS, st, a, rows_wg);
    return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
}

}  // namespace

// Does the sliding-window kernel take this launch?  (conv_igemm.hip asks before it picks a tile.)
bool conv_slide_supported(const ConvArgs& a, int KH, int stride, int ups, int pro) {
    if (!g_conv_slide || a.prec != TEDM_PREC_F16F4 || KH != 3 || stride != 1 || ups) return false;
    if (a.Cout != 64 || a.C1 != 64 || a.x2 || a.C2) return false;
    if (pro != PRO_NONE && pro != PRO_GN) return false;
    if (a.add || a.out_q16 || a.add_q16 || a.stats_mode == STATS_MAX) return false;
    if (a.H != a.Ho || a.W != a.Wo || a.W % 32 || a.H % 8 || a.H < 16) return false;
    return true;
}

void conv_slide_label(const ConvArgs& a, int pro, char* buf, int cap) {
    snprintf(buf, cap, "conv3x3_slide64_kernel<%d,%s>", pro, a.stats_mode == STATS_SUMSQ ? "true" : "false");
}

int conv_slide_launch(const ConvArgs& a_in, int pro, hipStream_t st) {
    ConvArgs a = a_in;
    a.tiles_x = a.W / 32;
    // rows per workgroup: conv_slide_common.h sl_pick_rows
    int rows = g_conv_slide_rows;
    if (rows <= 0 || rows % 8 || a.H % rows) {
        rows = sl_pick_rows(a.B, a.H, a.tiles_x);
    }
    const bool st_on = a.stats_mode == STATS_SUMSQ;
    if (pro == PRO_GN) return st_on ? launch_slide<PRO_GN, true>(a, rows, st) : launch_slide<PRO_GN, false>(a, rows, st);
    return st_on ? launch_slide<PRO_NONE, true>(a, rows, st) : launch_slide<PRO_NONE, false>(a, rows, st);
}
