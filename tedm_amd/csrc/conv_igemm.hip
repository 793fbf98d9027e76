// conv_igemm.hip -- NHWC implicit-GEMM convolution on gfx950 MFMA (v_mfma_f32_32x32x16_bf16).
//
// Replaces every nn.Conv2d on the UNet hot path except the 1-channel stem:
//   Block.proj 3x3            models/unet_model.py:122,127
//   Downsample 4x4 stride 2   models/unet_model.py:47-49
//   Upsample nearest x2 + 3x3 models/unet_model.py:39-44   (upsample fused into the halo gather)
//   last-level 3x3            models/unet_model.py:307-308,323-324
//   res_conv / to_qkv / to_out 1x1   models/unet_model.py:156-159,185,187-190,221-222
//
// GEMM view: D[pixel][cout] = sum_{tap,cin} A[pixel+tap][cin] * W[cout][cin][tap].
//   * One workgroup (4 waves) owns TR rows x 32 columns of output pixels of one image and
//     WN*32 output channels.  An MFMA M-tile is one 32-pixel row segment, so the A fragment of
//     tap (ky,kx) is the same LDS image shifted by a compile-time byte offset.
//   * Activations stay fp32 in HBM (the 1e-3 eps gate rules out bf16 storage);
//     while a 16*KS-channel chunk of the haloed input tile is staged into LDS the prologue of
//     the consumer is applied (GroupNorm-apply+FiLM+SiLU folded to silu(a*x+b), or channel
//     LayerNorm), the value is split into bf16 hi / bf16 lo planes, and the chunk is double
//     buffered against the MFMA loop.  LDS pixel stride is (2*KS+1)*16 B: odd multiple of 16 B
//     => ds_read_b128 of 16 consecutive pixels is bank-conflict free without address math.
//   * Weights are pre-packed (pack.hip) in MFMA B-fragment order, hi/lo planes, so each wave
//     fetches its own 32-cout slice straight from L2 with one coalesced 16-B load per lane per
//     fragment: weights never touch LDS and are never re-read by the same workgroup.
//   * bf16x3: acc += Ahi*Bhi + Alo*Bhi + Ahi*Blo (fp32 accumulate in the MFMA).
//   * Epilogue: +bias, optional (GN+SiLU'd) addend for the resnet residual, fp32 NHWC store in
//     128-B segments, and per-output-channel sum/sumsq (GroupNorm, l2norm) or max (k-softmax)
//     partials written to a deterministic slab (no atomics).
#include "common.h"
#include <stdio.h>
#include <string.h>
#include <type_traits>
#include "../../include/tedm_hip.h"

namespace {

template <int KH, int KW, int STRIDE, int WM, int MT, int KS, int NBUF, int PREC, int NTHR = 256>
struct Geo {
    static constexpr int TR = WM * MT;                       // output rows per tile
    static constexpr int HR = (TR - 1) * STRIDE + KH;        // halo rows / cols
    static constexpr int HC = 31 * STRIDE + KW;
    static constexpr int HR2 = (HR + 1) / 2, HC2 = (HC + 1) / 2;
    static constexpr int PLANE2 = HR2 * HC2;                 // stride-2: one parity plane
    static constexpr int NPIX = STRIDE == 1 ? HR * HC : 4 * PLANE2;
    static constexpr int PSTRIDE = KS * 32 + 16;             // bytes / pixel / precision plane
    static constexpr int REGION = NPIX * PSTRIDE;
    static constexpr int NPL = PREC == TEDM_PREC_BF16X3 ? 2 : 1;           // LDS planes of the activations
    static constexpr int BUFSZ = REGION * NPL;
    static constexpr int LDS = BUFSZ * NBUF;
    static constexpr int PIECES = HR * HC * 2 * KS;          // (halo pixel, 8-channel piece)
    static constexpr int NIT = (PIECES + NTHR - 1) / NTHR;
};

template <int KH, int KW, int STRIDE, int PAD, int UPS, int WM, int WN, int MT, int KS, int NBUF,
          int PRO, int PREC>
__global__ __launch_bounds__(WM * WN * 64, (MT * STRIDE >= 8 ? 1 : 2)) void conv_igemm_kernel(ConvArgs a) {
    constexpr int NTHR = WM * WN * 64;
    using G = Geo<KH, KW, STRIDE, WM, MT, KS, NBUF, PREC, NTHR>;
    constexpr int T = KH * KW;
    constexpr int NPL = G::NPL;                                  // activation planes (hi[,lo])
    constexpr int NPB = PREC == TEDM_PREC_BF16 ? 1 : 2;          // weight planes fetched (hi[,lo])
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // Two workgroups share a CU, i.e. two waves share each SIMD's matrix pipe.  Left alone they run in
    // lockstep (both in their MFMA segment at half speed, then both in staging / epilogue with the pipe idle:
    // measured with the diagnostic stamps).  A static priority for the wave in the even hardware slot turns the
    // pair into a pipeline: the prioritised wave runs its MFMA segments at full rate and the other one fills
    // the pipe during its partner's staging, barrier and epilogue phases.
    if (a.prio_mode == 1) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);      // HW_ID.wave_id
        if ((slot & 1) == 0) __builtin_amdgcn_s_setprio(1);
    } else if (a.prio_mode == 2) {
        if ((blockIdx.x & 1) == 0) __builtin_amdgcn_s_setprio(1);
    }

    // Block order: 1-D grid over (group of a.tpw consecutive spatial tiles, cout-tile group).  Workgroups
    // that read the SAME input tiles (different cout groups) get block ids 8 apart -- the dispatcher deals
    // blocks round-robin over the 8 XCDs, so they land on one XCD back to back and share the halo through
    // its L2 (speed only).  A workgroup walks its tiles with the (tile, chunk) loop flattened: the first
    // chunk of the next tile is in flight while the current tile's epilogue stores.
    const int tpi = a.tiles_x * a.tiles_y;
    const int ntile = a.B * tpi, ny = a.ny;
    const int nsuper = (ntile + a.tpw - 1) / a.tpw;
    int super, cty;
    {
        const int id = blockIdx.x;
        if (a.xcd_order && (nsuper & 7) == 0) {
            const int xcd = id & 7, slot = id >> 3;
            cty = slot % ny;
            super = (slot / ny) * 8 + xcd;
        } else {
            cty = id % ny;
            super = id / ny;
        }
    }
    const int tile_beg = super * a.tpw;
    const int tile_cnt = (tile_beg + a.tpw <= ntile) ? a.tpw : ntile - tile_beg;
    const int CT = (a.Cout + 31) >> 5;
    const int ct = cty * WN + wn;
    const bool active = ct < CT;                              // wave-uniform
    const int Cin = a.C1 + a.C2;
    const int Hin = UPS ? a.H * 2 : a.H, Win = UPS ? a.W * 2 : a.W;   // UPS 1: nearest x2, UPS 2: zero insertion x2

    // ---- per-thread staging geometry of the tile being STAGED (same for all its channel chunks) ----
    int gp[G::NIT];        // >= 0: valid source pixel, -1 = zero padding, -2 = no piece
    int gpc[G::NIT];       // source pixel index clamped into the image (what the unconditional loads read)
    int lofs[G::NIT];      // LDS byte offset of this piece inside a precision plane
    float ln_r[G::NIT], ln_m[G::NIT];
    const int piece = tid % (2 * KS);
    int ns = 0;            // image of the tile being staged
    auto setup_tile = [&](int tl) {
        ns = tl / tpi;
        const int trem = tl - ns * tpi;
        const int tys = trem / a.tiles_x, txs = trem - tys * a.tiles_x;
        const int iy0 = tys * G::TR * STRIDE - PAD, ix0 = txs * 32 * STRIDE - PAD;
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            const int i = tid + it * NTHR;
            const int pl = i / (2 * KS);
            gp[it] = -2; gpc[it] = ns * a.H * a.W; lofs[it] = 0; ln_r[it] = 0.f; ln_m[it] = 0.f;
            if (pl < G::HR * G::HC) {
                const int hy = pl / G::HC, hx = pl - hy * G::HC;
                const int iy = iy0 + hy, ix = ix0 + hx;
                const bool valid = iy >= 0 && iy < Hin && ix >= 0 && ix < Win && (UPS != 2 || !((iy | ix) & 1));
                const int sy = UPS ? (iy >> 1) : iy, sx = UPS ? (ix >> 1) : ix;
                const int syc = sy < 0 ? 0 : (sy >= a.H ? a.H - 1 : sy), sxc = sx < 0 ? 0 : (sx >= a.W ? a.W - 1 : sx);
                gpc[it] = (ns * a.H + syc) * a.W + sxc;
                gp[it] = valid ? gpc[it] : -1;
                const int lpix = STRIDE == 1 ? pl
                    : ((hy & 1) * 2 + (hx & 1)) * G::PLANE2 + (hy >> 1) * G::HC2 + (hx >> 1);
                lofs[it] = lpix * G::PSTRIDE + piece * 16;
                if (PRO == PRO_LN) {
                    const float2 st2 = *(const float2*)(a.pro_b + (size_t)gpc[it] * 2);
                    ln_r[it] = st2.x; ln_m[it] = st2.y;
                }
            }
        }
    };

    float raw[G::NIT][8];
    float pa[8], pb[8];

    auto issue_loads = [&](int c) {
        const int cb = c * (16 * KS) + piece * 8;             // channel in the virtual concat
        const float* src; int cs, ch;
        if (cb < a.C1) { src = a.x1; cs = a.C1; ch = cb; }
        else           { src = a.x2; cs = a.C2; ch = cb - a.C1; }
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            // unconditional loads (padding / unused pieces read the clamped pixel gpc[it] and are zeroed when
            // staged): VMEM inside lane-divergent branches mis-synchronised on gfx950, see linattn_fused.hip
            const float4* p = (const float4*)(src + (size_t)gpc[it] * cs + ch);
            const float4 v0 = p[0], v1 = p[1];
            raw[it][0] = v0.x; raw[it][1] = v0.y; raw[it][2] = v0.z; raw[it][3] = v0.w;
            raw[it][4] = v1.x; raw[it][5] = v1.y; raw[it][6] = v1.z; raw[it][7] = v1.w;
        }
        if (PRO == PRO_GN) {
            const float4* qa = (const float4*)(a.pro_a + (size_t)ns * Cin + cb);
            const float4* qb = (const float4*)(a.pro_b + (size_t)ns * Cin + cb);
            const float4 a0 = qa[0], a1 = qa[1], b0 = qb[0], b1 = qb[1];
            pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w;
            pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
            pb[0] = b0.x; pb[1] = b0.y; pb[2] = b0.z; pb[3] = b0.w;
            pb[4] = b1.x; pb[5] = b1.y; pb[6] = b1.z; pb[7] = b1.w;
        } else if (PRO == PRO_LN) {
            const float4* qa = (const float4*)(a.pro_a + cb);
            const float4 a0 = qa[0], a1 = qa[1];
            pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w;
            pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
        }
    };

    auto write_lds = [&](int buf) {
        char* base = smem + buf * G::BUFSZ;
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
            if (gp[it] == -2) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = raw[it][j];
                if (PRO == PRO_GN) x = silu_f(fmaf(pa[j], x, pb[j]));
                else if (PRO == PRO_LN) x = pa[j] * (x * ln_r[it] - ln_m[it]);
                v[j] = gp[it] >= 0 ? x : 0.f;              // zero padding applies AFTER the prologue
            }
            if (PREC == TEDM_PREC_FP16X2) {
                *(uint4*)(base + lofs[it]) = cvt8_f16(v);
            } else {
                uint4 hi, lo;
                split8(v, hi, lo);
                *(uint4*)(base + lofs[it]) = hi;
                if (NPL == 2) *(uint4*)(base + G::REGION + lofs[it]) = lo;
            }
        }
    };

    f32x16 acc[MT];
    const uint4* wq = (const uint4*)a.wpk;
    const int lane_a = (STRIDE == 1 ? (wm * MT * G::HC + r) : (wm * MT * G::HC2 + r)) * G::PSTRIDE + h * 16;
    const int NC = a.nk16 / KS;

    // Weight fragments are fetched one or two tap-column groups (all ky of one kx) AHEAD of the MFMAs that
    // consume them; the group's registers are bq[set].  Sets are indexed statically (kx, or the chunk
    // parity for 1x1) so they stay in VGPRs.
    constexpr int NSET = KW == 3 ? 3 : 2;   // 4x4: two sets, one tap column ahead (four sets of 4x4 fragments = 128 VGPRs spilled)
    uint4 bq[NSET][KH][KS][NPB];

    // Inactive waves (cout group past the end) run the same straight-line code on the last valid group and
    // skip only the epilogue: any branch inside the chunk loop splits it into basic blocks, and the waitcnt
    // pass then guards every MFMA group with `s_waitcnt vmcnt(0)`, which throws the prefetch distance away.
    const int ctw = active ? ct : CT - 1;
    auto load_b = [&](int c, int kx, uint4 (&b)[KH][KS][NPB]) {
#pragma unroll
        for (int ky = 0; ky < KH; ++ky)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int p = 0; p < NPB; ++p)   // packed layout always carries both planes (hi, lo)
                    b[ky][ks][p] = wq[((((size_t)ctw * a.nk16 + (c * KS + ks)) * T + (ky * KW + kx)) * 2 + p) * 64 + lane];
    };

    // All MFMAs of tap column kx for one staged chunk.  Each halo row is read from LDS once and feeds
    // every ky whose output row it belongs to.
    auto compute_kx = [&](int buf, int kx, const uint4 (&b)[KH][KS][NPB]) {
        const char* sb = smem + buf * G::BUFSZ + lane_a;
        constexpr int NPR = MT + (KH - 1) / STRIDE;           // plane rows touched by one row parity
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int py = 0; py < STRIDE; ++py)
#pragma unroll
                for (int pr = 0; pr < NPR; ++pr) {
                    const int poff = STRIDE == 1
                        ? (pr * G::HC + kx)
                        : ((py * 2 + (kx & 1)) * G::PLANE2 + pr * G::HC2 + (kx >> 1));
                    const int off = poff * G::PSTRIDE + ks * 32;
                    const uint4 ahq = *(const uint4*)(sb + off);
                    uint4 alq = ahq;
                    if (NPL == 2) alq = *(const uint4*)(sb + off + G::REGION);
#pragma unroll
                    for (int ky = py; ky < KH; ky += STRIDE) {
                        const int mt = pr - ky / STRIDE;
                        if (mt < 0 || mt >= MT) continue;
                        if (PREC == TEDM_PREC_FP16X2) {
                            const f16x8 af = __builtin_bit_cast(f16x8, ahq);
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(f16x8, b[ky][ks][0]), acc[mt], 0, 0, 0);
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(f16x8, b[ky][ks][NPB - 1]), acc[mt], 0, 0, 0);
                        } else {
                            const bf16x8 ah = __builtin_bit_cast(bf16x8, ahq);
                            const bf16x8 bh = __builtin_bit_cast(bf16x8, b[ky][ks][0]);
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mt], 0, 0, 0);
                            if (NPL == 2) {
                                const bf16x8 al = __builtin_bit_cast(bf16x8, alq);
                                const bf16x8 bl = __builtin_bit_cast(bf16x8, b[ky][ks][NPB - 1]);
                                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mt], 0, 0, 0);
                                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mt], 0, 0, 0);
                            }
                        }
                    }
                }
    };

    // ---- epilogue of one finished tile ------------------------------------------------------------
    // acc[mt][i]: pixel column ox0 + (i&3) + 8*(i>>2) + 4*h of output row oy0 + wm*MT + mt, channel
    // ct*32 + r.  A store instruction therefore writes two 128-B segments (32 channels x 2 pixels).
    const int cout = ct * 32 + r;
    const bool cok = active && cout < a.Cout;
    const float bias = (a.bias && cok) ? a.bias[cout] : 0.f;
    const int cstride = a.Cout;
    auto epilogue = [&](int tl) {
        if (!active) return;
#ifdef TEDM_DIAG_TRACE
        if (a.prio_mode == 3) return;                          // diagnostic: what the tile costs without its epilogue
#endif
        const int n = tl / tpi;
        const int trem = tl - n * tpi;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int oy0 = ty * G::TR, ox0 = tx * 32;
        float aa = 1.f, ab = 0.f;
        if (a.add_a && cok) { aa = a.add_a[(size_t)n * a.Cout + cout]; ab = a.add_b[(size_t)n * a.Cout + cout]; }
        float s1 = 0.f, s2 = 0.f, smax = -INFINITY;
        const bool full_cols = ox0 + 32 <= a.Wo;               // block-uniform
        // Addresses are (uniform row pointer + uniform pixel step) + a 32-bit per-lane offset, so a store is one
        // instruction with a scalar base and no per-lane 64-bit arithmetic; the statistics are only accumulated when a
        // slab was asked for.  (The epilogue is 15 % of a 64-channel conv's workgroup time: every VALU slot counts.)
        const unsigned lane_off = (unsigned)(4 * h * cstride + cout);
        auto emit = [&](auto mode, auto with_stats) {
            constexpr int MODE = decltype(mode)::value;       // 0: no addend, 1: plain addend, 2: silu(aa*add+ab)
            constexpr bool ST = decltype(with_stats)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int oy = oy0 + wm * MT + mt;
                if (oy >= a.Ho) continue;                      // wave-uniform
                const size_t rowbase = ((size_t)(n * a.Ho + oy) * a.Wo + ox0) * cstride;      // uniform
                float* op = a.out + rowbase;
                const float* ap = MODE ? a.add + rowbase : nullptr;
                if (full_cols) {
                    if (!cok) continue;
                    float ad[16];
                    if (MODE) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) ad[i] = (ap + (size_t)((i & 3) + 8 * (i >> 2)) * cstride)[lane_off];
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float v = acc[mt][i] + bias;
                        if (MODE == 1) v += ad[i];
                        if (MODE == 2) v += silu_f(fmaf(aa, ad[i], ab));
                        (op + (size_t)((i & 3) + 8 * (i >> 2)) * cstride)[lane_off] = v;
                        if (ST) { s1 += v; s2 = fmaf(v, v, s2); smax = fmaxf(smax, v); }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int dx = (i & 3) + 8 * (i >> 2);
                        if (ox0 + 4 * h + dx < a.Wo && cok) {
                            float v = acc[mt][i] + bias;
                            if (MODE) {
                                float t = (ap + (size_t)dx * cstride)[lane_off];
                                if (MODE == 2) t = silu_f(fmaf(aa, t, ab));
                                v += t;
                            }
                            (op + (size_t)dx * cstride)[lane_off] = v;
                            if (ST) { s1 += v; s2 = fmaf(v, v, s2); smax = fmaxf(smax, v); }
                        }
                    }
                }
            }
        };
        const bool st_on = a.stats_mode != STATS_NONE;
        if (!a.add) { if (st_on) emit(std::integral_constant<int, 0>{}, std::true_type{}); else emit(std::integral_constant<int, 0>{}, std::false_type{}); }
        else if (!a.add_a) emit(std::integral_constant<int, 1>{}, std::false_type{});      // addend and statistics never meet on this path
        else emit(std::integral_constant<int, 2>{}, std::false_type{});

        if (a.stats_mode != STATS_NONE) {
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); smax = fmaxf(smax, __shfl_xor(smax, 32));
            if (h == 0 && cok) {
                const int E = tpi * WM;
                const int e = (ty * a.tiles_x + tx) * WM + wm;
                if (a.stats_mode == STATS_SUMSQ) {
                    float2* p = (float2*)a.stats + ((size_t)n * E + e) * a.Cout + cout;
                    *p = make_float2(s1, s2);
                } else {
                    a.stats[((size_t)n * E + e) * a.Cout + cout] = smax;
                }
            }
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
    };

#ifdef TEDM_DIAG_TRACE
    auto stamp = [&](int ev) {
        if (a.trace && lane == 0 && blockIdx.x < 2048 && ev < 64)
            a.trace[((size_t)blockIdx.x * 4 + (wave & 3)) * 64 + ev] = __builtin_readcyclecounter();
    };
    if (a.trace && lane == 0 && blockIdx.x < 2048) {
        stamp(0);
        a.trace[((size_t)blockIdx.x * 4 + (wave & 3)) * 64 + 1] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
    }
#else
    auto stamp = [&](int) {};
#endif
    if (NBUF == 2) {
        // weight groups are fetched PD groups ahead (two tap columns ~ 2 x 36 MFMAs of cover for the L2
        // latency); the sequence simply wraps to chunk 0 at a tile boundary (same weights, next tile)
        constexpr int PD = KW == 3 ? 2 : 1;
        const int total = tile_cnt * NC;                      // flattened (tile, chunk) iterations
        // VMEM returns in order (vmcnt): the next chunk's ACTIVATION loads are issued after the last weight
        // group this chunk still has to wait for (the kx=0 prefetch of tap column PD), so no MFMA of this
        // chunk waits behind them; they only have to land by write_lds at the end of the chunk.
        auto chunk_body = [&](int c, int buf, int c_next, bool stage_next) {
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
                const int kx2 = (kx + PD) % KW;
                int c2 = c + (kx + PD) / KW;
                if (c2 >= NC) c2 -= NC;                        // wraps to the next tile's chunk 0 (same weights); at the very
                load_b(c2, kx2, bq[kx2 % NSET]);               // end of the workgroup the wrapped group is loaded and unused
                if (kx == (KW > PD ? KW - 1 - PD : 0) && stage_next) issue_loads(c_next);
                compute_kx(buf, kx, bq[kx % NSET]);
            }
        };
        setup_tile(tile_beg);
        load_b(0, 0, bq[0]);
        if (PD == 2) load_b(0, 1 % KW, bq[1 % NSET]);
        issue_loads(0);
        write_lds(0);
        __syncthreads();
        stamp(2);
        int c = 0, tl = tile_beg;
        for (int it = 0; it < total; ++it) {
            const bool more = it + 1 < total;
            const bool last_chunk = c == NC - 1;
            if (more && last_chunk) setup_tile(tl + 1);       // the next iteration stages a new tile
            if (c == 0) zero_acc();
            stamp(4 + 4 * it);
            chunk_body(c, it & 1, last_chunk ? 0 : c + 1, more);
            stamp(5 + 4 * it);
            if (more) write_lds((it + 1) & 1);
            stamp(6 + 4 * it);
            __syncthreads();
            stamp(7 + 4 * it);
            if (last_chunk) { epilogue(tl); ++tl; c = 0; } else ++c;
        }
        stamp(3);
    } else {
        // single LDS buffer (1x1, HBM-bound): keep registers low so 3-4 workgroups share a CU and
        // overlap each other's load / MFMA phases; the chunk's weights travel with its activations.
        for (int tl = tile_beg; tl < tile_beg + tile_cnt; ++tl) {
            setup_tile(tl);
            zero_acc();
            for (int c = 0; c < NC; ++c) {
                load_b(c, 0, bq[0]);
                issue_loads(c);
                __syncthreads();
                write_lds(0);
                __syncthreads();
                compute_kx(0, 0, bq[0]);
            }
            epilogue(tl);
        }
    }
}

struct TileSel { int wm, wn, mt; };

// Tile shapes (rows = wm*mt, always 32 columns, wn*32 output channels):
//   id 0 "wide8"  : 1x4 waves, 8 M-tiles -> 256 px x 128 cout, 1 wave/SIMD (uses >256 VGPRs)
//   id 1 "narrow" : 2x2 waves, 4 M-tiles -> 256 px x  64 cout
//   id 2 "wide4"  : 1x4 waves, 4 M-tiles -> 128 px x 128 cout
// stride-2 convs use half the rows (the halo doubles).
//   id 3 "wide8w"  : 2x4 waves (512 threads), 4 M-tiles -> 256 px x 128 cout
int g_tile_wide = 2;      // tile id used when Cout >= 128 (tedm_set_tuning("conv_tile_wide", v))
int g_tile_narrow = 1;
int g_prio_mode = 0;      // tedm_set_tuning("conv_prio", 0|1|2)
int g_solo = 0;           // tedm_set_tuning("conv_solo", 1): pad the LDS request so that workgroups do not share a CU (diagnostic)
int g_tpw_max = 1;        // max spatial tiles per workgroup (tedm_set_tuning("conv_tpw_max", v))
int g_xcd_order = 1;      // 1: cout-tile groups of one input tile share an XCD (tedm_set_tuning("conv_xcd_order", v))    // tile id used when Cout <  128 (tedm_set_tuning("conv_tile_narrow", v))

inline int select_tile_id(int Cout) { return Cout >= 128 ? g_tile_wide : g_tile_narrow; }
inline TileSel tile_geom(int id, int stride) {
    TileSel t;
    switch (id) {
        case 0: t = TileSel{1, 4, 8}; break;
        case 1: t = TileSel{2, 2, 4}; break;
        case 2: t = TileSel{1, 4, 4}; break;
        case 3: t = TileSel{2, 4, 4}; break;
        default: t = TileSel{2, 4, 4}; break;
    }
    if (stride == 2) t.mt = t.mt > 1 ? t.mt / 2 : t.mt;
    return t;
}

template <int KH, int KW, int STRIDE, int PAD, int UPS, int WM, int WN, int MT, int KS, int NBUF, int PRO, int PREC>
int launch_one(const ConvArgs& a, hipStream_t st) {
    using G = Geo<KH, KW, STRIDE, WM, MT, KS, NBUF, PREC, WM * WN * 64>;
    auto kern = conv_igemm_kernel<KH, KW, STRIDE, PAD, UPS, WM, WN, MT, KS, NBUF, PRO, PREC>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return TEDM_EHIP;
        attr_set = true;
    }
    const int CT = (a.Cout + 31) / 32;
    ConvArgs b = a;
    b.ny = (CT + WN - 1) / WN;
    b.xcd_order = g_xcd_order;
    // tiles per workgroup: keep >= ~2048 workgroups in the grid, at most g_tpw_max tiles each (NBUF==2 path)
    const int ntile = a.B * a.tiles_x * a.tiles_y;
    int tpw = NBUF == 2 ? (int)(((long long)ntile * b.ny) / 2048) : 1;
    if (tpw > g_tpw_max) tpw = g_tpw_max;
    if (tpw < 1) tpw = 1;
    b.tpw = tpw;
    b.prio_mode = g_prio_mode;
    dim3 grid(((ntile + tpw - 1) / tpw) * b.ny);
    const size_t lds = g_solo ? (size_t)(96 * 1024 > G::LDS ? 96 * 1024 : G::LDS) : (size_t)G::LDS;   // conv_solo: one workgroup per CU (diagnostic)
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, b);
    return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
}

template <int KH, int KW, int STRIDE, int PAD, int UPS, int KS, int NBUF, int PRO, int PREC>
int launch_tile(const ConvArgs& a, int tile_id, hipStream_t st) {
    constexpr int D = STRIDE == 2 ? 2 : 1;
    switch (tile_id) {
        case 0:  return launch_one<KH, KW, STRIDE, PAD, UPS, 1, 4, 8 / D, KS, NBUF, PRO, PREC>(a, st);
        case 1:  return launch_one<KH, KW, STRIDE, PAD, UPS, 2, 2, 4 / D, KS, NBUF, PRO, PREC>(a, st);
        case 2:  return launch_one<KH, KW, STRIDE, PAD, UPS, 1, 4, 4 / D, KS, NBUF, PRO, PREC>(a, st);
        default: return launch_one<KH, KW, STRIDE, PAD, UPS, 2, 4, 4 / D, KS, NBUF, PRO, PREC>(a, st);
    }
}

template <int PREC>
int launch_prec(const ConvArgs& a, int KH, int stride, int ups, int pro, int tile_id, hipStream_t st) {
    if (KH == 3 && stride == 1 && !ups && pro == PRO_NONE) return launch_tile<3, 3, 1, 1, 0, 1, 2, PRO_NONE, PREC>(a, tile_id, st);
    if (KH == 3 && stride == 1 && !ups && pro == PRO_GN)   return launch_tile<3, 3, 1, 1, 0, 1, 2, PRO_GN, PREC>(a, tile_id, st);
    if (KH == 3 && stride == 1 && ups == 1 && pro == PRO_NONE) return launch_tile<3, 3, 1, 1, 1, 1, 2, PRO_NONE, PREC>(a, tile_id, st);
    if (KH == 4 && stride == 2 && !ups && pro == PRO_NONE) return launch_tile<4, 4, 2, 1, 0, 1, 2, PRO_NONE, PREC>(a, tile_id, st);
    // dgrad of the stride-2 Downsample: 4x4 stride-1 pad-2 conv over the zero-inserted gradient
    if (KH == 4 && stride == 1 && ups == 2 && pro == PRO_NONE) return launch_tile<4, 4, 1, 2, 2, 1, 2, PRO_NONE, PREC>(a, tile_id, st);
    if (KH == 1 && stride == 1 && !ups) {
        if (pro == PRO_NONE) return launch_tile<1, 1, 1, 0, 0, 2, 1, PRO_NONE, PREC>(a, tile_id, st);
        if (pro == PRO_LN)   return launch_tile<1, 1, 1, 0, 0, 2, 1, PRO_LN, PREC>(a, tile_id, st);
    }
    return TEDM_EINVAL;
}

}  // namespace

int conv_set_tuning(const char* key, int v) {
    if (!strcmp(key, "conv_tile_wide") && v >= 0 && v <= 3) { g_tile_wide = v; return TEDM_OK; }
    if (!strcmp(key, "conv_prio") && v >= 0 && v <= 3) { g_prio_mode = v; return TEDM_OK; }
    if (!strcmp(key, "conv_solo") && (v == 0 || v == 1)) { g_solo = v; return TEDM_OK; }
    if (!strcmp(key, "conv_tpw_max") && v >= 1 && v <= 64) { g_tpw_max = v; return TEDM_OK; }
    if (!strcmp(key, "conv_xcd_order") && (v == 0 || v == 1)) { g_xcd_order = v; return TEDM_OK; }
    if (!strcmp(key, "conv_tile_narrow") && v >= 0 && v <= 3) { g_tile_narrow = v; return TEDM_OK; }
    return TEDM_EINVAL;
}

// Label == the rocprof kernel symbol (template argument list of conv_igemm_kernel).
void conv_kernel_label(const ConvArgs& a, int KH, int stride, int ups, int pro, char* buf, int cap) {
    const int id = select_tile_id(a.Cout);
    const TileSel t = tile_geom(id, stride);
    const int pad = KH == 1 ? 0 : (ups == 2 ? 2 : 1), ks = KH == 1 ? 2 : 1, nbuf = KH == 1 ? 1 : 2;
    snprintf(buf, cap, "conv_igemm_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d>", KH, KH, stride, pad, ups, t.wm, t.wn, t.mt,
             ks, nbuf, pro, a.prec);
}

int conv_stats_entries(int KH, int stride, int Cout, int Ho, int Wo) {
    (void)KH;
    const TileSel t = tile_geom(select_tile_id(Cout), stride);
    const int TR = t.wm * t.mt;
    return ((Ho + TR - 1) / TR) * ((Wo + 31) / 32) * t.wm;
}

size_t conv_packed_bytes(int Cout, int Cin, int taps) {
    const size_t CT = (Cout + 31) / 32;
    return CT * (size_t)(Cin / 16) * taps * 2 /*hi,lo*/ * 1024;
}

// a.tiles_x / a.tiles_y / a.nk16 are filled here.
int conv_launch(const ConvArgs& a_in, int KH, int stride, int ups, int pro, hipStream_t st) {
    ConvArgs a = a_in;
    const int Cin = a.C1 + a.C2;
    if (Cin % 16 || a.C1 % 16 || (a.x2 && a.C2 % 16)) return TEDM_EINVAL;
    if (KH == 1 && (a.C1 % 32 || a.C2 % 32)) return TEDM_EINVAL;   // 1x1 stages 32-channel chunks
    const int tile_id = select_tile_id(a.Cout);
    const TileSel t = tile_geom(tile_id, stride);
    const int TR = t.wm * t.mt;
    a.tiles_y = (a.Ho + TR - 1) / TR;
    a.tiles_x = (a.Wo + 31) / 32;
    a.nk16 = Cin / 16;
    if (a.prec == TEDM_PREC_BF16) return launch_prec<TEDM_PREC_BF16>(a, KH, stride, ups, pro, tile_id, st);
    if (a.prec == TEDM_PREC_FP16X2) return launch_prec<TEDM_PREC_FP16X2>(a, KH, stride, ups, pro, tile_id, st);
    return launch_prec<TEDM_PREC_BF16X3>(a, KH, stride, ups, pro, tile_id, st);
}
