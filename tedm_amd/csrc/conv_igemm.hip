// conv_igemm.hip -- host side of the NHWC implicit-GEMM convolution: tile table, tuning keys and the dispatch to the
// per-arithmetic-mode translation units (conv_igemm_p*.hip, each an instance of conv_igemm.inc).
#include "common.h"
#include <stdio.h>
#include <string.h>
#include "../../include/tedm_hip.h"

int conv_launch_prec0(const ConvArgs& a, int KH, int stride, int ups, int pro, int tile_id, hipStream_t st);
int conv_launch_prec1(const ConvArgs& a, int KH, int stride, int ups, int pro, int tile_id, hipStream_t st);
int conv_launch_prec2(const ConvArgs& a, int KH, int stride, int ups, int pro, int tile_id, hipStream_t st);
int conv_launch_prec3(const ConvArgs& a, int KH, int stride, int ups, int pro, int tile_id, hipStream_t st);
int conv_launch_prec4(const ConvArgs& a, int KH, int stride, int ups, int pro, int tile_id, hipStream_t st);

// conv_slide.hip: sliding-window kernel of the full-resolution 64-channel 3x3 convs (TEDM_PREC_F16F4)
bool conv_slide_supported(const ConvArgs& a, int KH, int stride, int ups, int pro);
int conv_slide_launch(const ConvArgs& a, int pro, hipStream_t st);
void conv_slide_label(const ConvArgs& a, int pro, char* buf, int cap);
extern int g_conv_slide, g_conv_slide_rows;
// conv_slide128.hip: the same for the 128 -> 64 convs over the decoder's channel concat
bool conv_slide128_supported(const ConvArgs& a, int KH, int stride, int ups, int pro);
int conv_slide128_launch(const ConvArgs& a, hipStream_t st);
void conv_slide128_label(const ConvArgs& a, char* buf, int cap);

int g_conv_prio_mode = 0;      // tedm_set_tuning("conv_prio", 0|1|2)
int g_conv_solo = 0;           // tedm_set_tuning("conv_solo", 1): pad the LDS request so that workgroups do not share a CU (diagnostic)
int g_conv_tpw_max = 4;        // max spatial tiles per workgroup (tedm_set_tuning("conv_tpw_max", v))
int g_conv_xcd_order = 1;      // 1: cout-tile groups of one input tile share an XCD (tedm_set_tuning("conv_xcd_order", v))

namespace {

struct TileSel { int wm, wn, mt; };

// Tile shapes (rows = wm*mt, always 32 columns, wn*32 output channels):
//   id 0 "wide8"  : 1x4 waves, 8 M-tiles -> 256 px x 128 cout, 1 wave/SIMD (uses >256 VGPRs)
//   id 1 "narrow" : 2x2 waves, 4 M-tiles -> 256 px x  64 cout
//   id 2 "wide4"  : 1x4 waves, 4 M-tiles -> 128 px x 128 cout
// stride-2 convs use half the rows (the halo doubles).
//   id 3 "wide8w"  : 2x4 waves (512 threads), 4 M-tiles -> 256 px x 128 cout
//   id 4 "wide256" : 1x8 waves (512 threads), 4 M-tiles -> 128 px x 256 cout
//   id 6 "narrow16": 4x2 waves (512 threads), 4 M-tiles -> 512 px x 64 cout, one workgroup per CU: a halo of 18 rows serves 16
//                    (2x2: 10 for 8) and eight waves share every weight fragment (+3-7 % on the full-resolution 3x3 convs)
//   id 5 "tall256" : 1x8 waves (512 threads), 8 M-tiles -> 256 px x 256 cout, one workgroup per CU (<= 256 VGPRs): every weight
//                    fragment feeds 8 MFMAs instead of 4 and a halo of 10 rows serves 8 (+4-9 % single fp16 product, +11-15 % f16f4
//                    on the 256- and 512-channel 3x3 convs: profiles/r03_conv_lab_tiles.txt)
int g_tile_wide = 3;      // tile id used when Cout >= 128 (tedm_set_tuning("conv_tile_wide", v)); 2x4 waves: two wave rows share
                          // every weight fragment (half the L2 weight traffic per FLOP of the 1x4 tile)
int g_tile_narrow = 6;
// g_tile_narrow: tile id used when Cout < 128 (tedm_set_tuning("conv_tile_narrow", v))

int g_tile_wide256 = 5;   // tile id used when Cout >= 256 (tedm_set_tuning("conv_tile_wide256", v)): 1x8 waves = 256 couts per
                          // workgroup: every staged input pixel feeds twice the MFMAs of the 128-cout tiles (+5-9 %, conv_lab)

// 1x1 convolutions with 128 <= Cout < 256 (HBM-bound res_conv / to_out) run 12 % faster on the 1x4-wave tile (conv_lab)
inline int select_tile_id(int Cout, int KH = 3) {
    // (1x1 convs keep the 4-row tile: single LDS buffer, several small workgroups per CU)
    return Cout >= 256 ? (KH == 1 && g_tile_wide256 == 5 ? 4 : g_tile_wide256) : (Cout >= 128 ? (KH == 1 && g_tile_wide == 3 ? 2 : g_tile_wide) : (KH == 1 && g_tile_narrow == 6 ? 1 : g_tile_narrow));
}
inline TileSel tile_geom(int id, int stride) {
    TileSel t;
    switch (id) {
        case 0: t = TileSel{1, 4, 8}; break;
        case 1: t = TileSel{2, 2, 4}; break;
        case 2: t = TileSel{1, 4, 4}; break;
        case 3: t = TileSel{2, 4, 4}; break;
        case 4: t = TileSel{1, 8, 4}; break;
        case 5: t = TileSel{1, 8, 8}; break;
        case 6: t = TileSel{4, 2, 4}; break;
        default: t = TileSel{2, 4, 4}; break;
    }
    if (stride == 2) t.mt = t.mt > 1 ? t.mt / 2 : t.mt;
    return t;
}

int g_conv_small_grid = 1;   // tedm_set_tuning("conv_small_grid", 0|1)
int g_conv_upw = 1;          // tedm_set_tuning("conv_upw", 0|1): the x2-upsample parity convs with the parity on the wave (PAD = -2 instances)

// The x2-upsample parity convs (ups == 3) with one parity per WAVE: 1x8-wave tile of 8 rows, 4 rows when that fills the chip better,
// -1 = keep one parity per tile (small grids).  The choice depends on the shape and B like every tile choice; all forms accumulate
// each output's taps and chunks in the same order (same bits).
inline int upw_tile(const ConvArgs& a) {
    if (!g_conv_upw || a.add || a.stats_mode != STATS_NONE || a.out_q16 || a.add_q16) return -1;
    const int CT = (a.Cout + 31) / 32, ny = (4 * CT + 7) / 8;
    auto wgs = [&](int rows) { return (long long)a.B * ((a.H + rows - 1) / rows) * ((a.W + 31) / 32) * ny; };
    return wgs(8) >= 256 ? 5 : (wgs(4) >= 256 ? 4 : -1);
}

// Small batches (the sampler, single images): when the default tile would launch fewer workgroups than the chip has CUs, halve the
// rows (tall256 -> wide256), then the output channels per workgroup.  The statistics slab is per 4 output rows whatever the tile
// (conv_igemm.inc) and every output accumulates its taps and channel chunks in the same order whatever shares its workgroup, so
// the results are the same bits.
inline long long tile_workgroups(const ConvArgs& a, int id, int stride, int ups) {
    const TileSel t = tile_geom(id, stride);
    const int TR = t.wm * t.mt, Ho = ups == 3 ? a.H : a.Ho, Wo = ups == 3 ? a.W : a.Wo;
    return (long long)a.B * ((Ho + TR - 1) / TR) * ((Wo + 31) / 32) * (ups == 3 ? 4 : 1) * (((a.Cout + 31) / 32 + t.wn - 1) / t.wn);
}
inline int effective_tile(const ConvArgs& a, int KH, int stride, int ups, int pro) {
    int id = select_tile_id(a.Cout, KH);
    if (id == 6 && stride == 2) id = 1;                           // (the 16-row halo of a stride-2 conv does not fit the LDS)
    if (a.out_q16 || a.add_q16 || pro == PRO_GNQ) { if (id == 5) id = 4; else if (id == 6) id = 1; }     // Q16 boundaries: instances of tiles 1-4 only
    if (!g_conv_small_grid) return id;
    if (id == 5 && tile_workgroups(a, id, stride, ups) < 256) id = 4;
    if (id == 6 && tile_workgroups(a, id, stride, ups) < 256) id = 1;
    if (tile_workgroups(a, id, stride, ups) < 256) { if (id == 4) id = 2; else if (id == 3) id = 1; }
    return id;
}

}  // namespace

int conv_set_tuning(const char* key, int v) {
    if (!strcmp(key, "conv_tile_wide") && v >= 0 && v <= 3) { g_tile_wide = v; return TEDM_OK; }
    if (!strcmp(key, "conv_tile_wide256") && v >= 0 && v <= 5) { g_tile_wide256 = v; return TEDM_OK; }
    if (!strcmp(key, "conv_prio") && v >= 0 && v <= 3) { g_conv_prio_mode = v; return TEDM_OK; }
    if (!strcmp(key, "conv_solo") && (v == 0 || v == 1)) { g_conv_solo = v; return TEDM_OK; }
    if (!strcmp(key, "conv_tpw_max") && v >= 1 && v <= 64) { g_conv_tpw_max = v; return TEDM_OK; }
    if (!strcmp(key, "conv_xcd_order") && (v == 0 || v == 1)) { g_conv_xcd_order = v; return TEDM_OK; }
    if (!strcmp(key, "conv_tile_narrow") && ((v >= 0 && v <= 3) || v == 6)) { g_tile_narrow = v; return TEDM_OK; }
    if (!strcmp(key, "conv_small_grid") && (v == 0 || v == 1)) { g_conv_small_grid = v; return TEDM_OK; }
    if (!strcmp(key, "conv_upw") && (v == 0 || v == 1)) { g_conv_upw = v; return TEDM_OK; }
    if (!strcmp(key, "conv_slide") && (v == 0 || v == 1)) { g_conv_slide = v; return TEDM_OK; }
    if (!strcmp(key, "conv_slide_rows") && v >= 0 && v % 8 == 0) { g_conv_slide_rows = v; return TEDM_OK; }
    return TEDM_EINVAL;
}

// Label == the rocprof kernel symbol (template argument list of conv_igemm_kernel).
void conv_kernel_label(const ConvArgs& a, int KH, int stride, int ups, int pro, char* buf, int cap) {
    if (conv_slide_supported(a, KH, stride, ups, pro)) { conv_slide_label(a, pro, buf, cap); return; }
    if (conv_slide128_supported(a, KH, stride, ups, pro)) { conv_slide128_label(a, buf, cap); return; }
    const int id = effective_tile(a, KH, stride, ups, pro);
    const TileSel t = tile_geom(id, stride);
    const int pad = KH == 1 ? 0 : (ups == 3 ? -1 : (ups == 2 ? 2 : 1)), ks = (KH == 1 || a.prec == TEDM_PREC_F16F4 || a.prec == TEDM_PREC_FP16) ? 2 : 1, nbuf = KH == 1 ? 1 : 2;
    const int qio = (a.out_q16 ? 1 : 0) | (a.add_q16 ? 2 : 0);
    if (qio) snprintf(buf, cap, "conv_igemm_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d>", KH, KH, stride, pad, ups, t.wm, t.wn, t.mt,
                      ks, nbuf, pro, a.prec, qio);
    else snprintf(buf, cap, "conv_igemm_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d>", KH, KH, stride, pad, ups, t.wm, t.wn, t.mt,
                  ks, nbuf, pro, a.prec);
    if (ups == 3) {
        const int tw = (KH == 2 && !a.x2) ? upw_tile(a) : -1;
        if (tw >= 0) snprintf(buf, cap, "conv_igemm_kernel<2,2,1,-2,0,1,8,%d,%d,2,%d,%d>", tw == 5 ? 8 : 4, ks, pro, a.prec);
        else snprintf(buf, cap, "conv_igemm_kernel<2,2,1,-1,0,%d,%d,%d,%d,2,%d,%d>", t.wm, t.wn, t.mt, ks, pro, a.prec);
    }
}

int conv_stats_entries(int KH, int stride, int Cout, int Ho, int Wo) {
    // one entry per group of RB output rows and tile of 32 columns, the same for every tile shape (conv_igemm.inc: RB / SG)
    (void)KH; (void)Cout;
    const int RB = stride == 1 ? 4 : 2;
    return ((Ho + RB - 1) / RB) * ((Wo + 31) / 32);
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
    if (conv_slide_supported(a, KH, stride, ups, pro)) return conv_slide_launch(a, pro, st);
    if (conv_slide128_supported(a, KH, stride, ups, pro)) return conv_slide128_launch(a, st);
    int tile_id = effective_tile(a, KH, stride, ups, pro);
    if (ups == 3) {                                               // parity convs: the kernel's grid is the low-resolution one
        if (KH != 2 || a.Ho != 2 * a.H || a.Wo != 2 * a.W || a.stats_mode != STATS_NONE || a.x2) return TEDM_EINVAL;
        const int tw = upw_tile(a);
        if (tw >= 0) { ups = 4; tile_id = tw; }                   // parity per wave (conv_igemm.inc PAD = -2)
        a.Ho = a.H; a.Wo = a.W;
        a.wpk_pstride = conv_packed_bytes(a.Cout, Cin, 4);
    }
    const TileSel t = tile_geom(tile_id, stride);
    const int TR = t.wm * t.mt;
    a.tiles_y = (a.Ho + TR - 1) / TR;
    a.tiles_x = (a.Wo + 31) / 32;
    a.nk16 = Cin / 16;
    if (a.prec == TEDM_PREC_BF16) return conv_launch_prec1(a, KH, stride, ups, pro, tile_id, st);
    if (a.prec == TEDM_PREC_FP16X2) return conv_launch_prec2(a, KH, stride, ups, pro, tile_id, st);
    if (a.prec == TEDM_PREC_FP16) {
        if (Cin % 32 || a.C1 % 32) return TEDM_EINVAL;            // 32-channel chunks
        return conv_launch_prec4(a, KH, stride, ups, pro, tile_id, st);
    }
    if (a.prec == TEDM_PREC_F16F4) {
        if (Cin % 32 || a.C1 % 32) return TEDM_EINVAL;            // 32-channel blocks must not straddle the two sources
        return conv_launch_prec3(a, KH, stride, ups, pro, tile_id, st);
    }
    return conv_launch_prec0(a, KH, stride, ups, pro, tile_id, st);
}
