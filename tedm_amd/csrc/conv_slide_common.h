// conv_slide_common.h -- shared by the sliding-window convolution kernels (conv_slide.hip, conv_slide128.hip).
#pragma once
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [0, N) -- every index below is a constant in the front end
template <typename F, int... I> TEDM_DEVFN void sl_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> TEDM_DEVFN void sl_for(F&& f) { sl_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SL_WAGPR = 48;                 // weight fragments (of 54) pinned to AGPRs
#ifndef SL_PF
#define SL_PF 2                              // fragment prefetch distance in groups
#endif
// s_waitcnt immediate (gfx9 layout) that waits for vmcnt <= n only
constexpr int sl_vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }
// Vector-memory operations of one band by MFMA gap t (0..215), as the band loop issues them: item j's two loads behind MFMA
// 24 j + 21, one output store behind MFMA 108 ph + 3 v + 1 (v < 32), the statistics store behind MFMA 97.
constexpr int sl_vm_in_gaps(int lo, int hi, bool st, int abl) {      // gaps lo <= t < hi
    int n = 0;
    for (int t = lo; t < hi; ++t) {
        if (!(abl & 1) && t % 24 == 21) n += 2;
        if (!(abl & 2) && !(abl & 4) && (t % 108) % 3 == 1 && (t % 108) / 3 < 32) n += 1;
        if (st && t == 97) n += 1;
    }
    return n;
}
// ... issued between item k's loads (previous band) and the stage that first reads them (gap 24 k + sp of this band)
constexpr int sl_vm_after(int k, int sp, bool st, int abl) { return sl_vm_in_gaps(24 * k + 22, 216, st, abl) + sl_vm_in_gaps(0, 24 * k + sp, st, abl); }
constexpr int SL_HC = 34;                    // halo columns of a 32-column strip
constexpr int SL_PST = 112;                  // bytes of one operand record (pixel, 32-channel chunk): conv_igemm.inc Geo::PSTRIDE
constexpr int SL_SEG = (SL_HC + 1) * SL_PST;  // one ring row of one chunk: 34 records + one spare slot that is never read (it takes
                                             // the stores of the columns outside the image, whose own records stay zero)

}  // namespace

// Rows of the image one workgroup walks (both sliding-window kernels; one workgroup per CU).  The kernel's time is
// (rounds of 256 workgroups) x (rows per workgroup + the ~10 rows a workgroup encodes before its first band), so among the divisors
// H, H/2, H/4, ... (multiples of 8, >= 16) the one that minimises that product wins: whole strips when B x column tiles is a multiple
// of the CU count (B = 32 at 256^2), shorter strips when whole ones would leave a partial round (B = 33, 48) or an empty chip
// (B = 1).  Every choice gives the same bits (a strip boundary only moves the rows a workgroup encodes twice).
inline int sl_pick_rows(int B, int H, int tiles_x, int cus = 256) {
    int best = H;
    long long best_cost = -1;
    for (int rows = H; rows >= 16 && rows % 8 == 0 && H % rows == 0; rows /= 2) {
        const long long wgs = (long long)B * tiles_x * (H / rows), rounds = (wgs + cus - 1) / cus;
        const long long cost = rounds * (rows + 10);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = rows; }
        if (rows % 16) break;
    }
    return best;
}
