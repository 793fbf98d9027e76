// attention.hip -- LinearAttention (models/unet_model.py:178-210) and the bottleneck Attention
// (:213-241) on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (fp32 in, fp32 fmaf
// chain out: no precision split needed; these contractions are 1.4 % of the FLOPs).
//
// Operand maps (cdna guide section 3): A[i][k]: lane = (k<<5)|i ; B[k][j]: lane = (k<<5)|j ;
// C/D: column j = lane&31, row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// A 32x32 accumulator is therefore directly the B operand of a following product that sums
// over its ROW index: k-step s of the next MFMA takes accumulator register s.
//
// qkv layout: (B, hw, 3*heads*32) fp32, channels [q | k | v], head-major, dim_head = 32.
#include "common.h"
#include "ops.h"
#include "../../include/tedm_hip.h"

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP)

TEDM_DEVFN int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// ---------------------------------------------------------------------------------------
// linear attention, pass 1: per (sample, head, pixel split)
//   ctx_p[d][e] = sum_n exp(k[n][d] - kmax[d]) * v[n][e],   z_p[d] = sum_n exp(k[n][d] - kmax[d])
// kmax comes from the to_qkv epilogue (max slab), so a single sweep suffices.  Each lane streams
// its own (pixel parity, channel) element straight from HBM: 2 x 128-B segments per load, no LDS.
__global__ __launch_bounds__(256) void linattn_ctx_kernel(const float* __restrict__ qkv, const float* __restrict__ kmax,
                                                          int B, long long hw, int heads, int S, long long pps,
                                                          float* __restrict__ part) {
    const int lane = threadIdx.x & 63, r = lane & 31, kk = lane >> 5;
    const long long gid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long long)B * S * heads) return;
    const int head = (int)(gid % heads);
    const int s = (int)((gid / heads) % S);
    const int b = (int)(gid / ((long long)heads * S));
    const int C3 = 3 * heads * 32;
    const long long p_beg = s * pps, p_end = (p_beg + pps < hw) ? p_beg + pps : hw;
    const float km = kmax[(size_t)b * C3 + heads * 32 + head * 32 + r];
    const float* kb = qkv + (size_t)b * hw * C3 + heads * 32 + head * 32 + r;
    const float* vb = kb + heads * 32;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float z = 0.f;
    constexpr int U = 8;
    for (long long p = p_beg; p < p_end; p += 2 * U) {
        float kv[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pix = p + 2 * u + kk;
            const bool ok = pix < p_end;
            kv[u] = ok ? kb[(size_t)pix * C3] : -INFINITY;
            vv[u] = ok ? vb[(size_t)pix * C3] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float e = __expf(kv[u] - km);                // exp(-inf) = 0 for masked pixels
            z += e;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e, vv[u], acc, 0, 0, 0);
        }
    }
    z += __shfl_xor(z, 32);
    float* cp = part + (size_t)gid * (1024 + 32);
#pragma unroll
    for (int i = 0; i < 16; ++i) cp[acc_row(i, kk) * 32 + r] = acc[i];
    if (kk == 0) cp[1024 + r] = z;
}

// pass 1 finalize: ctx[b][head][d][e] = sum_s ctx_p / (sum_s z_p[d] * hw)   (v / (h*w), :204)
__global__ __launch_bounds__(256) void linattn_ctx_finalize_kernel(const float* __restrict__ part, int heads, int S,
                                                                   long long hw, float* __restrict__ ctx,
                                                                   float* __restrict__ zout /*[B][heads][32] or null*/) {
    const int bh = blockIdx.x >> 2, b = bh / heads, head = bh - b * heads;      // 4 blocks per (sample, head)
    const int i = (blockIdx.x & 3) * 256 + threadIdx.x, d = i >> 5;
    float c = 0.f, z = 0.f;
    for (int s = 0; s < S; ++s) {
        const float* cp = part + ((size_t)((size_t)b * S + s) * heads + head) * (1024 + 32);
        c += cp[i];
        z += cp[1024 + d];
    }
    ctx[(size_t)bh * 1024 + i] = c / (z * (float)hw);
    if (zout && (i & 31) == 0) zout[(size_t)bh * 32 + d] = z;
}

// linear attention, pass 2: o[n][head*32+e] = sum_d ctx[d][e] * softmax_d(q[n][:])[d] * 32^-0.5
// O^T[e][pixel] tile per wave: A = ctx^T (16 VGPRs, resident across pixel tiles), B = q' with one
// pixel per lane so the softmax over d is lane-local plus one cross-half exchange.
__global__ __launch_bounds__(256) void linattn_out_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                          int B, long long hw, int heads, int tpw, float* __restrict__ o,
                                                          float* __restrict__ qprime /*qkv base: store q' over q, or null*/) {
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const long long tiles = (hw + 31) / 32;
    const long long chunks = (tiles + tpw - 1) / tpw;
    const long long gid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long long)B * chunks * heads) return;
    const int head = (int)(gid % heads);
    const long long ch = (gid / heads) % chunks;
    const int b = (int)(gid / ((long long)heads * chunks));
    const int C3 = 3 * heads * 32, CO = heads * 32;
    float a[16];
    const float* cb = ctx + ((size_t)b * heads + head) * 1024;
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = cb[(8 * (s >> 2) + 4 * hh + (s & 3)) * 32 + r];
    const float scale = 0.17677669529663688110f;   // 32^-0.5
    for (long long t = ch * tpw; t < (ch + 1) * tpw && t < tiles; ++t) {
        const long long p = t * 32 + r;
        const bool ok = p < hw;
        float q[16];
        const float* qp = qkv + ((size_t)b * hw + (ok ? p : 0)) * C3 + head * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *(const float4*)(qp + 4 * (2 * g + hh));
            q[4 * g] = v.x; q[4 * g + 1] = v.y; q[4 * g + 2] = v.z; q[4 * g + 3] = v.w;
        }
        float m = q[0];
#pragma unroll
        for (int s = 1; s < 16; ++s) m = fmaxf(m, q[s]);
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) { q[s] = __expf(q[s] - m); sum += q[s]; }
        sum += __shfl_xor(sum, 32);
        const float f = scale / sum;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) { q[s] *= f; acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], q[s], acc, 0, 0, 0); }
        if (ok && qprime) {                                   // training: keep q' = softmax_d(q)*scale for the backward
            float* qw = qprime + ((size_t)b * hw + p) * C3 + head * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(qw + 4 * (2 * g + hh)) = make_float4(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3]);
        }
        if (ok) {
            float* op = o + ((size_t)b * hw + p) * CO + head * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(op + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// bottleneck attention (models/unet_model.py:229-241): q,k l2-normalised along the TOKEN axis
// (inv_norm from the to_qkv epilogue), sim = scale * q^T k, softmax over keys, out = attn v.
// One wave per (sample, head, 32-query tile), ONE sweep over the key tiles with a running row max (online softmax:
// the accumulators are rescaled when it grows).  Both contractions run on v_mfma_f32_32x32x16_f16 with fp16 hi/lo
// operands (hi*hi + lo*hi + hi*lo: ~2^-21, i.e. fp32-grade) -- 12 issues of 32 cycles per (query tile, key tile) instead
// of the 48 x 64 cycles of the two-sweep v_mfma_f32_32x32x2_f32 form.  S^T[key][query] stays in the accumulator, so P is
// the B operand of the PV product as it stands (registers 8s..8s+7 = k-step s; V^T is loaded in that row order).
TEDM_DEVFN void split_frag8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 hj = (_Float16)v[j];
        hi[j] = hj;
        lo[j] = (_Float16)(v[j] - (float)hj);
    }
}
TEDM_DEVFN f32x16 mfma3_f16(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
}

__global__ __launch_bounds__(256) void full_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ inv,
                                                        int B, int ntok, int heads, float scale, float* __restrict__ o,
                                                        float* __restrict__ ml /*(B,ntok,heads,2) row max / row sum, or null*/) {
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int qtiles = (ntok + 31) / 32;
    const long long gid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long long)B * qtiles * heads) return;
    // fp16 conversions saturate at +-65504 for the rest of the kernel: an un-normalised V beyond fp16's range degrades to the
    // clamped hi + clamped lo instead of hi = inf, lo = v - inf = NaN (q, k are l2-normalised, P is in [0, 1]).  ADVICE r2.
    fp16_saturate_on();
    const int head = (int)(gid % heads);
    const int qt = (int)((gid / heads) % qtiles);
    const int b = (int)(gid / ((long long)heads * qtiles));
    const int C3 = 3 * heads * 32, CO = heads * 32;
    const float* base = qkv + (size_t)b * ntok * C3;
    const float* ivq = inv + (size_t)b * C3 + head * 32;
    const float* ivk = ivq + heads * 32;

    const int qi = qt * 32 + r;
    const bool qok = qi < ntok;
    // B operand of S^T = K Q^T: lane (query r, half hh), k-step s, element j <-> d = 16 s + 8 hh + j
    f16x8 qh[2], ql[2];
    float ik[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int d0 = 16 * s + 8 * hh;
        const float* qp = base + (size_t)(qok ? qi : 0) * C3 + head * 32 + d0;
        const float4 v0 = *(const float4*)qp, v1 = *(const float4*)(qp + 4);
        const float4 n0 = *(const float4*)(ivq + d0), n1 = *(const float4*)(ivq + d0 + 4);
        const float4 k0 = *(const float4*)(ivk + d0), k1 = *(const float4*)(ivk + d0 + 4);
        const float qv[8] = {v0.x * n0.x * scale, v0.y * n0.y * scale, v0.z * n0.z * scale, v0.w * n0.w * scale,
                             v1.x * n1.x * scale, v1.y * n1.y * scale, v1.z * n1.z * scale, v1.w * n1.w * scale};
        split_frag8(qv, qh[s], ql[s]);
        ik[s][0] = k0.x; ik[s][1] = k0.y; ik[s][2] = k0.z; ik[s][3] = k0.w;
        ik[s][4] = k1.x; ik[s][5] = k1.y; ik[s][6] = k1.z; ik[s][7] = k1.w;
    }
    const int ktiles = (ntok + 31) / 32;

    f32x16 oacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[i] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float* vbase = base + 2 * heads * 32 + head * 32 + r;
    for (int jt = 0; jt < ktiles; ++jt) {
        // ---- scores S^T[key][query]: A operand = K tile, lane (key r, half hh) ----
        const int kj = jt * 32 + r;
        const bool kok = kj < ntok;
        const float* kp = base + (size_t)(kok ? kj : 0) * C3 + heads * 32 + head * 32;
        f32x16 st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float4 v0 = *(const float4*)(kp + 16 * s + 8 * hh), v1 = *(const float4*)(kp + 16 * s + 8 * hh + 4);
            float kv[8] = {v0.x * ik[s][0], v0.y * ik[s][1], v0.z * ik[s][2], v0.w * ik[s][3],
                           v1.x * ik[s][4], v1.y * ik[s][5], v1.z * ik[s][6], v1.w * ik[s][7]};
            if (!kok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) kv[j] = 0.f;
            }
            f16x8 kh, kl;
            split_frag8(kv, kh, kl);
            st = mfma3_f16(kh, kl, qh[s], ql[s], st);
        }
        // ---- V^T rows in the accumulator's row order (issued before the softmax math hides their latency) ----
        float va[2][8];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = jt * 32 + 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
                va[s][j] = key < ntok ? vbase[(size_t)key * C3] : 0.f;
            }
        // ---- online softmax over the keys (rows of st); the query sits on the lane ----
        float tm = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (jt * 32 + acc_row(i, hh) < ntok) tm = fmaxf(tm, st[i]);
        tm = fmaxf(tm, __shfl_xor(tm, 32));
        const float m_new = fmaxf(m, tm);
        const float f = (m_new == m) ? 1.f : __expf(m - m_new);       // m = -inf on the first tile: f = 0 (accumulators are 0)
        l *= f;
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[i] *= f;
        m = m_new;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = 8 * s + j;
                pv[j] = (jt * 32 + acc_row(i, hh) < ntok) ? __expf(st[i] - m) : 0.f;
                l += pv[j];
            }
            f16x8 ph, pl, vh, vl;
            split_frag8(pv, ph, pl);
            split_frag8(va[s], vh, vl);
            oacc = mfma3_f16(vh, vl, ph, pl, oacc);                  // O^T[e][query] += V^T[e][key] P[key][query]
        }
    }
    l += __shfl_xor(l, 32);
    if (qok && ml && hh == 0) { float* mp = ml + (((size_t)b * ntok + qi) * heads + head) * 2; mp[0] = m; mp[1] = l; }
    if (qok) {
        const float il = 1.f / l;
        float* op = o + ((size_t)b * ntok + qi) * CO + head * 32 + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4*)(op + 8 * g) = make_float4(oacc[4 * g] * il, oacc[4 * g + 1] * il, oacc[4 * g + 2] * il,
                                                 oacc[4 * g + 3] * il);
    }
}

}  // namespace

// pixel splits of the context sweep: enough waves to fill 256 CUs x 4 SIMDs several times over.
int la_ctx_splits(long long hw) {
    long long s = hw / 512;
    if (s < 1) s = 1;
    if (s > 128) s = 128;
    return (int)s;
}
size_t la_ctx_partial_floats(int B, int heads, long long hw) {
    return (size_t)B * la_ctx_splits(hw) * heads * (1024 + 32);
}

int op_linattn_context(const float* qkv, const float* kmax, int B, long long hw, int heads, float* partial,
                       float* ctx, float* zout, hipStream_t st) {
    const int S = la_ctx_splits(hw);
    long long pps = (hw + S - 1) / S;
    pps += pps & 1;
    const long long waves = (long long)B * S * heads;
    hipLaunchKernelGGL(linattn_ctx_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, qkv, kmax, B, hw, heads,
                       S, pps, partial);
    hipLaunchKernelGGL(linattn_ctx_finalize_kernel, dim3(B * heads * 4), dim3(256), 0, st, partial, heads, S, hw, ctx, zout);
    return LAUNCH_OK();
}

int op_linattn_ctx_finalize(const float* partial, int B, int heads, int S, long long hw, float* ctx, hipStream_t st) {
    float* zout = nullptr;
    hipLaunchKernelGGL(linattn_ctx_finalize_kernel, dim3(B * heads * 4), dim3(256), 0, st, partial, heads, S, hw, ctx, nullptr);
    return LAUNCH_OK();
}

int op_linattn_out(const float* qkv, const float* ctx, int B, long long hw, int heads, float* o, float* qprime_inplace,
                   hipStream_t st) {
    const int tpw = 8;
    const long long tiles = (hw + 31) / 32, chunks = (tiles + tpw - 1) / tpw;
    const long long waves = (long long)B * chunks * heads;
    hipLaunchKernelGGL(linattn_out_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, qkv, ctx, B, hw, heads,
                       tpw, o, qprime_inplace);
    return LAUNCH_OK();
}

int op_full_attention(const float* qkv, const float* inv_norm, int B, int ntok, int heads, float scale, float* o, float* ml,
                      hipStream_t st) {
    const long long waves = (long long)B * ((ntok + 31) / 32) * heads;
    hipLaunchKernelGGL(full_attn_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, qkv, inv_norm, B, ntok,
                       heads, scale, o, ml);
    return LAUNCH_OK();
}
