// attention_bwd.hip -- backward of LinearAttention (models/unet_model.py:197-207) and of the bottleneck
// Attention (:229-241) on the exact-fp32 MFMA, mirroring attention.hip (same operand maps).
//
// Linear attention, per (sample, head), with q' = softmax_d(q)*s (s = 32^-0.5), k' = softmax_n(k),
// ctx[d][e] = sum_n k'[n][d] v[n][e] / hw, o[n][e] = sum_d ctx[d][e] q'[n][d]:
//   dctx[d][e] = sum_n q'[n][d] do[n][e]                     (la_bwd_dctx_kernel: reduction over pixels)
//   dq'[n][d]  = sum_e ctx[d][e] do[n][e]      dq = q' * (dq' - sum_d q' dq' / s)
//   dk'[n][d]  = sum_e dctx[d][e] v[n][e]/hw   dk = k' * (dk' - T[d]),  T[d] = sum_e dctx[d][e] ctx[d][e]
//   dv[n][e]   = sum_d k'[n][d] dctx[d][e]/hw                 (la_bwd_pix_kernel: one pixel per lane)
// (sum_n k'[n][d] dk'[n][d] collapses to T[d] because ctx already holds sum_n k' v / hw.)
// The training forward stores q' in place of q inside the qkv tensor; k' is rebuilt from k, kmax and Z.
#include "common.h"
#include "ops.h"
#include "../../include/tedm_hip.h"

namespace {

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP)

TEDM_DEVFN int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// dctx partials: A = q' (lane = d), B = do (lane = e), k = pixel parity.  Same sweep as linattn_ctx_kernel.
__global__ __launch_bounds__(256) void la_bwd_dctx_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                          int B, long long hw, int heads, int S, long long pps,
                                                          float* __restrict__ part) {
    const int lane = threadIdx.x & 63, r = lane & 31, kk = lane >> 5;
    const long long gid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long long)B * S * heads) return;
    const int head = (int)(gid % heads);
    const int s = (int)((gid / heads) % S);
    const int b = (int)(gid / ((long long)heads * S));
    const int C3 = 3 * heads * 32, CO = heads * 32;
    const long long p_beg = s * pps, p_end = (p_beg + pps < hw) ? p_beg + pps : hw;
    const float* qb = qkv + (size_t)b * hw * C3 + head * 32 + r;
    const float* db = dout + (size_t)b * hw * CO + head * 32 + r;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int U = 8;
    for (long long p = p_beg; p < p_end; p += 2 * U) {
        float qv[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pix = p + 2 * u + kk;
            const bool ok = pix < p_end;
            const long long pc = ok ? pix : p_end - 1;
            qv[u] = qb[(size_t)pc * C3];
            dv[u] = db[(size_t)pc * CO];
            if (!ok) { qv[u] = 0.f; dv[u] = 0.f; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[u], dv[u], acc, 0, 0, 0);
    }
    float* cp = part + (size_t)gid * 1024;
#pragma unroll
    for (int i = 0; i < 16; ++i) cp[acc_row(i, kk) * 32 + r] = acc[i];
}

// dctx = sum of partials; T[d] = sum_e dctx[d][e]*ctx[d][e].  4 blocks per (sample, head) like the forward finalize.
__global__ __launch_bounds__(256) void la_bwd_dctx_finalize_kernel(const float* __restrict__ part, const float* __restrict__ ctx,
                                                                   int heads, int S, float* __restrict__ dctx,
                                                                   float* __restrict__ T) {
    const int bh = blockIdx.x >> 2, b = bh / heads, head = bh - b * heads;
    const int i = (blockIdx.x & 3) * 256 + threadIdx.x;
    float c = 0.f;
    for (int s = 0; s < S; ++s) c += part[((size_t)((size_t)b * S + s) * heads + head) * 1024 + i];
    dctx[(size_t)bh * 1024 + i] = c;
    float t = c * ctx[(size_t)bh * 1024 + i];                 // row d = i>>5 : reduce over the 32 lanes of a row
    for (int m = 16; m >= 1; m >>= 1) t += __shfl_xor(t, m);
    if ((threadIdx.x & 31) == 0) T[(size_t)bh * 32 + (i >> 5)] = t;
}

// per-pixel part: one pixel per lane (column of every accumulator), rows = d or e in registers.
__global__ __launch_bounds__(256) void la_bwd_pix_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                         const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                         const float* __restrict__ T, const float* __restrict__ kmax,
                                                         const float* __restrict__ Z, int B, long long hw, int heads, int tpw,
                                                         float* __restrict__ dqkv) {
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const long long tiles = (hw + 31) / 32;
    const long long chunks = (tiles + tpw - 1) / tpw;
    const long long gid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long long)B * chunks * heads) return;
    const int head = (int)(gid % heads);
    const long long ch = (gid / heads) % chunks;
    const int b = (int)(gid / ((long long)heads * chunks));
    const int C3 = 3 * heads * 32, CO = heads * 32;
    const float inv_hw = 1.0f / (float)hw;
    const float scale = 0.17677669529663688110f, inv_scale = 5.65685424949238019521f;
    // A operands (constant per (b, head)): lane (i = r, k = hh) holds M[.][.] at the k-step's row pattern
    float a_ctx[16], a_dctx[16], a_dctxT[16], km[16], zi[16], tt[16];
    const float* cb = ctx + ((size_t)b * heads + head) * 1024;
    const float* gb = dctx + ((size_t)b * heads + head) * 1024;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int kidx = acc_row(s, hh);
        a_ctx[s] = cb[r * 32 + kidx];                  // A[i=d=r][k=e]   = ctx[d][e]
        a_dctx[s] = gb[r * 32 + kidx] * inv_hw;        // A[i=d=r][k=e]   = dctx[d][e]/hw
        a_dctxT[s] = gb[kidx * 32 + r] * inv_hw;       // A[i=e=r][k=d]   = dctx[d][e]/hw
        // row-pattern constants for k' (rows d = acc_row(s, hh) of the D[d][pixel] accumulators)
        km[s] = kmax[(size_t)b * C3 + heads * 32 + head * 32 + kidx];
        zi[s] = 1.0f / Z[((size_t)b * heads + head) * 32 + kidx];
        tt[s] = T[((size_t)b * heads + head) * 32 + kidx];
    }
    for (long long t = ch * tpw; t < (ch + 1) * tpw && t < tiles; ++t) {
        const long long p = t * 32 + r;
        const bool ok = p < hw;
        const size_t pb = (size_t)b * hw + (ok ? p : hw - 1);
        float qp[16], kv[16], vv[16], dd[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = head * 32 + 4 * (2 * g + hh);
            const float4 q4 = *(const float4*)(qkv + pb * C3 + c0);
            const float4 k4 = *(const float4*)(qkv + pb * C3 + heads * 32 + c0);
            const float4 v4 = *(const float4*)(qkv + pb * C3 + 2 * heads * 32 + c0);
            const float4 d4 = *(const float4*)(dout + pb * CO + c0);
            qp[4 * g] = q4.x; qp[4 * g + 1] = q4.y; qp[4 * g + 2] = q4.z; qp[4 * g + 3] = q4.w;
            kv[4 * g] = k4.x; kv[4 * g + 1] = k4.y; kv[4 * g + 2] = k4.z; kv[4 * g + 3] = k4.w;
            vv[4 * g] = v4.x; vv[4 * g + 1] = v4.y; vv[4 * g + 2] = v4.z; vv[4 * g + 3] = v4.w;
            dd[4 * g] = d4.x; dd[4 * g + 1] = d4.y; dd[4 * g + 2] = d4.z; dd[4 * g + 3] = d4.w;
        }
        f32x16 dq, dk, dvv;
#pragma unroll
        for (int i = 0; i < 16; ++i) { dq[i] = 0.f; dk[i] = 0.f; dvv[i] = 0.f; }
        float kp[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) kp[s] = __expf(kv[s] - km[s]) * zi[s];          // k'[d][pixel]
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            dq = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ctx[s], dd[s], dq, 0, 0, 0);     // dq'[d][n] = sum_e ctx[d][e] do[e][n]
            dk = __builtin_amdgcn_mfma_f32_32x32x2f32(a_dctx[s], vv[s], dk, 0, 0, 0);    // dk'[d][n] = sum_e dctx[d][e] v[e][n]/hw
            dvv = __builtin_amdgcn_mfma_f32_32x32x2f32(a_dctxT[s], kp[s], dvv, 0, 0, 0); // dv[e][n]  = sum_d dctx[d][e] k'[d][n]/hw
        }
        float sdot = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) sdot = fmaf(qp[s], dq[s], sdot);
        sdot += __shfl_xor(sdot, 32);
        sdot *= inv_scale;
        (void)scale;
        if (ok) {
            float* op = dqkv + pb * C3 + head * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 o1, o2, o3;
                o1.x = qp[4 * g] * (dq[4 * g] - sdot);         o1.y = qp[4 * g + 1] * (dq[4 * g + 1] - sdot);
                o1.z = qp[4 * g + 2] * (dq[4 * g + 2] - sdot); o1.w = qp[4 * g + 3] * (dq[4 * g + 3] - sdot);
                o2.x = kp[4 * g] * (dk[4 * g] - tt[4 * g]);             o2.y = kp[4 * g + 1] * (dk[4 * g + 1] - tt[4 * g + 1]);
                o2.z = kp[4 * g + 2] * (dk[4 * g + 2] - tt[4 * g + 2]); o2.w = kp[4 * g + 3] * (dk[4 * g + 3] - tt[4 * g + 3]);
                o3 = make_float4(dvv[4 * g], dvv[4 * g + 1], dvv[4 * g + 2], dvv[4 * g + 3]);
                *(float4*)(op + 8 * g) = o1;
                *(float4*)(op + heads * 32 + 8 * g) = o2;
                *(float4*)(op + 2 * heads * 32 + 8 * g) = o3;
            }
        }
    }
}

}  // namespace

// part: la_ctx_partial-sized (B*S*heads*1024 floats suffices); outputs dqkv (B,hw,384).  T: B*heads*32 scratch.
int op_linattn_backward(const float* qkv /*q slot holds q'*/, const float* dout, const float* ctx, const float* kmax,
                        const float* Z, int B, long long hw, int heads, float* part, float* dctx, float* T, float* dqkv,
                        hipStream_t st) {
    const int S = la_ctx_splits(hw);
    long long pps = (hw + S - 1) / S;
    pps += pps & 1;
    const long long waves = (long long)B * S * heads;
    hipLaunchKernelGGL(la_bwd_dctx_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, qkv, dout, B, hw, heads, S, pps, part);
    hipLaunchKernelGGL(la_bwd_dctx_finalize_kernel, dim3(B * heads * 4), dim3(256), 0, st, part, ctx, heads, S, dctx, T);
    const int tpw = 8;
    const long long tiles = (hw + 31) / 32, chunks = (tiles + tpw - 1) / tpw;
    const long long w2 = (long long)B * chunks * heads;
    hipLaunchKernelGGL(la_bwd_pix_kernel, dim3((unsigned)((w2 + 3) / 4)), dim3(256), 0, st, qkv, dout, ctx, dctx, T, kmax, Z, B, hw,
                       heads, tpw, dqkv);
    return LAUNCH_OK();
}

// =======================================================================================
// Bottleneck attention backward.  Forward (attention.hip): qn = q*iq, kn = k*ik (l2norm over TOKENS),
// S[i][j] = scale * sum_d qn[i][d] kn[j][d], P = softmax_j, out[i][d] = sum_j P[i][j] v[j][d].
// Saved by the training forward: qkv, inv norms, out, and per query (m_i, l_i) = row max / row sum.
// With delta_i = sum_d do[i][d]*out[i][d]:
//   dP[i][j] = sum_d do[i][d] v[j][d],   dS = P * (dP - delta_i)
//   dqn[i][d] = scale * sum_j dS[i][j] kn[j][d]      (fa_bwd_q_kernel: one wave per 32-query tile)
//   dkn[j][d] = scale * sum_i dS[i][j] qn[i][d],  dv[j][d] = sum_i P[i][j] do[i][d]   (fa_bwd_kv_kernel: per key tile)
//   l2norm backward over tokens: dq = iq * (dqn - qn * sum_i qn dqn)                 (fa_bwd_l2n_kernel)
namespace {

// delta[b][head][i] = sum_d do*out
__global__ void fa_bwd_delta_kernel(const float* __restrict__ dout, const float* __restrict__ out, long long total, float* __restrict__ delta) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // over B*ntok*heads
    if (i >= total) return;
    const float4* a = (const float4*)(dout + i * 32);
    const float4* b = (const float4*)(out + i * 32);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float4 x = a[k], y = b[k]; s += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w); }
    delta[i] = s;                                                              // layout (B, ntok, heads)
}

// dqn: S^T[key][query] orientation (query on the lane), exactly the forward's second sweep.
__global__ __launch_bounds__(256) void fa_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ inv,
                                                       const float* __restrict__ dout, const float* __restrict__ ml,
                                                       const float* __restrict__ delta, int B, int ntok, int heads, float scale,
                                                       float* __restrict__ dqn /*(B,ntok,heads*32)*/) {
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int qtiles = (ntok + 31) / 32;
    const long long gid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long long)B * qtiles * heads) return;
    const int head = (int)(gid % heads);
    const int qt = (int)((gid / heads) % qtiles);
    const int b = (int)(gid / ((long long)heads * qtiles));
    const int C3 = 3 * heads * 32, CO = heads * 32;
    const float* base = qkv + (size_t)b * ntok * C3;
    const float* ivq = inv + (size_t)b * C3 + head * 32;
    const float* ivk = ivq + heads * 32;
    const int qi = qt * 32 + r;
    const bool qok = qi < ntok;
    const int qic = qok ? qi : ntok - 1;
    float qv[16], ik[16], dd[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d0 = 4 * (2 * g + hh);
        const float4 v = *(const float4*)(base + (size_t)qic * C3 + head * 32 + d0);
        const float4 n = *(const float4*)(ivq + d0), nk = *(const float4*)(ivk + d0);
        const float4 d4 = *(const float4*)(dout + ((size_t)b * ntok + qic) * CO + head * 32 + d0);
        qv[4 * g] = v.x * n.x * scale; qv[4 * g + 1] = v.y * n.y * scale; qv[4 * g + 2] = v.z * n.z * scale; qv[4 * g + 3] = v.w * n.w * scale;
        ik[4 * g] = nk.x; ik[4 * g + 1] = nk.y; ik[4 * g + 2] = nk.z; ik[4 * g + 3] = nk.w;
        dd[4 * g] = d4.x; dd[4 * g + 1] = d4.y; dd[4 * g + 2] = d4.z; dd[4 * g + 3] = d4.w;
    }
    const float mi = ml[(((size_t)b * ntok + qic) * heads + head) * 2], li = 1.0f / ml[(((size_t)b * ntok + qic) * heads + head) * 2 + 1];
    const float dl = delta[((size_t)b * ntok + qic) * heads + head];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int ktiles = (ntok + 31) / 32;
    for (int jt = 0; jt < ktiles; ++jt) {
        const int kj = jt * 32 + r;
        const bool kok = kj < ntok;
        const int kjc = kok ? kj : ntok - 1;
        float kv[16], vv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 4 * (2 * g + hh);
            const float4 k4 = *(const float4*)(base + (size_t)kjc * C3 + heads * 32 + head * 32 + d0);
            const float4 v4 = *(const float4*)(base + (size_t)kjc * C3 + 2 * heads * 32 + head * 32 + d0);
            kv[4 * g] = kok ? k4.x * ik[4 * g] : 0.f; kv[4 * g + 1] = kok ? k4.y * ik[4 * g + 1] : 0.f;
            kv[4 * g + 2] = kok ? k4.z * ik[4 * g + 2] : 0.f; kv[4 * g + 3] = kok ? k4.w * ik[4 * g + 3] : 0.f;
            vv[4 * g] = kok ? v4.x : 0.f; vv[4 * g + 1] = kok ? v4.y : 0.f; vv[4 * g + 2] = kok ? v4.z : 0.f; vv[4 * g + 3] = kok ? v4.w : 0.f;
        }
        f32x16 st, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[s], qv[s], st, 0, 0, 0);      // S^T[j][i]
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[s], dd[s], dp, 0, 0, 0);      // dP^T[j][i] = sum_d v[j][d] do[i][d]
        }
        // A operand of the dqn product: kn[j][d] with lane = d, k = key row acc_row(s, hh)
        float ka[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int j = jt * 32 + acc_row(s, hh);
            const int jc = j < ntok ? j : ntok - 1;
            ka[s] = base[(size_t)jc * C3 + heads * 32 + head * 32 + r] * ivk[r];
            if (j >= ntok) ka[s] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bool jok = jt * 32 + acc_row(s, hh) < ntok;
            const float p = jok ? __expf(st[s] - mi) * li : 0.f;
            const float ds = p * (dp[s] - dl);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s], ds, acc, 0, 0, 0);       // dqn^T[d][i] += kn[j][d] dS^T[j][i]
        }
    }
    if (qok) {
        float* op = dqn + ((size_t)b * ntok + qi) * CO + head * 32 + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4*)(op + 8 * g) = make_float4(acc[4 * g] * scale, acc[4 * g + 1] * scale, acc[4 * g + 2] * scale, acc[4 * g + 3] * scale);
    }
}

// dkn, dv: S[query][key] orientation (key on the lane); row constants (m, 1/l, delta) per query row in registers.
__global__ __launch_bounds__(256) void fa_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ inv,
                                                        const float* __restrict__ dout, const float* __restrict__ ml,
                                                        const float* __restrict__ delta, int B, int ntok, int heads, float scale,
                                                        float* __restrict__ dkn, float* __restrict__ dv) {
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int ktiles = (ntok + 31) / 32;
    const long long gid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= (long long)B * ktiles * heads) return;
    const int head = (int)(gid % heads);
    const int kt = (int)((gid / heads) % ktiles);
    const int b = (int)(gid / ((long long)heads * ktiles));
    const int C3 = 3 * heads * 32, CO = heads * 32;
    const float* base = qkv + (size_t)b * ntok * C3;
    const float* ivq = inv + (size_t)b * C3 + head * 32;
    const float* ivk = ivq + heads * 32;
    const int kj = kt * 32 + r;
    const bool kok = kj < ntok;
    const int kjc = kok ? kj : ntok - 1;
    float kv[16], vv[16], iq[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d0 = 4 * (2 * g + hh);
        const float4 k4 = *(const float4*)(base + (size_t)kjc * C3 + heads * 32 + head * 32 + d0);
        const float4 v4 = *(const float4*)(base + (size_t)kjc * C3 + 2 * heads * 32 + head * 32 + d0);
        const float4 nk = *(const float4*)(ivk + d0), nq = *(const float4*)(ivq + d0);
        kv[4 * g] = kok ? k4.x * nk.x : 0.f; kv[4 * g + 1] = kok ? k4.y * nk.y : 0.f; kv[4 * g + 2] = kok ? k4.z * nk.z : 0.f; kv[4 * g + 3] = kok ? k4.w * nk.w : 0.f;
        vv[4 * g] = kok ? v4.x : 0.f; vv[4 * g + 1] = kok ? v4.y : 0.f; vv[4 * g + 2] = kok ? v4.z : 0.f; vv[4 * g + 3] = kok ? v4.w : 0.f;
        iq[4 * g] = nq.x; iq[4 * g + 1] = nq.y; iq[4 * g + 2] = nq.z; iq[4 * g + 3] = nq.w;
    }
    f32x16 akn, adv;
#pragma unroll
    for (int i = 0; i < 16; ++i) { akn[i] = 0.f; adv[i] = 0.f; }
    const int qtiles = (ntok + 31) / 32;
    for (int it = 0; it < qtiles; ++it) {
        // A operands with the QUERY on the lane (i' = query r): qn*scale and do, channels in the k-step pattern
        const int qi = it * 32 + r;
        const bool qok = qi < ntok;
        const int qic = qok ? qi : ntok - 1;
        float qa[16], da[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 4 * (2 * g + hh);
            const float4 q4 = *(const float4*)(base + (size_t)qic * C3 + head * 32 + d0);
            const float4 d4 = *(const float4*)(dout + ((size_t)b * ntok + qic) * CO + head * 32 + d0);
            qa[4 * g] = qok ? q4.x * iq[4 * g] * scale : 0.f; qa[4 * g + 1] = qok ? q4.y * iq[4 * g + 1] * scale : 0.f;
            qa[4 * g + 2] = qok ? q4.z * iq[4 * g + 2] * scale : 0.f; qa[4 * g + 3] = qok ? q4.w * iq[4 * g + 3] * scale : 0.f;
            da[4 * g] = qok ? d4.x : 0.f; da[4 * g + 1] = qok ? d4.y : 0.f; da[4 * g + 2] = qok ? d4.z : 0.f; da[4 * g + 3] = qok ? d4.w : 0.f;
        }
        f32x16 st, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { st[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], kv[s], st, 0, 0, 0);      // S[i][j]  (rows = queries)
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da[s], vv[s], dp, 0, 0, 0);      // dP[i][j] = sum_d do[i][d] v[j][d]
        }
        // operands of the accumulations: lane = channel d, k = query row acc_row(s, hh)
        float qd[16], dd2[16], ps[16], dss[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int i = it * 32 + acc_row(s, hh);
            const bool iok = i < ntok;
            const int ic = iok ? i : ntok - 1;
            const size_t row = (size_t)b * ntok + ic;
            const float mi = ml[(row * heads + head) * 2], li = 1.0f / ml[(row * heads + head) * 2 + 1];
            const float dl = delta[row * heads + head];
            ps[s] = iok ? __expf(st[s] - mi) * li : 0.f;
            dss[s] = ps[s] * (dp[s] - dl);
            qd[s] = iok ? base[(size_t)ic * C3 + head * 32 + r] * ivq[r] : 0.f;                       // qn[i][d=r]
            dd2[s] = iok ? dout[row * CO + head * 32 + r] : 0.f;                                     // do[i][d=r]
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            akn = __builtin_amdgcn_mfma_f32_32x32x2f32(qd[s], dss[s], akn, 0, 0, 0);   // dkn^T[d][j] += qn[i][d] dS[i][j]
            adv = __builtin_amdgcn_mfma_f32_32x32x2f32(dd2[s], ps[s], adv, 0, 0, 0);   // dv^T[d][j]  += do[i][d] P[i][j]
        }
    }
    if (kok) {
        float* o1 = dkn + ((size_t)b * ntok + kj) * CO + head * 32 + 4 * hh;
        float* o2 = dv + ((size_t)b * ntok + kj) * CO + head * 32 + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *(float4*)(o1 + 8 * g) = make_float4(akn[4 * g] * scale, akn[4 * g + 1] * scale, akn[4 * g + 2] * scale, akn[4 * g + 3] * scale);
            *(float4*)(o2 + 8 * g) = make_float4(adv[4 * g], adv[4 * g + 1], adv[4 * g + 2], adv[4 * g + 3]);
        }
    }
}

// l2norm-over-tokens backward for q and k (F.normalize(dim=-1), eps 1e-12) + assembly of dqkv (B,ntok,384):
//   d x[i][c] = inv[c] * (dxn[i][c] - xn[i][c] * sum_i xn dxn),  xn = x*inv.   One block per (sample, 32-channel group).
__global__ __launch_bounds__(256) void fa_bwd_l2n_kernel(const float* __restrict__ qkv, const float* __restrict__ inv,
                                                         const float* __restrict__ dqn, const float* __restrict__ dkn,
                                                         const float* __restrict__ dv, int ntok, int heads,
                                                         float* __restrict__ dqkv) {
    __shared__ float red[8][32];
    const int CO = heads * 32, C3 = 3 * CO;
    const int groups = C3 / 32;
    const int b = blockIdx.x / groups, cg = blockIdx.x - b * groups;
    const int c = cg * 32 + (threadIdx.x & 31), sub = threadIdx.x >> 5;               // 8 token lanes x 32 channels
    const int which = c / CO, cc = c - which * CO;                                     // 0 q, 1 k, 2 v
    const float* gsrc = which == 0 ? dqn : (which == 1 ? dkn : dv);
    const float iv = which < 2 ? inv[(size_t)b * C3 + c] : 1.f;
    float s = 0.f;
    if (which < 2)
        for (int i = sub; i < ntok; i += 8)
            s = fmaf(qkv[((size_t)b * ntok + i) * C3 + c] * iv, gsrc[((size_t)b * ntok + i) * CO + cc], s);
    red[sub][threadIdx.x & 31] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += red[k][threadIdx.x & 31];
    for (int i = sub; i < ntok; i += 8) {
        const size_t o = ((size_t)b * ntok + i) * C3 + c;
        const float g = gsrc[((size_t)b * ntok + i) * CO + cc];
        dqkv[o] = which < 2 ? iv * (g - qkv[o] * iv * tot) : g;
    }
}

}  // namespace

// Workspace (floats): delta B*ntok*heads, dqn/dkn/dv B*ntok*heads*32 each.  ml: (B,ntok,heads,2) from the forward.
int op_full_attention_backward(const float* qkv, const float* inv, const float* out, const float* ml, const float* dout, int B,
                               int ntok, int heads, float scale, float* delta, float* dqn, float* dkn, float* dv, float* dqkv,
                               hipStream_t st) {
    const long long rows = (long long)B * ntok * heads;
    hipLaunchKernelGGL(fa_bwd_delta_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, dout, out, rows, delta);
    const long long waves = (long long)B * ((ntok + 31) / 32) * heads;
    hipLaunchKernelGGL(fa_bwd_q_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, qkv, inv, dout, ml, delta, B, ntok, heads, scale, dqn);
    hipLaunchKernelGGL(fa_bwd_kv_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, qkv, inv, dout, ml, delta, B, ntok, heads, scale, dkn, dv);
    hipLaunchKernelGGL(fa_bwd_l2n_kernel, dim3(B * (3 * heads)), dim3(256), 0, st, qkv, inv, dqn, dkn, dv, ntok, heads, dqkv);
    return LAUNCH_OK();
}
