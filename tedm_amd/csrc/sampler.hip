// sampler.hip -- one ancestral sampling step behind the UNet forward as ONE kernel.
//
// Replaces the tensor passes of models/diffusion_model.py:205-267 that follow `self.model(x_t, t)`:
//   model_predictions       :237-257   x0 = sqrt_recip_acp[t] x_t - sqrt_recipm1_acp[t] eps   (or eps from a predicted x0)
//   p_mean_variance         :221-235   s = quantile(|x0|, 0.995) per sample (torch.quantile = a full sort), s = max(s, 1),
//                                      x0 = clip(x0, -s, s) / s
//   q_posterior             :259-267   mean = coef1[t] x0 + coef2[t] x_t,  log variance = posterior_log_variance_clipped[t]
//   sample_timestep         :205-219   x_{t-1} = mean + exp(0.5 logvar) noise   (no noise at t = 0)
// One workgroup per sample.  The quantile is an exact selection, not a sort: |x0| >= 0, so the fp32 bit patterns order like
// unsigned integers and a 4-pass radix select (8 + 8 + 8 + 7 bits, per-wave LDS histograms) finds the order statistic of rank
// floor(q (n - 1)); its upper neighbour is the same value or the minimum of the larger elements (one more pass).  The
// interpolation is torch.quantile's: ranks = q (n - 1) in fp32 (the host passes floor and weight), lerp as ATen/native/Lerp.h.
// Samples of up to 64 K values (256 x 256, the size the model is used at) take the register-resident instance: every thread keeps
// its 64 values, and the order statistic is built bit by bit -- 31 rounds of "how many values are below prefix | bit?", each one
// compare + add per value, a wave reduction and ONE barrier (LDS histograms with atomics cost 165 us per step here: the values
// of a pass crowd into a few bins and same-address LDS atomics serialise; this form takes ~20 us).
#include "common.h"
#include "ops.h"
#include "../../include/tedm_hip.h"

namespace {

constexpr int SP_T = 1024, SP_W = SP_T / 64;

struct SamplerArgs {
    const float* x_t; const float* model_out; const long long* t; const float* noise;
    const float* sqrt_recip_acp; const float* sqrt_recipm1_acp; const float* coef1; const float* coef2; const float* logvar;
    float* x0; float* mean; float* x_prev; float* s_out;
    long long chw; int T, objective /*0: model_out = eps, 1: model_out = x0*/, clip, k_lo; float w;
};

__global__ __launch_bounds__(SP_T) void sampler_step_kernel(SamplerArgs a) {
    __shared__ unsigned hist[SP_W][256];
    __shared__ unsigned sh_prefix, sh_rank, sh_cnt, sh_min;
    const int tid = threadIdx.x, wave = tid >> 6, b = blockIdx.x;
    const long long n = a.chw;
    long long ti = a.t[b];
    ti = ti < 0 ? 0 : (ti >= a.T ? a.T - 1 : ti);
    const float c_r = a.sqrt_recip_acp[ti], c_m = a.sqrt_recipm1_acp[ti];
    const float* xt = a.x_t + (size_t)b * n;
    const float* mo = a.model_out + (size_t)b * n;
    float* x0 = a.x0 + (size_t)b * n;

    // x0 (unclipped) into the output buffer: the selection passes re-read it (256 KB per 256x256 sample: L2)
    for (long long i = tid; i < n; i += SP_T) x0[i] = a.objective == 0 ? c_r * xt[i] - c_m * mo[i] : mo[i];
    __syncthreads();

    float s = 1.f;
    if (a.clip) {
        unsigned prefix = 0, mask = 0, rank = (unsigned)a.k_lo, cnt_eq = 0;
        constexpr int SH[4] = {23, 15, 7, 0}, NB[4] = {8, 8, 8, 7};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            for (int i = tid; i < SP_W * 256; i += SP_T) (&hist[0][0])[i] = 0;
            __syncthreads();
            for (long long i = tid; i < n; i += SP_T) {
                const unsigned u = __builtin_bit_cast(unsigned, x0[i]) & 0x7fffffffu;
                if ((u & mask) == prefix) atomicAdd(&hist[wave][(u >> SH[p]) & ((1u << NB[p]) - 1)], 1u);
            }
            __syncthreads();
            if (tid < 256) {                                  // fold the per-wave histograms
                unsigned c = 0;
#pragma unroll
                for (int w = 0; w < SP_W; ++w) c += hist[w][tid];
                hist[0][tid] = c;
            }
            __syncthreads();
            if (tid == 0) {
                unsigned cum = 0, d = 0;
                for (; d < (1u << NB[p]) - 1; ++d) { if (cum + hist[0][d] > rank) break; cum += hist[0][d]; }
                sh_prefix = prefix | (d << SH[p]); sh_rank = rank - cum; sh_cnt = hist[0][d];
            }
            __syncthreads();
            prefix = sh_prefix; rank = sh_rank; cnt_eq = sh_cnt;
            mask |= ((1u << NB[p]) - 1) << SH[p];
            __syncthreads();
        }
        const float v_lo = __builtin_bit_cast(float, prefix);
        float v_hi = v_lo;
        if (rank + 1 >= cnt_eq && a.w != 0.f) {                // the upper neighbour is the smallest LARGER element
            if (tid == 0) sh_min = 0x7fffffffu;
            __syncthreads();
            unsigned m = 0x7fffffffu;
            for (long long i = tid; i < n; i += SP_T) {
                const unsigned u = __builtin_bit_cast(unsigned, x0[i]) & 0x7fffffffu;
                if (u > prefix && u < m) m = u;
            }
            atomicMin(&sh_min, m);
            __syncthreads();
            v_hi = sh_min == 0x7fffffffu ? v_lo : __builtin_bit_cast(float, sh_min);
        }
        // torch.lerp (ATen/native/Lerp.h)
        const float q = a.w < 0.5f ? v_lo + a.w * (v_hi - v_lo) : v_hi - (v_hi - v_lo) * (1.f - a.w);
        s = fmaxf(q, 1.f);
    }
    if (tid == 0 && a.s_out) a.s_out[b] = s;

    const float k1 = a.coef1[ti], k2 = a.coef2[ti];
    const float sig = (a.noise && ti > 0) ? expf(0.5f * a.logvar[ti]) : 0.f;
    const float* nz = a.noise ? a.noise + (size_t)b * n : nullptr;
    float* mean = a.mean ? a.mean + (size_t)b * n : nullptr;
    float* xp = a.x_prev ? a.x_prev + (size_t)b * n : nullptr;
    for (long long i = tid; i < n; i += SP_T) {
        float v = x0[i];
        if (a.clip) v = fminf(fmaxf(v, -s), s) / s;
        x0[i] = v;
        const float m = k1 * v + k2 * xt[i];
        if (mean) mean[i] = m;
        if (xp) xp[i] = nz ? m + sig * nz[i] : m;
    }
}


constexpr int SP_EPT = 64;                                    // values per thread of the register-resident instance

__global__ __launch_bounds__(SP_T) void sampler_step_reg_kernel(SamplerArgs a) {
    __shared__ unsigned sh_c[2][SP_W];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int n = (int)a.chw;
    long long ti = a.t[b];
    ti = ti < 0 ? 0 : (ti >= a.T ? a.T - 1 : ti);
    const float c_r = a.sqrt_recip_acp[ti], c_m = a.sqrt_recipm1_acp[ti];
    const float* xt = a.x_t + (size_t)b * n;
    const float* mo = a.model_out + (size_t)b * n;
    float* x0 = a.x0 + (size_t)b * n;

    // |x0| bit patterns (0xffffffff past the end: never below a candidate, never the minimum), signs in a 64-bit mask
    unsigned u[SP_EPT];
    unsigned long long sg = 0ull;
#pragma unroll
    for (int i = 0; i < SP_EPT; ++i) {
        const int e = tid + i * SP_T;
        u[i] = 0xffffffffu;
        if (e < n) {
            const float v = a.objective == 0 ? c_r * xt[e] - c_m * mo[e] : mo[e];
            const unsigned bits = __builtin_bit_cast(unsigned, v);
            u[i] = bits & 0x7fffffffu;
            sg |= (unsigned long long)(bits >> 31) << i;
        }
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);     // eight values' loads in flight at a time (128-register budget)
    }
    int round = 0;
    auto block_sum = [&](unsigned c) -> unsigned {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += (unsigned)__shfl_xor((int)c, o);
        const int par = round++ & 1;                           // two buffers: one barrier per round
        if (lane == 0) sh_c[par][wave] = c;
        __syncthreads();
        unsigned t = 0;
#pragma unroll
        for (int w = 0; w < SP_W; ++w) t += sh_c[par][w];
        return t;
    };
    float s = 1.f;
    if (a.clip) {
        // largest v with count(u < v) <= k: the value of rank k (0-based) in ascending order
        unsigned v = 0u;
        const unsigned k = (unsigned)a.k_lo;
#pragma unroll 1
        for (int bit = 30; bit >= 0; --bit) {
            const unsigned cand = v | (1u << bit);
            unsigned c = 0;
#pragma unroll
            for (int i = 0; i < SP_EPT; ++i) c += u[i] < cand ? 1u : 0u;
            if (block_sum(c) <= k) v = cand;
        }
        const float v_lo = __builtin_bit_cast(float, v);
        float v_hi = v_lo;
        if (a.w != 0.f) {
            unsigned c = 0, m = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < SP_EPT; ++i) { c += u[i] <= v ? 1u : 0u; m = (u[i] > v && u[i] < m) ? u[i] : m; }
            const unsigned c_le = block_sum(c);
            if (k + 1 >= c_le) {                               // rank k + 1 lies past the copies of v: the smallest larger value
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o));
                const int par = round++ & 1;
                if (lane == 0) sh_c[par][wave] = m;
                __syncthreads();
                unsigned mm = 0xffffffffu;
#pragma unroll
                for (int w = 0; w < SP_W; ++w) mm = min(mm, sh_c[par][w]);
                v_hi = mm == 0xffffffffu ? v_lo : __builtin_bit_cast(float, mm);
            }
        }
        const float q = a.w < 0.5f ? v_lo + a.w * (v_hi - v_lo) : v_hi - (v_hi - v_lo) * (1.f - a.w);     // torch.lerp (ATen/native/Lerp.h)
        s = fmaxf(q, 1.f);
    }
    if (tid == 0 && a.s_out) a.s_out[b] = s;

    const float k1 = a.coef1[ti], k2 = a.coef2[ti];
    const float sig = (a.noise && ti > 0) ? expf(0.5f * a.logvar[ti]) : 0.f;
    const float* nz = a.noise ? a.noise + (size_t)b * n : nullptr;
    float* mean = a.mean ? a.mean + (size_t)b * n : nullptr;
    float* xp = a.x_prev ? a.x_prev + (size_t)b * n : nullptr;
#pragma unroll
    for (int i = 0; i < SP_EPT; ++i) {
        const int e = tid + i * SP_T;
        if (e < n) {
            float v = __builtin_bit_cast(float, u[i] | ((unsigned)((sg >> i) & 1ull) << 31));
            if (a.clip) v = fminf(fmaxf(v, -s), s) / s;
            x0[e] = v;
            const float m = k1 * v + k2 * xt[e];
            if (mean) mean[e] = m;
            if (xp) xp[e] = nz ? m + sig * nz[e] : m;
        }
        if ((i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

extern "C" int tedm_sampler_step(const float* x_t, const float* model_out, const int64_t* t, const float* noise, const float* sqrt_recip_acp,
                                 const float* sqrt_recipm1_acp, const float* coef1, const float* coef2, const float* logvar, int T, int B,
                                 long long chw, int objective, int clip, long long k_lo, float w, float* x0, float* mean, float* x_prev,
                                 float* s_out, void* stream) {
    if (!x_t || !model_out || !t || !sqrt_recip_acp || !sqrt_recipm1_acp || !coef1 || !coef2 || !logvar || !x0 || B < 1 || chw < 1 || T < 1 ||
        (objective != 0 && objective != 1) || k_lo < 0 || k_lo >= chw || chw > 0x7fffffffLL)
        return TEDM_EINVAL;
    SamplerArgs a{x_t, model_out, (const long long*)t, noise, sqrt_recip_acp, sqrt_recipm1_acp, coef1, coef2, logvar,
                  x0, mean, x_prev, s_out, chw, T, objective, clip, (int)k_lo, w};
    if (chw <= (long long)SP_EPT * SP_T) hipLaunchKernelGGL(sampler_step_reg_kernel, dim3(B), dim3(SP_T), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(sampler_step_kernel, dim3(B), dim3(SP_T), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? TEDM_OK : TEDM_EHIP;
}
