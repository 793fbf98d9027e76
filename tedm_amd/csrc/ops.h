// Host-side launchers of the non-conv kernels (elementwise.hip, attention.hip).  All tensors are
// fp32, device resident, NHWC unless stated; every launcher enqueues on `st` and returns TEDM_*.
#pragma once
#include <hip/hip_runtime.h>

int op_q_sample(const float* x0, const float* noise, const long long* t, const float* ca, const float* cb, int T,
                float* xt, int B, long long chw, int normalize, hipStream_t st);
int op_stem_conv(const float* x_nchw, const float* w, const float* bias, float* out, int B, int Cin, int H, int W,
                 int Cout, hipStream_t st);
// dim: width of the embedding fed to time_mlp.1; learned != 0: LearnedSinusoidalPosEmb (freqs = its `weights`, (dim - 1) / 2 of them)
int op_time_embed(const float* t, const float* freqs, const float* w1, const float* b1, const float* w2,
                  const float* b2, int B, int dim, int tdim, int learned, float* st_out, float* save /*tape or NULL*/, hipStream_t st);
// d weights of LearnedSinusoidalPosEmb from d emb [B][2 half + 1] and the saved embedding [B][2 half + 1] = (t, sin, cos)
int op_learned_sinusoid_backward(const float* demb, const float* emb, int B, int half, float* dw, hipStream_t st);
int op_film(const float* st_in, const float* wf, const float* bf, int B, int tdim, int J, float* film, hipStream_t st);
int op_gn_finalize(const float* slab, int E, int B, int C, int groups, long long hw, const float* gamma,
                   const float* beta, const float* film, int film_stride, float* oa, float* ob, float* mr_out,
                   hipStream_t st);
int op_resblock_out(const float* y, const float* ga, const float* gb, const float* x, float* out, float* lnstats,
                    int B, long long hw, int C, hipStream_t st);
// Q16 activation buffers (common.h): int16 data + block-scale bytes; C / 8 a power of two <= 64 for resblock_out
size_t q16_bytes(int B, int H, int W, int C);
int op_resblock_out_q16(const void* y_q16, const float* ga, const float* gb, const float* x, float* out, float* lnstats,
                        int B, int H, int W, int C, hipStream_t st);
int op_q16_encode(const float* x, void* q, int B, int H, int W, int C, hipStream_t st);       // C % 8 == 0
int op_q16_decode(const void* q, float* x, int B, int H, int W, int C, hipStream_t st);
int op_ln_stats(const float* x, float* lnstats, long long P, int C, hipStream_t st);
int op_ln_res(const float* z, const float* g, const float* x, float* out, long long P, int C, hipStream_t st);
int op_final_fused(const float* x1, const float* x2, const float* y2, const float* ga, const float* gb, const float* wres,
                   const float* bres, const float* wf, const float* bf, float* wfold, float* out_nchw, int B, long long hw,
                   int C, int OD, hipStream_t st);
bool final_fused_supported(int C1, int C2, int C);
int op_final_conv(const float* x, const float* w, const float* bias, float* out_nchw, int B, long long hw, int C,
                  int OD, hipStream_t st);
int op_colnorm_finalize(const float* slab, int E, int C, int B, float* inv, hipStream_t st);
int op_colmax_finalize(const float* slab, int E, int C, int B, float* mx, hipStream_t st);
int op_tap_to_features(const float* tap, int B, int h, int w, int C, float* feats_nchw, int C_total, int c_off, int S,
                       hipStream_t st);

// attention.hip  (qkv: (B, hw, 3*heads*32) fp32 with channel order [q | k | v], head-major inside each)
int la_ctx_splits(long long hw);
size_t la_ctx_partial_floats(int B, int heads, long long hw);
int op_linattn_context(const float* qkv, const float* kmax, int B, long long hw, int heads, float* partial,
                       float* ctx, float* zout, hipStream_t st);
int op_linattn_out(const float* qkv, const float* ctx, int B, long long hw, int heads, float* o, float* qprime_inplace,
                   hipStream_t st);
int op_full_attention(const float* qkv, const float* inv_norm, int B, int ntok, int heads, float scale, float* o, float* ml,
                      hipStream_t st);

int op_linattn_ctx_finalize(const float* partial, int B, int heads, int S, long long hw, float* ctx, hipStream_t st);

// linattn_fused.hip: LinearAttention without the qkv tensor (three recompute sweeps over x)
int la_fused_wgs_per_image(int sweep, long long hw);
// merge of the online-max context partials (sweep 1 called with kmax == NULL; partial stride 1024+64 floats per head)
int op_linattn_ctx_merge(const float* partial, int B, int heads, int S, long long hw, float* ctx, hipStream_t st);
bool la_tail_supported(int C, int heads, int prec);
bool la_self_ln_supported(int C, int heads, int prec, int sweep);   // lnstats == nullptr accepted (sweep: 1 = context, -1 = fused tail)
int op_linattn_qout_tail(const float* x, int C, const float* g, const float* lnstats, const void* wqkv, const float* ctx,
                         const void* wout, const float* bias, const float* g_out, int B, long long hw, int heads, float* out,
                         hipStream_t st);
int op_linattn_fused(int sweep /*0 kmax, 1 ctx, 2 q-out*/, const float* x, int C, const float* g, const float* lnstats,
                     const void* wpk, int B, long long hw, int heads, float* kmax_part, const float* kmax,
                     float* ctx_part, const float* ctx, float* o, int prec, hipStream_t st);

// attention_bwd.hip
int op_linattn_backward(const float* qkv_qprime, const float* dout, const float* ctx, const float* kmax, const float* Z, int B,
                        long long hw, int heads, float* part, float* dctx, float* T, float* dqkv, hipStream_t st);
int op_full_attention_backward(const float* qkv, const float* inv, const float* out, const float* ml, const float* dout, int B,
                               int ntok, int heads, float scale, float* delta, float* dqn, float* dkn, float* dv, float* dqkv,
                               hipStream_t st);

// wgrad.hip
size_t wgrad_partial_floats(int k, int stride, int Cout, int Cin, int B, int Ho, int Wo);
int op_conv_wgrad(const float* dy, int Cout, const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int Ho,
                  int Wo, int k, int stride, int ups, int pro, const float* pro_a, const float* pro_b, float* part,
                  float* dw, int accumulate, int fast /*1: single fp16 product on the 3x3 path (f16f4 training mode)*/,
                  const unsigned* dy_amax /*nullable: max |dy| bits if the producer of dy already has it*/, hipStream_t st);

// wgrad_mfma.hip (bf16x3 matrix-core path for stride-1 3x3 / 1x1; called by op_conv_wgrad)
size_t wgrad_mfma_floats(int k, int Cout, int Cin, int B, int Ho, int Wo);
int op_conv_wgrad_mfma(const float* dy, int Cout, const float* x1, int C1, const float* x2, int C2, int B, int H, int W, int Ho, int Wo,
                       int k, int ups, int pro, const float* pro_a, const float* pro_b, float* ws, int* S_out, int f16, const unsigned* dy_amax, hipStream_t st);
int wgrad_set_tuning(const char* key, int v);
int la_set_tuning(const char* key, int v);
int wgrad_mfma_set_tuning(const char* key, int v);

// backward.hip
// eval-mode classifier tail fused with the up-sampling sum: z_i (B,h_i,w_i,C1) -> logits (B,S,S); w2t [C1][32], BatchNorm folded by the host
int op_head_tail(const float* const* z, const int* h, const int* w, int n, int B, int S, int C1, const float* b1, const float* w2t,
                 const float* b2, const float* w3, const float* b3, float* logits, hipStream_t st);
int op_upsample_sum(const float* const* z, const int* h, const int* w, int n, int B, int S, int C, const float* bias, float* out,
                    hipStream_t st);
int op_pool_sum(const float* g, float* out, int B, int h, int w, int f, int C, hipStream_t st);
int op_pool2x2_sum(const float* g, float* out, int B, int h, int w, int C, hipStream_t st);
int linear_bwd_splits(int J);
int op_linear_backward(const float* dy, int dy_stride, const float* x, const float* W, int B, int J, int K, float* dW, float* db,
                       float* dx, float* part /*linear_bwd_splits(J)*B*K floats when dx*/, int accumulate, hipStream_t st);
int op_act_backward(float* g, const float* pre, int n, int mode /*0 silu, 1 gelu*/, hipStream_t st);
int op_add2(const float* a, const float* b, float* out, long long n, hipStream_t st);
int final_conv_bwd_blocks(long long P);
int op_final_conv_backward(const float* dY_nchw, const float* x, const float* w, float* dx, float* part, float* dw, float* db,
                           int B, long long hw, int C, int OD, int accumulate, hipStream_t st);
int stem_wgrad_blocks(int B, int H, int W);
int op_stem_wgrad(const float* dY, const float* x_nchw, int B, int Cin, int H, int W, int Cout, float* part, float* dW,
                  int accumulate, hipStream_t st);
int op_l1_loss(const float* eps, const float* target, const long long* t, const float* p2w, int T, int B, long long chw,
               float* grad, float* part, float* loss, hipStream_t st);
int gn_bwd_chunks(long long hw);
int op_gn_silu_backward(const float* G, const float* y, const float* fa, const float* fb, const float* mr, const float* gamma,
                        const float* beta, const float* film, int film_stride, float* dfilm, int B, long long hw, int C,
                        int groups, float* slab, float* ca, float* mcoef, float* dgam_n, float* dbet_n, float* dbias_n,
                        float* dy, float* dgamma, float* dbeta, float* dbias, int accumulate, unsigned* amax /*nullable: out, max |dy| bits*/, hipStream_t st);
int ln_bwd_blocks(long long P, int C);
int op_ln_backward(const float* dY, const float* x, const float* g, const float* add, float* dx, float* part, float* dg,
                   int accumulate, long long P, int C, hipStream_t st);
int op_colsum(const float* x, float* part, float* out, int accumulate, long long P, int C, hipStream_t st);

int conv_set_tuning(const char* key, int v);
