/*
 * tedm_hip_ops.h -- op-level entry points of libtedm_hip.so used by the parity tests and the
 * tuning harness.  They wrap single kernels of the hot path so a failure in the whole-UNet
 * parity test can be localised; the product path (tedm_hip.h) does not need them.
 */
#ifndef TEDM_HIP_OPS_H
#define TEDM_HIP_OPS_H

#ifdef __cplusplus
extern "C" {
#endif

/* One nn.Conv2d (models/unet_model.py:43,49,122,156,185,188,221,222,307,323) through the NHWC
 * implicit-GEMM MFMA kernel, including the fusions the UNet uses:
 *   x1 (B,H,W,C1) [+ x2 (B,H,W,C2) virtual channel concat], optional nearest x2 upsample,
 *   prologue: GroupNorm-apply+FiLM+SiLU folded to silu(gn_a[b][c]*x + gn_b[b][c])   (gn_a != NULL), or
 *             channel LayerNorm  ln_g[c]*(x*rstd - mean*rstd), ln_stats (B*H*W,2) = (rstd, mean*rstd);
 *   weight_oihw (Cout,C1+C2,k,k) fp32 in the reference layout (packed internally), bias (Cout) or NULL;
 *   epilogue: out = conv + bias + [add | silu(add_a[b][c]*add + add_b[b][c])];
 *   stats: per-channel partial slab, mode 1: (B,E,Cout,2) sum/sumsq, mode 2: (B,E,Cout) max; E is
 *          returned in *stats_entries (and by tedm_op_conv2d_stats_entries).
 * Supported geometries: k=3 s=1 p=1 (optionally upsample), k=4 s=2 p=1, k=1 s=1 p=0.
 * upsample: 0 none; 1 nearest x2 fused into the halo gather of the 3x3 kernel; 3 the same operator (Upsample,
 *   models/unet_model.py:39-44) as four 2x2 parity convolutions on the low-resolution image with pre-summed taps -- the form
 *   the UNet forward uses (one source, no statistics slab).
 * Synchronises the stream before returning (test helper). */
int tedm_op_conv2d(const float* x1, int C1, const float* x2, int C2, int B, int H, int W,
                   const float* weight_oihw, const float* bias, int Cout, int k, int stride, int upsample,
                   const float* gn_a, const float* gn_b, const float* ln_g, const float* ln_stats,
                   const float* add, const float* add_a, const float* add_b,
                   float* out, float* stats, int stats_mode, int* stats_entries, int prec, void* stream);
int tedm_op_conv2d_stats_entries(int Cout, int k, int stride, int Ho, int Wo);
/* Q16 activation storage (tedm_amd/csrc/common.h): int16 data [B][H][W][C] followed by one power-of-two scale byte per block
 * (row, 32-column tile, half, channel) = the 16 pixels one lane of the conv's MFMA accumulator holds.  The fast inference mode stores
 * the two conv outputs inside a ResnetBlock this way on the HBM-bound levels (tedm_set_tuning("act_q16", 0) turns it off;
 * "act_q16_max_c" bounds the block width it applies to).  Test surface:
 *   tedm_q16_bytes: size of the buffer of a (B, H, W, C) tensor (C % 8 == 0), 0 on bad arguments;
 *   tedm_q16_encode / tedm_q16_decode: fp32 NHWC <-> Q16; decode(encode(x)) is the format's rounding;
 *   tedm_op_conv2d_q16: one k x k (3 or 1) stride-1 conv; q16_flags bit 0: x is Q16 (needs the GroupNorm prologue coefficients),
 *   bit 1: out is Q16, bit 2: add is Q16 (the silu(add_a * add + add_b) resnet addend); emits the sum / sum-of-squares slab when `stats`;
 *   tedm_op_resblock_out_q16: out = silu(ga * y + gb) + x with y in Q16 (+ channel-LayerNorm statistics when lnstats != NULL). */
size_t tedm_q16_bytes(int B, int H, int W, int C);
int tedm_q16_encode(const float* x, void* q, int B, int H, int W, int C, void* stream);
int tedm_q16_decode(const void* q, float* x, int B, int H, int W, int C, void* stream);
int tedm_op_conv2d_q16(const void* x, int C, int B, int H, int W, const float* weight_oihw, const float* bias, int Cout, int k,
                       const float* gn_a, const float* gn_b, const void* add, const float* add_a, const float* add_b, void* out,
                       float* stats, int* stats_entries, int prec, int q16_flags, void* stream);
int tedm_op_resblock_out_q16(const void* y_q16, const float* ga, const float* gb, const float* x, float* out, float* lnstats, int B,
                             int H, int W, int C, void* stream);
/* Tuning harness: the same conv, weights packed once, 2 warm-up launches, then `iters` launches bracketed by HIP events on
 * `stream`; *ms_per_launch = their mean duration. */
int tedm_op_conv2d_bench(const float* x1, int C1, const float* x2, int C2, int B, int H, int W,
                         const float* weight_oihw, const float* bias, int Cout, int k, int stride, int upsample,
                         const float* gn_a, const float* gn_b, const float* add, float* out, float* stats, int stats_mode,
                         int prec, int iters, float* ms_per_launch, void* stream);

/* Backward of one nn.Conv2d (same geometries as tedm_op_conv2d; `upsample` = the forward conv read the
 * nearest-x2 upsampled x).  dy (B,Ho,Wo,Cout), x (B,H,W,Cin) NHWC; dx (B,H,W,Cin) and/or dw (Cout,Cin,k,k)
 * OIHW may be NULL.  dgrad = the forward implicit-GEMM kernel with flipped/transposed packed weights (zero
 * insertion for the stride-2 transpose, 2x2 sum-pool after the upsampled case); wgrad = wgrad.hip. */
int tedm_op_conv2d_backward(const float* dy, const float* x, const float* weight_oihw, int B, int H, int W, int Cin,
                            int Cout, int k, int stride, int upsample, float* dx, float* dw, int prec, void* stream);

/* Backward of Block's GroupNorm -> FiLM -> SiLU tail (models/unet_model.py:126-135) given the gradient G of the
 * activated output and the raw conv output y (both (B,hw,C) NHWC).  fa, fb (B,C): the forward's folded
 * coefficients (u = fa*y + fb); mr (B,groups,2) = (mean, rstd).  film: (1+scale) source, scale at
 * [n*film_stride + c], shift at +C, or NULL; dfilm receives d scale / d shift in the same layout.
 * Outputs: dy (B,hw,C) = gradient w.r.t. the conv output, dgamma, dbeta (C), dbias (C) = sum of dy. */
int tedm_op_gn_silu_backward(const float* G, const float* y, const float* fa, const float* fb, const float* mr,
                             const float* gamma, const float* beta, const float* film, int film_stride, float* dfilm,
                             int B, long long hw, int C, int groups, float* dy, float* dgamma, float* dbeta, float* dbias,
                             void* stream);
/* channel-LayerNorm backward (models/unet_model.py:52-61): dx = LN'(dY; x, g) [+ add], dg = sum_px dY * xhat. */
int tedm_op_ln_backward(const float* dY, const float* x, const float* g, const float* add, long long P, int C, float* dx,
                        float* dg, void* stream);

/* Attention cores on a materialised qkv tensor, forward + backward (tests of attention.hip / attention_bwd.hip):
 * linear attention (unet_model.py:197-207): qkv (B,hw,384) -> o (B,hw,128); dout (B,hw,128) -> dqkv (B,hw,384);
 * bottleneck attention (unet_model.py:229-239): qkv (B,ntok,384) -> o (B,ntok,128); dout -> dqkv. */
int tedm_op_linattn_fwd_bwd(const float* qkv, const float* dout, int B, long long hw, float* o, float* dqkv, void* stream);
int tedm_op_full_attention_fwd_bwd(const float* qkv, const float* dout, int B, int ntok, float* o, float* dqkv, void* stream);

/* LinearAttention core (models/unet_model.py:197-207) through the fused sweeps of linattn_fused.hip:
 *   x (B,hw,C) fp32, PreNorm gain g (C), lnstats (B*hw,2) = (rstd, mean*rstd), wqkv_oihw (384,C,1,1);
 *   outputs: kmax (B,128) = max over pixels of k, ctx (B,4,32,32) = softmax_n(k) v^T / hw,
 *   o (B,hw,128) = ctx^T softmax_d(q) * 32^-0.5 (the tensor to_out consumes).  heads = 4, dim_head = 32.
 * Synchronises the stream before returning (test helper). */
int tedm_op_linattn_fused(const float* x, int C, const float* g, const float* lnstats, const float* wqkv_oihw,
                          int B, long long hw, float* kmax_out, float* ctx_out, float* o_out, int prec, void* stream);

/* Tuning knobs (benchmark harness / A-B tests; defaults are the product's): "conv_tile_wide", "conv_tile_wide256",
 * "conv_tile_narrow" (workgroup tile per output-channel range), "conv_tpw_max", "conv_xcd_order", "conv_prio", "conv_solo"
 * (conv_igemm.hip; tile ids 0-6, of which 5 = 1x8 waves x 8 rows and 6 = 4x2 waves are the defaults for Cout >= 256 / Cout < 128;
 * every tile gives the same bits), "conv_small_grid" 0|1 (smaller tiles for launches of < 256 workgroups); "conv_up_split" 0|1
 * (Upsample as parity convolutions), "la_fused" 0|1|2, "la_tail" 0|1, "la_ctx_one" 0|1 (context sweep at C = 64 behind one barrier
 * per tile), "la_self_ln" 0|1 (the fused linear-attention kernels compute the PreNorm statistics themselves where they can),
 * "final_fuse" 0|1 (final_res_block's tail folded with final_conv; inference), "act_q16" / "act_q16_max_c" (Q16 branch storage,
 * off) (unet_host.hip, linattn_fused.hip); "deep_fp16" 0|1|2 (which convolutions run one fp16 product under TEDM_PREC_F16F4: none | the 3x3 convs
 * inside the ResnetBlocks below full resolution (default) | every k x k conv with >= 256 output channels; read by tedm_unet_pack,
 * so set it before packing); wgrad keys (wgrad.hip).  Unknown keys / values return TEDM_EINVAL. */
int tedm_set_tuning(const char* key, int value);

/* HIP-event profiler (bench.py's live roofline): brackets kernel launches on the launch stream with
 * event pairs.  level 0 off, 1 convolutions only, 2 every kernel of the forward.  After the timed
 * region call tedm_profile_collect() (synchronises on the recorded events), then read the entries:
 * kernel label (== rocprof symbol for the convolutions), launches, total ms, total algorithmic
 * flops and bytes. */
int tedm_profile_enable(int level);
int tedm_profile_reset(void);
/* level 1 only: bracket just the launches whose label equals `label` (NULL / "" = every convolution).  An event pair costs
 * ~5 us of launch gap, so a timed region that wants one kernel's duration should not pay for the other 74 convolutions. */
int tedm_profile_filter(const char* label);
int tedm_profile_collect(void);
int tedm_profile_count(void);
int tedm_profile_entry(int i, char* name, int name_cap, long long* launches, double* ms, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif
