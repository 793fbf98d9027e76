/*
 * tedm_hip.h -- C ABI of libtedm_hip.so, the MI355X (gfx950) implementation of TEDM's
 * diffusion-UNet denoising hot path.
 *
 * The reference (mmr12/TEDM) is pure Python and has no FFI; the boundary it offers for this
 * path is the nn.Module surface of models/unet_model.py:246-368 (`Unet`), wrapped by
 * models/diffusion_model.py:50-203 (`DiffusionModel`) and models/datasetDM_model.py:30-88
 * (`DatasetDM`).  Each entry point below names the reference interface it stands in for.
 * The Python facade `tedm_amd.unet.Unet` binds these with ctypes.
 *
 * Conventions: every function returns 0 on success or a negative TEDM_E* code and never
 * throws; `tedm_last_error()` gives a thread-local message.  All device pointers are
 * BORROWED for the duration of the call (the caller -- PyTorch's allocator -- owns them),
 * except packed weights, which the handle owns.  Kernels are enqueued on the caller's HIP
 * stream (`stream` is a hipStream_t passed as void*); nothing synchronises implicitly.
 * A handle is not thread-safe: one per process / GPU.
 */
#ifndef TEDM_HIP_H
#define TEDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEDM_OK            0
#define TEDM_EINVAL       -1   /* bad argument / unsupported configuration */
#define TEDM_ENOKEY       -2   /* unknown state-dict key */
#define TEDM_EHIP         -3   /* HIP runtime error */
#define TEDM_ENOMEM       -4   /* workspace too small / allocation failed */
#define TEDM_ESTATE       -5   /* call order (e.g. forward before pack) */

#define TEDM_MAX_LEVELS    8

/* Arithmetic mode of the MFMA contractions (convs, 1x1 projections). */
#define TEDM_PREC_BF16X3   0   /* default: bf16 hi/lo split, 3 MFMAs, fp32 accumulate (~2^-17 operand error) */
#define TEDM_PREC_BF16     1   /* single bf16 MFMA, fp32 accumulate (fast; eps error ~1e-2, fails the 1e-3 gate) */
#define TEDM_PREC_FP16X2   2   /* fp16 activations x (fp16 hi + fp16 lo) weights, 2 MFMAs, fp32 accumulate (experimental) */
#define TEDM_PREC_F16F4    3   /* fp16 x fp16 MFMA + ONE block-scaled fp4 MFMA (K = 64) carrying both cross terms
                                  A_lo*B + A*B_lo of the fp16 split: 1.5 bf16-MFMA issues per 16 channels instead of 3;
                                  eps error ~2-3e-4 (inside the 1e-3 gate) */
#define TEDM_PREC_FP16     4   /* ONE fp16 x fp16 MFMA per 16 channels (operand error 2^-12).  Not a whole-network mode (eps 9.5e-4):
                                  F16F4 uses it for the k x k convolutions with >= 256 output channels, where it costs 1e-4 of
                                  eps (tests/precision_study.py mixed256) and the matrix pipe / LDS / weight traffic fall by a third */

/* forward flags */
#define TEDM_FWD_TAPS_ONLY   (1 << 0)  /* skip ups[last][3], final_res_block, final_conv (feature extraction; eps_out may be NULL) */
#define TEDM_FWD_TRACE       (1 << 1)  /* keep every module output alive in the workspace and record it (tests) */
#define TEDM_FWD_PREC_SHIFT  8         /* bits 8..11: TEDM_PREC_* */

typedef struct tedm_unet tedm_unet;

/* Mirrors the constructor of models/unet_model.py:247-259 (`Unet.__init__`). */
typedef struct tedm_unet_cfg {
    int dim;                       /* base width (reference default 64) */
    int init_dim;                  /* 0 => dim */
    int out_dim;                   /* 0 => channels */
    int n_mults;                   /* len(dim_mults), <= TEDM_MAX_LEVELS */
    int dim_mults[TEDM_MAX_LEVELS];
    int channels;                  /* input image channels (TEDM forces 1, train.py:25) */
    int groups;                    /* resnet_block_groups (8) */
    int heads;                     /* attention heads (4)   -- unet_model.py:180,217 */
    int dim_head;                  /* attention head width (32) */
    int learned_sinusoidal_dim;    /* 0: SinusoidalPosEmb(dim) (unet_model.py:76-93).  > 0 (even): learned_sinusoidal_cond=True with this
                                      learned_sinusoidal_dim (:96-114,278-285): the parameter `time_mlp.0.weights` (dim/2) replaces the fixed
                                      frequency table and time_mlp.1 takes learned_sinusoidal_dim + 1 inputs */
} tedm_unet_cfg;

/* Unet.__init__ (models/unet_model.py:247-331): builds the layer plan (host only; the
 * handle-owned packed-weight buffers are allocated on the current HIP device by the first
 * tedm_unet_pack()). */
int  tedm_unet_create(const tedm_unet_cfg* cfg, tedm_unet** out);
void tedm_unet_destroy(tedm_unet* h);

/* Unet.state_dict() surface (276 tensors for the default model). */
int         tedm_unet_num_params(const tedm_unet* h);
const char* tedm_unet_param_key(const tedm_unet* h, int i);
long long   tedm_unet_param_numel(const tedm_unet* h, int i);

/* nn.Module.load_state_dict / parameter binding: `dev_ptr` is the fp32, contiguous device
 * tensor in the REFERENCE layout (conv OIHW, linear (out,in), norm (C), LayerNorm.g (1,C,1,1)).
 * The pointer is recorded, not copied; call tedm_unet_pack() after (re)binding or after the
 * optimizer changed the values. */
int tedm_unet_set_param(tedm_unet* h, const char* key, const void* dev_ptr, long long numel);

/* Repack all bound weights into the kernels' internal layouts (hi/lo MFMA fragment order in the
 * handle's precision mode -- bf16 planes for TEDM_PREC_BF16X3/BF16, fp16 planes for
 * TEDM_PREC_FP16X2 --, concatenated FiLM projection).  Enqueued on `stream`. */
int tedm_unet_pack(tedm_unet* h, void* stream);

/* Select the arithmetic mode of the MFMA contractions (TEDM_PREC_*; default BF16X3).  Changing the
 * weight encoding (bf16 <-> fp16) invalidates the packed weights: call tedm_unet_pack() again.
 * The precision bits of tedm_unet_forward's flags must name the same mode. */
int tedm_unet_set_precision(tedm_unet* h, int prec);

/* Workspace size in bytes for a forward of the given shape / flags. */
size_t tedm_unet_workspace_bytes(const tedm_unet* h, int B, int H, int W, int flags);

/* Unet.forward(x, timestep) (models/unet_model.py:333-368).
 *   x        (B, channels, H, W) fp32, NCHW contiguous
 *   t        (B,) fp32 timesteps, or NULL => no time conditioning (timestep=None, :337)
 *   eps_out  (B, out_dim, H, W) fp32, NCHW contiguous
 *   feat_out NULL, or 4 (n_mults) device pointers receiving the outputs of ups[i][2]
 *            (the tensors models/datasetDM_model.py:50-53 hooks) in NHWC: (B, h_i, w_i, C_i) fp32;
 *            individual entries may be NULL.
 * H and W must be divisible by 2^(n_mults-1). */
int tedm_unet_forward(tedm_unet* h, const float* x, const float* t, int B, int H, int W,
                      float* eps_out, float* const* feat_out,
                      void* workspace, size_t workspace_bytes, void* stream, int flags);

/* After a forward with TEDM_FWD_TRACE: named module outputs living in the workspace (NHWC). */
int         tedm_unet_trace_count(const tedm_unet* h);
const char* tedm_unet_trace_name(const tedm_unet* h, int i);
int         tedm_unet_trace_info(const tedm_unet* h, int i, size_t* byte_offset, int* shape4 /* B,H,W,C */);

/* DiffusionModel.forward_diffusion_model (models/diffusion_model.py:176-203) with the optional
 * x*2-1 of DiffusionModel.forward (:169-170) folded in:
 *   x_t[b] = sqrt_acp[t[b]] * (normalize ? 2*x0[b]-1 : x0[b]) + sqrt_1macp[t[b]] * noise[b]
 * x0, noise, x_t: (B, chw) fp32; t: (B,) int64; coefficient tables: (T,) fp32. */
int tedm_q_sample(const float* x0, const float* noise, const int64_t* t,
                  const float* sqrt_acp, const float* sqrt_1macp, int T,
                  float* x_t, int B, long long chw, int normalize, void* stream);

/* DatasetDM.extract_features' resize+concat (models/datasetDM_model.py:80-83): nearest-upsample
 * one NHWC tap (B,h,w,C) to (S,S) and write it into channels [c_off, c_off+C) of the NCHW
 * feature tensor (B, C_total, S, S) fp32. */
int tedm_tap_to_features(const float* tap, int B, int h, int w, int C,
                         float* feats, int C_total, int c_off, int S, void* stream);

/* ---- training path -----------------------------------------------
 * Replaces `loss.backward()` over Unet.forward in trainers/train_base_diffusion.py:36-47 (autograd through
 * models/unet_model.py:333-368).  Protocol per optimizer step:
 *   tedm_unet_pack + tedm_unet_pack_backward   (weights changed)       ->
 *   tedm_unet_forward_train  (records a tape inside `workspace`)       ->
 *   tedm_l1_loss             (loss value and d loss / d eps)           ->
 *   tedm_unet_backward       (writes every bound gradient; no accumulation)
 * Gradients use the reference parameter layouts (conv OIHW, Linear (out,in), vectors) so that a torch
 * optimizer (or an RCCL all-reduce over a flat buffer) consumes them directly. */

/* Bind the gradient buffer of one parameter (same keys / sizes as tedm_unet_set_param; every key except
 * "time_mlp.0.freqs" must be bound before tedm_unet_backward). */
int tedm_unet_set_grad(tedm_unet* h, const char* key, void* dev_ptr, long long numel);
/* (Re)encode the data-gradient weight packs from the bound parameters. */
int tedm_unet_pack_backward(tedm_unet* h, void* stream);
/* Workspace bytes for one forward_train + backward at this shape (tape + backward scratch). */
size_t tedm_unet_train_workspace_bytes(const tedm_unet* h, int B, int H, int W);
/* Forward that keeps every tensor the backward reads in `workspace`.  flags: precision bits only. */
int tedm_unet_forward_train(tedm_unet* h, const float* x, const float* t, int B, int H, int W, float* eps_out,
                            void* workspace, size_t workspace_bytes, void* stream, int flags);
/* Backward of the recorded forward.  d_eps: (B, out_dim, H, W) fp32 = d loss / d eps.  Must be given the
 * same workspace; consumes the tape (one backward per forward_train). */
int tedm_unet_backward(tedm_unet* h, const float* d_eps, void* workspace, size_t workspace_bytes, void* stream);
/* The handle keeps ONE tape.  Every tedm_unet_forward_train bumps a generation counter (tedm_unet_tape_generation returns
 * the generation of the tape now held); tedm_unet_backward_gen fails with TEDM_ESTATE ("tape overwritten by a later
 * forward") when `generation` is not the tape's -- use it whenever two forwards can be outstanding (autograd). */
long long tedm_unet_tape_generation(const tedm_unet* h);
/* Where the training forward left the output of ups[level][2] (the tensor models/datasetDM_model.py:50-53 hooks), inside the
 * training workspace: byte offset and NHWC shape (B, H, W, C).  Valid until the tape is consumed or overwritten. */
int tedm_unet_tape_tap(const tedm_unet* h, int level, size_t* offset_bytes, int shape[4]);
int tedm_unet_backward_gen(tedm_unet* h, const float* d_eps, void* workspace, size_t workspace_bytes, void* stream,
                           long long generation);
/* Data-parallel training (not in the reference, which has no distributed code): `fn(user, group)` is called from
 * inside tedm_unet_backward, on the calling thread, each time every gradient of one top-level module has been ENQUEUED on the
 * stream -- group = its state-dict prefix, in the order final_conv, final_res_block, ups.{n-1..0}, mid_block2, mid_attn,
 * mid_block1, downs.{n-1..0}, init_conv, time_mlp (the flat gradient buffer completes from its end towards its start).  The host
 * records an event there and launches the all-reduce of the finished slice on a side stream, so that communication hides under
 * the rest of the backward (tedm_amd/dp.py).  fn = NULL removes the hook.  The callback must not call back into the handle. */
typedef void (*tedm_backward_mark_fn)(void* user, const char* group);
int tedm_unet_set_backward_callback(tedm_unet* h, tedm_backward_mark_fn fn, void* user);
/* DiffusionModel.train_step's loss (models/diffusion_model.py:138-143):
 *   loss = mean_b( mean_chw |eps - target| * p2w[t_b] ),  grad = d loss / d eps  (grad may be NULL).
 * scratch: B*64 floats; loss: 1 float; all device pointers. */
int tedm_l1_loss(const float* eps, const float* target, const int64_t* t, const float* p2w, int T, int B,
                 long long chw, float* grad, float* scratch, float* loss, void* stream);

/* One ancestral sampling step behind the UNet forward (models/diffusion_model.py:205-267: model_predictions -> p_mean_variance with
 * the per-sample 99.5 % dynamic thresholding -> q_posterior -> sample_timestep) as one kernel, one workgroup per sample.
 *   model_out : what the UNet returned, (B, chw); objective 0: it is eps ('pred_noise'), 1: it is x0 ('pred_x_start')
 *   noise     : NULL (no noise term: p_mean_variance alone) or (B, chw); ignored for samples with t = 0
 *   clip      : `clip_denoised`; k_lo / w: floor and fractional part of q * (chw - 1) evaluated in fp32 as torch.quantile does
 *   outputs   : x0 (B, chw) clipped and normalised, mean (may be NULL), x_prev (may be NULL), s_out [B] thresholds (may be NULL)
 * The five coefficient tables are the module's buffers (sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod,
 * posterior_mean_coef1/2, posterior_log_variance_clipped), T entries each. */
int tedm_sampler_step(const float* x_t, const float* model_out, const int64_t* t, const float* noise, const float* sqrt_recip_acp,
                      const float* sqrt_recipm1_acp, const float* coef1, const float* coef2, const float* logvar, int T, int B,
                      long long chw, int objective, int clip, long long k_lo, float w, float* x0, float* mean, float* x_prev,
                      float* s_out, void* stream);

/* ---- stand-alone sub-module forward --------------------------------------------------------------------------------------
 * The reference's Unet subclasses (models/global_local_cl.py:33-47,75-104) call `block(x, t)` and `attn(x)` on the modules of
 * `self.downs / mid_* / self.ups` directly.  `module` is the state-dict prefix of a ResnetBlock ("downs.0.0", "mid_block1",
 * "ups.1.1", "final_res_block") or of a Residual(PreNorm(attention)) block ("downs.0.2", "mid_attn", "ups.3.2").
 * x1 (| x2): NHWC inputs (x2: second half of a virtual channel concat, or NULL); st_silu: SiLU(time embedding) (B, 4*dim) or
 * NULL = no FiLM (models/unet_model.py:168); out: NHWC (B,H,W,Cout).  Inference only. */
int tedm_unet_module_channels(const tedm_unet* h, const char* module, int* cin, int* cout);
size_t tedm_unet_module_workspace_bytes(const tedm_unet* h, const char* module, int B, int H, int W);
int tedm_unet_module_forward(tedm_unet* h, const char* module, const float* x1, int C1, const float* x2, int C2,
                             const float* st_silu, int B, int H, int W, float* out, void* workspace,
                             size_t workspace_bytes, void* stream, int flags);

/* ---- pixel-classifier head (models/datasetDM_model.py:57-64,85-88) -------------------
 * The head's first layer is a 1x1 conv over the (B, 960*T, S, S) concat of nearest-upsampled decoder taps.  A 1x1 conv
 * commutes with nearest upsampling and distributes over the channel concat:
 *     W1 * cat_i(up(f_i)) = sum_i up(W1_i * f_i)
 * so the projection runs at every tap's NATIVE resolution on the NHWC tap buffers the UNet forward already produced and
 * the 960*T-channel tensor is never materialised (40 GB -> 1 GB at B=32, T=5, S=256).  These are the building blocks;
 * tedm_amd/head.py wires them (and their adjoints) behind DatasetDM.forward. */
size_t tedm_conv1x1_packed_bytes(int Cout, int Cin);                         /* 0 unless Cout%16==0 and Cin%32==0 */
int tedm_conv1x1_pack(const float* w_out_in, int Cout, int Cin, void* wpk, int prec, void* stream);
/* out (B,H,W,Cout) = x (B,H,W,Cin) * W^T [+ bias] [+ add]; `add` may alias `out` (accumulation over taps/timesteps). */
int tedm_conv1x1_nhwc(const float* x, int B, int H, int W, int Cin, const void* wpk, int Cout, const float* bias,
                      const float* add, float* out, int prec, void* stream);
/* dw (Cout,Cin) (+)= sum_pixels dy^T x  (fp32 MFMA, deterministic split reduction). */
size_t tedm_conv1x1_wgrad_workspace_bytes(int Cout, int Cin, int B, int H, int W);
int tedm_conv1x1_wgrad_nhwc(const float* dy, const float* x, int B, int H, int W, int Cin, int Cout, float* dw,
                            int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* out (B,S,S,C) = bias + sum_i nearest_upsample(z_i (B,h_i,w_i,C));  n <= 8 inputs. */
int tedm_upsample_sum_nhwc(const float* const* z, const int* h, const int* w, int n, int B, int S, int C,
                           const float* bias, float* out, void* stream);
/* Inference-mode tail of the classifier in one pass (models/datasetDM_model.py:59-64 with eval-mode BatchNorm folded into the
 * following conv by the caller): logits (B,S,S) = b3 + w3 . relu(b2 + W2 relu(b1 + sum_i nearest_upsample(z_i))), W2 given
 * TRANSPOSED as w2t [C1][32] (C1 % 32 == 0, second width fixed at the reference's 32).  Neither the C1-channel nor the
 * 32-channel tensor reaches HBM.  All pointers device; b3: 1 float. */
int tedm_head_tail_nhwc(const float* const* z, const int* h, const int* w, int n, int B, int S, int C1, const float* b1,
                        const float* w2t, const float* b2, const float* w3, const float* b3, float* logits, void* stream);
/* TRAINING-mode tail of the classifier behind its first conv (models/datasetDM_model.py:59-64: ReLU, BatchNorm2d(C1), Conv2d(C1,32,1),
 * ReLU, BatchNorm2d(32), Conv2d(32,1,1)) on h1 (P pixels x C1, NHWC) in three forward and three backward passes; only z2 (P x 32) is
 * kept, everything else is recomputed (tedm_amd/csrc/head_train.hip; tedm_amd/head.py derives the coefficient vectors between the
 * passes).  `blocks` = number of per-block partial results the caller allocated = workgroups launched; partials are summed by the
 * caller in a fixed order (deterministic).  a / b: BatchNorm as y = a x + b; k [3][32], m [3][C1]: coefficients of
 * dz2 = [z2 > 0] (k0 dl + k1 + k2 z2) and dh1 = [h1 > 0] (m0 du + m1 + m2 h1), du = W2^T dz2. */
 /* shift / shift2 (per channel, may be NULL = 0): the (sum, sumsq) partials are taken of (x - shift) -- a shift near the channel mean
 * keeps the batch variance exact in fp32 for channels with |mean| >> std. */
int tedm_head_train_stats1(const float* h1, long long P, int C1, const float* shift /*[C1]*/, float* part /*[blocks][C1][2]*/, int blocks, void* stream);
int tedm_head_train_mid(const float* h1, long long P, int C1, const float* a1, const float* b1, const float* w2t /*[C1][32]*/,
                        const float* bias2, const float* shift2 /*[32]*/, float* z2 /*[P][32]*/, float* part2 /*[blocks][32][2]*/, int blocks, void* stream);
int tedm_head_train_out(const float* z2, long long P, const float* a2, const float* b2, const float* w3, const float* b3,
                        float* logits /*[P]*/, void* stream);
int tedm_head_train_bwd1(const float* z2, const float* dl, long long P, float* part /*[blocks][33]: A1[32], A0*/, int blocks, void* stream);
int tedm_head_train_bwd2(const float* h1, const float* z2, const float* dl, long long P, int C1 /*128 | 256*/, const float* k,
                         float* part /*[blocks][32*C1 + 32]: G[32][C1], db2[32]*/, int blocks, void* stream);
int tedm_head_train_bwd3(const float* h1, const float* z2, const float* dl, long long P, int C1, const float* k, const float* w2 /*[32][C1]*/,
                         const float* m, float* dh1, void* stream);
/* adjoint of one nearest upsample by `factor`: out (B,h,w,C) = block sums of g (B,h*factor,w*factor,C). */
int tedm_pool_sum_nhwc(const float* g, int B, int h, int w, int factor, int C, float* out, void* stream);

const char* tedm_last_error(void);
const char* tedm_version(void);
/* sha256 (hex) of the sources this binary was compiled from (tedm_amd/build.py: every file under tedm_amd/csrc and include/);
 * the Python loader refuses a library whose digest differs from the tree it sits in. */
const char* tedm_source_digest(void);

#ifdef __cplusplus
}
#endif
#endif /* TEDM_HIP_H */
