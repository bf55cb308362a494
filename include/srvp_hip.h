/*
 * srvp_hip.h -- C ABI of libsrvp_hip.so, the MI355X (gfx950 / CDNA4) kernel library behind the SRVP
 * training / rollout hot path.
 *
 * The reference (edouardelasalles/srvp) has no FFI layer: its hot path is PyTorch modules calling cuDNN/cuBLAS
 * implicitly.  Each entry point below replaces the implicit device op(s) behind the cited reference lines
 * (paths relative to the reference root).  Conventions:
 *   - extern "C", plain pointers and sizes, no torch types; every call returns 0 on success or a non-zero
 *     code with a message available from srvp_last_error(); no exception crosses the ABI.
 *   - the caller owns every buffer (device pointers); the library allocates nothing.
 *   - every launch is asynchronous on the hipStream_t passed as `stream` (void*), no host sync inside.
 *   - activations are NHWC bfloat16 with an optional 1-pixel zero border ("padded" tensors);
 *     channel counts are padded to a multiple of 32 by the caller (zero weights / zero gamma,beta).
 */
#ifndef SRVP_HIP_H
#define SRVP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRVP_MAX_TAPS 16

#define SRVP_ACT_NONE 0
#define SRVP_ACT_LRELU 1    /* LeakyReLU(0.2), module/utils.py:41 */
#define SRVP_ACT_TANH 2
#define SRVP_ACT_RELU 3
#define SRVP_ACT_SIGMOID 4

int srvp_version(void);
/* a non-blocking HIP stream of the lowest priority of the device (the product's second stream: weight gradients / packing, work off the
 * critical path); *priority_out (optional) receives that priority.  The caller owns the stream (hipStreamDestroy). */
int srvp_stream_create_low_priority(void** stream_out, int* priority_out);
const char* srvp_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Tap-table implicit-GEMM convolution on MFMA (bf16 operands, fp32 accumulate).
 * Replaces nn.Conv2d / nn.ConvTranspose2d forward (module/conv.py:174-179,200-223,299-304,330-353) and, with
 * transposed packed weights and the gradient as input, their data-gradient.
 *   out[n, oy*so+ooy, ox*so+oox, j] = sum_t sum_c in_t[n, oy, ox, c] * W[t][j][c]
 *   in_t[n, oy, ox, c] = src{0|1}[n', (oy*si+dy[t]) (>>1 after +1 if ups), (ox*si+dx[t]) (...), c]
 * src0 supplies channels [0,C0), src1 (optional: the skip connection of conv.py:270, selected per sample through
 * map1[n], module/srvp.py:185-190,222-223) channels [C0, C0+C1).  dy/dx are in padded coordinates of the source.
 * Optional epilogue: per-column sum / sum-of-squares (fp64 atomics) for BatchNorm batch statistics
 * (conv.py:104); column j contributes to stats[j % stat_mod] and stats[stat_mod + j % stat_mod].
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const void* src0; const void* src1; const int32_t* map1;
    int32_t C0, C1;
    int32_t H0p, W0p, H1p, W1p;   /* physical (padded) spatial dims of each source */
    int32_t ups0, ups1;           /* nearest x2 upsample folded into the gather (conv.py:331-349) */
    int32_t si;                   /* input stride */
    int32_t ntaps; int32_t dy[SRVP_MAX_TAPS]; int32_t dx[SRVP_MAX_TAPS];
    const void* wt;               /* bf16 [ntaps][Cout][C0+C1] */
    int32_t Cout;
    int32_t N, OH, OW;            /* output (sub-)grid */
    void* dst; int32_t DHp, DWp, so, ooy, oox, Cdst, cdst_off;
    double* stats; int32_t stat_mod;
    /* image-side output layer (conv.py:304,353 + sigmoid :273-274): when out_f32 != NULL the first out_nc columns are
     * written as fp32 into the (N, out_nc, DHp, DWp) frame tensor (optionally through a sigmoid) instead of `dst` */
    float* out_f32; int32_t out_nc, out_sigmoid;
    /* Hoisted skip half (the skip connection is identical for every time step, module/srvp.py:222-223):
     *   map0      : image indirection for src0 (n -> map0[n]), NULL = identity
     *   dst_is_f32: write the fp32 accumulators to `dst` as fp32 [N][DHp][DWp][Cdst] instead of bf16
     *   add_f32   : fp32 [add_mod][OH][OW][Cout] added to the accumulators (image n uses row n % add_mod) before
     *               the statistics / store -- conv([h, skip]) = conv_h(h) + conv_s(skip), conv_s computed once per sample */
    const int32_t* map0; int32_t dst_is_f32; const float* add_f32; int32_t add_mod;
    int32_t wt_fragmajor;         /* wt is packed MFMA-fragment-major (srvp_pack_desc.layout 1): required by, and only valid
                                   * for, launches that srvp_conv_wants_fragmajor() accepts */
    int32_t tap_phase_chunks;     /* > 0 (halo kernel only): src0 is a SPACE-TO-DEPTH tensor whose 64-channel chunk cc belongs to
                                   * phase cc / tap_phase_chunks (0..3) and is convolved with THAT phase's taps only: dy/dx hold
                                   * 4 x ntaps entries, [phase * ntaps + t].  The data-gradient of a sub-pixel upsample convolution
                                   * (conv.py:331-349) over the output gradient stored [N][H/2+2][W/2+2][4C] is then one launch with
                                   * K = 4 C ntaps exactly.  wt: [ntaps][Cout][C0] as usual (chunk -> phase implied) */
    int32_t elem_f32;             /* 1: precision = 'fp32' parity mode -- src0 / src1 / dst / wt are FP32 tensors of the same NHWC
                                   * shapes (wt tap-major, srvp_pack_desc.dst_f32) and the contraction runs in exact fp32 on the
                                   * matrix cores (v_mfma_f32_32x32x2_f32 = a k-ordered fmaf chain), csrc/conv_f32.hip */
    int32_t splitk;               /* > 1 (generic kernel; dst_is_f32 = 1, no stats / add_f32): the K steps are shared out over `splitk`
                                   * workgroups per output tile, split z writing its partial sums to the fp32 slab
                                   * dst + z * N*DHp*DWp*Cdst -- for the tiny-M, long-K launches (the 4x4 -> 1x1 encoder output
                                   * layer, conv.py:176-179 / 224, and the data gradient of the 1x1 -> 4x4 decoder input layer,
                                   * conv.py:299-301): K = 8192 is 128 dependent K steps on a handful of workgroups otherwise.
                                   * srvp_splitk_finish sums the slabs in a fixed order (deterministic) */
    int32_t f32_quad;             /* q > 0: the fp32 tensor of the hoisted skip half -- `dst` of the dst_is_f32 launch that writes it,
                                   * `add_f32` of the launches that read it -- is stored pixel-quad-major for a consumer of output stride q:
                                   * element (n, Y, X, c) at ((((n DHp + Y) q + X % q) (DWp / q / 4) + (X / q) / 4) C + c) 4 + (X / q) % 4,
                                   * so that four consecutive output columns of a consumer lane are one 16-byte load.  Consumers need
                                   * so == q and OW % 4 == 0; same values, another layout */
    /* BatchNorm-backward reduction of the PRODUCER layer fused into a data-gradient launch (halo kernel only; conv.py:103-104 backward):
     * this launch writes dA = gradient wrt the activated output of a conv -> BN -> LeakyReLU block whose raw (pre-BN) output is
     * bnr_raw [N][OH][OW][Cout] (same shape as dst).  With bnr_red != NULL the epilogue also accumulates, per channel,
     *   bnr_red[0][c] += sum g,  bnr_red[1][c] += sum g * (raw - mean) * invstd,   g = bf16(dA) * f'(scale * raw + shift)
     * i.e. exactly what srvp_bn_bwd_reduce (da_mode 0) computes from the stored dA and raw -- one read of dA and one launch less
     * per layer.  bnr_coef: fp32 [4][Cout] = scale, shift, mean, invstd of that layer. */
    const void* bnr_raw; const float* bnr_coef; double* bnr_red;
    /* Inference (eval-mode BatchNorm, conv.py:103-104 with running statistics): ep_coef != NULL folds the block's normalisation and
     * activation into this launch's epilogue -- dst receives  act( ep_coef[c] * acc + ep_coef[Cout + c] )  (scale, shift of output
     * channel c; ep_act = ACT_* id) computed from the fp32 accumulators (after add_f32) and rounded to bf16 once, written where the
     * consumer reads it: dst is an activation tensor [N][DHp + 2 ep_border][DWp + 2 ep_border][Cdst] whose interior the launch
     * addresses through DHp / DWp / so / ooy / oox as usual (add_f32 keeps the unbordered geometry).  No raw tensor, no srvp_bn_act
     * pass.  bf16 launches without stats / dst_is_f32 / out_f32 / splitk. */
    const float* ep_coef; int32_t ep_act, ep_border;
} srvp_conv_desc;
int srvp_conv_mfma(const srvp_conv_desc* d, void* stream);
/* Image-side OUTPUT layer of the VGG decoder (conv.py:353 + 273-274: ConvTranspose2d(64 -> nc <= 3, 3x3, stride 1, pad 1) + sigmoid) as a
 * streaming kernel (csrc/conv_out.hip): act = bf16 [N][66][66][64] (1-pixel zero border), wt_tapmajor = bf16 [9][32][64] (srvp_pack_desc
 * layout 0 of the block's forward weights, Cout padded to 32), out = fp32 (N, Cout_real, 64, 64).  One persistent workgroup per CU walks
 * whole images with a rolling LDS window of input rows: every activation byte crosses HBM -> LDS once.  srvp_conv_out_eligible: 1 if
 * the geometry is the one this kernel covers (SRVP_CONV_OUT_STREAM=0 switches it off: the layer then runs through srvp_conv_mfma). */
int srvp_conv_out_eligible(int C0, int H, int W, int Cout_real, int k, int s, int p);
int srvp_conv_out_fwd(const void* act, const void* wt_tapmajor, float* out, int N, int Cout_real, int sigmoid, void* stream);
/* Image-side OUTPUT layer of the DCGAN decoder (conv.py:304-305: ConvTranspose2d(64 -> nc <= 3, 4x4, stride 2, pad 1) + sigmoid, 32x32 -> 64x64)
 * as a streaming kernel of the same kind (csrc/conv_out.hip): act = bf16 [N][34][34][64] (1-pixel zero border), w_f32 = the layer's fp32 master
 * weight (Cin = 64, nc, 4, 4) -- rounded to bf16 by the kernel itself (round to nearest even, as srvp_pack_weight does), out = fp32
 * (N, Cout_real, 64, 64).  Sub-pixel form: the 4 nc (phase, channel) pairs are the N dimension of the 16x16x32 MFMA, K = 64 channels x the
 * 3x3 window of low-resolution pixels.  srvp_conv_up_out_eligible: 1 for the geometry covered (SRVP_CONV_OUT_STREAM=0: off). */
int srvp_conv_up_out_eligible(int C0, int Hin, int Win, int Cout_real, int k, int s, int p);
int srvp_conv_up_out_fwd(const void* act, const float* w_f32, float* out, int N, int Cout_real, int sigmoid, void* stream);
/* 1 (default): 3x3 stride-1 single-source convolutions run on the halo-tiled kernel (input patch staged in LDS once
 * per channel chunk, taps = LDS offsets); 0: every convolution on the generic tap-gather kernel.  Same results, bit for bit. */
int srvp_conv_set_halo(int on);
/* 1 (default): 3x3 stride-1 64 -> 64 channel launches on 64x64 images (N >= 96, plain bf16 destination, optional stats / bnr_*) run on the
 * streaming kernel (csrc/conv_stream.hip: persistent workgroup per CU, rolling LDS row window, register-resident weights); 0: on the
 * tile kernels.  Same results up to fp32 summation order. */
int srvp_conv_set_stream64(int on);
/* Launches taken by the streaming kernels of csrc/conv_stream.hip since the library was loaded (host-side counters, for tests that must
 * know a case really ran on them): which = 0: 64 -> 64 channel forward / plain data gradient, 1: data gradient with fused BatchNorm-backward
 * sums (bnr_*), 2: the 64-channel sub-pixel stage entry (srvp_conv_mfma_multi).  -1 for another `which`. */
long long srvp_conv_stream_count(int which);
/* 1 (default; env SRVP_CONV_IN_STREAM): srvp_conv_in_fwd / srvp_conv_in_fwd_bnr serve 3x3 stride-1 layers with 64 output channels on 64x64
 * frames through the streaming kernel of csrc/conv_in_stream.hip; 0: through the 128-pixel tile kernel (exact fp32 MFMA).  A/B switch of the tests. */
int srvp_conv_set_in_stream(int on);
/* 1 if this descriptor will run on the halo-tiled kernel, which wants its weights fragment-major (pack layout 1) */
int srvp_conv_wants_fragmajor(const srvp_conv_desc* d);
/* 0, or the pixel-tile size (>= 256: eligible for bnr_red) of the halo-tiled kernel variant this descriptor will run on */
int srvp_conv_runs_on_halo(const srvp_conv_desc* d);
/* d[0..n-1]: as n calls of srvp_conv_mfma; launches that run on the same halo-kernel variant with the same grid (the four
 * output phases of a sub-pixel upsample convolution) are issued as ONE grid */
int srvp_conv_mfma_multi(const srvp_conv_desc* d, int n, void* stream);

/* Weight gradient of the same tap-table convolution (autograd of the modules above):
 *   dW[t][j][c] += sum_{n,oy,ox} dout[n, oy*so+ooy[t], ox*so+oox[t], j] * in_t[n, oy, ox, c]     (fp32 atomics)
 */
typedef struct {
    const void* src0; const void* src1; const int32_t* map1;
    int32_t C0, C1;
    int32_t H0p, W0p, H1p, W1p;
    int32_t ups0, ups1;
    int32_t si;
    int32_t ntaps; int32_t dy[SRVP_MAX_TAPS]; int32_t dx[SRVP_MAX_TAPS];
    const void* dout; int32_t DHp, DWp, so; int32_t ooy[SRVP_MAX_TAPS]; int32_t oox[SRVP_MAX_TAPS];
    int32_t Cout;                 /* channels of dout (its physical channel count) */
    int32_t N, OH, OW;
    float* dw;                    /* fp32 [ntaps][Cout][C0+C1], accumulated into */
    int32_t splitk;
    const int32_t* map0;          /* image indirection for src0 (NULL = identity) */
    int32_t elem_f32;             /* 1: src0 / src1 / dout are fp32 tensors (fp32 parity mode, exact-fp32 MFMA) */
    int32_t dout_cstride, dout_coff; /* dout is a channel slice of a wider tensor: pixel stride dout_cstride channels (0 = Cout),
                                   * first channel dout_coff (one phase of a space-to-depth gradient) */
    int32_t dout_phase_taps;      /* > 0 (per-tap kernel): tap t reads the channel slice of phase t / dout_phase_taps, i.e. first
                                   * channel dout_coff + (t / dout_phase_taps) * Cout -- all 16 (phase, tap) weight gradients of a
                                   * sub-pixel block in ONE launch over its space-to-depth output gradient */
} srvp_wgrad_desc;
int srvp_wgrad_mfma(const srvp_wgrad_desc* d, void* stream);
/* 1: fragments through ds_read_b64_tr_b16 (default), 0: 16-bit LDS reads (conservative fallback) */
int srvp_wgrad_set_tr(int on);
/* 1 (default): 3x3 stride-1 single-source weight gradients with 64-multiple channel counts run on the halo-tiled kernel
 * (all 9 taps per workgroup, operands staged once); 0: per-tap kernel for everything */
int srvp_wgrad_set_halo(int on);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (training / eval) + activation, split around the grid-wide reduction (conv.py:103-106).
 * ------------------------------------------------------------------------------------------------ */
/* stats(double [2][C]) -> scale, shift, mean, invstd (fp32 [C]); updates running stats (momentum 0.1, unbiased
 * variance) and num_batches_tracked when running_mean != NULL.  C_real <= C: padded channels get scale=shift=0. */
int srvp_bn_finalize(const double* stats, double count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float* scale, float* shift, float* mean, float* invstd,
                     int C, int C_real, float eps, float momentum, void* stream);
/* eval mode: scale/shift from running statistics */
int srvp_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float* scale, float* shift, int C, int C_real, float eps, void* stream);
/* act = f(scale*raw + shift) -> bf16 padded tensor (+ optional 2x2 max-pooled copy, conv.py:204-222;
 * + optional fp32 unpadded copy used for the encoder output hx). */
int srvp_bn_act(const void* raw, const float* scale, const float* shift, int act,
                int N, int H, int W, int C,
                void* dst, int dst_border, void* dst_pool, int pool_border, float* dst_f32, void* stream);
/* the same; keep_frames (int32 [N], may be NULL) applies to launches with dst_pool: frames n with keep_frames[n] == 0 get only
 * their pooled output -- the full-resolution activation of a pooled layer is read by nothing but the skip connections */
int srvp_bn_act_keep(const void* raw, const float* scale, const float* shift, int act,
                     int N, int H, int W, int C,
                     void* dst, int dst_border, void* dst_pool, int pool_border, float* dst_f32, const int32_t* keep_frames,
                     void* stream);

/* fp32 parity mode: raw / dst / dst_pool are fp32 tensors of the same shapes */
int srvp_bn_act_keep_f32(const void* raw, const float* scale, const float* shift, int act,
                         int N, int H, int W, int C,
                         void* dst, int dst_border, void* dst_pool, int pool_border, float* dst_f32, const int32_t* keep_frames,
                         void* stream);

/* srvp_bn_finalize + srvp_bn_act_keep(_f32) as ONE launch (replaces the two-kernel sequence placed by conv.py:103-106 in training
 * mode): every workgroup derives the coefficients from `stats` / `count` once into LDS (same fp64 expressions as
 * srvp_bn_finalize: bit-identical values), workgroup 0 stores scale / shift / mean / invstd and updates the running statistics. */
int srvp_bn_finalize_act(const void* raw, const double* stats, double count, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale, float* shift,
                         float* mean, float* invstd, int C_real, float eps, float momentum, int act, int N, int H, int W, int C,
                         void* dst, int dst_border, void* dst_pool, int pool_border, float* dst_f32, const int32_t* keep_frames,
                         void* raw_pool, int elem_f32, int dst_s2d, void* stream);
/* raw_pool (with dst_pool; may be NULL): [N][H/2][W/2][C], unbordered, element type of raw -- the RAW value at the position each pooled
 * activation was taken from (first maximum in scan order, torch's tie rule).  It is what lets the BatchNorm-backward sums of a pooled
 * layer ride the consumer's data-gradient launch: that launch writes the pooled gradient dA_p, only the arg-max position of a window
 * receives it, so  sum g = sum dA_p * f'(scale * raw_pool + shift)  and likewise the second sum -- srvp_conv_desc.bnr_raw = raw_pool
 * -- instead of a pass that re-derives the arg-max from four raw pixels per window (srvp_bn_bwd_reduce da_mode 2).  The skip-connection
 * gradient of such a layer (da2) is added by srvp_bn_bwd_reduce with da_mode 3. */
/* dst_s2d = 1: `dst` is written SPACE-TO-DEPTH, [N][H/2+2][W/2+2][4C] with a 1-pixel zero border -- pixel (y, x) at position (y/2, x/2),
 * channel group (y&1)*2 + (x&1) -- the layout a 4x4 stride-2 consumer (DCGAN encoder, conv.py:174-179) convolves as a 2x2-tap-per-phase
 * stride-1 halo convolution (srvp_conv_desc.tap_phase_chunks); no pooling then.  srvp_bn_act_s2d: the same store for the two-launch /
 * eval-mode path (bf16). */
int srvp_bn_act_s2d(const void* raw, const float* scale, const float* shift, int act, int N, int H, int W, int C, void* dst, void* stream);

/* backward of activation+BN.  dA comes from one or two places:
 *   main: tensor `da` with channel stride da_cstride / offset da_coff, mode 0 = same resolution,
 *         mode 1 = gradient of the nearest-x2-upsampled tensor (sum the 2x2 block),
 *         mode 2 = gradient of the 2x2-max-pooled tensor (routed to the arg-max pixel, first-max tie rule);
 *   extra: optional second same-resolution gradient `da2` (skip-connection gradient), bf16 [N][H][W][C].
 * pass 1 accumulates red[0][c] = sum g, red[1][c] = sum g*xhat (fp64 atomics), g = dA * f'(scale*raw+shift). */
typedef struct {
    const void* raw; const void* act; int32_t act_border;   /* act: padded activated tensor (mode 2 only) */
    const float* scale; const float* shift; const float* mean; const float* invstd; int32_t act_kind;
    const void* da; int32_t da_mode, da_cstride, da_coff, da_border; int32_t da_is_f32;   /* da_mode 3 (srvp_bn_bwd_reduce only): the pooled consumer's
                                                             * term is accumulated elsewhere (raw_pool above): only the da2 term is summed -- N = the number of da2 rows,
                                                             * da2_idx[j] = the FRAME row j belongs to (the inverse map of modes 0-2), `da` unused */
    const void* da2; const int32_t* da2_idx;                /* da2: compact [B][H][W][C]; da2_idx[n] = row or -1 */
    int32_t N, H, W, C;
    void* tsum; int32_t tsum_T;                             /* apply only: bf16 [N/T][H+2b][W+2b][C] = sum over the T time steps (frames
                                                             * ordered t*B + b) of the written gradient, or NULL */
    int32_t elem_f32;                                       /* 1: raw / act / da / da2 / tsum / draw are fp32 tensors (fp32 parity mode) */
    int32_t draw_s2d;                                       /* apply only: draw is written SPACE-TO-DEPTH, [N][H/2+2][W/2+2][4C] with a
                                                             * 1-pixel border: pixel (y, x) -> position (y/2, x/2), channel group
                                                             * (y&1)*2 + (x&1) (consumed by tap_phase_chunks launches); da_mode 0 / 1 */
} srvp_bnbwd_desc;
int srvp_bn_bwd_reduce(const srvp_bnbwd_desc* d, double* red, void* stream);
/* red -> dgamma, dbeta (accumulated into fp32 grads when non-NULL, times param_grad_scale) and the per-channel coefficients
 * used by apply.  Data parallel (SyncBatchNorm, train.py:283): `red` and `count` are the all-reduced GLOBAL sums, and
 * param_grad_scale = 1 / world -- every rank then holds the global parameter gradient divided by world, so that the DDP
 * average over ranks of the world-times-too-large per-rank gradients (loss / LOCAL batch, train.py:106) is the single-process
 * gradient, exactly like torch's SyncBatchNorm, which forms dgamma / dbeta from the local sums. */
int srvp_bn_bwd_finalize(const double* red, double count, const float* scale, const float* mean, const float* invstd,
                         float* dgamma, float* dbeta, float* coef, int C, int C_real, int has_bn, float param_grad_scale,
                         void* stream);
/* pass 2: draw = scale*(g - mean_g - xhat*mean_gx) -> bf16 tensor with border `dst_border` */
int srvp_bn_bwd_apply(const srvp_bnbwd_desc* d, const float* coef, void* draw, int dst_border, void* stream);
/* srvp_bn_bwd_finalize (has_bn = 1) + srvp_bn_bwd_apply as ONE launch: coefficients derived per workgroup into LDS, workgroup 0
 * adds dgamma / dbeta (times param_grad_scale) and stores `coef` (may be NULL); d->mean / d->invstd required */
int srvp_bn_bwd_finalize_apply(const srvp_bnbwd_desc* d, const double* red, double count, float* dgamma, float* dbeta, float* coef,
                               int C_real, float param_grad_scale, void* draw, int dst_border, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small-channel layers (first encoder conv, last decoder conv; conv.py:174,200,304,353), direct fp32 VALU
 * kernels on the reference's own (T*B, C, 64, 64) fp32 frame layout -- these layers are HBM-bound.
 * ------------------------------------------------------------------------------------------------ */
/* x fp32 NCHW [N][Cin<=4][H][W] -> raw bf16 NHWC [N][OH][OW][Cout]; w fp32 [Cout_real][Cin][k][k] (OIHW) */
int srvp_conv_in_fwd(const float* x, const float* w, void* raw, double* stats,
                     int N, int Cin, int H, int W, int Cout, int Cout_real, int k, int s, int p, void* stream);
/* dW[Cout_real][Cin][k][k] += sum draw * x   (draw: bf16 padded(border 1) [N][OH+2][OW+2][Cout]) */
/* srvp_conv_in_fwd used as the DATA GRADIENT of the image-side output layer (conv.py:353 backward: x = gradient frames (N, nc, 64, 64),
 * w = the ConvTranspose weight (Cin_layer, nc, 3, 3) read as (O, I, k, k), raw = dA of the producer block [N][64][64][Cout]) with the
 * producer's BatchNorm-backward sums accumulated in the same launch: bnr_raw / bnr_coef / bnr_red as in srvp_conv_desc.bnr_*. */
int srvp_conv_in_fwd_bnr(const float* x, const float* w, void* raw, int N, int Cin, int H, int W, int Cout, int Cout_real, int k, int s,
                         int p, const void* bnr_raw, const float* bnr_coef, double* bnr_red, void* stream);
/* 1 if srvp_conv_in_fwd_bnr serves this shape (3x3 stride 1 on 64x64 frames, Cout 32 / 64, MFMA image-side kernels enabled) */
int srvp_conv_in_fwd_bnr_ok(int Cin, int H, int W, int Cout, int k, int s, int p);
int srvp_conv_in_wgrad(const float* x, const void* draw, float* dw,
                       int N, int Cin, int H, int W, int Cout, int Cout_real, int k, int s, int p, void* stream);
/* srvp_bn_bwd_finalize_apply + srvp_conv_in_wgrad of the FIRST block (conv.py:200) as one launch: the gradient wrt the block's pre-BatchNorm
 * output feeds nothing but this weight gradient (there is no data gradient wrt the frames), so it is formed on the way into LDS from dA and
 * raw and never stored.  d as for srvp_bn_bwd_finalize_apply (da_mode 0, bf16, LeakyReLU, unbordered 64-channel dA of 64x64 frames, scale /
 * shift / mean / invstd the rows of one [4][64] tensor); red / count / dgamma / dbeta / coef / C_real / param_grad_scale as there.
 * srvp_conv_in_wgrad_bn_ok: 1 if the descriptor and shape are served (SRVP_IN_WGRAD_BN=0 switches it off). */
int srvp_conv_in_wgrad_bn_ok(const srvp_bnbwd_desc* d, int Cin, int H, int W, int Cout, int k, int s, int p);
int srvp_conv_in_wgrad_bn(const float* x, const srvp_bnbwd_desc* d, const double* red, double count, float* dgamma, float* dbeta,
                          float* coef, int C_real, float param_grad_scale, float* dw, int N, int Cin, int Cout_real, void* stream);
/* fp32 parity mode: raw / draw are fp32 NHWC tensors (direct fp32 kernels: an fmaf chain in (ci, kh, kw) order) */
int srvp_conv_in_fwd_f32(const float* x, const float* w, void* raw, double* stats,
                         int N, int Cin, int H, int W, int Cout, int Cout_real, int k, int s, int p, void* stream);
int srvp_conv_in_wgrad_f32(const float* x, const void* draw, float* dw,
                           int N, int Cin, int H, int W, int Cout, int Cout_real, int k, int s, int p, void* stream);
int srvp_out_dpre_f32(const float* x_out, const float* dx_out, void* draw, float* dpre_f32,
                      int N, int nc, int H, int W, int C, int apply_sigmoid, void* stream);
/* sigmoid backward of the last decoder layer (conv.py:273-274): dpre = dx_ * x_ * (1 - x_) from the fp32
 * (N, nc, H, W) frame tensors into a bf16 NHWC tensor [N][H+2][W+2][C] (zero border, channels >= nc zero) */
/* draw == NULL: only the fp32 copy is produced (the caller runs both gradients of the layer on srvp_conv_in_fwd / _wgrad) */
int srvp_out_dpre(const float* x_out, const float* dx_out, void* draw, float* dpre_f32 /* optional fp32 (N, nc, H, W) copy */,
                  int N, int nc, int H, int W, int C, int apply_sigmoid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weight packing between the reference state-dict layouts (OIHW conv / IOHW convT, fp32) and tap-major bf16.
 * ------------------------------------------------------------------------------------------------ */
/* packed bf16 [ntaps][J][K]: packed[t][j][k] = w[ jr*sj + kr*sk + tap_off[t] ], 0 for padding.  Each of the J / K
 * axes is one or two zero-padded segments (two when the layer input is the concat of two padded tensors):
 * index i < X0 maps to real i (if < X0r), index X0 + i maps to real X0r + i (if i < X1r). */
typedef struct {
    int32_t ntaps; int32_t tap_off[SRVP_MAX_TAPS];
    int32_t J, K;
    int32_t J0, J0r, J1r;
    int32_t K0, K0r, K1r;
    int64_t sj, sk;
    /* optional: packed tap t = SUM of the source taps s (element offset s) whose bit is set in tap_set[t] (0 = plain
     * tap_off[t] mapping).  Used by the sub-pixel form of "nearest x2 upsample then 3x3 conv" (conv.py:331-349), where
     * each of the 2x2 / 4x4 effective taps is a sum of original taps; unpack adds a packed gradient to every tap of its set. */
    int32_t tap_set[SRVP_MAX_TAPS];
    /* 0: tap-major [t][J][K].  1: MFMA-fragment-major [t][K/64][4 (16-wide k slices)][J/32][64 lanes][8]: lane l of a
     * fragment holds row j = 32*jt + (l & 31), k = 64*cc + 16*kk + 8*(l >> 5) .. +7 -- one B operand of
     * v_mfma_f32_32x32x16_bf16 is then ONE contiguous 1 KiB load (the halo-tiled convolution reads its weights straight
     * from L2 into registers in this order).  Needs J % 32 == 0 and K % 64 == 0. */
    int32_t layout;
    int32_t dst_f32;              /* 1: the packed tensor is fp32 instead of bf16 (fp32 parity mode; layout 0 only) */
    int32_t kc_total, kc_off;     /* layout 1: this job fills the 64-wide K chunks [kc_off, kc_off + K/64) of a destination that has
                                   * kc_total chunks per tap (0 = K/64, i.e. the whole tensor): one job per phase of a space-to-depth K axis */
} srvp_pack_desc;
int srvp_pack_weight(const float* src, void* dst, const srvp_pack_desc* d, void* stream);
/* fp32 gradient: w_grad[ jr*sj + kr*sk + tap_off[t] ] += packed_grad[t][j][k]  (inverse mapping, for dW) */
int srvp_unpack_wgrad(const float* src, float* dst, const srvp_pack_desc* d, void* stream);
/* every layer of a network in one launch: `jobs_dev` is a DEVICE-resident array of njobs jobs (the caller builds it once per
 * plan -- buffers are static), total_wgs = sum over jobs of srvp_pack_job_wgs(ntaps*J*K).  Same per-element semantics as the calls above. */
typedef struct { const void* src; void* dst; srvp_pack_desc d; } srvp_pack_job;
int srvp_pack_weight_multi(const srvp_pack_job* jobs_dev, int njobs, int64_t total_wgs, void* stream);
int srvp_unpack_wgrad_multi(const srvp_pack_job* jobs_dev, int njobs, int64_t total_wgs, void* stream);
/* workgroups the multi launches give a job of `total` = ntaps*J*K elements; total_wgs above = the sum over the jobs */
int srvp_pack_job_wgs(int64_t total);
/* Tile form of the two launches above (round 4): one workgroup per small tile (pack: 32 j x 16 k x all taps, unpack: 16 j x 32 k x all
 * taps; halved j for 4x4 kernels) with the job descriptor staged in LDS -- 8 workgroups per CU instead of 2-3, one dependent round of
 * global loads per workgroup.  srvp_pack_job_tiles(d, unpack) = tiles of a job, 0 if the job is not eligible (fp32 packed tensors,
 * taps not innermost, channel counts not multiples of the tile): such jobs go through the multi launches.  `jobs_dev` of the _tiles
 * launches holds eligible jobs only (at most 256), total_tiles = the sum of their tile counts.  Byte-identical results. */
int srvp_pack_job_tiles(const srvp_pack_desc* d, int unpack);
int srvp_pack_weight_tiles(const srvp_pack_job* jobs_dev, int njobs, int64_t total_tiles, void* stream);
int srvp_unpack_wgrad_tiles(const srvp_pack_job* jobs_dev, int njobs, int64_t total_tiles, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Latent path: Linear / MLP / LSTM / residual Euler rollout (module/mlp.py, module/srvp.py:229-413), fp32.
 * ------------------------------------------------------------------------------------------------ */
/* C[M][N] (+)= act( A[M][K] (strides as,ak) * B[K][N] (strides bk,bn) + bias[N] ), fp32, generic strides */
int srvp_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs,
                  const float* bias, float* C, int64_t c_rs, int M, int N, int K, int act, int accumulate,
                  void* stream);
/* weight + bias gradient of one Linear layer in ONE launch: gw[M][N] += delta[:, :M]^T act[:, :N], gb[M] += column sums of delta
 * (gb may be NULL); delta [K][ld_delta], act [K][ld_act] row-major over the K = (steps x batch) rows (autograd of mlp.py:21-45) */
int srvp_linear_wgrad_f32(const float* delta, int64_t ld_delta, const float* act, int64_t ld_act, float* gw, int64_t ld_gw,
                          float* gb, int M, int N, int K, void* stream);
/* dst[blk * dst_stride + i] += src[blk * n + i], i < n, blk < nblk (p_z input gradients added onto the frame-start states) */
int srvp_add_blocks_f32(float* dst, int64_t dst_stride, const float* src, int nblk, int64_t n, void* stream);
/* Glue between the latent path and the conv stacks (srvp.py:216-221 and its backward; srvp.py:246-250, 268 backward):
 *  srvp_latent_to_z: decoder input rows dst[t*B+b] = [w[b] | y[t][b] | 0-padding to Cz] (bf16, or fp32 with dst_f32), y rows of frame t
 *                    at y + t * y_tstride;
 *  srvp_dz_split:    d_w[b] = sum_t dz[t*B+b][:nh] (+ d_w_add), d_y[t][b] = dz[t*B+b][nh:nh+ny] (+ d_y_add), d_y rows at d_y + t * dy_tstride;
 *  srvp_rows_scatter_add_f32: dst[idx[r]] += src[r] (idx int64 or int32). */
int srvp_latent_to_z(const float* w, const float* y, int64_t y_tstride, void* dst, int nt, int B, int nh, int ny, int Cz,
                     int dst_f32, void* stream);
int srvp_dz_split(const void* dz, int Cz, int elem_f32, int nt, int B, int nh, int ny, const float* d_w_add, const float* d_y_add,
                  float* d_w, float* d_y, int64_t dy_tstride, void* stream);
int srvp_rows_scatter_add_f32(float* dst, const void* idx, int idx_is_i64, const float* src, int64_t rows, int C, void* stream);

/* out = a*x + b*y (y may be NULL) */
int srvp_axpby_f32(float* out, float a, const float* x, float b, const float* y, int64_t n, void* stream);
/* column sums: out[n] (+)= sum_m A[m][n] */
int srvp_colsum_f32(const float* A, int64_t a_rs, float* out, int M, int N, int accumulate, void* stream);
/* elementwise helpers on fp32 rows */
int srvp_act_bwd_f32(const float* pre_or_out, const float* dy, float* dx, int64_t n, int act, int from_output,
                     void* stream);

/* LSTM (nn.LSTM(nhx, nh, 1), srvp.py:132,366): gates_x = x W_ih^T + b (precomputed by srvp_gemm_f32),
 * recurrent part here.  Saves gate activations for backward.  Layout: [T][B][4*nh], gate order i,f,g,o. */
int srvp_lstm_fwd(const float* gates_x, const float* w_hh, float* h_out, float* c_out, float* gates_act,
                  int T, int B, int nh, void* stream);
/* the same forward as ONE persistent launch over the T steps (csrc/rollout_fused.hip: clusters of nh / 16 workgroups per 16-row batch
 * tile -- nh / 32 per 32-row tile where the batch has more 16-row tiles than one co-resident launch holds or nh is not a multiple of 64 --,
 * W_hh slices resident in registers, h exchanged through global memory, one counter barrier per step).
 * (both persistent kernels: 0 is also returned when 8 * nh / 32 workgroups cannot be co-resident on the device; a cluster barrier
 *  that waits longer than ~seconds -- workgroups not co-resident after all -- gives up and counts the event in word 1 of the tile's
 *  256-byte counter block at the start of the workspace: results are then invalid, the device is not hung)
 * srvp_lstm_fused_ws_bytes: 0 = shape not eligible (nh in {64, 128, 256}), else the size of the zero-initialised-by-the-call workspace;
 * gates_x is not modified (srvp_lstm_fwd copies it into gates_act first) */
int64_t srvp_lstm_fused_ws_bytes(int T, int B, int nh);
int srvp_lstm_fwd_fused(const float* gates_x, const float* w_hh, float* h_out, float* c_out, float* gates_act,
                        int T, int B, int nh, void* ws, int64_t ws_bytes, void* stream);
/* the backward recurrence likewise as ONE persistent launch (same clusters, W_hh[gate rows][own 32 units] slices in registers, the gate
 * gradients of a step exchanged through dgates itself, one counter barrier per step); same eligibility and workspace as the forward.
 * dh_out: gradient wrt the LSTM outputs [T][B][nh]; writes dgates [T][B][4 nh] (what srvp_lstm_bwd writes). */
int srvp_lstm_bwd_fused(const float* dh_out, const float* w_hh, const float* c_out, const float* gates_act, float* dgates,
                        int T, int B, int nh, void* ws, int64_t ws_bytes, void* stream);
/* scratch: 2*B*nh floats */
int srvp_lstm_bwd(const float* dh_out, const float* w_hh, const float* c_out, const float* gates_act,
                  float* dgates, float* scratch, int T, int B, int nh, void* stream);

/* Residual Euler rollout (srvp.py:300-323,370-405) for the MLP `dynamics` and the prior MLP `p_z` (nl linear layers,
 * ReLU between; weights [out][in] row-major exactly as in the state dict).  Step i (0-based) belongs to frame slot
 * f = i / n_euler (frame index f+1); the first sub-step of a frame evaluates p_z on the current state and draws z from
 * the posterior parameters while f+1 < n_data_frames, from the prior afterwards (srvp.py:383-393). F = #frame slots. */
typedef struct {
    int32_t B, ny, nz, nh, nl;
    int32_t nsteps, n_euler, n_data_frames;
    float dt;
    const float* dyn_w[8]; const float* dyn_b[8];
    const float* pz_w[8]; const float* pz_b[8];
    const float* y0;                       /* [B][ny] */
    const float* q_z_params;               /* [F][B][2nz] (entries used only for posterior frames) or NULL */
    const float* eps_z;                    /* [F][B][nz] */
    float* y_all;                          /* [nsteps+1][B][ny] every state incl. y0 */
    float* z;                              /* [F][B][nz] */
    float* p_z_params;                     /* [F][B][2nz] */
    float* res;                            /* [nsteps][B][ny] */
    float* inp_all;                        /* [nsteps][B][ny+nz] dynamics inputs (saved for the weight gradient) */
    float* hid_dyn;                        /* [nl-1][nsteps][B][nh] post-ReLU hidden activations, or NULL (inference) */
    float* hid_pz;                         /* [nl-1][F][B][nh] or NULL */
    float* scratch_hid;                    /* [nl-1][B][nh], used when hid_* is NULL */
    float* scratch_out;                    /* [B][max(ny, 2nz)] */
    int32_t pz_external;                   /* 1: every frame is a posterior frame and the caller evaluates p_z (forward and
                                            * backward) BATCHED over all frames outside the serial chain -- the prior MLP only
                                            * feeds the KL term then (srvp.py:383-390), its input is the stored state */
    void* fused_ws; int64_t fused_ws_bytes; /* optional workspace of srvp_rollout_fused_ws_bytes(d) bytes: with it, posterior-only
                                            * training chains (pz_external, hid_dyn) run as ONE persistent kernel forward and ONE
                                            * backward (csrc/rollout_fused.hip) instead of nl + 1 launches per Euler step.  The
                                            * library owns its contents (two cluster-counter blocks, cleared by the forward's
                                            * preparation launch, then the split-K slabs): no initialisation by the caller */
} srvp_rollout_desc;
int srvp_rollout_fwd(const srvp_rollout_desc* d, void* stream);
/* bytes of fused_ws the persistent kernels need for this chain, 0 if it must run unfused (dimensions / LDS budget / mode) */
int64_t srvp_rollout_fused_ws_bytes(const srvp_rollout_desc* d);
/* The INFERENCE chain (pz_external = 0: p_z inside the loop, posterior samples while n_data_frames lasts, prior samples afterwards; reference
 * module/srvp.py:377-405, test.py:237-246) as persistent launches (csrc/rollout_fused.hip rollout_gen_kernel): bytes of fused_ws it needs, 0 if
 * the chain must run as per-layer launches (dimensions, LDS budget, workgroup placement not round-robin over the XCDs, SRVP_ROLLOUT_GEN_FUSED=0).
 * srvp_rollout_fwd takes it when fused_ws / fused_ws_bytes hold at least that much; hid_* / inp_all / scratch_* are not touched then. */
int64_t srvp_rollout_gen_ws_bytes(const srvp_rollout_desc* d);
/* Cluster-barrier timeouts of the persistent latent kernels since the library was loaded (a cluster whose workgroups were not
 * co-resident gives up after a bounded spin and leaves garbage behind): copied to pinned host memory in stream order; the host polls
 * it every few steps and raises if it is non-zero. */
int srvp_cluster_timeouts_read(unsigned* host_word, void* stream);
/* The persistent latent kernels deal the workgroups of a cluster to ONE XCD and VERIFY it per launch (HW_REG_XCC_ID of every member): where it
 * holds, the exchanged tiles are written with plain stores and stay in that XCD's L2 for the other members' L1-bypassing loads; otherwise
 * agent-scope write-through stores (csrc/rollout_fused.hip "XCD-LOCAL EXCHANGE").  srvp_cluster_set_xcd_local(0) forces the latter (A/B switch,
 * default 1 / env SRVP_CLUSTER_XCD_LOCAL; same results bit for bit).  srvp_cluster_stats_read: host_words2[0] = cluster-barrier timeouts,
 * [1] = clusters that ran XCD-local (summed over launches), since the library was loaded; pinned host memory, stream order. */
int srvp_cluster_set_xcd_local(int on);
int srvp_cluster_stats_read(unsigned* host_words2, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Deterministic mode of the fp32 parity mode (elem_f32 launches): the reference's CPU path is bit-reproducible run to run; with this
 * switch on, so is the parity mode.  Every cross-workgroup sum that is otherwise formed with atomics in arrival order -- the
 * BatchNorm-backward sums, the image-side weight gradient, the latent weight gradients (single split), the ELBO accumulators (single
 * workgroup) -- is formed in a fixed order; the BatchNorm FORWARD statistics are not taken in the convolution epilogues (pass
 * stats = NULL) but by srvp_bn_stats_f32_det from the stored fp32 raw output (= the accumulators).  `workspace`: >= 8 MiB of device
 * memory owned by the caller for the per-workgroup partial sums; all launches from one stream while the mode is on.  Process-wide.
 * ------------------------------------------------------------------------------------------------ */
int srvp_set_deterministic(int on, void* workspace, int64_t workspace_bytes);
int srvp_get_deterministic(void);
/* stats[0][c] += sum_r raw[r][c], stats[1][c] += sum_r raw[r][c]^2 over the rows of an fp32 tensor [rows][C], fp64, fixed order */
int srvp_bn_stats_f32_det(const float* raw, int64_t rows, int C, double* stats, void* stream);
typedef struct {
    srvp_rollout_desc f;
    const float* d_y_all;                  /* [nsteps+1][B][ny] gradient wrt every stored state (zeros where unused) */
    const float* d_z;                      /* [F][B][nz] or NULL */
    const float* d_pz;                     /* [F][B][2nz] direct gradient wrt the prior parameters (KL) or NULL */
    const float* d_res;                    /* [nsteps][B][ny] or NULL */
    float* d_y0;                           /* [B][ny] */
    float* d_qz;                           /* [F][B][2nz] gradient wrt the posterior parameters through the samples */
    float* dhid_dyn;                       /* [nl][nsteps][B][max(nh,ny)] per-layer pre-activation deltas */
    float* dhid_pz;                        /* [nl][F][B][max(nh,2nz)] */
    float* work;                           /* 3*B*ny + B*(ny+nz) + B*nz floats */
    float* dinp_all;                       /* [nsteps][B][ny+nz] per-step input gradients (pz_external chains) or NULL */
} srvp_rollout_bwd_desc;
int srvp_rollout_bwd(const srvp_rollout_bwd_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ELBO terms (train.py:90-106) and Adam (train.py:289; torch.optim.Adam defaults)
 * ------------------------------------------------------------------------------------------------ */
/* nll = sum (x-x_)^2/(2 s^2) + log s + .5 log 2pi ; optionally d_x_ = gscale*(x_-x)/s^2.  out: double[1] += */
int srvp_nll(const float* x_, const float* x, float* d_x_, int64_t n, float scale, float gscale, double* out,
             void* stream);
/* KL( N(q) || N(p) ) summed (p == NULL -> N(0,1)); raw params [rows][2d]; grads scaled by gscale. out: double[1] += */
int srvp_kl(const float* q, const float* p, float* dq, float* dp, int64_t rows, int d, float gscale, double* out,
            void* stream);
/* sum over rows of ||res_row||_2 ; d_res = gscale * res/||res|| (0 at 0).  out: double[1] += */
int srvp_l2rows(const float* res, float* d_res, int64_t rows, int d, float gscale, double* out, void* stream);
/* rsample: out = loc + eps*(softplus(raw)+1e-8) ; backward: dparams += [dout, dout*eps*sigmoid(raw)] */
int srvp_rsample_fwd(const float* params, const float* eps, float* out, int64_t rows, int d, void* stream);
int srvp_rsample_bwd(const float* params, const float* eps, const float* dout, float* dparams, int64_t rows, int d,
                     int accumulate, void* stream);
/* fused Adam over one flat fp32 buffer; step_size = lr/(1-b1^t), bc2_sqrt = sqrt(1-b2^t) */
int srvp_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
              int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data parallelism over the GPUs of a node: RCCL (xGMI) collectives enqueued on the caller's stream -- the gradient
 * all-reduce of DistributedDataParallel (train.py:309-314) and the SyncBatchNorm statistics exchange (train.py:278-283).
 * RCCL is bound at run time (dlopen), so the library has no link-time dependency on it.  One process per GPU; the 128-byte
 * id made by rank 0 (srvp_comm_unique_id) is handed to the other ranks by the host (e.g. torch.distributed's store).
 * ------------------------------------------------------------------------------------------------ */
int srvp_comm_unique_id(void* id128);
int srvp_comm_init(const void* id128, int rank, int world, void** comm_out);
int srvp_comm_destroy(void* comm);
/* What RCCL reports about a communicator (diagnostics of `bench.py --gpus N`): info3[0] = ncclCommCount, [1] = ncclCommUserRank,
 * [2] = ncclGetVersion; -1 where the loaded librccl lacks the call. */
int srvp_comm_info(void* comm, int* info3);
/* in-place sum over ranks */
int srvp_allreduce_f64(void* comm, double* buf, int64_t n, void* stream);
int srvp_allreduce_f32(void* comm, float* buf, int64_t n, void* stream);
/* generic in-place all-reduce: dtype 0 = fp32, 1 = fp64, 2 = bf16; op 0 = sum, 1 = average over ranks (ncclAvg -- DistributedDataParallel's
 * gradient averaging, train.py:309-314, without a separate 1/world pass over the buffer) */
int srvp_allreduce(void* comm, void* buf, int64_t n, int dtype, int op, void* stream);
int srvp_bcast_bytes(void* comm, void* buf, int64_t nbytes, int root, void* stream);
/* PROTOTYPE, SRVP_COMM=peer: the SyncBatchNorm statistics exchange as a one-sided peer read (csrc/comm.hip).  Every rank creates a slab
 * (device memory, exported through hipIpc: ipc_handle64 receives the 64-byte handle), opens the other ranks' slabs, and a collective is one
 * single-workgroup launch per rank: publish buf[0..n) at data_off of the own slab, raise the flag at flag_off to `seq`, wait (bounded: 5 s,
 * then *err_flag = 1) until every peer's flag reaches `seq`, replace buf by the sum over ranks taken in rank order.  The caller alternates
 * two (data_off, flag_off) parities per call site and increases seq with every use.  world <= 8; slabs: HOST array of `world` device pointers. */
int srvp_peer_slab_create(int64_t bytes, void** dev_ptr, void* ipc_handle64);
int srvp_peer_slab_open(const void* ipc_handle64, void** dev_ptr);
int srvp_peer_slab_close(void* dev_ptr, int owned);
int srvp_peer_allreduce_f64(double* buf, int n, int rank, int world, void* const* slabs, int64_t data_off, int64_t flag_off,
                            uint64_t seq, int* err_flag, void* stream);

/* misc */
int srvp_fill_f64(double* p, int64_t n, double v, void* stream);
/* input pipeline (replaces the CPU collate of data/base.py:54-84): uint8 videos stacked [B][T][H][W][C]
 * (C = 1 for grey-scale arrays without channel axis) -> float32 frames (T, B, C, H, W) in [0, 1] = value / 255 */
int srvp_frames_u8_to_f32(const void* in_u8, float* out, int B, int T, int H, int W, int C, void* stream);
/* Stochastic Moving-MNIST batch rasteriser (data/mmnist.py:116-124 + data/base.py:71-84): digits_u8
 * [n_digits][dh][dw]; idx int32 [B][num_digits] digit of each object; pos int32 [B][num_digits][T][2] = (row, column) offset of
 * the object at frame t (the rounded trajectory of mmnist.py:165).  out float32 (T, B, 1, nx, nx) = min(255, sum) / 255 and/or
 * out_u8 uint8 [B][T][nx][nx] (the reference's per-video arrays); either may be NULL. */
int srvp_mmnist_render(const void* digits_u8, int n_digits, int dh, int dw, const int* idx, const int* pos, int B, int T,
                       int num_digits, int nx, float* out, void* out_u8, void* stream);
/* The trajectories themselves on the device (data/mmnist.py:113-237: digit index, start, speed, bounces with a fresh speed at every
 * wall contact): one thread per object, counter-based Philox4x32-10 keyed by `seed`, counter (draw block, object, batch_counter) --
 * batch k of a run is a function of (seed, k) alone.  idx int32 [B][num_digits], pos int32 [B][num_digits][T][2] (row, column offset:
 * the inputs of srvp_mmnist_render), contacts (optional) int32 [B][num_digits] = wall contacts of each object.  Equal to the
 * reference generator in distribution, not in its np.random stream. */
int srvp_mmnist_trajectories(uint64_t seed, uint64_t batch_counter, int B, int num_digits, int T, int nx, int dh, int dw,
                             int max_speed, int deterministic, int n_digits, int* idx, int* pos, int* contacts, void* stream);
int srvp_cast_f32_bf16(const float* src, void* dst, int64_t rows, int cols, int dst_cols, void* stream);
/* dst fp32 [n] = scale * src bf16 [n] (the way back of the opt-in bf16 gradient payload of the data-parallel exchange) */
int srvp_cast_bf16_f32(const void* src, float* dst, int64_t n, float scale, void* stream);
/* dst bf16 [M][C] = sum_z parts[z * slab_elems + m * C + c] (z ascending), the slabs of an srvp_conv_mfma launch with splitk > 1;
 * stats (may be NULL): += per-column sum and sum of squares of the fp32 sums (column c -> stats[c % stat_mod], stats[stat_mod + c %
 * stat_mod]), i.e. what the convolution's own BatchNorm-statistics epilogue computes (conv.py:104) */
int srvp_splitk_finish(const float* parts, int splitk, int64_t slab_elems, int64_t M, int C, void* dst, double* stats,
                       int stat_mod, void* stream);
/* fp32 parity mode: the same zero-padding row copy into an fp32 [rows][dst_cols] tensor */
int srvp_pad_f32(const float* src, float* dst, int64_t rows, int cols, int dst_cols, void* stream);
/* evaluation metrics: x, y = `planes` float32 planes of H x W (<= 64 x 64; (nt*B*C) planes of NCHW frames).
 * mse[p] = mean((x-y)^2) (test.py:249; PSNR = 10 log10(1/mse), test.py:251, train.py:175-176);
 * ssim[p] = mean over the (H-F+1) x (W-F+1) window positions of the SSIM map of metrics/ssim.py:92-110 (gaussian window
 * F = filter_size, sigma; constants (k1 max_val)^2, (k2 max_val)^2) = test.py:56-57.  Either output may be NULL. */
int srvp_frame_metrics(const float* x, const float* y, int64_t planes, int H, int W, float max_val, int filter_size,
                       float sigma, float k1, float k2, float* mse, float* ssim, void* stream);
/* dsel[b][hw][c] = sum_t dcat[t*B+b][hw][coff+c]  (gradient of the skip expand over time, srvp.py:222-223; the
 * gather of srvp.py:187 is undone by srvp_bn_bwd_* through da2_idx) */
int srvp_skip_grad_reduce(const void* dcat, int cstride, int coff, int C, int HW, int T, int B, void* dsel, void* stream);
int srvp_skip_grad_reduce_f32(const void* dcat, int cstride, int coff, int C, int HW, int T, int B, void* dsel, void* stream);

#ifdef __cplusplus
}
#endif
#endif
