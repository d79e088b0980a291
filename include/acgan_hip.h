/*
 * acgan_hip.h — C ABI of libacgan_hip.so: the MI355X (gfx950) kernels behind the
 * Augmented CycleGAN training step of adrianalbert/domain-transfer-GAN.
 *
 * The reference has NO native layer: its hot path reaches the
 * device through torch.nn modules.  Each entry point below therefore cites the
 * reference call site(s) whose arithmetic it replaces (paths relative to
 * /root/reference/augmented_cyclegan/).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - plain C: pointers + ints, no torch types.  All pointers are DEVICE pointers
 *    owned by the caller; the library never allocates device memory, never
 *    synchronises, and only enqueues on `stream` (a hipStream_t passed as void*).
 *  - return 0 on success, a negative acg_status otherwise; acg_last_error()
 *    returns a thread-local message for the last failure on this thread.
 *  - activations are fp32 NHWC with the channel count padded to a multiple of 16
 *    ("C16"); padded channels hold zeros.  IMAGE tensors — up to 4 real channels: the
 *    networks' inputs and outputs (networks.py:159-160, 187-188, 321, 365, 445), their
 *    gradients, the PatchGAN maps — may be stored with 4 channels ("C4", 16 bytes per pixel)
 *    where they are the thin side of a thin layer: see acg_conv_desc.  Weights are passed in
 *    the packed forms produced by acg_pack_conv_weight().
 */
#ifndef ACGAN_HIP_H
#define ACGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACG_VERSION 117

typedef enum {
    ACG_OK = 0,
    ACG_ERR_INVALID = -1,   /* bad argument / unsupported shape */
    ACG_ERR_WORKSPACE = -2, /* workspace too small */
    ACG_ERR_LAUNCH = -3,    /* hip launch error */
    ACG_ERR_COMM = -4       /* RCCL unavailable or an RCCL call failed */
} acg_status;

typedef enum { ACG_ACT_NONE = 0, ACG_ACT_RELU = 1, ACG_ACT_LRELU = 2 /* slope 0.2 */, ACG_ACT_TANH = 3 } acg_act;
typedef enum { ACG_PAD_ZERO = 0, ACG_PAD_REFLECT = 1 } acg_pad_mode;
typedef enum { ACG_IMPL_MFMA = 0, ACG_IMPL_DIRECT = 1 } acg_conv_impl;
typedef enum { ACG_PREC_F32 = 0, ACG_PREC_BF16 = 1, ACG_PREC_BF16X3 = 2 } acg_precision;

/* Geometry of one nn.Conv2d (or of the Conv2d whose adjoint an nn.ConvTranspose2d is).
 * Ci / Co are the STORED channel counts of the NHWC tensors: multiples of 16, or 4 for an image tensor (Cir resp. Cor <= 4)
 * on the thin side of a layer with K > 1 whose other side is wider — the layers whose kernels gather / write 4 channels per
 * pixel anyway.  Packed weights do not depend on it (acg_pack_conv_weight takes widths padded to 16). */
typedef struct {
    int N, Hi, Wi, Ci;
    int Ho, Wo, Co;
    int K, stride, pad, pad_mode;
    int Cir, Cor; /* REAL (unpadded) channel counts, 0 = unknown.  Cir <= 4 / Cor <= 4 select the thin-channel
                   * K-flattening (8 taps x 4 channels per 32-deep stage) in the fp32 kernels. */
} acg_conv_desc;

int acg_version(void);
const char *acg_last_error(void);
/* Name, template arguments included, of the convolution kernel the calling thread dispatched last (e.g.
 * "igemm_conv_x3_ws<REFLECT=1,STATS=1,ROWP=1>").  Diagnostic: the reference delegates algorithm choice to
 * torch.nn.Conv2d / cuDNN (networks.py:158-189) and cannot say what ran; bench.py labels its roofline with this. */
const char *acg_last_kernel(void);
/* Measurement hooks (bench.py; the reference has no counterpart — its timing is train.py:243's wall clock per image).
 * acg_debug_mid_event: `event` (a hipEvent_t) is recorded by the NEXT weight-gradient entry point of the calling thread
 * between its main kernel and its split-K reduction launch, then forgotten: the two are timed apart.
 * acg_probe_mfma_rate: enqueues a register-only v_mfma_f32_16x16x32_bf16 loop (`iters` x 16 MFMAs per wave, 8 waves per
 * CU, random operands, no LDS / memory traffic) and returns the FLOPs it executes in *flops_out; timed by the caller it
 * gives the matrix rate the device sustains in its present clock / power state (scratch: >= 512 floats per CU, never written). */
int acg_debug_mid_event(void *event);
int acg_probe_mfma_rate(float *scratch, size_t scratch_floats, int iters, double *flops_out, void *stream);
/* selects the convolution implementation for subsequent calls on this process
 * (MFMA implicit GEMM = product path; DIRECT = naive one-thread-per-output kernels kept
 * as an on-device cross-check).  Both are HIP kernels; there is no CPU path. */
int acg_set_conv_impl(int impl);
/* arithmetic of the MFMA convolution kernels (tensors stay fp32 in HBM in every mode):
 *  ACG_PREC_BF16X3 (default) each fp32 operand is split into bf16 hi + lo on its way into LDS and every product is
 *                  formed as lo*hi + hi*lo + hi*hi by three v_mfma_f32_32x32x16_bf16 with fp32 accumulation: operands
 *                  keep 16 mantissa bits (~4e-6 rms error per convolution), inside the 1e-3 parity bar, at bf16 rate;
 *  ACG_PREC_F32    exact fp32 FMA chain on v_mfma_f32_32x32x2_f32 — the strict mode;
 *  ACG_PREC_BF16   operands rounded to bf16, fp32 accumulate — throughput mode, NOT a parity path.
 * Packed weights must be re-packed (acg_pack_conv_weight) after a switch. */
int acg_set_conv_precision(int prec);

/* ---- data path (dataloader.py:17-35): raw[N,H,W,Craw] -> first C channels, NaN -> 0, per-sample / per-channel min-max
 *      scaling to [-1, 1] (constant planes -> 0), written NCHW.  Once per data set, on the device. ---- */
int acg_minmax_scale_nhwc_to_nchw(const float *raw, float *out_nchw, int N, int H, int W, int Craw, int C, void *stream);

/* ---- layout at the API edge (reference tensors are NCHW: dataloader.py:26, model.py:404) ---- */
int acg_nchw_to_nhwc16(const float *src, float *dst, int N, int C, int H, int W, int Cp, void *stream);
int acg_nhwc16_to_nchw(const float *src, float *dst, int N, int C, int H, int W, int Cp, void *stream);
/* torch.cat((a, b), 1) on C16 tensors — model.py:410, 472 */
int acg_concat_channels(const float *a, int Ca, int Cap, const float *b, int Cb, int Cbp, float *dst, int Cdp,
                        size_t npix, void *stream);
/* adjoint of the above: splits d(dst) into d(a), d(b) (either may be NULL) */
int acg_split_channels(const float *gdst, int Cdp, float *ga, int Ca, int Cap, float *gb, int Cb, int Cbp,
                       size_t npix, void *stream);

/* ---- weights: OIHW (torch layout, real channel counts Or x Ir) -> packed, zero padded ----
 * wf: [K*K][Ci/8][CoP][8]  (B operand of forward / ConvTranspose-backward GEMMs)
 * wb: [K*K][Co/8][CiP][8]  (B operand of data-gradient / ConvTranspose-forward GEMMs)
 * CoP = acg_ncols_pad(Co), CiP = acg_ncols_pad(Ci); Ci / Co here are widths padded to 16.  Either output may be NULL.
 * Size the buffers with acg_packed_w{f,b}_elems: some layers carry a second packed form behind the first. */
int acg_ncols_pad(int c);
size_t acg_packed_wf_elems(int K, int Ci, int Co);
size_t acg_packed_wb_elems(int K, int Ci, int Co);
int acg_pack_conv_weight(const float *w_oihw, int Or, int Ir, int K, int Ci, int Co, float *wf, float *wb, void *stream);
/* The regular (non-thin) layers of a whole network in ONE launch — the packed copies are refreshed once per optimiser step
 * (model.py:447-452, 510-515 `optimizer.step()` changes every weight), and one launch per layer was 68 five-microsecond
 * kernels per training step.  bf16 / bf16x3 arithmetic; a layer qualifies where acg_pack_conv_weights_multi_supported
 * (thin layers and the exact-fp32 mode keep acg_pack_conv_weight).  Same bytes as acg_pack_conv_weight writes. */
typedef struct {
    const float *w;      /* OIHW, real Or x Ir */
    float *wf, *wb;      /* acg_packed_w{f,b}_elems floats each */
    int Or, Ir, K, Ci, Co;
} acg_pack_item;
int acg_pack_conv_weights_multi_supported(int Or, int Ir, int K);
int acg_pack_conv_weights_multi(const acg_pack_item *items, int n, void *stream);
int acg_pad_vector(const float *src, int n, float *dst, int np, void *stream); /* bias -> C16 */

/* ---- nn.Conv2d forward (+bias, + fused activation) — networks.py:159-188, 211-243, 321-338,
 *      365-382, 445-471; modules.py:162,180,211,227 (reflection pad folded into the loader). */
int acg_conv2d_fwd(const acg_conv_desc *d, const float *x, const float *wf, const float *bias, float *y, int act,
                   void *stream);
/* acg_conv2d_fwd with act = NONE that also writes, from its epilogue, the per-tile partial statistics of y that the
 * InstanceNorm / CondInstanceNorm behind the convolution needs (modules.py:24-31, 46-55): stats[n][tile][{mean,M2}][Co]
 * over tiles of 128 consecutive output pixels, to be merged by acg_norm_stats_from_partials(rows_per_chunk = 128).
 * Supported (query first) for the bf16x3 128-column tile: Co >= 128, Ci % 32 == 0, Ho*Wo % 128 == 0. */
int acg_conv2d_fwd_stats_supported(const acg_conv_desc *d);
int acg_conv2d_fwd_stats(const acg_conv_desc *d, const float *x, const float *wf, const float *bias, float *y,
                         float *stats, void *stream);
/* data gradient of the above (autograd of nn.Conv2d / ReflectionPad2d): dy -> dx.
 * Reflect padding needs a workspace for the padded gradient image. */
size_t acg_conv2d_bwd_data_workspace_bytes(const acg_conv_desc *d);
int acg_conv2d_bwd_data(const acg_conv_desc *d, const float *dy, const float *wb, float *dx, void *workspace,
                        size_t ws_bytes, void *stream);
/* dx = data gradient + addend (same shape as dx), fused into the convolution epilogue: the gradient arriving over a
 * ResnetBlock's skip connection (modules.py:185-188, 232-235) joins the gradient of the block's first convolution without
 * a separate element-wise pass.  Supported (query first) where the reflect data gradient takes its frame path. */
int acg_conv2d_bwd_data_add_supported(const acg_conv_desc *d);
int acg_conv2d_bwd_data_add(const acg_conv_desc *d, const float *dy, const float *wb, const float *addend,
                            const unsigned *addend_sign_mask, float *dx, void *ws, size_t ws_bytes, void *stream);
/* addend_sign_mask (may be NULL): acg_norm_apply's sign bitmask of the block output; only elements whose bit is set are
 * added, i.e. the skip gradient dy * (y > 0) is formed here from dy itself and never written to memory. */
/* dx = data gradient * (x > 0), x = this convolution's own INPUT when that input is the ReLU output of the layer in front
 * (ResnetBlock: pad-conv-ReLU-pad-conv, modules.py:211-227): dx is then the gradient w.r.t. that layer's PRE-activation and
 * its separate activation-backward pass is skipped (torch runs threshold_backward there).  Same support query as _add. */
int acg_conv2d_bwd_data_relu(const acg_conv_desc *d, const float *dy, const float *wb, const float *x, float *dx,
                             void *ws, size_t ws_bytes, void *stream);
/* ---- Pre-split ("S16") activation storage for the MFMA-bound 3x3 layers (the ResnetBlock / CINResnetBlock convolutions,
 * modules.py:139-235, in the bf16x3 arithmetic).  An S16 tensor has the shape and byte size of its fp32 NHWC twin; per pixel
 * and 8-channel group it holds 16 bytes of bf16 hi (RNE of x) followed by 16 bytes of bf16 lo (RNE of x - hi): the operand
 * form the convolutions split every fp32 value into inside their loaders otherwise.  Written once by the producer of an
 * activation (norm apply, convolution epilogue), it lets both operands of the convolution travel global -> LDS by LDS-DMA.
 * acg_conv2d_s16_supported: forward, data gradient and weight gradient of this layer all take S16 operands. */
int acg_s16_encode(const float *x, void *y_s16, size_t n, void *stream);   /* n elements, n % 8 == 0 */
int acg_s16_decode(const void *x_s16, float *y, size_t n, void *stream);   /* y = hi + lo */
int acg_conv2d_s16_supported(const acg_conv_desc *d);
/* x S16; y fp32, or S16 when out_s16 != 0; stats (may be NULL): the per-tile statistics of acg_conv2d_fwd_stats (fp32 y,
 * act NONE only) */
int acg_conv2d_fwd_s16(const acg_conv_desc *d, const void *x_s16, const float *wf, const float *bias, void *y, int act,
                       float *stats, int out_s16, void *stream);
/* dy S16.  out_s16 == 0: dx fp32, optional addend (+ sign bitmask) as acg_conv2d_bwd_data_add.  out_s16 != 0: dx S16,
 * optional relu_src_s16 as acg_conv2d_bwd_data_relu (the convolution's own S16 input; only its sign is read). */
int acg_conv2d_bwd_data_s16(const acg_conv_desc *d, const void *dy_s16, const float *wb, void *dx, void *ws, size_t ws_bytes,
                            const float *addend, const unsigned *addend_sign_mask, const void *relu_src_s16, int out_s16,
                            void *stream);
/* ... that also produces the first pass of the backward of the (Cond)InstanceNorm whose OUTPUT gradient dx is
 * (modules.py:83-97, 121-131: y = act(norm(x) * gamma + beta [+ res]); fp32 dx only): per 128-pixel tile of dx the sums
 * part[((n * (Hi*Wi/128) + tile) * 2 + {0, 1}) * Ci + c] = (sum gy, sum gy * xhat), gy = dx * act'(y), xhat = (x - mean) *
 * rstd — what acg_norm_bwd computes in a pass of its own over dx and x; hand them to acg_norm_bwd_partials.  The activation
 * mask comes from sign_mask (the bitmask acg_norm_apply stored) or, when that is NULL, is recomputed from x with gamma /
 * beta (gstride 0 or Ci); act NONE or RELU. */
typedef struct acg_norm_sums {
    const float *x, *mean, *rstd;     /* the norm's input (N,Hi,Wi,Ci) and its statistics [N*Ci] */
    const float *gamma, *beta;        /* read only when act != NONE and sign_mask == NULL */
    int gstride;
    const unsigned *sign_mask;
    int act;
    float *part;
} acg_norm_sums;
int acg_conv2d_bwd_data_s16_sums_supported(const acg_conv_desc *d);
int acg_conv2d_bwd_data_s16_sums(const acg_conv_desc *d, const void *dy_s16, const float *wb, float *dx, void *ws,
                                 size_t ws_bytes, const float *addend, const unsigned *addend_sign_mask,
                                 const acg_norm_sums *ns, void *stream);
/* ... and on fp32 operands: the persistent row pipeline (zero-padded 3x3 stride 1, Co == 32, Ci == 64, Wi % 128 == 0), and
 * from round 6 the generic row-patch tile (3x3 stride 1, Ci == 32, Co == 64, Wi % 128 == 0), the four-phase tile of the stride-2
 * 3x3 layer (Ci == 64, Wo % 128 == 0) and the thin-row kernel of the 7x7 head (Ci == 32, a C4 image on the output side, Hi % 8 == 0, Wi % 16 == 0).  ns->sign_mask must be NULL (the activation
 * mask is recomputed from ns->x); part[N][Hi * Wi / 128][2][Ci], every entry written. */
int acg_conv2d_bwd_data_sums_supported(const acg_conv_desc *d);
int acg_conv2d_bwd_data_sums(const acg_conv_desc *d, const float *dy, const float *wb, float *dx, void *ws, size_t ws_bytes,
                             const acg_norm_sums *ns, void *stream);
/* conv + ReLU with S16 output that also stores (y > 0) as a sign bitmask (layout of acg_norm_apply's: bit e % 32 of word e / 32
 * for float index e; Co % 32 == 0), and the S16-output data gradient of the convolution BEHIND it masked by that bitmask
 * instead of by the sign of its S16 input (where acg_conv2d_bwd_data_s16_sums_supported(d)): the pad-conv-ReLU-pad-conv chain
 * of modules.py:211-227 without reading the activation a second time */
int acg_conv2d_fwd_s16_mask(const acg_conv_desc *d, const void *x_s16, const float *wf, const float *bias, void *y_s16,
                            unsigned *sign_mask, void *stream);
int acg_conv2d_bwd_data_s16_mask(const acg_conv_desc *d, const void *dy_s16, const float *wb, void *dx_s16, void *ws,
                                 size_t ws_bytes, const unsigned *relu_sign_mask, void *stream);
/* x and dy S16; dw / db as acg_conv2d_bwd_weight */
int acg_conv2d_bwd_weight_s16(const acg_conv_desc *d, const void *x_s16, const void *dy_s16, float *dw, float *db, int Or,
                              int Ir, void *ws, size_t ws_bytes, int accumulate, void *stream);
/* weight (+bias) gradient: x, dy -> dw in torch OIHW layout (Or x Ir real channels), db[Or] (may be NULL).
 * Deterministic split-K over pixels with a second-stage reduction (no atomics).  accumulate != 0: the result is ADDED to
 * dw / db — pass the parameter's .grad (what autograd's AccumulateGrad does after loss.backward(), model.py:445, 509,
 * with one extra element-wise kernel per parameter); 0: dw / db are overwritten. */
size_t acg_conv2d_bwd_weight_workspace_bytes(const acg_conv_desc *d);
int acg_conv2d_bwd_weight(const acg_conv_desc *d, const float *x, const float *dy, float *dw_oihw, float *db,
                          int Or, int Ir, void *workspace, size_t ws_bytes, int accumulate, void *stream);

/* ---- nn.ConvTranspose2d(k3,s2,p1,op1) — networks.py:178-179, 231-234.  `d` describes the
 *      Conv2d it is the adjoint of (Hi,Wi,Ci = the LARGE side = ConvTranspose output). ---- */
int acg_conv_transpose2d_fwd(const acg_conv_desc *d, const float *x, const float *wb, const float *bias, float *y,
                             int act, void *stream);
/* ... that also emits the per-tile statistics of acg_conv2d_fwd_stats for an InstanceNorm behind the transposed convolution
 * (networks.py:178-181, 231-236): stats [N][Hi*Wi/128][2][Ci], Hi x Wi x Ci = the transposed convolution's OUTPUT. */
int acg_conv_transpose2d_fwd_stats_supported(const acg_conv_desc *d);
int acg_conv_transpose2d_fwd_stats(const acg_conv_desc *d, const float *x, const float *wb, const float *bias, float *y,
                                   float *stats, void *stream);
int acg_conv_transpose2d_bwd_data(const acg_conv_desc *d, const float *dy, const float *wf, float *dx, void *stream);
int acg_conv_transpose2d_bwd_weight(const acg_conv_desc *d, const float *x, const float *dy, float *dw_oihw,
                                    float *db, int Or, int Ir, void *workspace, size_t ws_bytes, int accumulate,
                                    void *stream);

/* ---- normalisation: InstanceNorm (modules.py:64-97, biased var), CondInstanceNorm
 *      (modules.py:104-132, UNBIASED var, per-sample affine), BatchNorm2d/1d train mode
 *      (networks.py:407-415, 450-466).  Tensor viewed as [G groups][P pixels][C]:
 *      IN/CIN: G=N, P=H*W;  BN: G=1, P=N*H*W. ---- */
size_t acg_norm_workspace_bytes(int G, size_t P, int C);
/* mean[G*C], rstd[G*C]; unbiased!=0 selects var*P/(P-1).  If run_mean/run_var are non-NULL
 * (BatchNorm) they are updated with momentum (running_var takes the unbiased variance). */
int acg_norm_stats(const float *x, int G, size_t P, int C, float eps, int unbiased, float *mean, float *rstd,
                   float *run_mean, float *run_var, float momentum, void *workspace, size_t ws_bytes, void *stream);
/* the same statistics from per-chunk (mean, M2) partials a producer already holds (acg_conv2d_fwd_stats), merged with
 * Chan's formula; x is not read.  part[((g*nchunks + chunk)*2 + {0,1})*C + c], nchunks = ceil(P / rows_per_chunk). */
int acg_norm_stats_from_partials(const float *part, int G, size_t P, int C, int rows_per_chunk, float eps, int unbiased,
                                 float *mean, float *rstd, void *stream);
/* BatchNorm eval mode: mean/rstd (length Cp) from the running buffers (length C) */
int acg_bn_eval_stats(const float *run_mean, const float *run_var, int C, int Cp, float eps, float *mean, float *rstd,
                      void *stream);
/* y = act((x-mean)*rstd*gamma + beta [+ res]); gamma/beta indexed [g*gstride + c] (gstride 0 or C) */
int acg_norm_apply(const float *x, const float *mean, const float *rstd, const float *gamma, const float *beta,
                   int gstride, const float *res, float *y, unsigned *sign_mask, int G, size_t P, int C, int act, int fmt,
                   void *stream);
/* fmt: 0 = fp32 tensors; bit 1 set: y is written pre-split (S16, see acg_s16_encode), bit 0 set: res is read pre-split
 * (fmt 2 or 3; ReLU, C % 8 == 0): the producer of a residual-block activation writes the operand form of the next
 * convolution once, instead of every convolution loader splitting it again.  fmt 1 (res pre-split, y fp32) is the output
 * norm of the LAST block of a trunk: the layer behind it reads fp32, so no acg_s16_decode pass is needed. */
/* sign_mask (may be NULL; needs res, ReLU/LeakyReLU, P*C/4 % 8 == 0): ceil(G*P*C/32) words, bit e%32 of word e/32 =
 * (y[e] > 0).  acg_norm_bwd takes it in place of y: the backward of ReLU(x + IN(conv(..))) (modules.py:185-188, 232-235)
 * needs only the sign of y, and reads 1/32 of a tensor instead of the tensor, twice. */
/* backward: dy (w.r.t. y), y, x -> dx, dres (if has_res; = dy*act'(y)), dgamma/dbeta.
 * gstride == 0 (InstanceNorm / BatchNorm): dgamma/dbeta hold the first `nparam` (real, unpadded) channels, summed over the
 * groups; accumulate != 0 ADDS them to the destination (pass the parameter's .grad: no separate accumulation kernel).
 * gstride == C (CondInstanceNorm): dgamma/dbeta are [G*C] (gradients of the per-sample scale / shift), nparam and accumulate
 * are ignored (must be 0).  y may be NULL when no residual was added: the activation mask is then recomputed from x with
 * gamma/beta (one tensor stream less per pass).  unbiased as in acg_norm_stats; unbiased == 2 means the statistics were
 * constants (BatchNorm eval mode): dx = gamma*rstd*dy*act'. */
int acg_norm_bwd(const float *dy, const float *y, const unsigned *sign_mask, const float *x, const float *mean,
                 const float *rstd,
                 const float *gamma, const float *beta, int gstride, float *dx, float *dres, float *dgamma, float *dbeta,
                 int nparam, int accumulate, int G, size_t P, int C, int act, int unbiased, int dx_s16, void *workspace,
                 size_t ws_bytes, void *stream);
/* dx_s16 != 0: dx is written pre-split (S16) for the convolution gradients that consume it (ReLU, no dres, C % 8 == 0) */
/* the same with the first pass already done: part[((g * nchunks + chunk) * 2 + {0, 1}) * C + c] partial sums over any
 * partition of the P rows into nchunks chunks (acg_conv2d_bwd_data_s16_sums: 128-pixel tiles) */
int acg_norm_bwd_partials(const float *dy, const float *y, const unsigned *sign_mask, const float *x, const float *mean,
                          const float *rstd, const float *gamma, const float *beta, int gstride, float *dx, float *dres,
                          float *dgamma, float *dbeta, int nparam, int accumulate, int G, size_t P, int C, int act,
                          int unbiased, int dx_s16, const float *part, int nchunks, void *workspace, size_t ws_bytes,
                          void *stream);

/* the two halves of acg_norm_bwd, for SyncBN: local sums[(g*2+{0,1})*C+c] = (sum gy, sum gy*xhat), then — after the
 * caller has all-reduced them — the apply pass with the GLOBAL pixel count Ptot. */
int acg_norm_bwd_sums(const float *dy, const float *y, const float *x, const float *mean, const float *rstd, float *sums,
                      int G, size_t P, int C, int act, void *workspace, size_t ws_bytes, void *stream);
int acg_norm_bwd_apply(const float *dy, const float *y, const float *x, const float *mean, const float *rstd,
                       const float *gamma, int gstride, const float *sums, float *dx, float *dres, int G, size_t P,
                       size_t Ptot, int C, int act, int unbiased, void *stream);

/* out = x where the sign-bitmask bit (acg_norm_apply layout) is set, else 0 — the materialised form of a masked skip
 * gradient, for the paths that cannot fuse it (n % 4 == 0) */
int acg_mask_apply(const float *x, const unsigned *sign_mask, float *out, size_t n, void *stream);
/* nn.Dropout(p), training mode (replaces modules.py:167-168, 214-215 `nn.Dropout(0.5)` behind the first ReLU of a residual
 * block): out = keep ? x * scale : 0, scale = 1 / (1 - p); the Bernoulli(1 - p) draw is GIVEN, one bit per element in the
 * sign-bitmask layout (bit e % 32 of word e / 32), so forward and backward (the same call on the gradient) share it and a
 * test can inject it.  n % 4 == 0. */
int acg_dropout_apply(const float *x, const unsigned *keep_bits, float scale, float *out, size_t n, void *stream);

/* ---- elementwise ---- */
int acg_act_bwd(const float *dy, const float *y, float *dx, size_t n, int act, void *stream); /* dx = dy*act'(y) */

/* ---- small dense layers: nn.Linear (networks.py:406-418), the 1x1 convs on the (N,nl,1,1) latent
 *      inside CondInstanceNorm (modules.py:111-118).  y[N][Op] = act(x[N][I] W[O][I]^T + b); columns
 *      O..Op-1 are written as zeros. ---- */
int acg_linear_fwd(const float *x, const float *w, const float *b, float *y, int N, int I, int ldx, int O, int Op,
                   int act, void *stream);
/* g = dy*act'(y) is applied inside; dx may be NULL; dw[O][I], db[O] are overwritten */
int acg_linear_bwd(const float *dy, const float *y, const float *x, const float *w, float *dx, float *dw, float *db,
                   int N, int I, int ldx, int O, int Op, int act, void *stream);
/* dst[s][i] (+)= src[off[s] + i] for i < len[s], every segment in ONE launch: hands the slices of a concatenated gradient
 * (the scale / shift layers of all CondInstanceNorms of a generator run as one dense layer, networks.py:98-132) to the
 * layers' own gradient tensors.  accumulate != 0 adds (the parameter's .grad), 0 overwrites. */
#define ACG_MAX_SEGMENTS 96
typedef struct acg_segments {
    void *dst[ACG_MAX_SEGMENTS];
    int off[ACG_MAX_SEGMENTS], len[ACG_MAX_SEGMENTS];
    int n;
} acg_segments;
int acg_segments_accumulate(const float *src, const acg_segments *segs, int accumulate, void *stream);

/* ---- DiscriminatorLatent (networks.py:396-433) fused: Linear(I->H) BatchNorm1d LeakyReLU(0.2), two more H->H stages, then
 *      Linear(H->1), BatchNorm in train mode (batch statistics; running buffers updated when non-NULL).  One launch per
 *      direction; the batch must fit one workgroup's LDS (acg_latent_mlp_supported), else use acg_linear_* + acg_norm_*.
 *      w[l] are torch nn.Linear weights [out][in]; z is [N][ldz] with I valid columns; out is [N][4] (column 0 = the
 *      prediction).  a_save [3][N][H] and stats_save [3][2][H] carry the pre-norm activations and (mean, rstd) to the
 *      backward pass.  Gradient pointers may be NULL; accumulate != 0 adds to them (parameter .grad). ---- */
typedef struct {
    const float *w[4], *b[4], *gamma[3], *beta[3];
    float *run_mean[3], *run_var[3];
} acg_latent_mlp_params;
typedef struct {
    float *dw[4], *db[4], *dgamma[3], *dbeta[3];
} acg_latent_mlp_grads;
int acg_latent_mlp_supported(int N, int I, int H);
int acg_latent_mlp_fwd(const acg_latent_mlp_params *params, const float *z, int ldz, int N, int I, int H, float eps,
                       float momentum, float *a_save, float *stats_save, float *out, void *stream);
int acg_latent_mlp_bwd(const acg_latent_mlp_params *params, const acg_latent_mlp_grads *grads, const float *z, int ldz, int N,
                       int I, int H, const float *a_save, const float *stats_save, const float *dout, float *dz, int accumulate,
                       void *stream);

/* ---- spatial mean over H*W of a C16 map -> [N][Cp] (LatentEncoder extension for S != 64,
 *      identity at the reference's 1x1 map — networks.py:482) ---- */
int acg_spatial_mean_fwd(const float *x, float *y, int N, size_t P, int Cp, void *stream);
int acg_spatial_mean_bwd(const float *dy, float *dx, int N, size_t P, int Cp, void *stream);

/* ---- losses: results are written to device scalars (no host sync) ----
 * tensors are C16; only channels < C count.  out[0] = mean((p-target)^2)  (model.py:65-70) */
size_t acg_reduce_workspace_bytes(size_t n);
int acg_mse_const_fwd(const float *p, size_t npix, int C, int Cp, float target, float *out, void *workspace,
                      size_t ws_bytes, void *stream);
int acg_mse_const_bwd(const float *p, size_t npix, int C, int Cp, float target, const float *gout, float *dp,
                      void *stream);
/* out[0] = mean(|a-b|) (F.l1_loss, model.py:391,468,486,494) */
int acg_l1_fwd(const float *a, const float *b, size_t npix, int C, int Cp, float *out, void *workspace,
               size_t ws_bytes, void *stream);
int acg_l1_bwd(const float *a, const float *b, size_t npix, int C, int Cp, const float *gout, float *da, float *db,
               void *stream);
/* out[0] = mean(x) over valid channels (P_t_A ... monitors, model.py:522-523) */
int acg_mean_fwd(const float *x, size_t npix, int C, int Cp, float *out, void *workspace, size_t ws_bytes,
                 void *stream);

/* ---- optimiser: torch.nn.utils.clip_grad_norm + torch.optim.Adam.step (model.py:447-452, 510-515)
 *      on one flat fp32 buffer per network. ---- */
int acg_sumsq(const float *g, size_t n, float *out, void *workspace, size_t ws_bytes, void *stream);
/* coef = min(1, max_norm/(sqrt(*sumsq)+1e-6)) read on device; p,m,v updated in place; g is NOT modified
 * unless scale_grads != 0 (then g *= coef, matching the reference's in-place clip). */
int acg_adam_step(float *p, float *g, float *m, float *v, size_t n, const float *sumsq, float max_norm, float lr,
                  float beta1, float beta2, float eps, int step, int scale_grads, void *stream);

/* The same for every network of an optimiser phase in three launches (multi-tensor clip + Adam: model.py:447-452 clips and
 * steps D_A, D_B (and D_z_B); model.py:510-515 G_A_B, G_B_A (and E_B)).  Per group: *sumsq = sum(g^2), then
 * coef = min(1, max_norm/(sqrt(*sumsq)+1e-6)), g *= coef, Adam on (p, m, v).  Bit-identical to acg_sumsq + acg_adam_step
 * (scale_grads = 1) per group.  The group array is host memory, read during the call.  step_dev (device, may be NULL): when
 * given, the step number of the bias corrections is *step_dev + 1, read by the kernel — for a launch recorded into a HIP
 * graph, whose arguments are fixed at capture (`step` is then ignored). */
#define ACG_ADAM_MAX_GROUPS 8
typedef struct {
    float *p, *g, *m, *v; /* device: parameters, gradients, first and second moments of one network, n floats each */
    size_t n;
    float *sumsq;         /* device: receives the network's gradient sum of squares (1 float) */
} acg_adam_group;
size_t acg_clip_adam_multi_workspace_bytes(int ngroups);
int acg_clip_adam_multi(const acg_adam_group *groups, int ngroups, float max_norm, float lr, float beta1, float beta2, float eps,
                        int step, const int *step_dev, void *workspace, size_t ws_bytes, void *stream);

/* ---- gradient exchange of the data-parallel step (replaces nn.parallel.data_parallel, networks.py:193-197 etc.): one
 *      process per GPU; rank 0 makes an id and ships its ACG_COMM_ID_BYTES to the other ranks by any side channel; every
 *      rank then joins with its current HIP device.  acg_comm_allreduce_mean averages a flat fp32 buffer (a network's
 *      gradients, model.py:447-449 / 510-512 clip AFTER it) in place across the ranks, enqueued on `stream` (ncclAvg over
 *      xGMI).  RCCL is bound at run time; the rest of the library does not need it. ---- */
#define ACG_COMM_ID_BYTES 128
int acg_comm_unique_id(void *id);
int acg_comm_init(void **comm, const void *id, int nranks, int rank);
int acg_comm_allreduce_mean(void *comm, float *buf, size_t n, void *stream);
int acg_comm_destroy(void *comm);

#ifdef __cplusplus
}
#endif
#endif /* ACGAN_HIP_H */
