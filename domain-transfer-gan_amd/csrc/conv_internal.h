// Internal geometry shared by the convolution kernels (not part of the C ABI).
#pragma once
#include "common.h"

#define ACG_KC 16 // K-channels per implicit-GEMM stage

struct Taps {
    int n;
    short dy[64];
    short dx[64];
    short w[64];
    // (dy & 0xff) | (dx & 0xff) << 8 | w << 16 per tap: ONE dword the kernels fetch with a scalar load per stage
    // (16-bit array elements go through vector memory and leave the weight-slab offset in a VGPR).  Filled by
    // acg_taps_pack() in the launchers.
    int pk[64];
    // row-patch stages of the generic bf16 tile (conv_bf16.hip, filled by its launcher): runs of up to 3 consecutive taps
    // with one dy and dx values within a span of 3.  gpk[16 ph + i] = first tap (relative to the phase's first) | taps << 8 |
    // (smallest dx & 0xff) << 16; ngrp = the group counts of the (up to four) phases, one byte each
    int gpk[64];
    int ngrp;
};

static inline Taps acg_taps_pack(const Taps &t)
{
    Taps r = t;
    for (int i = 0; i < 64; ++i)
        r.pk[i] = i < t.n ? ((t.dy[i] & 0xff) | ((t.dx[i] & 0xff) << 8) | ((int)t.w[i] << 16)) : 0;
    return r;
}

struct Geom {
    int Hin, Win, Cin;    // gathered tensor (K-channels = Cin, multiple of 16)
    int Hout, Wout, Cout; // written tensor; columns >= Cout are masked
    int GH, GW;           // grid of output positions per image covered by this launch
    int os, oy0, ox0;     // output pixel = (gy*os + oy0, gx*os + ox0)
    int is;               // gathered pixel = (gy*is + dy[t], gx*is + dx[t])
    int reflect;          // 1: mirror out-of-range coordinates (ReflectionPad2d), 0: zeros
    int act;
    int ncols_pad;        // packed-weight column count (multiple of the N tile)
    long long w_elems;    // element count of the packed weight array (bf16 modes: where the lo part starts)
    // Data gradient of a reflection-padded convolution, computed on the padded grid (Hout x Wout = fold_H+2p x fold_W+2p).
    // fold_p > 0 (wave-specialised kernel only): padded-grid pixels whose input pixel receives no mirrored contribution
    // are stored straight into out2 (N, fold_H, fold_W, Cout); only the frame (pad ring + the rows/columns it mirrors
    // onto) goes to `out`, and reflect_fold_frame_kernel sums that 3 % of the pixels afterwards.
    int fold_p = 0, fold_H = 0, fold_W = 0;
    float *out2 = nullptr;
    const float *addend = nullptr; // fold_p > 0 only: tensor of out2's shape added to the result (residual-path gradient)
    // optional sign bitmask of the addend (norm.hip layout: bit e%32 of word e/32 for float index e): only elements whose
    // bit is set are added — the skip gradient dy * (y > 0) of ReLU(x + IN(..)) without materialising it
    const unsigned *addend_mask = nullptr;
    // fold_p > 0 only: tensor of out2's shape whose SIGN masks the result (value > 0 ? keep : 0) before the addend — the
    // convolution's own input when that input is the ReLU output of the layer in front: dx is then already the gradient
    // w.r.t. that layer's pre-activation and its separate activation-backward pass (3 tensor streams) disappears
    const float *relu_src = nullptr;
    // pre-split ("S16") storage, conv_x3_pre.hip: per pixel and 8-channel group 16 bytes of bf16 hi then 16 bytes of bf16 lo
    // (x = hi + lo + O(2^-17 |x|); the same 4 bytes per element as fp32).  out_s16: the written tensor (out2 on the frame
    // path) takes that form; relu_s16: relu_src is stored that way (its sign = the sign of the hi halves).
    int out_s16 = 0, relu_s16 = 0;
    // out_s16 only.  mask_out: the launch also stores (output > 0) as one bit per element (norm.hip layout: bit e % 32 of word
    // e / 32 for float index e) — all a later data gradient needs of a conv + ReLU output; relu_mask: such a bitmask takes the
    // place of relu_src (1/32 of the bytes)
    unsigned *mask_out = nullptr;
    const unsigned *relu_mask = nullptr;
    // unpad (pre-split kernel only): data gradient of a ReflectionPad2d(1) 3x3 convolution computed on the UN-padded grid,
    // every tile one grid row.  The adjoint of the mirror folds pad row -1 onto row 1 and pad row H onto row H-2: for the
    // tiles of those two rows the kernel row that reads the mirrored dy row takes the packed slabs 9 + kw = w[0][kw] +
    // w[2][kw] instead of its own (acg_packed_wb_elems), and the column part of the fold — two pixels per grid row — is
    // a separate small GEMM (dgrad_colfix_kernel) whose result colfix[(image row * 2 + {left, right}) * Cout + c] is
    // added to pixels 1 and W-2 before the side inputs.  No padded scratch tensor, no fold pass.
    int unpad = 0;
    const float *colfix = nullptr;
    // unpad only (acg_norm_sums): the tile's share of the backward sums of the norm whose output gradient this launch writes
    const float *ns_x = nullptr, *ns_mean = nullptr, *ns_rstd = nullptr, *ns_gamma = nullptr, *ns_beta = nullptr;
    const unsigned *ns_mask = nullptr;
    float *ns_part = nullptr;
    int ns_gstride = 0, ns_act = 0;
    // per-tile statistics for an InstanceNorm behind the convolution (acg_conv2d_fwd_stats): (mean, M2) of every output
    // channel over the 128 output pixels of a tile, written to stats[((img * stats_cpi + stats_chunk0 + tile) * 2 + {0,1}) *
    // Cout + c], tile = the tile's index within its image in THIS launch (stats_cpi = chunks per image over all launches
    // that fill the tensor: the four sub-pixel phases of a ConvTranspose2d each add their own chunks)
    float *stats = nullptr;
    int stats_cpi = 0, stats_chunk0 = 0;
    // nphase == 4 (generic bf16 tile only): the four sub-pixel phases of a stride-2 data gradient / ConvTranspose2d forward
    // in ONE launch.  Phase ph = 2 py + px writes output pixels (2 gy + py, 2 gx + px) from the taps pk[16 ph .. 16 ph +
    // ph_ntaps[ph]); all phases share GH x GW (even output sizes).  Workgroup w handles phase w % 4 of tile w / 4, so the four
    // phases of a tile — which gather the same rows of the input — run side by side on one XCD and share its L2.
    int nphase = 0;
    int ph_ntaps = 0;     // tap counts of the four phases, one byte each (an array here would be indexed at run time in the
                          // kernel and drag the by-value argument structs into scratch memory: 2-3x slower, measured)
    int thin;             // 1: K flattened over (tap, 4 channels): stage s = taps 8s..8s+7, channels 0..3 of each
    int bk8;              // 8-float k-chunks per packed weight slab (Cin/8; thin: 4*ceil(ntaps/8), single slab)
    long long Mtot;       // N*GH*GW
};

int acg_igemm_launch(const float *in, const float *wp, const float *bias, float *out, const Geom &g, const Taps &t,
                     hipStream_t st);

// weight-gradient implicit GEMM: dw_part[split][tap][CiP][CoP] partial sums over pixel ranges
struct WGeom {
    int Hin, Win, Cin;    // gathered (conv-input side) tensor
    int Hg, Wg, Cg;       // gradient (conv-output side) tensor, one row per GEMM-K pixel
    int is;               // gathered pixel = (gy*is + dy[t], gx*is + dx[t]) for gradient pixel (gy, gx)
    int reflect;
    int CiP, CoP;         // padded dims of the partial buffer
    int nsplit;
    long long Mtot;       // N*Hg*Wg
    long long m_per_split;
    int thin;             // 1: gathered operand columns = (tap, 4 channels) flattened, 8 taps per 32-column tile
    int bias_from;        // 0: none, 1: column sums of the gradient-side operand, 2: of the gathered-side operand
    float *bias_part;     // [nsplit][CoP or CiP] partial column sums (written by the tap-0 / tile-0 blocks)
};
int acg_wgrad_launch(const float *x, const float *dy, float *part, const WGeom &g, const Taps &t, hipStream_t st);
void acg_wgrad_tiles(int Ci, int Co, int *bci, int *bco, int ntaps);
int acg_wgrad_taps_per_wg(int Ci, int Co, int ntaps, int thin);
// conv_wgrad_k4.hip: kernel-row weight gradient of the zero-padded 4x4 layers (D_B) and the stride-2 3x3 pair of the generators
bool acg_wgrad_krowg_shape_ok(int K, int stride, int pad, int reflect, int Wi, int Wo, int Cx, int Cg);
bool acg_wgrad_krowg_ok(const WGeom &g, const Taps &t);
int acg_wgrad_krowg_launch(const float *x, const float *dy, float *part, const WGeom &g, const Taps &t, hipStream_t st);

extern int g_acg_precision;
extern int g_acg_conv_impl;
int acg_igemm_x3_ws_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &t,
                           long long n_w_elems, hipStream_t st, float *stats = nullptr);
bool acg_igemm_uses_ws(const Geom &g);
// conv_x3_pre.hip: the same tile on a pre-split (S16) gathered tensor, both operands by LDS-DMA
bool acg_igemm_x3_pre_ok(const Geom &g, const Taps &t);
int acg_igemm_x3_pre_launch(const void *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &t,
                            long long n_w_elems, hipStream_t st, float *stats = nullptr);
// conv_x3_pp.hip: its persistent form (one workgroup per CU walks over its tiles; the epilogue of a tile is drained by two
// dedicated waves during the next tile's loop)
bool acg_igemm_x3_pp_ok(const Geom &g, const Taps &t);
int acg_igemm_x3_pp_launch(const void *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &t,
                           long long n_w_elems, hipStream_t st, float *stats = nullptr);
bool acg_conv_patch16_ok(const Geom &g, const Taps &t);
int acg_conv_patch16_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &t,
                            long long n_w_elems, hipStream_t st);
// conv_rows.hip: persistent row pipeline for the full-resolution 3x3 stride-1 32 -> 64 channel layers
bool acg_conv_rows_ok(const Geom &g, const Taps &t);
int acg_conv_rows_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &t,
                         long long n_w_elems, hipStream_t st);
bool acg_conv_patchn_ok(const Geom &g, const Taps &t);   // ... with C4 output and N-packed weights (conv_patchn_x3)
int acg_conv_patchn_launch(const float *in, const void *wn, const float *bias, float *out, const Geom &g, const Taps &t, hipStream_t st);
bool acg_conv_thinrow_ok(const Geom &g, const Taps &t);  // C4 gathered tensor -> 32 channels, one K step per kernel row (conv_thinrow_x3)
int acg_conv_thinrow_launch(const float *in, const void *wr, const float *bias, float *out, const Geom &g, const Taps &t, hipStream_t st);
// conv_ph4.hip: stride-2 data gradient / ConvTranspose2d forward with the four sub-pixel phases in one tile (64 columns)
bool acg_igemm_ph4_ok(const Geom &g);
bool acg_ph4_plan(const Taps &t, const int nt[4], Taps *out);
int acg_igemm_ph4_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &plan,
                         long long n_w_elems, hipStream_t st);
int acg_igemm_bf16_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &t,
                          int bn, long long n_w_elems, hipStream_t st);
int acg_wgrad_bf16_launch(const float *x, const float *dy, float *part, const WGeom &g, const Taps &t, int bci, int bco,
                          hipStream_t st);
// conv_wgrad_tr.hip: one kernel row per workgroup (stride-1 3x3, 128-multiple channels, bf16x3)
bool acg_wgrad_krow_ok(const WGeom &g, const Taps &t);
int acg_wgrad_krow_launch(const float *x, const float *dy, float *part, const WGeom &g, hipStream_t st);
int acg_wgrad_krow_s16_launch(const void *x, const void *dy, float *part, const WGeom &g, hipStream_t st); // pre-split operands
bool acg_wgrad_krow_s_ok(const WGeom &g, const Taps &t);   // its 32 <-> 64 channel, 128-pixel-run variant
// conv_wgrad_thin.hip: the 7x7 image layers (C4 tensor on one side, 32 channels on the other), persistent over 8 x 16 tiles
bool acg_wgrad_thin_patch_ok(const WGeom &g, const Taps &t, int *K, int *flip);
int acg_wgrad_thin_patch_tiles(const WGeom &g);
int acg_wgrad_thin_patch_launch(const float *thin, const float *wide, float *part, const WGeom &g, const Taps &t, hipStream_t st);
int acg_wgrad_krow_s_launch(const float *x, const float *dy, float *part, const WGeom &g, hipStream_t st);

#ifdef __HIPCC__
// bf16x3 operand split of 8 fp32 values: hi = RNE bf16(x), lo = RNE bf16(x - hi), each returned as 8 packed bf16
// (x = hi + lo + O(2^-17 |x|)).  Written pairwise so every conversion is one v_cvt_pk_bf16_f32 and the residuals one
// v_pk_add_f32: 20 VALU instructions per 8 values (a whole-vector __builtin_convertvector on gathered scalars
// compiles to one conversion per element).
typedef unsigned acg_u32x4 __attribute__((ext_vector_type(4)));
// byte offset of a buffer load, or 0xFFFFFFFF (range-checked by the buffer resource -> zeros) for a masked lane.
// Bit arithmetic on purpose: with `ok ? off : ~0u` LLVM sinks the load into both arms of a branch and separates the two
// copies with s_waitcnt vmcnt(0) (same destination registers), so every load of a stage pays a full memory latency —
// measured on the weight-gradient loaders: 0.89 -> 0.72 ms.
__device__ __forceinline__ unsigned acg_masked_off(unsigned off, bool ok)
{
    const unsigned keep = 0u - (unsigned)ok;
    return (off & keep) | ~keep;
}
__device__ __forceinline__ void acg_split8(const float (&v)[8], acg_u32x4 &hi, acg_u32x4 &lo)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2 * q], v[2 * q + 1]}, bf16x2_t));
        const float ha = __builtin_bit_cast(float, h << 16), hb = __builtin_bit_cast(float, h & 0xffff0000u);
        hi[q] = h;
        lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2 * q] - ha, v[2 * q + 1] - hb}, bf16x2_t));
    }
}
__device__ __forceinline__ acg_u32x4 acg_round8(const float (&v)[8]) // bf16 mode: the hi part only
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    acg_u32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        r[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2 * q], v[2 * q + 1]}, bf16x2_t));
    return r;
}
#endif
