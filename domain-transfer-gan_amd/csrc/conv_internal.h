// Internal geometry shared by the convolution kernels (not part of the C ABI).
#pragma once
#include "common.h"

#define ACG_KC 16 // K-channels per implicit-GEMM stage

struct Taps {
    int n;
    short dy[64];
    short dx[64];
    short w[64];
};

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
void acg_wgrad_tiles(int Ci, int Co, int *bci, int *bco);

extern int g_acg_precision;
int acg_igemm_bf16_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &t,
                          int bn, long long n_w_elems, hipStream_t st);
int acg_wgrad_bf16_launch(const float *x, const float *dy, float *part, const WGeom &g, const Taps &t, int bci, int bco,
                          hipStream_t st);
