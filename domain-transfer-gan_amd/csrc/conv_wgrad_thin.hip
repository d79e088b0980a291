// Weight gradient of the 7x7 image layers (bf16x3 arithmetic): one side of the layer is an image tensor stored C4, the other
// has 32 channels — the nc -> 32 stem (networks.py:159-160; the C4 tensor is the gathered input x) and, mirrored, the
// 32 -> nc head (networks.py:187-188; the C4 tensor is the gradient dy, gathered at p - tap + pad).
//
//   part[(tap, c)][w] = sum over pixels p of  thin[p + window(tap)][c] * wide[p][w]          c < 4, w < 32
//
// The per-tap kernel (conv_wgrad.hip, thin mode) gathers the thin operand from global memory once per (pixel, tap) — 49 x
// 16 bytes per pixel — and re-reads the wide operand once per 8-tap group: 0.3 ms for 0.3 GB of tensors.  Here a persistent
// workgroup walks 8 x 16 pixel tiles; per tile the thin PATCH (14 x 24 pixels, 5 KB as bf16 hi + lo) and the wide tile
// (128 pixels x 32 channels, 16 KB) are split once into pixel-major LDS images, and — K being the PIXEL dimension — both MFMA
// operands come out of them through ds_read_b64_tr_b16 (conv_wgrad_tr.hip has the lane map): with 8 bytes per patch pixel the
// 32 rows of an A tile, (window column kw, channel c), kw = 0 .. 7, are 64 contiguous bytes behind the row's pixel, so one
// 32 x 32 accumulator per kernel ROW holds all of its taps.  The four waves split the kernel rows (2, 2, 2, 1 + the bias
// column sums); a workgroup writes ONE partial slab when its tiles are done, the split-K reduction is wgrad_reduce_kernel's.
#include "common.h"
#include "conv_internal.h"
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int TH = 8, TW = 16;                 // pixel tile
constexpr int PW = TW + 8, PHMAX = TH + 6;     // patch: 8 window columns (the eighth has no tap), up to 7 kernel rows
constexpr int TPLANE = PHMAX * PW * 8;         // bytes of one hi (or lo) patch image: 8 bytes per pixel
constexpr int WPLANE = TH * TW * 64;           // bytes of one hi (or lo) wide image: 64 bytes per pixel
constexpr int LDSB = 2 * TPLANE + 2 * WPLANE;  // 21.8 KB
typedef __attribute__((address_space(3))) char lds_char;

template <int STEP>   // STEP: bytes between pixel rows of the image
__device__ __forceinline__ bf16x8 tr_frag(const lds_char *p)
{
    // K elements (pixels) 0..3 from the block at p, 4..7 from the block four pixel rows below
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 4 * STEP));
    const s16x8 v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void split4(const f32x4 &v, uint2 &hi, uint2 &lo)   // the arithmetic of acg_split8
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    unsigned *hp = &hi.x, *lp = &lo.x;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2 * q], v[2 * q + 1]}, bf16x2_t));
        const float ha = __builtin_bit_cast(float, h << 16), hb = __builtin_bit_cast(float, h & 0xffff0000u);
        hp[q] = h;
        lp[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2 * q] - ha, v[2 * q + 1] - hb}, bf16x2_t));
    }
}
}

// thin: the C4 tensor (N, Hin, Win, 4), gathered at (y + ry + dymin, x + kw + dxmin); wide: (N, Hg, Wg, 32), one row per pixel.
// flip: window position (ry, kw) is tap (K-1-ry) K + (K-1-kw) of the partial layout instead of ry K + kw (mirrored case).
__global__ __launch_bounds__(256) void wgrad_thin_patch_x3(const float *__restrict__ thin, const float *__restrict__ wide,
                                                           float *__restrict__ part, WGeom g, int K, int dymin, int dxmin,
                                                           int flip, int ntiles, unsigned thin_bytes, unsigned wide_bytes)
{
    __shared__ __attribute__((aligned(16))) char lds[LDSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (g.Wg + TW - 1) / TW, tiles_y = (g.Hg + TH - 1) / TH;
    const int PH = TH + K - 1;
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void *)thin, 0, thin_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wide, 0, wide_bytes, 0x00020000);

    // this wave's kernel rows: 2 w, 2 w + 1 (the last wave: row 6 and, when the bias gradient is asked for, the column sums)
    const int ry0 = 2 * wave;
    const bool bias_wave = wave == 3 && g.bias_from == 1;
    f32x16 acc[2], accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; accb[r] = 0.f; }
    // A operand of the column sums: row 0 = ones (lane l holds row l & 31 of the tile)
    const short one = (lane & 31) == 0 ? (short)0x3F80 : (short)0;   // bf16 1.0
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (s16x8){one, one, one, one, one, one, one, one});

    // transposed-read lane parts (group gq = lane >> 4 reads K rows 8 (gq >> 1) + q, columns 16 (gq & 1) + 4 p ..)
    const int gq = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    const int a_lane = (8 * (gq >> 1) + q + 4 * (gq & 1) + p) * 8;                  // thin patch: pixel x + window column
    const int b_lane = (8 * (gq >> 1) + q) * 64 + (16 * (gq & 1) + 4 * p) * 2;      // wide tile: pixel row, channel

    f32x4 pw[4], pt[2];
    auto load_tile = [&](int tile) {
        int b = tile;
        const int tx = b % tiles_x; b /= tiles_x;
        const int ty = b % tiles_y;
        const int n = b / tiles_y;
        const int gy0 = ty * TH, gx0 = tx * TW;
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // wide tile: 8 lanes per pixel, 16 bytes each
            const int pix = (tid >> 3) + 32 * k, c4 = tid & 7;
            const int y = gy0 + pix / TW, x = gx0 + pix % TW;
            const bool ok = tile < ntiles && y < g.Hg && x < g.Wg;
            const unsigned off = (unsigned)(((n * g.Hg + y) * g.Wg + x) * 32 + c4 * 4) * 4u;
            pw[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, acg_masked_off(off, ok), 0, 0));
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {   // thin patch: one pixel per lane
            const int pp = tid + 256 * it;
            const int py = pp / PW, px = pp - py * PW;
            int iy = gy0 + py + dymin, ix = gx0 + px + dxmin;
            bool ok = tile < ntiles && py < PH;
            if (g.reflect) {
                ok = ok && iy > -g.Hin && iy < 2 * g.Hin - 1 && ix > -g.Win && ix < 2 * g.Win - 1;
                iy = iy < 0 ? -iy : iy;
                iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                ix = ix < 0 ? -ix : ix;
                ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
            } else {
                ok = ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
            }
            const unsigned off = (unsigned)(((n * g.Hin + iy) * g.Win + ix) * 4) * 4u;
            pt[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, acg_masked_off(off, ok), 0, 0));
        }
    };

    int tile = blockIdx.x;
    load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();   // the previous tile's fragments are read
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pix = (tid >> 3) + 32 * k, c4 = tid & 7;
            uint2 hi, lo;
            split4(pw[k], hi, lo);
            *(uint2 *)(lds + 2 * TPLANE + pix * 64 + c4 * 8) = hi;
            *(uint2 *)(lds + 2 * TPLANE + WPLANE + pix * 64 + c4 * 8) = lo;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int pp = tid + 256 * it;
            if (pp < PHMAX * PW) {
                uint2 hi, lo;
                split4(pt[it], hi, lo);
                *(uint2 *)(lds + pp * 8) = hi;
                *(uint2 *)(lds + TPLANE + pp * 8) = lo;
            }
        }
        __syncthreads();
        load_tile(tile + gridDim.x);   // in flight under the MFMAs below (masked past the last tile)
        const lds_char *tb = (const lds_char *)lds, *wb = (const lds_char *)lds + 2 * TPLANE;
#pragma unroll 2
        for (int ks = 0; ks < TH; ++ks) {   // K step = one tile row of 16 pixels
            const bf16x8 bh = tr_frag<64>(wb + ks * TW * 64 + b_lane);
            const bf16x8 bl = tr_frag<64>(wb + WPLANE + ks * TW * 64 + b_lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (ry0 + j < K) {   // (wave-uniform)
                    const lds_char *ap = tb + (ks + ry0 + j) * PW * 8 + a_lane;
                    const bf16x8 ah = tr_frag<8>(ap), al = tr_frag<8>(ap + TPLANE);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
                }
            }
            if (bias_wave) {
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bl, accb, 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bh, accb, 0, 0, 0);
            }
        }
    }

    // ---- this workgroup's partial slab: rows (tap, c), 32 columns; accumulator register r of lane l = row (r & 3) + 8 (r >> 2) +
    // 4 (l >> 5) = (kw, c) of the kernel row, column l & 31
    float *o = part + (long long)blockIdx.x * g.CiP * g.CoP;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ry = ry0 + j;
        if (ry >= K) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), kw = m >> 2, c = m & 3;
            if (kw < K) {
                const int tap = flip ? (K - 1 - ry) * K + (K - 1 - kw) : ry * K + kw;
                o[(long long)(tap * 4 + c) * g.CoP + (lane & 31)] = acc[j][r];
            }
        }
    }
    if (bias_wave && lane < 32) g.bias_part[(long long)blockIdx.x * g.CoP + lane] = accb[0];
}

// K x K window (K <= 7), stride 1, thin tensor C4, wide tensor 32 channels; the tap list in kernel order (forward: dy = kh -
// pad; mirrored: dy = pad - kh)
bool acg_wgrad_thin_patch_ok(const WGeom &g, const Taps &t, int *Kout, int *flip)
{
    static const bool off = acg_debug_switch("ACG_NO_WGRAD_THIN_PATCH"); // A/B switch
    if (off || g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA || !g.thin || g.is != 1) return false;
    if (g.Cin != 4 || g.Cg != 32 || g.CoP != 32 || g.Hin != g.Hg || g.Win != g.Wg || g.bias_from == 2) return false;
    int K = 1;
    while (K * K < t.n) ++K;
    if (K * K != t.n || K < 2 || K > 7 || g.CiP < 4 * K * K) return false;
    bool fwd = true, bwd = true;
    for (int i = 0; i < t.n; ++i) {
        const int kh = i / K, kw = i - kh * K;
        fwd = fwd && t.dy[i] - t.dy[0] == kh && t.dx[i] - t.dx[0] == kw;
        bwd = bwd && t.dy[0] - t.dy[i] == kh && t.dx[0] - t.dx[i] == kw;
    }
    if (!fwd && !bwd) return false;
    *Kout = K; *flip = fwd ? 0 : 1;
    return true;
}

int acg_wgrad_thin_patch_tiles(const WGeom &g)
{
    const long long nimg = g.Mtot / ((long long)g.Hg * g.Wg);
    return (int)(nimg * ((g.Hg + TH - 1) / TH) * ((g.Wg + TW - 1) / TW));
}

int acg_wgrad_thin_patch_launch(const float *thin, const float *wide, float *part, const WGeom &g, const Taps &t, hipStream_t st)
{
    int K, flip;
    ACG_REQUIRE(acg_wgrad_thin_patch_ok(g, t, &K, &flip), "wgrad_thin_patch_x3: unsupported geometry");
    int ymin = t.dy[0], xmin = t.dx[0];
    for (int i = 1; i < t.n; ++i) { ymin = t.dy[i] < ymin ? t.dy[i] : ymin; xmin = t.dx[i] < xmin ? t.dx[i] : xmin; }
    const long long nimg = g.Mtot / ((long long)g.Hg * g.Wg);
    const long long tb = nimg * g.Hin * g.Win * 4 * 4, wbts = g.Mtot * 32 * 4;
    ACG_REQUIRE(tb < (1LL << 32) && wbts < (1LL << 32), "wgrad_thin_patch_x3: operand exceeds the 4 GiB buffer-addressing limit");
    const int ntiles = acg_wgrad_thin_patch_tiles(g);
    ACG_REQUIRE(g.nsplit >= 1 && g.nsplit <= ntiles, "wgrad_thin_patch_x3: split plan (%d workgroups for %d tiles)", g.nsplit, ntiles);
    hipLaunchKernelGGL(wgrad_thin_patch_x3, dim3(g.nsplit), dim3(256), 0, st, thin, wide, part, g, K, ymin, xmin, flip, ntiles,
                       (unsigned)tb, (unsigned)wbts);
    ACG_CHECK_LAUNCH("wgrad_thin_patch_x3");
    acg_note_kernel("wgrad_thin_patch_x3<K=%d,flip=%d>", K, flip);
    return ACG_OK;
}
