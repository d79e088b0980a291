// Weight gradient of the stride-1 3x3 convolutions (bf16x3 arithmetic), one KERNEL ROW per workgroup.
//
//   dW[ky][kx][ci][co] = sum over output pixels m=(n,oy,ox) of  x[n, oy+ky-1, ox+kx-1][ci] * dy[m][co]
//
// The per-tap kernel (conv_bf16.hip, wgrad_bf16) gathers an x tile and a dy tile per tap and transposes both into
// [channel][8 pixels] LDS rows in registers; its counters and ablations show three pipes near-critical at
// once — the L1/TA path (64 KB of loads per 64-pixel stage), the VGPR->LDS store path (the same 64 KB again) and the
// matrix pipe — and a time equal to the SUM of the matrix time and the load/store time.  Here the three taps of a kernel
// row share ONE dy tile and ONE x window (32 consecutive output pixels of an image row need input pixels ox-1 .. ox+32):
//   * the LDS images stay PIXEL-major, [pixel][128 channels] bf16 with 320-byte rows, exactly as the NHWC tensors deliver
//     them (a thread converts 8 channels of one pixel: two ds_write_b128, no register transpose);
//   * the K-contiguous MFMA operands are read with ds_read_b64_tr_b16 (the hardware transposes 4 pixels x 16 channels
//     per 16-lane group; rows 320 B apart put the four pixels of a block on distinct 64-byte bank segments);
//   * a tap shift kx is a ROW offset of the x image, so the three taps read the same window at rows kx + k.
// Per 32-pixel stage a workgroup loads 17 + 16 KB for 3 x 2 x 128 x 128 x 32 MACs — a third of the per-tap kernel's loads,
// splits and LDS stores per MAC.  8 waves as 4 (ci) x 2 (co), wave tile 32 x 64 per tap, v_mfma_f32_32x32x16_bf16,
// 96 accumulator registers; two stage buffers, one barrier per stage.
#include "common.h"
#include "conv_internal.h"
#include <cstdlib>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int KP = 32;                 // output pixels per stage (one run inside an image row)
constexpr int XW = KP + 2;             // x window: the run with one halo pixel on each side
constexpr int BC = 128;                // channel tile on both sides
constexpr int PITCH = 320;             // bytes per pixel row: 128 bf16 + 64 B (rows 0..3 of a transposed block -> 4 bank segments)
constexpr int XIMG = XW * PITCH, DIMG = KP * PITCH;    // one hi (or lo) image
constexpr int BUF = 2 * XIMG + 2 * DIMG;               // [x hi][x lo][dy hi][dy lo]
typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ bf16x8 tr_frag(const lds_char *p)
{
    // K elements 0..3 from the block at p, 4..7 from the block four pixel rows below
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 4 * PITCH));
    const s16x8 v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
}

__global__ __launch_bounds__(512) void wgrad_x3_krow(const float *__restrict__ x, const float *__restrict__ dy,
                                                     float *__restrict__ part, WGeom g, unsigned x_bytes, unsigned d_bytes)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int tiles_ci = g.CiP / BC, tiles_co = g.CoP / BC;
    // XCD-aware order (bijective, any grid size): workgroups are dealt to the 8 XCDs round-robin, so the three kernel rows
    // of one pixel range — which stream the same dy rows and overlapping x rows — are made neighbours on ONE XCD and share
    // its L2 (dealt across three XCDs each of them pulled its own copy from HBM: 1.6 GB per launch)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    int b = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tco = b % tiles_co; b /= tiles_co;
    const int tci = b % tiles_ci; b /= tiles_ci;
    const int ky = b % 3;
    const int split = b / 3;
    const int ci0 = tci * BC, co0 = tco * BC;
    const long long mbeg = (long long)split * g.m_per_split;
    long long mend = mbeg + g.m_per_split;
    if (mend > g.Mtot) mend = g.Mtot;
    const int nst = mbeg < mend ? (int)((mend - mbeg) / KP) : 0;
    const int H = g.Hg, W = g.Wg;

    f32x16 acc[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd_ = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, d_bytes, 0x00020000);

    // position of the NEXT stage to load (wave-uniform), advanced by one run of KP pixels per stage
    int ln, loy, lox;
    {
        const long long mm = mbeg < g.Mtot ? mbeg : 0;
        ln = (int)(mm / ((long long)H * W));
        const int rr = (int)(mm - (long long)ln * H * W);
        loy = rr / W;
        lox = rr - loy * W;
    }
    long long lm = mbeg;
    // this thread's units: 8 channels of pixel pj of the x window (threads 0..31: also of pixel 32 + pj) and of the dy run
    const int c8 = tid & 15, pj = tid >> 4;
    const bool extra = tid < 2 * 16;
    u32x4 rxa[2], rxb[2], rda[2];
    const bool do_bias = g.bias_from == 1 && ky == 0 && tci == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto load_stage = [&]() {
        int iy = loy + ky - 1;
        bool rowok = true;
        if (g.reflect) {
            iy = iy < 0 ? -iy : iy;
            iy = iy >= H ? 2 * (H - 1) - iy : iy;
        } else {
            rowok = (unsigned)iy < (unsigned)H;
        }
        const int rowbase = (ln * H + iy) * W;
        auto xoff = [&](int j) {
            int ix = lox - 1 + j;
            bool ok = rowok;
            if (g.reflect) {
                ix = ix < 0 ? -ix : ix;
                ix = ix >= W ? 2 * (W - 1) - ix : ix;
            } else {
                ok = ok && (unsigned)ix < (unsigned)W;
            }
            return acg_masked_off((unsigned)((rowbase + ix) * g.Cin + ci0 + 8 * c8) * 4u, ok);
        };
        const unsigned o0 = xoff(pj);
        rxa[0] = __builtin_amdgcn_raw_buffer_load_b128(rx_, o0, 0, 0);
        rxa[1] = __builtin_amdgcn_raw_buffer_load_b128(rx_, o0, 16, 0);
        if (wave == 0) { // the two halo pixels past the run: 32 units, the first half of wave 0 (the other lanes masked)
            const unsigned o1 = acg_masked_off(xoff(KP + pj), extra);
            rxb[0] = __builtin_amdgcn_raw_buffer_load_b128(rx_, o1, 0, 0);
            rxb[1] = __builtin_amdgcn_raw_buffer_load_b128(rx_, o1, 16, 0);
        }
        const unsigned od = (unsigned)(((int)lm + pj) * g.Cg + co0 + 8 * c8) * 4u;
        rda[0] = __builtin_amdgcn_raw_buffer_load_b128(rd_, od, 0, 0);
        rda[1] = __builtin_amdgcn_raw_buffer_load_b128(rd_, od, 16, 0);
        lm += KP;
        lox += KP;
        if (lox == W) { lox = 0; if (++loy == H) { loy = 0; ++ln; } }
    };
    auto put = [&](char *img_hi, char *img_lo, int row, const u32x4 (&r)[2]) {
        const f32x4 a = __builtin_bit_cast(f32x4, r[0]), c = __builtin_bit_cast(f32x4, r[1]);
        const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        acg_u32x4 hi, lo;
        acg_split8(v, hi, lo);
        *(acg_u32x4 *)(img_hi + row * PITCH + c8 * 16) = hi;
        *(acg_u32x4 *)(img_lo + row * PITCH + c8 * 16) = lo;
    };
    auto store_stage = [&](int buf) {
        char *base = lds + buf * BUF;
        put(base, base + XIMG, pj, rxa);
        if (extra) put(base, base + XIMG, KP + pj, rxb);
        put(base + 2 * XIMG, base + 2 * XIMG + DIMG, pj, rda);
        if (do_bias) {
            const f32x4 a = __builtin_bit_cast(f32x4, rda[0]), c = __builtin_bit_cast(f32x4, rda[1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bsum[e] += a[e]; bsum[4 + e] += c[e]; }
        }
    };

    // transposed-read lane map (tools/probes/tr_read.hip): lane 4q+p of 16-lane group gq supplies pixel row 8*(gq>>1) + q,
    // channels 16*(gq&1) + 4p .. 4p+3, and receives channel 16*(gq&1) + (lane&15) of the four rows: the 32x32x16 operand
    // (row lane&31, K group lane>>5)
    const int gq = lane >> 4, li = lane & 15;
    const int frag_row = 8 * (gq >> 1) + (li >> 2), frag_col = 16 * (gq & 1) + 4 * (li & 3);
    const int xlane = frag_row * PITCH + (wi * 32 + frag_col) * 2;
    const int dlane = frag_row * PITCH + (wj * 64 + frag_col) * 2;

    if (nst > 0) {
        load_stage();
        store_stage(0);
        if (nst > 1) load_stage();
    }
    __syncthreads();
#ifndef ACG_KROW_LATE
#define ACG_KROW_LATE 1
#endif
    // waves 4-7 (the second wave of every SIMD) convert and store the next stage AFTER their MFMAs, waves 0-3 before: one
    // wave of a SIMD is in its VALU/LDS-store phase while its partner feeds the matrix pipe
    const bool late = ACG_KROW_LATE && wave >= 4;
    for (int s = 0; s < nst; ++s) {
        const int cur = s & 1;
        if (!late && s + 1 < nst) {
            store_stage(cur ^ 1);
            if (s + 2 < nst) load_stage();
        }
        const lds_char *xb = (const lds_char *)(lds + cur * BUF) + xlane;
        const lds_char *db = (const lds_char *)(lds + cur * BUF + 2 * XIMG) + dlane;
#pragma unroll
        for (int ks = 0; ks < KP / 16; ++ks) {
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = tr_frag(db + ks * 16 * PITCH + j * 64);
                bl[j] = tr_frag(db + DIMG + ks * 16 * PITCH + j * 64);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const bf16x8 ah = tr_frag(xb + (ks * 16 + t) * PITCH);
                const bf16x8 al = tr_frag(xb + XIMG + (ks * 16 + t) * PITCH);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[t][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[t][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[t][j], 0, 0, 0);
            }
        }
        if (late && s + 1 < nst) {
            store_stage(cur ^ 1);
            if (s + 2 < nst) load_stage();
        }

        __syncthreads();
    }

    if (do_bias) { // the 32 threads that share a channel group fold their fp32 column sums in fixed order through LDS
        float *red = (float *)lds;
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(pj * 16 + c8) * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < BC) {
            float s = 0.f;
            for (int r = 0; r < KP; ++r) s += red[(r * 16 + (tid >> 3)) * 8 + (tid & 7)];
            g.bias_part[(long long)split * g.CoP + co0 + tid] = s;
        }
    }

#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float *o = part + ((long long)split * 9 + ky * 3 + t) * g.CiP * g.CoP;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int co = co0 + wj * 64 + j * 32 + (lane & 31);
                o[(long long)ci * g.CoP + co] = acc[t][j][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same scheme for the full-resolution 32 <-> 64 channel layers (networks.py:164, 183: 3x3 at 256 x 256).  Their whole
// (Cin x Cout) tile is one or two 32x32 MFMA tiles, so the waves of a workgroup cannot divide channels: they divide the
// PIXELS of a 128-pixel run instead (wave w: pixels 32w .. 32w+31, all three taps of the kernel row, the whole channel
// tile) and add their accumulators through LDS once at the end.  These layers are HBM-bound (805 MB of x and dy per
// launch against 77 GFLOP): the per-tap kernel pulled both tensors through L2 nine times (0.8 ms); here the three kernel
// rows of a pixel range share an XCD and each reads them once.  One stage buffer, two barriers per stage, next stage's
// loads in flight under the MFMAs; 66 KB of LDS: two workgroups per CU.
namespace {
constexpr int SKP = 128;                                      // pixels per stage
constexpr int SXW = SKP + 2;
constexpr int spitch(int C) { return C * 2 + ((C * 2) % 128 == 0 ? 64 : 0); } // 32 ch: 64 B, 64 ch: 192 B (4 rows -> 4 segments)
}

template <int BCI, int BCO>
__global__ __launch_bounds__(256) void wgrad_x3_krow_s(const float *__restrict__ x, const float *__restrict__ dy,
                                                       float *__restrict__ part, WGeom g, unsigned x_bytes, unsigned d_bytes)
{
    constexpr int PX = spitch(BCI), PD = spitch(BCO);
    constexpr int XI = SXW * PX, DI = SKP * PD;                 // one hi (or lo) image
    constexpr int UX = BCI / 8, UD = BCO / 8;                   // 8-channel units per pixel
    constexpr int XS = SKP * UX / 256, DS = SKP * UD / 256;     // unit slots per thread (run pixels; the 2 halo pixels go extra)
    constexpr int TI = BCI / 32, TJ = BCO / 32;
    static_assert(TI * TJ == 2 && XS >= 1 && DS >= 1, "tile config");
    constexpr int LDSB = 2 * XI + 2 * DI;
    constexpr int REDB = 3 * TI * TJ * 16 * 64 * 4;             // one tap's accumulators of three waves
    __shared__ __attribute__((aligned(16))) char lds[LDSB > REDB ? LDSB : REDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int b = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int ky = b % 3, split = b / 3;
    const long long mbeg = (long long)split * g.m_per_split;
    long long mend = mbeg + g.m_per_split;
    if (mend > g.Mtot) mend = g.Mtot;
    const int nst = mbeg < mend ? (int)((mend - mbeg) / SKP) : 0;
    const int H = g.Hg, W = g.Wg;

    f32x16 acc[3][TI][TJ];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd_ = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, d_bytes, 0x00020000);

    int ln, loy, lox;
    {
        const long long mm = mbeg < g.Mtot ? mbeg : 0;
        ln = (int)(mm / ((long long)H * W));
        const int rr = (int)(mm - (long long)ln * H * W);
        loy = rr / W;
        lox = rr - loy * W;
    }
    long long lm = mbeg;
    u32x4 rx[XS][2], rh[2], rd[DS][2];
    const bool halo = tid < 2 * UX;                              // unit (pixel -1 or SKP, channel group tid % UX)
    const bool do_bias = g.bias_from == 1 && ky == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto load_stage = [&]() {
        int iy = loy + ky - 1;
        bool rowok = true;
        if (g.reflect) {
            iy = iy < 0 ? -iy : iy;
            iy = iy >= H ? 2 * (H - 1) - iy : iy;
        } else {
            rowok = (unsigned)iy < (unsigned)H;
        }
        const int rowbase = (ln * H + iy) * W;
#pragma unroll
        for (int k = 0; k < XS; ++k) { // run pixels lox .. lox+127: always inside the row
            const int u = tid + 256 * k;
            const unsigned off = acg_masked_off((unsigned)((rowbase + lox + u / UX) * g.Cin + 8 * (u % UX)) * 4u, rowok);
            rx[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rx_, off, 0, 0);
            rx[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rx_, off, 16, 0);
        }
        if (wave == 0) { // the two halo pixels
            int ix = tid < UX ? lox - 1 : lox + SKP;
            bool ok = rowok && halo;
            if (g.reflect) {
                ix = ix < 0 ? -ix : ix;
                ix = ix >= W ? 2 * (W - 1) - ix : ix;
            } else {
                ok = ok && (unsigned)ix < (unsigned)W;
            }
            const unsigned off = acg_masked_off((unsigned)((rowbase + ix) * g.Cin + 8 * (tid % UX)) * 4u, ok);
            rh[0] = __builtin_amdgcn_raw_buffer_load_b128(rx_, off, 0, 0);
            rh[1] = __builtin_amdgcn_raw_buffer_load_b128(rx_, off, 16, 0);
        }
#pragma unroll
        for (int k = 0; k < DS; ++k) {
            const int u = tid + 256 * k;
            const unsigned off = (unsigned)(((int)lm + u / UD) * g.Cg + 8 * (u % UD)) * 4u;
            rd[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rd_, off, 0, 0);
            rd[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rd_, off, 16, 0);
        }
        lm += SKP;
        lox += SKP;
        if (lox == W) { lox = 0; if (++loy == H) { loy = 0; ++ln; } }
    };
    auto put = [&](char *img_hi, char *img_lo, int byte_off, const u32x4 (&r)[2]) {
        const f32x4 a = __builtin_bit_cast(f32x4, r[0]), c = __builtin_bit_cast(f32x4, r[1]);
        const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        acg_u32x4 hi, lo;
        acg_split8(v, hi, lo);
        *(acg_u32x4 *)(img_hi + byte_off) = hi;
        *(acg_u32x4 *)(img_lo + byte_off) = lo;
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int k = 0; k < XS; ++k) {
            const int u = tid + 256 * k;
            put(lds, lds + XI, (1 + u / UX) * PX + (u % UX) * 16, rx[k]);
        }
        if (halo) put(lds, lds + XI, (tid < UX ? 0 : SKP + 1) * PX + (tid % UX) * 16, rh);
#pragma unroll
        for (int k = 0; k < DS; ++k) {
            const int u = tid + 256 * k;
            put(lds + 2 * XI, lds + 2 * XI + DI, (u / UD) * PD + (u % UD) * 16, rd[k]);
            if (do_bias) {
                const f32x4 a = __builtin_bit_cast(f32x4, rd[k][0]), c = __builtin_bit_cast(f32x4, rd[k][1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bsum[e] += a[e]; bsum[4 + e] += c[e]; }
            }
        }
    };

    const int gq = lane >> 4, li = lane & 15;
    const int frag_row = 32 * wave + 8 * (gq >> 1) + (li >> 2), frag_col = 16 * (gq & 1) + 4 * (li & 3);
    const lds_char *xb = (const lds_char *)lds + frag_row * PX + frag_col * 2;
    const lds_char *db = (const lds_char *)lds + 2 * XI + frag_row * PD + frag_col * 2;
    auto frag = [&](const lds_char *p, int pitch) {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 4 * pitch));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    if (nst > 0) load_stage();
    for (int s = 0; s < nst; ++s) {
        __syncthreads(); // every wave has read the previous stage
        store_stage();
        __syncthreads();
        if (s + 1 < nst) load_stage(); // in flight under the MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bh[TJ], bl[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                bh[j] = frag(db + ks * 16 * PD + j * 64, PD);
                bl[j] = frag(db + DI + ks * 16 * PD + j * 64, PD);
            }
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const bf16x8 ah = frag(xb + (ks * 16 + t) * PX + i * 64, PX);
                    const bf16x8 al = frag(xb + XI + (ks * 16 + t) * PX + i * 64, PX);
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[t][i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[t][i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < TJ; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[t][i][j], 0, 0, 0);
                }
        }
    }
    __syncthreads();

    if (do_bias) { // threads with the same channel group (tid % UD) fold their column sums in fixed order
        float *red = (float *)lds;
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < BCO) {
            float sum = 0.f;
            for (int r = 0; r < 256 / UD; ++r) sum += red[(r * UD + (tid >> 3)) * 8 + (tid & 7)];
            g.bias_part[(long long)split * g.CoP + tid] = sum;
        }
        __syncthreads();
    }

    // the four waves hold partial sums over their pixel slices: waves 1-3 park one tap at a time in LDS, wave 0 adds them
    // in wave order and writes the tap out
    float *red = (float *)lds;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(((wave - 1) * TI * TJ + i * TJ + j) * 16 + r) * 64 + lane] = acc[t][i][j][r];
        }
        __syncthreads();
        if (wave == 0) {
            float *o = part + ((long long)split * 9 + ky * 3 + t) * g.CiP * g.CoP;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[t][i][j][r];
#pragma unroll
                        for (int w = 0; w < 3; ++w) v += red[((w * TI * TJ + i * TJ + j) * 16 + r) * 64 + lane];
                        const int ci = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        const int co = j * 32 + (lane & 31);
                        o[(long long)ci * g.CoP + co] = v;
                    }
        }
        __syncthreads();
    }
}

static bool krow_taps_ok(const WGeom &g, const Taps &t)
{
    if (g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA || g.thin) return false;
    if (t.n != 9 || g.is != 1 || g.Hin != g.Hg || g.Win != g.Wg || g.bias_from == 2) return false;
    for (int i = 0; i < 9; ++i)
        if (t.dy[i] != i / 3 - 1 || t.dx[i] != i % 3 - 1) return false;
    return true;
}

// the 32 <-> 64 channel variant: whole-tensor channel tiles, width a multiple of the 128-pixel run
bool acg_wgrad_krow_s_ok(const WGeom &g, const Taps &t)
{
    static const bool off = acg_debug_switch("ACG_NO_KROW") || acg_debug_switch("ACG_NO_KROW_S"); // A/B switches
    if (off || !krow_taps_ok(g, t) || g.Wg % SKP != 0 || g.m_per_split % SKP != 0) return false;
    const bool c = (g.Cin == 32 && g.Cg == 64) || (g.Cin == 64 && g.Cg == 32);
    return c && g.CiP == g.Cin && g.CoP == g.Cg;
}

int acg_wgrad_krow_s_launch(const float *x, const float *dy, float *part, const WGeom &g, hipStream_t st)
{
    const int blocks = g.nsplit * 3;
    const long long nimg = g.Mtot / ((long long)g.Hg * g.Wg);
    const long long xbytes = nimg * g.Hin * g.Win * g.Cin * 4, dbytes = g.Mtot * g.Cg * 4;
    ACG_REQUIRE(xbytes < (1LL << 32) && dbytes < (1LL << 32), "wgrad_x3_krow_s: operand exceeds the 4 GiB buffer-addressing limit");
    if (g.Cin == 32) hipLaunchKernelGGL((wgrad_x3_krow_s<32, 64>), dim3(blocks), dim3(256), 0, st, x, dy, part, g, (unsigned)xbytes, (unsigned)dbytes);
    else hipLaunchKernelGGL((wgrad_x3_krow_s<64, 32>), dim3(blocks), dim3(256), 0, st, x, dy, part, g, (unsigned)xbytes, (unsigned)dbytes);
    ACG_CHECK_LAUNCH("wgrad_x3_krow_s");
    acg_note_kernel("wgrad_x3_krow_s<%d,%d>", g.Cin, g.Cin == 32 ? 64 : 32);
    return ACG_OK;
}

// stride-1 3x3, pad 1, same-size maps whose width is a multiple of the 32-pixel run, 128-multiple channels on both sides
bool acg_wgrad_krow_ok(const WGeom &g, const Taps &t)
{
    static const bool off = acg_debug_switch("ACG_NO_KROW"); // A/B switch
    if (off || g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA || g.thin) return false;
    if (t.n != 9 || g.is != 1 || g.Hin != g.Hg || g.Win != g.Wg || g.Wg % KP != 0) return false;
    if (g.Cin % BC != 0 || g.Cg % BC != 0 || g.CiP != g.Cin || g.CoP != g.Cg || g.bias_from == 2) return false;
    if (g.m_per_split % KP != 0) return false;
    for (int i = 0; i < 9; ++i)
        if (t.dy[i] != i / 3 - 1 || t.dx[i] != i % 3 - 1) return false;
    return true;
}

int acg_wgrad_krow_launch(const float *x, const float *dy, float *part, const WGeom &g, hipStream_t st)
{
    const int blocks = g.nsplit * 3 * (g.CiP / BC) * (g.CoP / BC);
    const long long nimg = g.Mtot / ((long long)g.Hg * g.Wg);
    const long long xbytes = nimg * g.Hin * g.Win * g.Cin * 4, dbytes = g.Mtot * g.Cg * 4;
    ACG_REQUIRE(xbytes < (1LL << 32) && dbytes < (1LL << 32), "wgrad_x3_krow: operand exceeds the 4 GiB buffer-addressing limit");
    hipLaunchKernelGGL(wgrad_x3_krow, dim3(blocks), dim3(512), 0, st, x, dy, part, g, (unsigned)xbytes, (unsigned)dbytes);
    ACG_CHECK_LAUNCH("wgrad_x3_krow");
    acg_note_kernel("wgrad_x3_krow");
    return ACG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad_x3_krow on PRE-SPLIT ("S16", conv_internal.h) operands: x and dy arrive as bf16 hi / lo halves, so the stage images
// are filled by LDS-DMA (buffer_load_dwordx4 ... lds) — no VGPR staging, no split, no ds_write_b128 — and three stage
// buffers keep two stages of loads in flight behind the MFMAs.  An LDS row is one pixel: [128 ch hi: 256 B][128 ch lo:
// 256 B][64 B pad], 576 bytes = 36 DMA lanes (the four pad lanes are masked): a pixel's 512 bytes are contiguous in
// memory, rows 576 B apart put the four pixel rows of a transposed-read block on distinct 64-byte bank segments exactly
// like the 320-byte rows above.
//
// M16 (namespace krow16; opt-in through ACG_KROW_M16, measured slower: DESIGN_LOG.md R5.2): the MFMAs are v_mfma_f32_16x16x32_bf16 — the shape that holds the higher clock under load (1.88-1.97 against
// 1.73-1.77 PFLOP/s in register-only loops, tools/probes/mfma_rate.hip).  One MFMA spans the whole 32-pixel stage in K; its
// K group g (lanes 16 g .. 16 g + 15) takes pixels 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3 (two transposed reads, 16 rows
// apart — the same permutation of K on both operands), a 16-lane group supplies four pixel rows x 16 channels, and rows are
// 544 bytes: the eight rows the two groups of a 32-lane pass read fall on eight distinct 32-byte bank segments.  The six steps
// of a stage are (output-channel half) x (tap) instead of (K half) x (tap): the same MFMA cycles and registers per step, the x
// fragments read once per half.
#ifndef ACG_KROW_NDW
#define ACG_KROW_NDW 4
#endif
#define KS_M16 1
#define KS_NS krow16
#include "conv_wgrad_tr_s16.inc"
#undef KS_M16
#undef KS_NS
#define KS_M16 0
#define KS_NS krow32
#include "conv_wgrad_tr_s16.inc"
#undef KS_M16
#undef KS_NS

int acg_wgrad_krow_s16_launch(const void *x, const void *dy, float *part, const WGeom &g, hipStream_t st)
{
    const int blocks = g.nsplit * 3 * (g.CiP / BC) * (g.CoP / BC);
    const long long nimg = g.Mtot / ((long long)g.Hg * g.Wg);
    const long long xbytes = nimg * g.Hin * g.Win * g.Cin * 4, dbytes = g.Mtot * g.Cg * 4;
    ACG_REQUIRE(xbytes < (1LL << 32) && dbytes < (1LL << 32), "wgrad_x3_krow_s16: operand exceeds the 4 GiB buffer-addressing limit");
    // A/B switch (read per call): the v_mfma_f32_16x16x32_bf16 form (krow16) — measured 0.36-0.38 ms against 0.332 ms for the
    // 32x32x16 form at batch 32 (DESIGN_LOG.md R5.2: its x fragments are read once per output-channel half), so not the default
    const bool m16 = acg_debug_switch("ACG_KROW_M16");
#define KROW_S16(R, NS) hipLaunchKernelGGL((NS::wgrad_x3_krow_s16<R>), dim3(blocks), dim3(512 + 64 * ACG_KROW_NDW), 0, st, (const char *)x, (const char *)dy, part, g, (unsigned)xbytes, (unsigned)dbytes)
    if (g.reflect) { if (m16) KROW_S16(true, krow16); else KROW_S16(true, krow32); }
    else { if (m16) KROW_S16(false, krow16); else KROW_S16(false, krow32); }
#undef KROW_S16
    ACG_CHECK_LAUNCH("wgrad_x3_krow_s16");
    acg_note_kernel(m16 ? "wgrad_x3_krow_s16<16x16x32>" : "wgrad_x3_krow_s16");
    return ACG_OK;
}
