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

// stride-1 3x3, pad 1, same-size maps whose width is a multiple of the 32-pixel run, 128-multiple channels on both sides
bool acg_wgrad_krow_ok(const WGeom &g, const Taps &t)
{
    static const bool off = getenv("ACG_NO_KROW") != nullptr; // A/B switch
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
    return ACG_OK;
}
