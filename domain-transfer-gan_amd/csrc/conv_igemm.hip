// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// One kernel family serves: Conv2d forward (zero or reflection padding folded into the
// gather), the data gradient of stride-1 and stride-2 convolutions, and ConvTranspose2d
// forward/backward — all expressed as   out[m, co] = sum_{tap, ci} in[pix(m, tap), ci] * W[tap][ci][co]
// over a per-launch tap list (im2col-free: the A operand is gathered row-wise from NHWC).
//
// Tiling: block = 256 threads = 4 waves; block tile BM(pixels) x BN(channels); K advances in
// stages of KC=32 input channels (one 128-B line per gathered pixel; KC=16 for 16-channel tensors) of one tap.  LDS images are [kchunk(8)][row][8 floats] so a
// lane's A (or B) fragment for four consecutive MFMAs is ONE ds_read_b128; A/B agree on the
// k order (8*kc + 4*(lane>>5) + j).  Global loads for stage s+1 are issued before the MFMAs of
// stage s (register staging); LDS is double-buffered: ONE barrier per stage.  K order is channel-chunk
// major / tap minor and M tiles are 8x16 spatial patches, which keeps the per-XCD working set inside L2.
#include <stdlib.h>
#include "conv_internal.h"

// pixel (n, gy, gx) of tile-local row `ml` : 2-D spatial tiles (tw x BM/tw) when g.tw > 0, else flattened
template <int BM>
__device__ __forceinline__ bool tile_pixel(const Geom &g, int tile_m, int ml, int &n, int &gy, int &gx)
{
    if (g.tw > 0) {
        const int th = BM / g.tw;
        const int tx_n = (g.GW + g.tw - 1) / g.tw, ty_n = (g.GH + th - 1) / th;
        const int per_img = tx_n * ty_n;
        n = tile_m / per_img;
        const int r = tile_m - n * per_img;
        const int tyi = r / tx_n, txi = r - tyi * tx_n;
        gy = tyi * th + ml / g.tw;
        gx = txi * g.tw + ml % g.tw;
        return gy < g.GH && gx < g.GW;
    }
    const long long m = (long long)tile_m * BM + ml;
    const int GHW = g.GH * g.GW;
    const bool ok = m < g.Mtot;
    const long long mm = ok ? m : 0;
    n = (int)(mm / GHW);
    const int r = (int)(mm - (long long)n * GHW);
    gy = r / g.GW;
    gx = r - gy * g.GW;
    return ok;
}

template <int BM, int BN, int WM, int WN, int KC, int DB>
__global__ __launch_bounds__(256) void igemm_conv_f32(const float *__restrict__ in, const float *__restrict__ wp,
                                                      const float *__restrict__ bias, float *__restrict__ out,
                                                      Geom g, Taps taps)
{
    constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int NKC = KC / 8;              // 8-float k-chunks per stage
    constexpr int QPR = KC / 4;              // float4 chunks per gathered pixel row
    constexpr int RPP = 256 / QPR;           // pixel rows loaded per pass
    constexpr int AL = BM / RPP;             // float4 A loads per thread per stage
    constexpr int BCH = BN * KC / 4;
    constexpr int BL = (BCH + 255) / 256;
    constexpr int AKS = BM * 8 + 8, BKS = BN * 8 + 8; // k-chunk strides, +32 B pad against LDS bank aliasing
    constexpr int ASZ = NKC * AKS, BSZ = NKC * BKS;
    static_assert(WM * WN == 4 && AL >= 1 && MB >= 1 && NB >= 1, "tile config");

    __shared__ __attribute__((aligned(16))) float As[(DB ? 2 : 1) * ASZ];
    __shared__ __attribute__((aligned(16))) float Bs[(DB ? 2 : 1) * BSZ];
    __shared__ long long out_off[BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: blocks that share an XCD (bid % 8) walk neighbouring tiles, so halo pixels and
    // both N-tiles of one pixel tile hit the same L2.  Bijective for any grid size.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tiles_n = g.ncols_pad / BN;
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int n0 = tile_n * BN;

    // per-thread gather rows (fixed across the K loop)
    const int q = tid % QPR, rrow = tid / QPR;
    int a_img[AL], a_by[AL], a_bx[AL];
    bool a_ok[AL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        int n, gy, gx;
        a_ok[j] = tile_pixel<BM>(g, tile_m, rrow + RPP * j, n, gy, gx);
        a_img[j] = n;
        a_by[j] = gy * g.is;
        a_bx[j] = gx * g.is;
    }
    if (tid < BM) {
        int n, gy, gx;
        long long off = -1;
        if (tile_pixel<BM>(g, tile_m, tid, n, gy, gx))
            off = (((long long)n * g.Hout + (gy * g.os + g.oy0)) * g.Wout + (gx * g.os + g.ox0)) * g.Cout;
        out_off[tid] = off;
    }

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // K order: channel chunk OUTER, taps INNER — the 128-B (KC=32) segment of every halo pixel is re-read by
    // all taps back to back while it is hot in L1/L2, so the per-XCD working set stays ~halo x 128 B per block.
    const int S = g.thin ? (taps.n + 7) / 8 : taps.n * (g.Cin / KC);
    f32x4 ra[AL], rb[BL];

    auto load_stage = [&](int s) {
        int cc = s / taps.n;
        int t = s - cc * taps.n;
        int c0 = cc * KC, tw, ty, tx, coff = 4 * q;
        bool tok = true;
        if (KC == 32 && g.thin) { // this thread's 16-byte chunk q is tap 8s+q, channels 0..3
            t = s * 8 + q;
            tok = t < taps.n;
            t = tok ? t : 0;
            c0 = s * KC;
            coff = -c0; // gather channel offset 0 (c0 + coff == 0); c0 still addresses the flattened weight rows
            tw = 0;
        } else {
            tw = taps.w[t];
        }
        ty = taps.dy[t];
        tx = taps.dx[t];
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            int iy = a_by[j] + ty, ix = a_bx[j] + tx;
            bool ok = a_ok[j] && tok;
            if (g.reflect) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                ix = ix < 0 ? -ix : ix;
                ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
            } else {
                ok = ok && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win;
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4 *)(in + (((long long)a_img[j] * g.Hin + iy) * g.Win + ix) * g.Cin + c0 + coff);
            ra[j] = v;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BCH) {
                const int kc = idx / (BN * 2);
                const int rem = idx - kc * BN * 2;
                rb[i] = *(const f32x4 *)(wp + (((long long)tw * g.bk8 + (c0 >> 3) + kc) * g.ncols_pad + n0) * 8 +
                                         rem * 4);
            }
        }
    };
    auto store_stage = [&](int buf) {
        float *A = As + buf * ASZ, *B = Bs + buf * BSZ;
#pragma unroll
        for (int j = 0; j < AL; ++j) *(f32x4 *)&A[(q >> 1) * AKS + (rrow + RPP * j) * 8 + (q & 1) * 4] = ra[j];
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BCH) {
                const int kc = idx / (BN * 2);
                *(f32x4 *)&B[kc * BKS + (idx - kc * BN * 2) * 4] = rb[i];
            }
        }
    };

    auto compute = [&](int cur) {
        const float *A = As + cur * ASZ, *B = Bs + cur * BSZ;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            f32x4 a[MB], b[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i)
                a[i] = *(const f32x4 *)&A[kc * AKS + (wm * TM + i * 32 + (lane & 31)) * 8 + (lane >> 5) * 4];
#pragma unroll
            for (int j = 0; j < NB; ++j)
                b[j] = *(const f32x4 *)&B[kc * BKS + (wn * TN + j * 32 + (lane & 31)) * 8 + (lane >> 5) * 4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
        }
    };

    load_stage(0);
    if (DB) {
        store_stage(0);
        __syncthreads();
        for (int s = 0; s < S; ++s) {
            const int cur = s & 1;
            if (s + 1 < S) load_stage(s + 1); // global loads in flight under the MFMAs below
            compute(cur);
            // the other buffer was last read in iteration s-1; every wave has passed that iteration's barrier
            if (s + 1 < S) store_stage(cur ^ 1);
            __syncthreads();
        }
    } else {
        for (int s = 0; s < S; ++s) {
            __syncthreads(); // every wave finished reading the previous stage
            store_stage(0);
            __syncthreads();
            if (s + 1 < S) load_stage(s + 1);
            compute(0);
        }
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int co = n0 + wn * TN + j * 32 + (lane & 31);
        const bool cok = co < g.Cout;
        const float bv = (bias != nullptr && cok) ? bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const long long off = out_off[row];
                if (cok && off >= 0) out[off + co] = acg_apply_act(acc[i][j][r] + bv, g.act);
            }
        }
    }
}

static int bn_for(int c) { return c >= 128 ? 128 : (c >= 64 ? 64 : 32); }

extern "C" int acg_ncols_pad(int c)
{
    const int bn = bn_for(c);
    return (c + bn - 1) / bn * bn;
}

template <int KC, int DB>
static void launch_kc(int bn, dim3 grid, hipStream_t st, const float *in, const float *wp, const float *bias, float *out,
                      const Geom &g, const Taps &t)
{
    dim3 block(256);
    if (bn == 128)
        hipLaunchKernelGGL((igemm_conv_f32<128, 128, 2, 2, KC, DB>), grid, block, 0, st, in, wp, bias, out, g, t);
    else if (bn == 64)
        hipLaunchKernelGGL((igemm_conv_f32<128, 64, 2, 2, KC, DB>), grid, block, 0, st, in, wp, bias, out, g, t);
    else
        hipLaunchKernelGGL((igemm_conv_f32<128, 32, 4, 1, KC, DB>), grid, block, 0, st, in, wp, bias, out, g, t);
}

// tuning switches (read once): ACG_IGEMM_KC=16|32, ACG_IGEMM_DB=0|1, ACG_IGEMM_TILE2D=0|1
static int env_int(const char *k, int dflt)
{
    const char *v = getenv(k);
    return v ? atoi(v) : dflt;
}

int acg_igemm_launch(const float *in, const float *wp, const float *bias, float *out, const Geom &g0, const Taps &t,
                     hipStream_t st)
{
    static const int kKC = env_int("ACG_IGEMM_KC", 32), kDB = env_int("ACG_IGEMM_DB", 0), k2D = env_int("ACG_IGEMM_TILE2D", 0);
    if (g0.Mtot <= 0 || t.n <= 0) return ACG_OK;
    Geom g = g0;
    const int bn = bn_for(g.Cout);
    if (g_acg_precision == ACG_PREC_BF16 && g_acg_conv_impl == ACG_IMPL_MFMA) return acg_igemm_bf16_launch(in, wp, bias, out, g0, t, bn, st);
    int tiles_m;
    if (k2D && g.GW >= 16 && g.GH >= 8) { // 2-D spatial tiles 8 x 16: halo 10 x 18 instead of 3 full rows
        g.tw = 16;
        tiles_m = (int)(g.Mtot / ((long long)g.GH * g.GW)) * acg_cdiv(g.GH, 8) * acg_cdiv(g.GW, 16);
    } else {
        g.tw = 0;
        tiles_m = acg_cdiv(g.Mtot, 128);
    }
    const int tiles_n = g.ncols_pad / bn;
    dim3 grid(tiles_m * tiles_n);
    if (g.thin) g.bk8 = 4 * ((t.n + 7) / 8); else g.bk8 = g.Cin / 8;
    const bool kc32 = ((g.Cin % 32 == 0) && kKC == 32) || g.thin;
    if (kc32 && kDB) launch_kc<32, 1>(bn, grid, st, in, wp, bias, out, g, t);
    else if (kc32) launch_kc<32, 0>(bn, grid, st, in, wp, bias, out, g, t);
    else if (kDB) launch_kc<16, 1>(bn, grid, st, in, wp, bias, out, g, t);
    else launch_kc<16, 0>(bn, grid, st, in, wp, bias, out, g, t);
    ACG_CHECK_LAUNCH("igemm_conv_f32");
    return ACG_OK;
}
