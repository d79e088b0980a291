// Implicit-GEMM convolution for gfx950 on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// One kernel family serves: Conv2d forward (zero or reflection padding folded into the
// gather), the data gradient of stride-1 and stride-2 convolutions, and ConvTranspose2d
// forward/backward — all expressed as   out[m, co] = sum_{tap, ci} in[pix(m, tap), ci] * W[tap][ci][co]
// over a per-launch tap list (im2col-free: the A operand is gathered row-wise from NHWC).
//
// Tiling: block = 256 threads = 4 waves; block tile BM(pixels) x BN(channels); K advances in
// stages of KC=16 input channels of one tap.  LDS images are [kchunk(8)][row][8 floats] so a
// lane's A (or B) fragment for four consecutive MFMAs is ONE ds_read_b128; A/B agree on the
// k order (8*kc + 4*(lane>>5) + j).  Global loads for stage s+1 are issued before the MFMAs of
// stage s (register staging), LDS is single-buffered with two barriers per stage.
#include "conv_internal.h"

#define KC ACG_KC

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_conv_f32(const float *__restrict__ in, const float *__restrict__ wp,
                                                      const float *__restrict__ bias, float *__restrict__ out,
                                                      Geom g, Taps taps)
{
    constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int AL = BM * KC / 4 / 256;
    constexpr int BCH = BN * KC / 4;
    constexpr int BL = (BCH + 255) / 256;
    static_assert(WM * WN == 4 && AL >= 1 && MB >= 1 && NB >= 1, "tile config");

    __shared__ __attribute__((aligned(16))) float As[2 * BM * 8];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BN * 8];
    __shared__ long long out_off[BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: blocks that share an XCD (bid % 8) walk neighbouring tiles, so the
    // halo rows / both N-tiles of one pixel tile hit the same L2.  Bijective for any grid size.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tiles_n = g.ncols_pad / BN;
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int n0 = tile_n * BN;
    const long long m0 = (long long)tile_m * BM;
    const int GHW = g.GH * g.GW;

    // per-thread gather rows (fixed across the K loop)
    const int q = tid & 3;
    int a_img[AL], a_by[AL], a_bx[AL];
    bool a_ok[AL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const long long m = m0 + (tid >> 2) + 64 * j;
        a_ok[j] = m < g.Mtot;
        const long long mm = a_ok[j] ? m : 0;
        const int n = (int)(mm / GHW);
        const int r = (int)(mm - (long long)n * GHW);
        const int gy = r / g.GW, gx = r - gy * g.GW;
        a_img[j] = n;
        a_by[j] = gy * g.is;
        a_bx[j] = gx * g.is;
    }
    if (tid < BM) {
        const long long m = m0 + tid;
        long long off = -1;
        if (m < g.Mtot) {
            const int n = (int)(m / GHW);
            const int r = (int)(m - (long long)n * GHW);
            const int gy = r / g.GW, gx = r - gy * g.GW;
            off = (((long long)n * g.Hout + (gy * g.os + g.oy0)) * g.Wout + (gx * g.os + g.ox0)) * g.Cout;
        }
        out_off[tid] = off;
    }

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nci = g.Cin / KC;
    const int S = taps.n * nci;
    f32x4 ra[AL], rb[BL];

    auto load_stage = [&](int s) {
        const int t = s / nci;
        const int c0 = (s - t * nci) * KC;
        const int ty = taps.dy[t], tx = taps.dx[t], tw = taps.w[t];
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            int iy = a_by[j] + ty, ix = a_bx[j] + tx;
            bool ok = a_ok[j];
            if (g.reflect) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                ix = ix < 0 ? -ix : ix;
                ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
            } else {
                ok = ok && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win;
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *(const f32x4 *)(in + (((long long)a_img[j] * g.Hin + iy) * g.Win + ix) * g.Cin + c0 + 4 * q);
            ra[j] = v;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BCH) {
                const int kc = idx / (BN * 2);
                const int rem = idx - kc * BN * 2;
                rb[i] = *(const f32x4 *)(wp + (((long long)tw * (g.Cin >> 3) + (c0 >> 3) + kc) * g.ncols_pad + n0) * 8 +
                                         rem * 4);
            }
        }
    };

    load_stage(0);
    for (int s = 0; s < S; ++s) {
        __syncthreads(); // every wave finished reading the previous stage
#pragma unroll
        for (int j = 0; j < AL; ++j)
            *(f32x4 *)&As[((q >> 1) * BM + (tid >> 2) + 64 * j) * 8 + (q & 1) * 4] = ra[j];
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int idx = tid + 256 * i;
            if (idx < BCH) *(f32x4 *)&Bs[idx * 4] = rb[i];
        }
        __syncthreads();
        if (s + 1 < S) load_stage(s + 1); // in flight under the MFMAs below
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            f32x4 a[MB], b[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i)
                a[i] = *(const f32x4 *)&As[(kc * BM + wm * TM + i * 32 + (lane & 31)) * 8 + (lane >> 5) * 4];
#pragma unroll
            for (int j = 0; j < NB; ++j)
                b[j] = *(const f32x4 *)&Bs[(kc * BN + wn * TN + j * 32 + (lane & 31)) * 8 + (lane >> 5) * 4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
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

int acg_igemm_launch(const float *in, const float *wp, const float *bias, float *out, const Geom &g, const Taps &t,
                     hipStream_t st)
{
    if (g.Mtot <= 0 || t.n <= 0) return ACG_OK;
    const int bn = bn_for(g.Cout);
    const int tiles_m = acg_cdiv(g.Mtot, 128);
    const int tiles_n = g.ncols_pad / bn;
    dim3 grid(tiles_m * tiles_n), block(256);
    if (bn == 128)
        hipLaunchKernelGGL((igemm_conv_f32<128, 128, 2, 2>), grid, block, 0, st, in, wp, bias, out, g, t);
    else if (bn == 64)
        hipLaunchKernelGGL((igemm_conv_f32<128, 64, 2, 2>), grid, block, 0, st, in, wp, bias, out, g, t);
    else
        hipLaunchKernelGGL((igemm_conv_f32<128, 32, 4, 1>), grid, block, 0, st, in, wp, bias, out, g, t);
    ACG_CHECK_LAUNCH("igemm_conv_f32");
    return ACG_OK;
}
