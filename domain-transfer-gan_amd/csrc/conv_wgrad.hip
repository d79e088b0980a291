// Weight-gradient implicit GEMM for gfx950 (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   dW[tap][ci][co] = sum_m  x[pix(m, tap), ci] * dy[m, co]        (GEMM: M'=ci, N'=co, K'=pixels)
//
// NHWC makes both operands K-major rows ([pixel][channel]), which is exactly the natural LDS
// image for the MFMA A/B fragments (lane l: A[ci = l&31][k = l>>5] -> consecutive lanes read
// consecutive floats, conflict-free ds_read_b32).  The pixel reduction is split over the grid
// (deterministic split-K): each block writes its partial tile, a second kernel sums the
// partials in a fixed order and scatters into torch's OIHW layout — no float atomics, so
// results are bitwise reproducible run to run.
#include "conv_internal.h"
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int BCI, int BCO, int WI, int WJ, int WK, int KP, bool X3 = false>
__global__ __launch_bounds__(256) void wgrad_f32(const float *__restrict__ x, const float *__restrict__ dy,
                                                 float *__restrict__ part, WGeom g, Taps taps, unsigned x_bytes,
                                                 unsigned d_bytes)
{
    constexpr int TI = BCI / WI, TJ = BCO / WJ, MI = TI / 32, MJ = TJ / 32;
    constexpr int XCH = KP * BCI / 4, DCH = KP * BCO / 4;
    constexpr int XL = XCH / 256, DL = DCH / 256;
    constexpr int KW = KP / WK; // pixels of a stage handled by one wave
    static_assert(WI * WJ * WK == 4 && XL >= 1 && DL >= 1 && MI >= 1 && MJ >= 1, "tile config");

    __shared__ __attribute__((aligned(16))) float Xs[KP * BCI];
    __shared__ __attribute__((aligned(16))) float Ds[KP * BCO];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave % WK, wj = (wave / WK) % WJ, wi = wave / (WK * WJ);

    const int tiles_ci = g.CiP / BCI, tiles_co = g.CoP / BCO;
    // XCD-aware order: the blocks of one pixel range (all taps / tiles of a split) share an XCD and stream the
    // same x / dy rows through one L2 instead of eight (bijective remap, any grid size)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    int b = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tco = b % tiles_co; b /= tiles_co;
    const int tci = b % tiles_ci; b /= tiles_ci;
    const int ntap_blocks = g.thin ? 1 : taps.n;
    const int tap = b % ntap_blocks;
    const int split = b / ntap_blocks;
    const int ci0 = tci * BCI, co0 = tco * BCO;
    int ty = taps.dy[tap], tx = taps.dx[tap];
    const long long mbeg = (long long)split * g.m_per_split;
    long long mend = mbeg + g.m_per_split;
    if (mend > g.Mtot) mend = g.Mtot;
    const int GHW = g.Hg * g.Wg;
    // branch-free gathers: 32-bit offsets into buffer resources; masked lanes use offset 0xFFFFFFFF (-> zeros)
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd_ = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, d_bytes, 0x00020000);

    f32x16 acc[MI][MJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 rx[XL], rd[DL];
    const bool do_bias = g.bias_from != 0 && tap == 0 && (g.bias_from == 1 ? tci == 0 : tco == 0);
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    // pixel coordinates of this thread's XL gathered rows: decoded ONCE (two integer divisions), then advanced by KP
    // pixels per stage with carries — the K dimension is pixels, so a per-stage re-decode would cost ~1.3k VALU
    // cycles per wave and stage in the same in-order stream that has to issue the MFMAs.
    int xn[XL], xy[XL], xx[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        const long long m = mbeg + (tid + 256 * j) / (BCI / 4);
        const long long mm = m < g.Mtot ? m : 0;
        xn[j] = (int)(mm / GHW);
        const int rr = (int)(mm - (long long)xn[j] * GHW);
        xy[j] = rr / g.Wg;
        xx[j] = rr - xy[j] * g.Wg;
    }
    // thin layout: column quad c4 of tile tci = tap 8*tci + c4 (channels 0..3) — a per-lane tap, looked up ONCE here:
    // inside the stage loop the table fetch is a vector load whose s_waitcnt vmcnt(0) would serialize the gathers
    int tyj[XL], txj[XL];
    bool tokj[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        tyj[j] = ty; txj[j] = tx; tokj[j] = true;
        if (BCI == 32 && g.thin) {
            const int t = tci * 8 + (tid + 256 * j) % (BCI / 4);
            tokj[j] = t < taps.n;
            tyj[j] = taps.dy[tokj[j] ? t : 0];
            txj[j] = taps.dx[tokj[j] ? t : 0];
        }
    }
    auto load_stage = [&](long long k0) {
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx / (BCI / 4), c4 = idx - r * (BCI / 4);
            const long long m = k0 + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            int ci = ci0 + c4 * 4;
            bool cok = ci < g.Cin;
            if (BCI == 32 && g.thin) {
                cok = tokj[j];
                ci = 0;
            }
            {
                int iy = xy[j] * g.is + tyj[j], ix = xx[j] * g.is + txj[j];
                bool ok = m < mend && cok;
                if (g.reflect) {
                    iy = iy < 0 ? -iy : iy;
                    iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                    ix = ix < 0 ? -ix : ix;
                    ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
                } else {
                    ok = ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
                }
                const unsigned off = acg_masked_off((unsigned)(((xn[j] * g.Hin + iy) * g.Win + ix) * g.Cin + ci) * 4u, ok);
                v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx_, off, 0, 0));
            }
            rx[j] = v;
            // advance this row by KP pixels for the next stage
            xx[j] += KP;
            while (xx[j] >= g.Wg) { xx[j] -= g.Wg; if (++xy[j] == g.Hg) { xy[j] = 0; ++xn[j]; } }
        }
#pragma unroll
        for (int j = 0; j < DL; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx / (BCO / 4), c4 = idx - r * (BCO / 4);
            const long long m = k0 + r;
            const int co = co0 + c4 * 4;
            const unsigned off = acg_masked_off((unsigned)((int)m * g.Cg + co) * 4u, m < mend && co < g.Cg);
            rd[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd_, off, 0, 0));
        }
    };

    if (mbeg < mend) load_stage(mbeg);
    for (long long k0 = mbeg; k0 < mend; k0 += KP) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < XL; ++j) *(f32x4 *)&Xs[(tid + 256 * j) * 4] = rx[j];
#pragma unroll
        for (int j = 0; j < DL; ++j) *(f32x4 *)&Ds[(tid + 256 * j) * 4] = rd[j];
        if (do_bias) { // every j of a thread holds the same channel quad (256 % (BC/4) == 0)
            if (g.bias_from == 1) {
#pragma unroll
                for (int j = 0; j < DL; ++j) bsum += rd[j];
            } else {
#pragma unroll
                for (int j = 0; j < XL; ++j) bsum += rx[j];
            }
        }
        __syncthreads();
        if (k0 + KP < mend) load_stage(k0 + KP);
        if constexpr (X3) {
            // thin layers outside the strict fp32 mode: products as bf16x3 (8 consecutive pixels of a column per lane, split
            // hi/lo in registers); loader, fp32 LDS tiles and partial-sum layout unchanged
            static_assert(!X3 || KW % 16 == 0, "16 pixels per bf16 MFMA step");
#pragma unroll 2
            for (int kk = wk * KW; kk < (wk + 1) * KW; kk += 16) {
                bf16x8_t ah[MI], al[MI], bh[MJ], bl[MJ];
                const int row = kk + 8 * (lane >> 5);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = Xs[(row + e) * BCI + wi * TI + i * 32 + (lane & 31)];
                    acg_u32x4 hi, lo;
                    acg_split8(v, hi, lo);
                    ah[i] = __builtin_bit_cast(bf16x8_t, hi); al[i] = __builtin_bit_cast(bf16x8_t, lo);
                }
#pragma unroll
                for (int j = 0; j < MJ; ++j) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = Ds[(row + e) * BCO + wj * TJ + j * 32 + (lane & 31)];
                    acg_u32x4 hi, lo;
                    acg_split8(v, hi, lo);
                    bh[j] = __builtin_bit_cast(bf16x8_t, hi); bl[j] = __builtin_bit_cast(bf16x8_t, lo);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < MJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else
#pragma unroll 4
        for (int kk = wk * KW; kk < (wk + 1) * KW; kk += 2) {
            float a[MI], bb[MJ];
            const int row = kk + (lane >> 5);
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = Xs[row * BCI + wi * TI + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < MJ; ++j) bb[j] = Ds[row * BCO + wj * TJ + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < MJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
    }

    if (do_bias) { // fixed-order fold of the per-thread column sums through LDS -> bias_part[split][channel]
        __syncthreads();
        const int BC = g.bias_from == 1 ? BCO : BCI;
        const int q4 = BC / 4, rows = 256 / q4;
        *(f32x4 *)&Xs[tid * 4] = bsum;
        __syncthreads();
        if (tid < q4) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < rows; ++r) s += *(const f32x4 *)&Xs[(r * q4 + tid) * 4];
            const int cpad = g.bias_from == 1 ? g.CoP : g.CiP;
            const int c0 = g.bias_from == 1 ? co0 : ci0;
            *(f32x4 *)&g.bias_part[(long long)split * cpad + c0 + tid * 4] = s;
        }
        __syncthreads();
    }

    if (WK > 1) {
        // the WK waves of one (wi, wj) sub-tile hold partial sums of the SAME 32x32 outputs: fold them through LDS
        // in wave order (deterministic)
        static_assert(WK == 1 || (MI == 1 && MJ == 1), "split-K-in-block only for 32x32 wave tiles");
        static_assert(WK == 1 || KP * BCI >= WI * WJ * (WK - 1) * 1024, "LDS fold space");
        __syncthreads();
        float *red = Xs;
        const int grp = wi * WJ + wj;
        if (wk > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((grp * (WK - 1) + wk - 1) * 16 + r) * 64 + lane] = acc[0][0][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int w = 0; w < WK - 1; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] += red[((grp * (WK - 1) + w) * 16 + r) * 64 + lane];
        }
        if (wk != 0) return;
    }

    float *o = part + ((long long)split * ntap_blocks + tap) * g.CiP * g.CoP;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wi * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int co = co0 + wj * TJ + j * 32 + (lane & 31);
                o[(long long)ci * g.CoP + co] = acc[i][j][r];
            }
}

// taps per workgroup of the bf16 weight-gradient kernel (they share the gradient-side tile): the 64 -> 128-multiple channel
// 3x3 layers (stride-2 downsample / ConvTranspose, networks.py:168, 178) take a 64 x 128 tile and a whole kernel row
int acg_wgrad_taps_per_wg(int Ci, int Co, int ntaps, int thin)
{
    static const bool off = acg_debug_switch("ACG_NO_WGRAD_NT"); // A/B switch
    return (!off && !thin && g_acg_precision != ACG_PREC_F32 && g_acg_conv_impl == ACG_IMPL_MFMA && ntaps == 9 && Ci == 64 &&
            Co >= 128 && Co % 128 == 0) ? 3 : 1;
}

void acg_wgrad_tiles(int Ci, int Co, int *bci, int *bco, int ntaps)
{
    if (acg_wgrad_taps_per_wg(Ci, Co, ntaps, 0) == 3) { *bci = 64; *bco = 128; return; }
    // 128x128 for the dense 128/256-channel layers, 64x64 mid; thin layers 32x32; 32<->64-channel layers get a
    // rectangular 32x64 / 64x32 tile (two 32x32 wave tiles, two waves splitting K each)
    const int mn = Ci < Co ? Ci : Co, mx = Ci < Co ? Co : Ci;
    if (mn >= 128) { *bci = *bco = 128; return; }
    if (mn >= 64) { *bci = *bco = 64; return; }
    if (mn == 32 && mx >= 64) { *bci = Ci == 32 ? 32 : 64; *bco = Co == 32 ? 32 : 64; return; }
    *bci = *bco = 32;
}

int acg_wgrad_launch(const float *x, const float *dy, float *part, const WGeom &g, const Taps &t, hipStream_t st)
{
    int bci, bco;
    acg_wgrad_tiles(g.Cin, g.Cg, &bci, &bco, g.thin ? 0 : t.n);
    int tpk, tpf;
    if (acg_wgrad_thin_patch_ok(g, t, &tpk, &tpf)) return acg_wgrad_thin_patch_launch(x, dy, part, g, t, st);
    if (acg_wgrad_krowg_ok(g, t)) return acg_wgrad_krowg_launch(x, dy, part, g, t, st);
    if (acg_wgrad_krow_ok(g, t)) return acg_wgrad_krow_launch(x, dy, part, g, st);
    if (acg_wgrad_krow_s_ok(g, t)) return acg_wgrad_krow_s_launch(x, dy, part, g, st);
    if (g_acg_precision != ACG_PREC_F32 && !g.thin) return acg_wgrad_bf16_launch(x, dy, part, g, t, bci, bco, st);
    const int blocks = g.nsplit * (g.thin ? 1 : t.n) * (g.CiP / bci) * (g.CoP / bco);
    dim3 grid(blocks), block(256);
    const long long nimg = g.Mtot / ((long long)g.Hg * g.Wg);
    const long long xbytes = nimg * g.Hin * g.Win * g.Cin * 4, dbytes = g.Mtot * g.Cg * 4;
    ACG_REQUIRE(xbytes < (1LL << 32) && dbytes < (1LL << 32), "wgrad: operand exceeds the 4 GiB buffer-addressing limit");
    const unsigned xb = (unsigned)xbytes, db = (unsigned)dbytes;
    if (bci == 128)
        hipLaunchKernelGGL((wgrad_f32<128, 128, 2, 2, 1, 32>), grid, block, 0, st, x, dy, part, g, t, xb, db);
    else if (bci == 64 && bco == 64)
        hipLaunchKernelGGL((wgrad_f32<64, 64, 2, 2, 1, 32>), grid, block, 0, st, x, dy, part, g, t, xb, db);
    else if (bci == 32 && bco == 64)
        hipLaunchKernelGGL((wgrad_f32<32, 64, 1, 2, 2, 64>), grid, block, 0, st, x, dy, part, g, t, xb, db);
    else if (bci == 64 && bco == 32)
        hipLaunchKernelGGL((wgrad_f32<64, 32, 2, 1, 2, 64>), grid, block, 0, st, x, dy, part, g, t, xb, db);
    else if (g.thin && g_acg_precision != ACG_PREC_F32 && g_acg_conv_impl == ACG_IMPL_MFMA && !acg_debug_switch("ACG_NO_THIN_X3"))
        hipLaunchKernelGGL((wgrad_f32<32, 32, 1, 1, 4, 128, true>), grid, block, 0, st, x, dy, part, g, t, xb, db);
    else
        hipLaunchKernelGGL((wgrad_f32<32, 32, 1, 1, 4, 128>), grid, block, 0, st, x, dy, part, g, t, xb, db);
    ACG_CHECK_LAUNCH("wgrad_f32");
    acg_note_kernel("wgrad_f32");
    return ACG_OK;
}
