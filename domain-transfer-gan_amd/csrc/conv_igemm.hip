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
// stage s (register staging, single LDS buffer, two barriers per stage).  K order is channel-chunk major /
// tap minor.  (Measured and dropped: LDS double buffering and 8x16 spatial M tiles — both lose to occupancy.)
#include <stdlib.h>
#include "conv_internal.h"
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int reflect_coord(int i, int n)
{
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// REFLECT / THIN are compile-time so the plain zero-pad path keeps wave-uniform tap lookups (scalar loads) and a
// branch-free gather: 32-bit offsets into a buffer resource, masked lanes get offset 0xFFFFFFFF, which the
// hardware range check turns into zeros.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int BM, int BN, int WM, int WN, int KC, bool REFLECT, bool THIN, bool X3 = false>
__global__ __launch_bounds__(256) void igemm_conv_f32(const float *__restrict__ in, const float *__restrict__ wp,
                                                      const float *__restrict__ bias, float *__restrict__ out,
                                                      Geom g, Taps taps, unsigned in_bytes, unsigned w_bytes)
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
    static_assert(WM * WN == 4 && AL >= 1 && MB >= 1 && NB >= 1 && BCH % 64 == 0, "tile config");
    static_assert(!THIN || KC == 32, "thin mode flattens 8 taps x 4 channels into one 32-deep stage");

    __shared__ __attribute__((aligned(16))) float As[ASZ];
    __shared__ __attribute__((aligned(16))) float Bs[BSZ];
    __shared__ __attribute__((aligned(16))) unsigned out_rel[BM];  // byte offset of a tile row's output pixel from the tile's first, ~0u: none
    __shared__ int tap_dy[64], tap_dx[64];

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
    const long long m0 = (long long)tile_m * BM;
    const int GHW = g.GH * g.GW;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, w_bytes, 0x00020000);

    if (THIN && tid < 64) {
        tap_dy[tid] = tid < taps.n ? taps.dy[tid] : 0;
        tap_dx[tid] = tid < taps.n ? taps.dx[tid] : 0;
    }

    // per-thread gather rows (fixed across the K loop)
    const int q = tid % QPR, rrow = tid / QPR;
    int a_row[AL], a_by[AL], a_bx[AL]; // a_row: pixel index of (n, by, bx) [zero pad] or n*Hin [reflect]
    bool a_ok[AL];
#pragma unroll
    for (int j = 0; j < AL; ++j) {
        const long long m = m0 + rrow + RPP * j;
        a_ok[j] = m < g.Mtot;
        const long long mm = a_ok[j] ? m : 0;
        const int n = (int)(mm / GHW);
        const int r = (int)(mm - (long long)n * GHW);
        const int gy = r / g.GW, gx = r - gy * g.GW;
        a_by[j] = gy * g.is;
        a_bx[j] = gx * g.is;
        a_row[j] = REFLECT ? n * g.Hin : (n * g.Hin + a_by[j]) * g.Win + a_bx[j];
    }
    auto pix_off = [&](long long m) { // element offset of grid position m's output pixel
        const int n = (int)(m / GHW);
        const int r = (int)(m - (long long)n * GHW);
        const int gy = r / g.GW, gx = r - gy * g.GW;
        return (((long long)n * g.Hout + (gy * g.os + g.oy0)) * g.Wout + (gx * g.os + g.ox0)) * g.Cout;
    };
    const long long off0 = pix_off(m0);   // uniform; the rows of a tile lie less than 4 GiB behind it
    if (tid < BM) out_rel[tid] = m0 + tid < g.Mtot ? (unsigned)((pix_off(m0 + tid) - off0) * 4) : ~0u;
    // per-thread B chunk offsets (bytes) inside one stage's weight block; the stage part is a scalar offset
    unsigned b_voff[BL];
    int b_lds[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int idx = tid + 256 * i;
        const int kc = idx / (BN * 2);
        const int rem = idx - kc * BN * 2;
        b_voff[i] = (unsigned)(((kc * g.ncols_pad + n0) * 8 + rem * 4) * 4);
        b_lds[i] = kc * BKS + rem * 4;
    }

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // K order: channel chunk OUTER, taps INNER — the 128-B (KC=32) segment of every halo pixel is re-read by all
    // taps back to back while it is hot in L1/L2.  Thin mode: stage s = taps 8s..8s+7, channels 0..3 of each.
    const int S = THIN ? (taps.n + 7) / 8 : taps.n * (g.Cin / KC);
    u32x4 ra[AL], rb[BL];
    if (THIN) __syncthreads(); // tap table visible

    auto load_stage = [&](int s) {
        int c0, ty, tx, tw;
        bool tok = true;
        if (THIN) { // this lane's 16-byte chunk q is tap 8s+q, channels 0..3; weight rows are the flattened k
            const int t = s * 8 + q;
            tok = t < taps.n;
            ty = tap_dy[t & 63];
            tx = tap_dx[t & 63];
            tw = 0;
            c0 = s * KC;
        } else {
            const int cc = s / taps.n;
            const int t = s - cc * taps.n;
            c0 = cc * KC;
            const int pk = taps.pk[t];
            ty = (pk << 24) >> 24;
            tx = (pk << 16) >> 24;
            tw = pk >> 16;
        }
        const int coff = THIN ? 0 : c0 + 4 * q;
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            int pix;
            bool ok = a_ok[j] && tok;
            if (REFLECT) {
                const int iy = reflect_coord(a_by[j] + ty, g.Hin), ix = reflect_coord(a_bx[j] + tx, g.Win);
                pix = (a_row[j] + iy) * g.Win + ix;
            } else {
                const int iy = a_by[j] + ty, ix = a_bx[j] + tx;
                ok = ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
                pix = a_row[j] + ty * g.Win + tx;
            }
            const unsigned off = acg_masked_off((unsigned)(pix * g.Cin + coff) * 4u, ok);
            ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
        }
        const unsigned soff = (unsigned)(((tw * g.bk8 + (c0 >> 3)) * g.ncols_pad) * 8) * 4u;
#pragma unroll
        for (int i = 0; i < BL; ++i)
            if (tid + 256 * i < BCH) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff, 0);
    };

    load_stage(0);
    for (int s = 0; s < S; ++s) {
        __syncthreads(); // every wave finished reading the previous stage
#pragma unroll
        for (int j = 0; j < AL; ++j) *(u32x4 *)&As[(q >> 1) * AKS + (rrow + RPP * j) * 8 + (q & 1) * 4] = ra[j];
#pragma unroll
        for (int i = 0; i < BL; ++i)
            if (tid + 256 * i < BCH) *(u32x4 *)&Bs[b_lds[i]] = rb[i];
        __syncthreads();
        if (s + 1 < S) load_stage(s + 1); // in flight under the MFMAs below
        if constexpr (X3) {
            // thin layers outside the strict fp32 mode: same loader, fp32 LDS tiles and packed weights, but the products run
            // as bf16x3 on the bf16 matrix pipe (a lane's 8 consecutive k of a chunk are split hi/lo in registers): the
            // fp32 MFMA made these HBM-sized layers matrix-bound (16 x 64 cycles per 32-deep stage and tile against 6 x 32)
            static_assert(!X3 || KC == 32, "two 16-deep bf16 MFMA steps per stage");
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t ah[MB], al[MB], bh[NB], bl[NB];
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const float *p = &As[(2 * ks + (lane >> 5)) * AKS + (wm * TM + i * 32 + (lane & 31)) * 8];
                    const f32x4 v0 = *(const f32x4 *)p, v1 = *(const f32x4 *)(p + 4);
                    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    acg_u32x4 hi, lo;
                    acg_split8(v, hi, lo);
                    ah[i] = __builtin_bit_cast(bf16x8_t, hi); al[i] = __builtin_bit_cast(bf16x8_t, lo);
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float *p = &Bs[(2 * ks + (lane >> 5)) * BKS + (wn * TN + j * 32 + (lane & 31)) * 8];
                    const f32x4 v0 = *(const f32x4 *)p, v1 = *(const f32x4 *)(p + 4);
                    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    acg_u32x4 hi, lo;
                    acg_split8(v, hi, lo);
                    bh[j] = __builtin_bit_cast(bf16x8_t, hi); bl[j] = __builtin_bit_cast(bf16x8_t, lo);
                }
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            f32x4 a[MB], b[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i)
                a[i] = *(const f32x4 *)&As[kc * AKS + (wm * TM + i * 32 + (lane & 31)) * 8 + (lane >> 5) * 4];
#pragma unroll
            for (int j = 0; j < NB; ++j)
                b[j] = *(const f32x4 *)&Bs[kc * BKS + (wn * TN + j * 32 + (lane & 31)) * 8 + (lane >> 5) * 4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Branch-free, as in
    // conv_bf16.hip: bias and activation in a pass over the accumulators, then buffer stores relative to the tile's first
    // pixel whose masked lanes carry an out-of-range offset (a predicated 64-bit store with a run-time activation switch per
    // element was ~45 instructions and a branch each: as many instructions as the whole main loop).
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int co = n0 + wn * TN + j * 32 + (lane & 31);
        const float bv = (bias != nullptr && co < g.Cout) ? bias[co] : 0.f;
        if (g.act == ACG_ACT_NONE) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += bv;
        } else if (g.act == ACG_ACT_RELU) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] + bv > 0.f ? acc[i][j][r] + bv : 0.f;
        } else if (g.act == ACG_ACT_LRELU) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acc[i][j][r] + bv > 0.f ? acc[i][j][r] + bv : 0.2f * (acc[i][j][r] + bv);
        } else {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = acg_apply_act(acc[i][j][r] + bv, g.act);
        }
    }
    typedef unsigned epi_u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + off0), 0, 0xFFFFFFF0u, 0x00020000);
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            // accumulator registers 4 r4 .. 4 r4 + 3 are tile rows 8 r4 + 4 (lane >> 5) + 0 .. 3
            const epi_u32x4 rel = *(const epi_u32x4 *)&out_rel[wm * TM + i * 32 + 8 * r4 + 4 * (lane >> 5)];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int co = n0 + wn * TN + j * 32 + (lane & 31);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = acc[i][j][4 * r4 + q]; // (a bit cast of the vector-element lvalue itself reads element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, acg_masked_off(rel[q] + (unsigned)co * 4u, rel[q] != ~0u && co < g.Cout), 0, 0);
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

template <int KC, bool REFLECT, bool THIN, bool X3 = false>
static void launch_v(int bn, dim3 grid, hipStream_t st, const float *in, const float *wp, const float *bias, float *out,
                     const Geom &g, const Taps &t, unsigned inb, unsigned wb)
{
    dim3 block(256);
    if (bn == 128)
        hipLaunchKernelGGL((igemm_conv_f32<128, 128, 2, 2, KC, REFLECT, THIN, X3>), grid, block, 0, st, in, wp, bias, out, g, t, inb, wb);
    else if (bn == 64)
        hipLaunchKernelGGL((igemm_conv_f32<128, 64, 2, 2, KC, REFLECT, THIN, X3>), grid, block, 0, st, in, wp, bias, out, g, t, inb, wb);
    else
        hipLaunchKernelGGL((igemm_conv_f32<128, 32, 4, 1, KC, REFLECT, THIN, X3>), grid, block, 0, st, in, wp, bias, out, g, t, inb, wb);
}

int acg_igemm_launch(const float *in, const float *wp, const float *bias, float *out, const Geom &g0, const Taps &t0,
                     hipStream_t st)
{
    if (g0.Mtot <= 0 || t0.n <= 0) return ACG_OK;
    Geom g = g0;
    const Taps t = acg_taps_pack(t0);
    const int bn = bn_for(g.Cout);
    if (g_acg_precision != ACG_PREC_F32 && g_acg_conv_impl == ACG_IMPL_MFMA && !g0.thin) return acg_igemm_bf16_launch(in, wp, bias, out, g0, t, bn, g0.w_elems, st);
    if (g0.thin && acg_conv_thinrow_ok(g0, t))   // C4 image -> 32 channels: the row-packed weights sit behind the regular region
        return acg_conv_thinrow_launch(in, wp + g0.w_elems, bias, out, g0, t, st);
    ACG_REQUIRE(g0.ns_part == nullptr, "igemm_conv: norm-backward sums requested on a kernel that does not emit them");
    const long long nimg = g.Mtot / ((long long)g.GH * g.GW);
    const long long in_bytes = nimg * g.Hin * g.Win * g.Cin * 4;
    ACG_REQUIRE(in_bytes < (1LL << 32), "igemm_conv: gathered tensor of %lld bytes exceeds the 4 GiB buffer-addressing limit", in_bytes);
    g.bk8 = g.thin ? 4 * ((t.n + 7) / 8) : g.Cin / 8;
    long long nslab = 1;
    if (!g.thin)
        for (int i = 0; i < t.n; ++i) nslab = t.w[i] + 1 > nslab ? t.w[i] + 1 : nslab;
    const long long w_bytes = nslab * g.bk8 * g.ncols_pad * 8 * 4;
    ACG_REQUIRE(w_bytes < (1LL << 32), "igemm_conv: packed weights exceed 4 GiB");
    const int tiles_m = acg_cdiv(g.Mtot, 128);
    const int tiles_n = g.ncols_pad / bn;
    dim3 grid(tiles_m * tiles_n);
    const unsigned inb = (unsigned)in_bytes, wb = (unsigned)w_bytes;
    const bool kc32 = (g.Cin % 32 == 0) || g.thin;
    static const bool no_thin_x3 = acg_debug_switch("ACG_NO_THIN_X3"); // A/B switch
    const bool x3 = g_acg_precision != ACG_PREC_F32 && g_acg_conv_impl == ACG_IMPL_MFMA && !no_thin_x3;
    if (g.thin && x3) {
        if (g.reflect) launch_v<32, true, true, true>(bn, grid, st, in, wp, bias, out, g, t, inb, wb);
        else launch_v<32, false, true, true>(bn, grid, st, in, wp, bias, out, g, t, inb, wb);
    } else if (g.thin) {
        if (g.reflect) launch_v<32, true, true>(bn, grid, st, in, wp, bias, out, g, t, inb, wb);
        else launch_v<32, false, true>(bn, grid, st, in, wp, bias, out, g, t, inb, wb);
    } else if (kc32) {
        if (g.reflect) launch_v<32, true, false>(bn, grid, st, in, wp, bias, out, g, t, inb, wb);
        else launch_v<32, false, false>(bn, grid, st, in, wp, bias, out, g, t, inb, wb);
    } else {
        if (g.reflect) launch_v<16, true, false>(bn, grid, st, in, wp, bias, out, g, t, inb, wb);
        else launch_v<16, false, false>(bn, grid, st, in, wp, bias, out, g, t, inb, wb);
    }
    ACG_CHECK_LAUNCH("igemm_conv_f32");
    acg_note_kernel("igemm_conv_f32<128,%d,KC=%d,REFLECT=%d,THIN=%d,X3=%d>", bn, (kc32 ? 32 : 16), g.reflect ? 1 : 0, g.thin ? 1 : 0, (g.thin && x3) ? 1 : 0);
    return ACG_OK;
}
