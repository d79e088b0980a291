// bf16-operand variants of the implicit-GEMM convolution and weight-gradient kernels
// (v_mfma_f32_32x32x16_bf16: bf16 x bf16 products, fp32 accumulate — 16x the fp32 matrix rate).
//
// Tensors stay fp32 in HBM.  Activations are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) in registers
// on their way into LDS; weights are pre-packed as bf16 once per optimiser step.  Same tap-list
// formulation, tiling and C/D epilogue as the fp32 kernels (conv_igemm.hip / conv_wgrad.hip); the LDS
// row slot is the same 32 bytes, now holding 16 k-values, so one ds_read_b128 is exactly one MFMA
// operand (lane l: row l&31, k = 8*(l>>5) .. +7).
//
// This is the throughput mode (`acg_set_conv_precision(ACG_PREC_BF16)`); the fp32 kernels remain the
// parity path (1e-3 bar).  Error model: each product carries two 2^-9 roundings, sums stay fp32.
#include "conv_internal.h"
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// SPLIT: operands as bf16 hi + lo, products hi*hi + hi*lo + lo*hi (fp32-grade: ~2^-16 per product).  The lo images
// live behind the hi images in LDS; packed weights carry their lo part at +w_lo_off elements.
//
// RP ("row patch", stride-1 gathers on grids whose rows are whole tiles): a stage is one ROW of the kernel instead of one
// tap.  The tile's 128 output pixels are consecutive in x, so the taps (ty, tx0 .. tx0 + 2) read the same BM + 2 input
// pixels shifted by one row of the LDS image each: the patch is fetched, split and stored once per kernel row (a 3x3 layer:
// 3 instead of 9 passes over the input through L2, a third of the barriers), and the MFMAs of the row's taps read it at
// row offsets 0 / 1 / 2 — 16 bytes in the [k8][row][8] planes, which keeps ds_read_b128 aligned and conflict-free.  Taps
// are consumed in list order, so every accumulator sees the MFMA sequence of the tap-per-stage kernel: bit-identical.
template <int BM, int BN, int WM, int WN, int KC, bool REFLECT, bool SPLIT, bool RP = false>
__global__ __launch_bounds__(256) void igemm_conv_bf16(const float *__restrict__ in, const __bf16 *__restrict__ wp,
                                                       const float *__restrict__ bias, float *__restrict__ out,
                                                       Geom g, Taps taps, unsigned in_bytes, unsigned w_bytes,
                                                       unsigned w_lo_bytes)
{
    constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int NKC = KC / 16;             // 16-wide k-chunks per stage
    constexpr int UPR = KC / 8;              // 8-float units per gathered pixel row
    constexpr int RPP = 256 / UPR;           // pixel rows per pass
    constexpr int AL = BM / RPP;             // units per thread per stage
    constexpr int BCH = NKC * BN * 2;        // 16-byte chunks in the B tile
    constexpr int BL = (BCH + 255) / 256;
    // LDS images are planes of 8 consecutive k: [k8][row][8 bf16], rows 16 B apart.  ds_read_b128 serves a wave in
    // four 16-lane groups ({0-3,12-15,20-27}, ...) over a 256-B bank row: 32 consecutive rows of one plane put every
    // group on 16 distinct 16-byte slots (a [row][16 k] image with 32-B rows is 2-way conflicted: 42 % of the LDS
    // cycles by SQ_LDS_BANK_CONFLICT).  ds_write_b128 goes by 8 contiguous lanes over a 128-B bank row: the plane
    // pads put the (rows x planes) / (columns x 2 planes) that a lane group stores on 8 distinct slots.
    constexpr int NK8 = KC / 8;
    constexpr int NTX = RP ? 3 : 1;          // taps per stage
    constexpr int ALX = RP ? AL + 1 : AL;    // gather passes per stage (row patch: one more for the NTX - 1 extra pixels)
    // plane strides (bf16 elements); the row patch's BM + 2 rows of 16 B are the same 2080 B as the padded BM rows of KC = 32
    constexpr int APL = RP ? (BM + NTX - 1) * 8 : BM * 8 + (NK8 == 4 ? 16 : NK8 == 8 ? 8 : 32), BPL = BN * 8 + 32;
    static_assert(WM * WN == 4 && AL >= 1 && MB >= 1 && NB >= 1, "tile config");
    static_assert(!RP || KC == 32, "row-patch stages are 32 channels deep");

    constexpr int NIMG = SPLIT ? 2 : 1;
    constexpr int A_IMG = NK8 * APL, B_IMG = NK8 * BPL; // one hi (or lo) image
    __shared__ __attribute__((aligned(16))) __bf16 As[NIMG * A_IMG];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[NTX * NIMG * B_IMG];
    __shared__ __attribute__((aligned(16))) unsigned out_rel[BM];  // byte offset of a tile row's output pixel from the tile's first, ~0u: no such pixel
    __shared__ float sred[(RP && SPLIT && BM / WM == 32 && BN / WN == 32 ? 2 : 1) * WM * BN];   // cross-wave fold of the per-tile statistics (Geom.stats) / norm-backward sums (Geom.ns_part)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    int swz = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    // phased launch (Geom.nphase): phase = the low two bits of the tile number
    const int ph = g.nphase ? (swz & 3) : 0;
    if (g.nphase) swz >>= 2;
    const int tbase = ph * 16, ntap = g.nphase ? ((g.ph_ntaps >> (8 * ph)) & 0xff) : taps.n;
    const int oy0 = g.nphase ? (ph >> 1) : g.oy0, ox0 = g.nphase ? (ph & 1) : g.ox0;
    const int tiles_n = g.ncols_pad / BN;
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int n0 = tile_n * BN;
    const long long m0 = (long long)tile_m * BM;
    const int GHW = g.GH * g.GW;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, w_bytes, 0x00020000);
    const int u = tid % UPR, rrow = tid / UPR;
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
    unsigned b_voff[BL];
    int b_lds[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int idx = tid + 256 * i;
        const int kc = idx / (BN * 2);
        const int rem = idx - kc * BN * 2;
        b_voff[i] = (unsigned)(((kc * g.ncols_pad + n0) * 16 + rem * 8) * 2);
        b_lds[i] = (kc * 2 + (rem & 1)) * BPL + (rem >> 1) * 8; // chunk = (column rem/2, k-half rem&1)
    }
    auto pix_off = [&](long long m) { // element offset of grid position m's output pixel
        const int n = (int)(m / GHW);
        const int r = (int)(m - (long long)n * GHW);
        const int gy = r / g.GW, gx = r - gy * g.GW;
        return (((long long)n * g.Hout + (gy * g.os + oy0)) * g.Wout + (gx * g.os + ox0)) * g.Cout;
    };
    const long long off0 = pix_off(m0);   // uniform; the rows of a tile lie less than 4 GiB behind it
    if (tid < BM) out_rel[tid] = m0 + tid < g.Mtot ? (unsigned)((pix_off(m0 + tid) - off0) * 4) : ~0u;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // row patch: the tile is pixels gx0 .. gx0 + BM - 1 of row gy0 of image img; stage = (channel chunk, tap group)
    const int img = (int)(m0 / GHW), gy0 = (int)(m0 - (long long)img * GHW) / g.GW, gx0 = (int)(m0 - (long long)img * GHW) - gy0 * g.GW;
    const int ngrp = RP ? (taps.ngrp >> (8 * ph)) & 0xff : 0;
    const int S = (RP ? ngrp : ntap) * (g.Cin / KC);
    f32x8 ra[ALX];
    u32x4 rb[NTX][BL], rbl[NTX][BL]; // 8 bf16 each (hi, lo)

    auto load_stage = [&](int s) {
        if constexpr (RP) {
            const int cc = s / ngrp;
            const int c0 = cc * KC;
            const int gd = taps.gpk[ph * 16 + (s - cc * ngrp)]; // first tap | taps << 8 | smallest dx << 16
            const int first = tbase + (gd & 0xff), nt = (gd >> 8) & 0xff, tx0 = (gd << 8) >> 24;
            const int pk0 = taps.pk[first];
            int iy = gy0 + ((pk0 << 24) >> 24);
            bool rok = true;
            if (REFLECT) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
            } else {
                rok = (unsigned)iy < (unsigned)g.Hin;
            }
            const int rowbase = (img * g.Hin + iy) * g.Win;
#pragma unroll
            for (int j = 0; j < ALX; ++j) {
                int ix = gx0 + tx0 + rrow + RPP * j;
                bool ok = rok && (j < AL || rrow < nt - 1);
                if (REFLECT) {
                    ix = ix < 0 ? -ix : ix;
                    ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
                } else {
                    ok = ok && (unsigned)ix < (unsigned)g.Win;
                }
                const unsigned off = (unsigned)((rowbase + ix) * g.Cin + c0 + 8 * u) * 4u;
                const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off, ok), 0, 0);
                const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off + 16u, ok), 0, 0);
                const f32x4 flo = __builtin_bit_cast(f32x4, lo), fhi = __builtin_bit_cast(f32x4, hi);
                ra[j] = (f32x8){flo[0], flo[1], flo[2], flo[3], fhi[0], fhi[1], fhi[2], fhi[3]};
            }
#pragma unroll
            for (int q = 0; q < NTX; ++q)
                if (q < nt) {
                    const int tw = taps.pk[first + q] >> 16;
                    const unsigned soff = (unsigned)(((tw * (g.Cin >> 4) + (c0 >> 4)) * g.ncols_pad) * 16) * 2u;
#pragma unroll
                    for (int i = 0; i < BL; ++i)
                        if (tid + 256 * i < BCH) {
                            rb[q][i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff, 0);
                            if (SPLIT) rbl[q][i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff + w_lo_bytes, 0);
                        }
                }
            return;
        }
        const int cc = s / ntap;
        const int t = s - cc * ntap;
        const int c0 = cc * KC;
        const int pk = taps.pk[tbase + t]; // scalar load: (dy, dx, weight slab) of this stage
        const int ty = (pk << 24) >> 24, tx = (pk << 16) >> 24, tw = pk >> 16;
#pragma unroll
        for (int j = 0; j < AL; ++j) {
            int pix;
            bool ok = a_ok[j];
            if (REFLECT) {
                int iy = a_by[j] + ty, ix = a_bx[j] + tx;
                iy = iy < 0 ? -iy : iy;
                iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                ix = ix < 0 ? -ix : ix;
                ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
                pix = (a_row[j] + iy) * g.Win + ix;
            } else {
                const int iy = a_by[j] + ty, ix = a_bx[j] + tx;
                ok = ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
                pix = a_row[j] + ty * g.Win + tx;
            }
            const unsigned off = (unsigned)(pix * g.Cin + c0 + 8 * u) * 4u;
            const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off, ok), 0, 0);
            const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off + 16u, ok), 0, 0);
            const f32x4 flo = __builtin_bit_cast(f32x4, lo), fhi = __builtin_bit_cast(f32x4, hi);
            ra[j] = (f32x8){flo[0], flo[1], flo[2], flo[3], fhi[0], fhi[1], fhi[2], fhi[3]};
        }
        const unsigned soff = (unsigned)(((tw * (g.Cin >> 4) + (c0 >> 4)) * g.ncols_pad) * 16) * 2u;
#pragma unroll
        for (int i = 0; i < BL; ++i)
            if (tid + 256 * i < BCH) {
                rb[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff, 0);
                if (SPLIT) rbl[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff + w_lo_bytes, 0);
            }
    };

    // Norm-backward sums (Geom.ns_part, below): only the one-column-block row-patch instance carries the code (the data gradient of
    // the 3x3 32 -> 64 layer); the norm input x at the tile's output positions is requested during the LAST stage's MFMAs
    constexpr bool NS = RP && SPLIT && MB == 1 && NB == 1;
    float ns_xv[NS ? 16 : 1];
    auto ns_load = [&]() {
        if constexpr (NS) {
            const __amdgpu_buffer_rsrc_t rnx = __builtin_amdgcn_make_buffer_rsrc((void *)(g.ns_x + off0), 0, 0xFFFFFFF0u, 0x00020000);
            const int co = n0 + wn * TN + (lane & 31);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const u32x4 rel = *(const u32x4 *)&out_rel[wm * TM + 8 * r4 + 4 * (lane >> 5)];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ns_xv[4 * r4 + q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rnx, acg_masked_off(rel[q] + (unsigned)co * 4u, co < g.Cout && rel[q] != ~0u), 0, 0));
            }
        }
    };
    load_stage(0);
    for (int s = 0; s < S; ++s) {
        int gfirst = 0, gnt = 1, gtx0 = 0;   // this stage's tap group (row patch)
        if constexpr (RP) {
            const int gd = taps.gpk[ph * 16 + s % ngrp];
            gfirst = tbase + (gd & 0xff); gnt = (gd >> 8) & 0xff; gtx0 = (gd << 8) >> 24;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ALX; ++j) {
            if (RP && j == AL && rrow >= NTX - 1) continue;
            const int a_at = u * APL + (rrow + RPP * j) * 8;
            const float v[8] = {ra[j][0], ra[j][1], ra[j][2], ra[j][3], ra[j][4], ra[j][5], ra[j][6], ra[j][7]};
            if (SPLIT) {
                acg_u32x4 hi, lo;
                acg_split8(v, hi, lo);
                *(acg_u32x4 *)&As[a_at] = hi;
                *(acg_u32x4 *)&As[A_IMG + a_at] = lo;
            } else {
                *(acg_u32x4 *)&As[a_at] = acg_round8(v);
            }
        }
#pragma unroll
        for (int q = 0; q < NTX; ++q)
            if (q < gnt) {
#pragma unroll
                for (int i = 0; i < BL; ++i)
                    if (tid + 256 * i < BCH) {
                        *(u32x4 *)&Bs[q * NIMG * B_IMG + b_lds[i]] = rb[q][i];
                        if (SPLIT) *(u32x4 *)&Bs[q * NIMG * B_IMG + B_IMG + b_lds[i]] = rbl[q][i];
                    }
            }
        __syncthreads();
        if (s + 1 < S) load_stage(s + 1);
        else if (NS && g.ns_part != nullptr) ns_load();
        // (a run-time trip count on purpose: with `if (q < gnt)` bodies the accumulators are merged through VGPRs and every
        // tap pays 2 x 32 v_accvgpr moves)
#pragma unroll 1
        for (int q = 0; q < gnt; ++q) {
        // row patch: tap q of the group reads the patch (its dx - the group's smallest dx) rows further on
        const int dxo = RP ? ((taps.pk[gfirst + q] << 16) >> 24) - gtx0 : 0;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            bf16x8 a[MB], b[NB], al[MB], bl[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int at = (kc * 2 + (lane >> 5)) * APL + (wm * TM + i * 32 + (lane & 31) + dxo) * 8;
                a[i] = *(const bf16x8 *)&As[at];
                if (SPLIT) al[i] = *(const bf16x8 *)&As[A_IMG + at];
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int bt = q * NIMG * B_IMG + (kc * 2 + (lane >> 5)) * BPL + (wn * TN + j * 32 + (lane & 31)) * 8;
                b[j] = *(const bf16x8 *)&Bs[bt];
                if (SPLIT) bl[j] = *(const bf16x8 *)&Bs[B_IMG + bt];
            }
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (SPLIT) { // small cross terms first, the leading term last
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], b[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                }
        }
        }
    }

    if (g.stats != nullptr) {
        // Per-tile (mean, M2) of every output column over the tile's 128 pixels, for the InstanceNorm that follows (the norm's
        // statistics pass then merges these partials instead of re-reading the tensor; full tiles inside one image and
        // act == NONE are guaranteed by the launcher).  A column's 128 values sit in 16 x MB registers of the two lanes
        // l, l + 32 of the WM waves that share the column range: register sums, one shuffle, one LDS fold; twice (mean,
        // then squared deviations, like the two-pass formula of modules.py:83-97).
        float mu[NB];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int co = n0 + wn * TN + j * 32 + (lane & 31);
                const float bv = (bias != nullptr && co < g.Cout) ? bias[co] : 0.f;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r] + bv;
                        s += pass == 0 ? v : (v - mu[j]) * (v - mu[j]);
                    }
                s += __shfl_xor(s, 32);
                if (lane < 32) sred[wm * BN + wn * TN + j * 32 + lane] = s;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int cl = wn * TN + j * 32 + (lane & 31);
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) s += sred[w * BN + cl];
                if (pass == 0) {
                    mu[j] = s * (1.f / BM);
                } else if (wm == 0 && lane < 32 && n0 + cl < g.Cout) {
                    const long long tile = m0 / BM, tpi = (long long)GHW / BM;     // tile index, tiles per image in this launch
                    const long long chunk = (tile / tpi) * g.stats_cpi + g.stats_chunk0 + ph * tpi + tile % tpi;
                    float *o = g.stats + chunk * 2 * g.Cout + n0 + cl;
                    o[0] = mu[j];
                    o[g.Cout] = s;
                }
            }
            __syncthreads();
        }
    }
    if constexpr (NS) {
        if (g.ns_part != nullptr) {
            // Norm-backward sums (acg_conv2d_bwd_data_sums, round 6): this launch is a data gradient whose output is the gradient
            // w.r.t. the OUTPUT of a norm (+ ReLU) with input ns_x — the first pass of that norm's backward, sum gy and sum gy * xhat
            // per channel over the tile's 128 pixels (gy = dx * act'(y), the mask recomputed from x like norm_bwd_partial<.., 1>),
            // leaves from here.  Whole tiles inside one image, no bias / activation (launcher).
            const bool ns_relu = g.ns_act == ACG_ACT_RELU;
            const int co = n0 + wn * TN + (lane & 31);
            const bool cok = co < g.Cout;
            float mu = 0.f, rs = 0.f, ga = 1.f, be = 1.f;
            if (cok) {
                mu = g.ns_mean[img * g.Cout + co];
                rs = g.ns_rstd[img * g.Cout + co];
                if (ns_relu) { ga = g.ns_gamma[g.ns_gstride * img + co]; be = g.ns_beta[g.ns_gstride * img + co]; }
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float xh = (ns_xv[r] - mu) * rs;
                const float gy = (ns_relu && !(xh * ga + be > 0.f)) ? 0.f : acc[0][0][r];   // same expression as norm_apply_kernel
                s1 += gy;
                s2 += gy * xh;
            }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lane < 32) {
                sred[wm * BN + wn * TN + lane] = s1;
                sred[WM * BN + wm * BN + wn * TN + lane] = s2;
            }
            __syncthreads();
            if (tid < BN && n0 + tid < g.Cout) {
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { a += sred[w * BN + tid]; b += sred[WM * BN + w * BN + tid]; }
                float *o = g.ns_part + ((m0 / BM) * 2) * g.Cout + n0 + tid;   // part[N][GHW / BM][2][Cout], tile-major like m0
                o[0] = a;
                o[g.Cout] = b;
            }
        }
    }
    // Stores: 32 lanes x 4 B = one 128-byte line of an output pixel per half wave.  Branch-free: bias and activation in a
    // pass over the accumulators, then buffer stores relative to the tile's first pixel whose masked lanes (no such row /
    // column) carry an out-of-range offset — a predicated 64-bit store per element cost ~45 instructions and a branch.
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
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + off0), 0, 0xFFFFFFF0u, 0x00020000);
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            // accumulator registers 4 r4 .. 4 r4 + 3 are tile rows 8 r4 + 4 (lane >> 5) + 0 .. 3
            const u32x4 rel = *(const u32x4 *)&out_rel[wm * TM + i * 32 + 8 * r4 + 4 * (lane >> 5)];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int co = n0 + wn * TN + j * 32 + (lane & 31);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#ifdef ACG_ABL_NOSTORE   // diagnostic build: keep the arithmetic alive, store (almost) nothing
                    const bool ok = rel[q] != ~0u && co < g.Cout && acc[i][j][4 * r4 + q] == 12345.678f;
#else
                    const bool ok = rel[q] != ~0u && co < g.Cout;
#endif
                    const float v = acc[i][j][4 * r4 + q]; // (a bit cast of the vector-element lvalue itself reads element 0)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, acg_masked_off(rel[q] + (unsigned)co * 4u, ok), 0, 0);
                }
            }
        }
}

template <int KC, bool REFLECT, bool SPLIT, bool RP = false>
static void launch_bf16_kc(int bn, dim3 grid, hipStream_t st, const float *in, const __bf16 *wp, const float *bias,
                           float *out, const Geom &g, const Taps &t, unsigned inb, unsigned wb, unsigned wlo)
{
    dim3 block(256);
    if (bn == 128)
        hipLaunchKernelGGL((igemm_conv_bf16<128, 128, 2, 2, KC, REFLECT, SPLIT, RP>), grid, block, 0, st, in, wp, bias, out, g, t, inb, wb, wlo);
    else if (bn == 64)
        hipLaunchKernelGGL((igemm_conv_bf16<128, 64, 2, 2, KC, REFLECT, SPLIT, RP>), grid, block, 0, st, in, wp, bias, out, g, t, inb, wb, wlo);
    else
        hipLaunchKernelGGL((igemm_conv_bf16<128, 32, 4, 1, KC, REFLECT, SPLIT, RP>), grid, block, 0, st, in, wp, bias, out, g, t, inb, wb, wlo);
}

// Row-patch stages (see the kernel): do the tap lists of this launch decompose into runs of <= 3 taps on one input row, and
// is every tile 128 consecutive pixels of one grid row?  Fills t.gpk / t.ngrp.
static bool rp_groups(const Geom &g, Taps &t)
{
    // (the four sub-pixel phases of a stride-2 data gradient have 1 / 2 / 2 / 4 taps: too little to share for the bigger
    // LDS image and the lower occupancy — measured 0.48 -> 0.54 ms on the 128 -> 64 channel layers)
    if (g.is != 1 || g.GW % 128 != 0 || g.Mtot % 128 != 0 || g.Cin % 32 != 0 || g.thin || g.nphase) return false;
    t.ngrp = 0;
    bool shared = false;
    const int nph = g.nphase ? 4 : 1;
    for (int ph = 0; ph < nph; ++ph) {
        const int base = g.nphase ? 16 * ph : 0, ntap = g.nphase ? (g.ph_ntaps >> (8 * ph)) & 0xff : t.n;
        int ng = 0;
        for (int i = 0; i < ntap;) {
            int j = i + 1, lo = t.dx[base + i], hi = lo;
            while (j < ntap && j - i < 3 && t.dy[base + j] == t.dy[base + i]) {
                const int d = t.dx[base + j], l2 = d < lo ? d : lo, h2 = d > hi ? d : hi;
                if (h2 - l2 > 2) break;
                lo = l2; hi = h2; ++j;
            }
            if (ng == 16 || i > 255 || lo < -128 || lo > 127) return false;
            t.gpk[16 * ph + ng++] = i | ((j - i) << 8) | ((lo & 0xff) << 16);
            i = j;
        }
        shared |= ng < ntap;
        t.ngrp |= ng << (8 * ph);
    }
    return shared; // one tap per stage everywhere: nothing to share
}

// does this launch go to the wave-specialised bf16x3 kernel (conv_x3.hip)?  128-column tiles, 32-channel stages
bool acg_igemm_uses_ws(const Geom &g)
{
    static const bool no_ws = acg_debug_switch("ACG_NO_WS"); // A/B switch
    return !no_ws && g_acg_precision == ACG_PREC_BF16X3 && g_acg_conv_impl == ACG_IMPL_MFMA && !g.thin && g.Cout >= 128 &&
           g.Cin % 32 == 0;
}

// n_w_elems: element count of the packed weight array (hi part); the lo part of BF16X3 sits right behind it
int acg_igemm_bf16_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g0, const Taps &t,
                          int bn, long long n_w_elems, hipStream_t st)
{
    Geom g = g0;
    g.thin = 0;
    dim3 grid(acg_cdiv(g.Mtot, 128) * (g.ncols_pad / bn) * (g.nphase ? 4 : 1));
    const __bf16 *w = (const __bf16 *)wp;
    ACG_REQUIRE(g.nphase == 0 || (g.nphase == 4 && !acg_igemm_uses_ws(g0) && g.Mtot % 128 == 0),
                "igemm_conv_bf16: phased launches are for the generic tile, whole tiles per phase");
    const bool split = g_acg_precision == ACG_PREC_BF16X3;
    const long long nimg = g.Mtot / ((long long)g.GH * g.GW);
    const long long in_bytes = nimg * g.Hin * g.Win * g.Cin * 4;
    const long long w_bytes = n_w_elems * 2 * (split ? 2 : 1);
    ACG_REQUIRE(in_bytes < (1LL << 32) && w_bytes < (1LL << 32), "igemm_conv_bf16: operand exceeds the 4 GiB buffer-addressing limit");
    const unsigned inb = (unsigned)in_bytes, wb = (unsigned)w_bytes, wlo = (unsigned)(n_w_elems * 2);
#define BF16_DISPATCH(KCV, SP)                                                                              \
    do {                                                                                                    \
        if (g.reflect) launch_bf16_kc<KCV, true, SP>(bn, grid, st, in, w, bias, out, g, t, inb, wb, wlo);   \
        else launch_bf16_kc<KCV, false, SP>(bn, grid, st, in, w, bias, out, g, t, inb, wb, wlo);            \
    } while (0)
    // only the row pipeline (and the pre-split trunk kernels, which do not come through here) emits norm-backward sums: a launch
    // that asked for them must not land on a kernel that ignores Geom.ns_part (the caller's `part` buffer is uninitialised)
    // the row pipeline and the generic tile (whole 128-pixel tiles inside one image, plain data gradient) emit norm-backward sums;
    // a launch that asked for them must not land on a kernel that ignores Geom.ns_part (the caller's `part` buffer is uninitialised)
    const bool ns_tile = split && !acg_igemm_uses_ws(g0) && g0.nphase == 0 && g0.stats == nullptr && bias == nullptr && g0.act == ACG_ACT_NONE &&
                         g0.os == 1 && ((long long)g0.GH * g0.GW) % 128 == 0 && g0.ns_x != nullptr && g0.ns_mean != nullptr && g0.ns_rstd != nullptr &&
                         g0.ns_mask == nullptr && (g0.ns_act == ACG_ACT_NONE || (g0.ns_act == ACG_ACT_RELU && g0.ns_gamma != nullptr && g0.ns_beta != nullptr)) &&
                         (g0.ns_gstride == 0 || g0.ns_gstride >= g0.Cout) && !acg_conv_patchn_ok(g0, t) && !acg_conv_patch16_ok(g0, t) &&
                         bn == 32 && g0.Cin % 32 == 0 && !g0.reflect && !acg_debug_switch("ACG_NO_RP") && [&]() { Taps tq = t; return rp_groups(g, tq); }();
    ACG_REQUIRE(g0.ns_part == nullptr || (split && !acg_igemm_uses_ws(g0) && (acg_conv_rows_ok(g0, t) || ns_tile)),
                "igemm_conv_bf16: norm-backward sums requested on a geometry neither the row pipeline nor the generic tile takes");
    if (acg_igemm_uses_ws(g0)) {
        ACG_REQUIRE(g0.stats == nullptr || (g0.stats_chunk0 == 0 && g0.stats_cpi == (int)(((long long)g0.GH * g0.GW) / 128)),
                    "igemm_conv_x3_ws: per-tile statistics of a phased launch");
        return acg_igemm_x3_ws_launch(in, wp, bias, out, g0, t, n_w_elems, st, g0.stats);
    }
    if (split && acg_conv_rows_ok(g0, t)) return acg_conv_rows_launch(in, wp, bias, out, g0, t, n_w_elems, st);   // persistent row pipeline
    if (split && acg_conv_patchn_ok(g0, t))   // C4 output: the N-packed weights sit behind the regular hi + lo images
        return acg_conv_patchn_launch(in, (const __bf16 *)wp + 2 * n_w_elems, bias, out, g0, t, st);
    static const bool no_patch = acg_debug_switch("ACG_NO_PATCH"); // A/B switch
    if (!no_patch && acg_conv_patch16_ok(g0, t)) {
        ACG_REQUIRE(g0.stats == nullptr, "conv_patch16: no statistics epilogue");
        return acg_conv_patch16_launch(in, wp, bias, out, g0, t, n_w_elems, st);
    }
    ACG_REQUIRE(g0.stats == nullptr || (((long long)g0.GH * g0.GW) % 128 == 0 && g0.act == ACG_ACT_NONE && g0.Mtot % 128 == 0),
                "igemm_conv_bf16: per-tile statistics need whole 128-pixel tiles per image and no activation");
    ACG_REQUIRE(g0.fold_p == 0, "igemm_conv_bf16: the fold bypass is implemented by the wave-specialised kernel only");
    static const bool no_rp = acg_debug_switch("ACG_NO_RP"); // A/B switch
    Taps tr = t;
    if (!no_rp && split && rp_groups(g, tr)) {
        if (g.reflect) launch_bf16_kc<32, true, true, true>(bn, grid, st, in, w, bias, out, g, tr, inb, wb, wlo);
        else launch_bf16_kc<32, false, true, true>(bn, grid, st, in, w, bias, out, g, tr, inb, wb, wlo);
        ACG_CHECK_LAUNCH("igemm_conv_bf16 (row patch)");
        acg_note_kernel("igemm_conv_bf16<128,%d,KC=32,REFLECT=%d,SPLIT=1,RP=1>", bn, g.reflect ? 1 : 0);
        return ACG_OK;
    }
    if (split) { // hi+lo images double the LDS: 32-channel stages keep 3-4 blocks per CU
        if (g.Cin % 32 == 0) BF16_DISPATCH(32, true);
        else BF16_DISPATCH(16, true);
    } else if (g.Cin % 64 == 0) BF16_DISPATCH(64, false);
    else if (g.Cin % 32 == 0) BF16_DISPATCH(32, false);
    else BF16_DISPATCH(16, false);
#undef BF16_DISPATCH
    ACG_CHECK_LAUNCH("igemm_conv_bf16");
    acg_note_kernel("igemm_conv_bf16<128,%d,KC=%d,REFLECT=%d,SPLIT=%d>", bn,
                    split ? (g.Cin % 32 == 0 ? 32 : 16) : (g.Cin % 64 == 0 ? 64 : (g.Cin % 32 == 0 ? 32 : 16)), g.reflect ? 1 : 0, split ? 1 : 0);
    return ACG_OK;
}

// ------------------------------------------------------------------------------------------------
// weight gradient, bf16 operands.  GEMM K = pixels: both operands need 8 consecutive PIXELS per lane for a
// fixed channel, i.e. the transpose of the NHWC rows.  The transpose happens in registers on the way into
// LDS: a thread loads the same channel quad of 8 consecutive pixels (8 x float4, each instruction covering
// whole 512-B pixel rows across the wave) and writes four 16-byte [channel][8 pixels] fragments.
// LDS images: Xs[ci][KP + 8], Ds[co][KP + 8] (bf16; +16 B row pad -> conflict-free ds_read_b128).
// ------------------------------------------------------------------------------------------------
// SPLIT: both operands as bf16 hi + lo (lo images behind the hi images), x_lo*d_hi + x_hi*d_lo + x_hi*d_hi.
// NT: consecutive taps per workgroup.  They share the gradient-side tile (one load, split and LDS image for NT taps) and each
// brings its own gathered tile: per tap and stage a workgroup pulls (BCI + BCO / NT) instead of (BCI + BCO) channel rows
// through L2 — these kernels move 4-5 GB per launch through L2 for 0.8 GB of operands, and that, not HBM, is their limit.
template <int BCI, int BCO, int WI, int WJ, int WK, int KP, bool SPLIT, int NT = 1>
__global__ __launch_bounds__(256) void wgrad_bf16(const float *__restrict__ x, const float *__restrict__ dy,
                                                  float *__restrict__ part, WGeom g, Taps taps, unsigned x_bytes,
                                                  unsigned d_bytes)
{
    constexpr int TI = BCI / WI, TJ = BCO / WJ, MI = TI / 32, MJ = TJ / 32;
    // LDS rows hold KP pixels of one channel.  KP == 64 (128-B rows): the eight 16-byte pixel groups of a row are
    // XOR-permuted by g(row) = (row>>2 & 3) | ((row>>1 ^ row>>4) & 1) << 2, which makes BOTH the transposed
    // ds_write_b128 (8-lane groups, rows 4 apart) and the fragment ds_read_b128 (16-lane groups) conflict-free — a
    // padded linear row leaves the stores 4-way conflicted (60 % of the LDS cycles).  Other KP: +16 B row pad.
    constexpr bool SWZ = KP == 64;
    constexpr int RS = SWZ ? KP : KP + 8;               // LDS row stride (elements)
#define ACG_WG_AT(row, pgq) ((row) * RS + (((pgq) ^ (SWZ ? ((((row) >> 2) & 3) | (((((row) >> 1) ^ ((row) >> 4)) & 1) << 2)) : 0)) * 8))
    constexpr int XU = (KP / 8) * (BCI / 4), DU = (KP / 8) * (BCO / 4); // 8-pixel x 4-channel units
    constexpr int XL = (NT * XU + 255) / 256, DL = (DU + 255) / 256;   // x units: tap-major, unit U = tid + 256 l -> tap U / XU
    constexpr int KW = KP / WK;                         // pixels of a stage per wave
    constexpr int DOFF = (NT * XU + DU <= 256) ? NT * XU : 0; // threads [XU, XU+DU) load the dy units when both fit
    static_assert(WI * WJ * WK == 4 && MI >= 1 && MJ >= 1 && KW % 16 == 0, "tile config");
    static_assert(NT == 1 || (WK == 1 && XU <= 256), "several taps per workgroup: no split-K inside the block");

    constexpr int NIMG = SPLIT ? 2 : 1;
    constexpr int XIMG = NIMG * BCI * RS;               // one tap's gathered image(s)
    __shared__ __attribute__((aligned(16))) __bf16 Xs[NT * XIMG];
    __shared__ __attribute__((aligned(16))) __bf16 Ds[NIMG * BCO * RS];

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
    const int tap = (b % (taps.n / NT)) * NT;   // first of this workgroup's NT taps
    const int split = b / (taps.n / NT);
    const int ci0 = tci * BCI, co0 = tco * BCO;
    const long long mbeg = (long long)split * g.m_per_split;
    long long mend = mbeg + g.m_per_split;
    if (mend > g.Mtot) mend = g.Mtot;
    const int GHW = g.Hg * g.Wg;

    f32x16 acc[NT][MI][MJ];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < MJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

    f32x4 rx[XL][8], rd[DL][8];
    const bool do_bias = g.bias_from != 0 && tap == 0 && (g.bias_from == 1 ? tci == 0 : tco == 0);
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd_ = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, d_bytes, 0x00020000);

    // first pixel of each of this thread's 8-pixel units, decoded ONCE and advanced by KP per stage with carries
    int un[XL], uy[XL], ux[XL];
#pragma unroll
    for (int l = 0; l < XL; ++l) {
        const int unit = (tid + 256 * l) % XU;
        const long long m = mbeg + (unit / (BCI / 4)) * 8;
        const long long mm = m < g.Mtot ? m : 0;
        un[l] = (int)(mm / GHW);
        const int rr = (int)(mm - (long long)un[l] * GHW);
        uy[l] = rr / g.Wg;
        ux[l] = rr - uy[l] * g.Wg;
    }

    auto load_stage = [&](long long k0) {
#pragma unroll
        for (int l = 0; l < XL; ++l) {
            const int U = tid + 256 * l, tt = NT == 1 ? 0 : U / XU, unit = NT == 1 ? U : U - tt * XU;
            const int ty = taps.dy[tap + (tt < NT ? tt : 0)], tx = taps.dx[tap + (tt < NT ? tt : 0)];
            const int c4 = unit % (BCI / 4), pg = unit / (BCI / 4);
            const int ci = ci0 + c4 * 4;
            const long long m = k0 + pg * 8;
            int n = un[l], gy = uy[l], gx = ux[l];
            const bool uok = U < NT * XU && ci < g.Cin;
            // the 8 pixels of a unit are consecutive output positions: when no lane's unit crosses a row end (always
            // so for W % 8 == 0), the row part of the address is formed once and each pixel costs a few VALU ops
            const bool in_row = gx + 8 <= g.Wg && m + 8 <= mend;
            const int ixf = gx * g.is + tx; // first input column of the run; its last is ixf + 7 * is
            const bool interior = in_row && ixf >= 0 && ixf + 7 * g.is < g.Win;
            if (__builtin_amdgcn_ballot_w64(uok && !interior) == 0) {
                // no lane's run touches a row end or the padding columns (15 of 16 runs of a 128-wide map): one masked
                // VGPR offset per unit, the pixel step rides in the scalar offset — no per-pixel VALU work at all
                int iy = gy * g.is + ty;
                bool oky = uok;
                if (g.reflect) {
                    iy = iy < 0 ? -iy : iy;
                    iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                } else {
                    oky = oky && (unsigned)iy < (unsigned)g.Hin;
                }
                const unsigned off = acg_masked_off((unsigned)(((n * g.Hin + iy) * g.Win + ixf) * g.Cin + ci) * 4u, oky);
                const unsigned xstep = (unsigned)(g.is * g.Cin) * 4u;
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    rx[l][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx_, off, p * xstep, 0));
            } else if (__builtin_amdgcn_ballot_w64(uok && !in_row) == 0) {
                int iy = gy * g.is + ty;
                bool oky = uok && in_row;
                if (g.reflect) {
                    iy = iy < 0 ? -iy : iy;
                    iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                } else {
                    oky = oky && (unsigned)iy < (unsigned)g.Hin;
                }
                const int rowbase = (n * g.Hin + iy) * g.Win;
                const int ix0 = gx * g.is + tx;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    int ix = ix0 + p * g.is;
                    bool ok = oky;
                    if (g.reflect) {
                        ix = ix < 0 ? -ix : ix;
                        ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
                    } else {
                        ok = ok && (unsigned)ix < (unsigned)g.Win;
                    }
                    const unsigned off = acg_masked_off((unsigned)((rowbase + ix) * g.Cin + ci) * 4u, ok);
                    rx[l][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx_, off, 0, 0));
                }
            } else {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    int iy = gy * g.is + ty, ix = gx * g.is + tx;
                    bool ok = uok && m + p < mend;
                    if (g.reflect) {
                        iy = iy < 0 ? -iy : iy;
                        iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                        ix = ix < 0 ? -ix : ix;
                        ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
                    } else {
                        ok = ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
                    }
                    const unsigned off = acg_masked_off((unsigned)(((n * g.Hin + iy) * g.Win + ix) * g.Cin + ci) * 4u, ok);
                    rx[l][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx_, off, 0, 0));
                    if (++gx == g.Wg) { gx = 0; if (++gy == g.Hg) { gy = 0; ++n; } }
                }
            }
            ux[l] += KP;
            while (ux[l] >= g.Wg) { ux[l] -= g.Wg; if (++uy[l] == g.Hg) { uy[l] = 0; ++un[l]; } }
        }
#pragma unroll
        for (int l = 0; l < DL; ++l) {
            const int unit = DOFF ? (tid >= DOFF ? tid - DOFF : DU) : tid + 256 * l;
            const int c4 = unit % (BCO / 4), pg = unit / (BCO / 4);
            const int co = co0 + c4 * 4;
            const long long m = k0 + pg * 8;
            const bool uok = unit < DU && co < g.Cg;
            const bool full = m + 8 <= mend;
            const unsigned base = (unsigned)((int)m * g.Cg + co) * 4u, step = (unsigned)g.Cg * 4u;
            if (__builtin_amdgcn_ballot_w64(uok && !full) == 0) { // whole units: one masked offset, scalar pixel step
                const unsigned off = acg_masked_off(base, uok);
#pragma unroll
                for (int p = 0; p < 8; ++p)
                    rd[l][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd_, off, p * step, 0));
            } else {
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const unsigned off = acg_masked_off(base + p * step, uok && (full || m + p < mend));
                    rd[l][p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd_, off, 0, 0));
                }
            }
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int l = 0; l < XL; ++l) {
            const int U = tid + 256 * l, tt = NT == 1 ? 0 : U / XU, unit = NT == 1 ? U : U - tt * XU;
            if (U < NT * XU) {
                const int c4 = unit % (BCI / 4), pg = unit / (BCI / 4);
                __bf16 *Xt = Xs + tt * XIMG;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v[8];
#pragma unroll
                    for (int p = 0; p < 8; ++p) v[p] = rx[l][p][c];
                    if (SPLIT) {
                        acg_u32x4 hi, lo;
                        acg_split8(v, hi, lo);
                        *(acg_u32x4 *)&Xt[ACG_WG_AT(c4 * 4 + c, pg)] = hi;
                        *(acg_u32x4 *)&Xt[BCI * RS + ACG_WG_AT(c4 * 4 + c, pg)] = lo;
                    } else {
                        *(acg_u32x4 *)&Xt[ACG_WG_AT(c4 * 4 + c, pg)] = acg_round8(v);
                    }
                }
            }
        }
#pragma unroll
        for (int l = 0; l < DL; ++l) {
            const int unit = DOFF ? (tid >= DOFF ? tid - DOFF : DU) : tid + 256 * l;
            if (unit < DU) {
                const int c4 = unit % (BCO / 4), pg = unit / (BCO / 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v[8];
#pragma unroll
                    for (int p = 0; p < 8; ++p) v[p] = rd[l][p][c];
                    if (SPLIT) {
                        acg_u32x4 hi, lo;
                        acg_split8(v, hi, lo);
                        *(acg_u32x4 *)&Ds[ACG_WG_AT(c4 * 4 + c, pg)] = hi;
                        *(acg_u32x4 *)&Ds[BCO * RS + ACG_WG_AT(c4 * 4 + c, pg)] = lo;
                    } else {
                        *(acg_u32x4 *)&Ds[ACG_WG_AT(c4 * 4 + c, pg)] = acg_round8(v);
                    }
                }
            }
        }
    };

    if (mbeg < mend) load_stage(mbeg);
    for (long long k0 = mbeg; k0 < mend; k0 += KP) {
        __syncthreads();
        store_stage();
        if (do_bias) { // fp32 column sums of the operand rows as they stream through registers (exact, not bf16)
            if (g.bias_from == 1) {
#pragma unroll
                for (int l = 0; l < DL; ++l)
#pragma unroll
                    for (int p = 0; p < 8; ++p) bsum += rd[l][p];
            } else { // the first tap's units are the first XU threads of pass 0
#pragma unroll
                for (int p = 0; p < 8; ++p) bsum += rx[0][p];
            }
        }
        __syncthreads();
        if (k0 + KP < mend) load_stage(k0 + KP);
#pragma unroll
        for (int kk = wk * KW; kk < (wk + 1) * KW; kk += 16) {
            bf16x8 bb[MJ], bl[MJ];
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                const int bt = ACG_WG_AT(wj * TJ + j * 32 + (lane & 31), (kk >> 3) + (lane >> 5));
                bb[j] = *(const bf16x8 *)&Ds[bt];
                if (SPLIT) bl[j] = *(const bf16x8 *)&Ds[BCO * RS + bt];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bf16x8 a[MI], al[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int at = t * XIMG + ACG_WG_AT(wi * TI + i * 32 + (lane & 31), (kk >> 3) + (lane >> 5));
                    a[i] = *(const bf16x8 *)&Xs[at];
                    if (SPLIT) al[i] = *(const bf16x8 *)&Xs[BCI * RS + at];
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < MJ; ++j) {
                        if (SPLIT) {
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bb[j], acc[t][i][j], 0, 0, 0);
                            acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bl[j], acc[t][i][j], 0, 0, 0);
                        }
                        acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bb[j], acc[t][i][j], 0, 0, 0);
                    }
            }
        }
    }

    if (do_bias) { // threads of one unit column share a channel quad: fold them in fixed order through LDS
        __syncthreads();
        float *red = (float *)Xs;
        const bool fromD = g.bias_from == 1;
        const int q4 = (fromD ? BCO : BCI) / 4, nun = fromD ? DU : XU, off = fromD ? DOFF : 0;
        // unit index of this thread for the chosen operand (first pass only: XL == DL == 1 whenever units <= 256)
        const int unit = fromD ? (DOFF ? (tid >= DOFF ? tid - DOFF : nun) : tid) : tid;
        static_assert((XL == 1 || NT > 1) && DL == 1, "bias fusion assumes one unit per thread per operand (per tap)");
        (void)off;
        if (unit < nun) *(f32x4 *)&red[unit * 4] = bsum;
        __syncthreads();
        if (tid < q4) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int r = 0; r < nun / q4; ++r) s += *(const f32x4 *)&red[(r * q4 + tid) * 4];
            const int cpad = fromD ? g.CoP : g.CiP;
            const int c0 = fromD ? co0 : ci0;
            *(f32x4 *)&g.bias_part[(long long)split * cpad + c0 + tid * 4] = s;
        }
        __syncthreads();
    }

    if (WK > 1) {
        static_assert(WK == 1 || (MI == 1 && MJ == 1), "split-K-in-block only for 32x32 wave tiles");
        static_assert(WK == 1 || sizeof(__bf16) * BCI * RS + sizeof(__bf16) * BCO * RS >= (size_t)WI * WJ * (WK - 1) * 1024 * sizeof(float),
                      "LDS fold space");
        __syncthreads();
        float *red = (float *)Xs; // Xs and Ds are contiguous static LDS objects of this kernel; the fold may spill into Ds
        const int grp = wi * WJ + wj;
        if (wk > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((grp * (WK - 1) + wk - 1) * 16 + r) * 64 + lane] = acc[0][0][0][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int w = 0; w < WK - 1; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][0][r] += red[((grp * (WK - 1) + w) * 16 + r) * 64 + lane];
        }
        if (wk != 0) return;
    }

#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float *o = part + ((long long)split * taps.n + tap + t) * g.CiP * g.CoP;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < MJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = ci0 + wi * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int co = co0 + wj * TJ + j * 32 + (lane & 31);
                    o[(long long)ci * g.CoP + co] = acc[t][i][j][r];
                }
    }
}

int acg_wgrad_bf16_launch(const float *x, const float *dy, float *part, const WGeom &g, const Taps &t, int bci, int bco,
                          hipStream_t st)
{
    const int nt = acg_wgrad_taps_per_wg(g.Cin, g.Cg, t.n, g.thin);
    const int blocks = g.nsplit * (t.n / nt) * (g.CiP / bci) * (g.CoP / bco);
    dim3 grid(blocks), block(256);
    const long long nimg = g.Mtot / ((long long)g.Hg * g.Wg);
    const long long xbytes = nimg * g.Hin * g.Win * g.Cin * 4, dbytes = g.Mtot * g.Cg * 4;
    ACG_REQUIRE(xbytes < (1LL << 32) && dbytes < (1LL << 32), "wgrad_bf16: operand exceeds the 4 GiB buffer-addressing limit");
    const unsigned xb = (unsigned)xbytes, db = (unsigned)dbytes;
#define WG_BF16(BCI_, BCO_, WI_, WJ_, WK_, KP_)                                                                          \
    do {                                                                                                                 \
        if (split) hipLaunchKernelGGL((wgrad_bf16<BCI_, BCO_, WI_, WJ_, WK_, KP_, true>), grid, block, 0, st, x, dy, part, g, t, xb, db); \
        else hipLaunchKernelGGL((wgrad_bf16<BCI_, BCO_, WI_, WJ_, WK_, KP_, false>), grid, block, 0, st, x, dy, part, g, t, xb, db);      \
    } while (0)
    const bool split = g_acg_precision == ACG_PREC_BF16X3;
    if (nt == 3) { // 64 x 128 tile, the three taps of a kernel row per workgroup (acg_wgrad_tiles gave bco = 128 for it)
        ACG_REQUIRE(bci == 64 && bco == 128, "wgrad_bf16: tile of the three-tap variant");
        if (split) hipLaunchKernelGGL((wgrad_bf16<64, 128, 2, 2, 1, 64, true, 3>), grid, block, 0, st, x, dy, part, g, t, xb, db);
        else hipLaunchKernelGGL((wgrad_bf16<64, 128, 2, 2, 1, 64, false, 3>), grid, block, 0, st, x, dy, part, g, t, xb, db);
    } else if (bci == 128) WG_BF16(128, 128, 2, 2, 1, 64);
    else if (bci == 64 && bco == 64) WG_BF16(64, 64, 2, 2, 1, 64);
    else if (bci == 32 && bco == 64) WG_BF16(32, 64, 1, 2, 2, 128);
    else if (bci == 64 && bco == 32) WG_BF16(64, 32, 2, 1, 2, 128);
    else WG_BF16(32, 32, 1, 1, 4, 256);
#undef WG_BF16
    ACG_CHECK_LAUNCH("wgrad_bf16");
    acg_note_kernel("wgrad_bf16<%d,%d,SPLIT=%d,NT=%d>", bci, bco, split ? 1 : 0, nt);
    return ACG_OK;
}
