// Stride-2 data gradient / ConvTranspose2d forward (networks.py:168, 178-179, 231-234): the four sub-pixel phases in ONE
// TILE.
//
// A stride-2 transposed convolution is four dense small-tap convolutions, one per output phase (py, px) = output pixel
// (2 gy + py, 2 gx + px); conv_api.hip ran them as four workgroups per tile of the generic kernel (conv_bf16.hip), each
// gathering and splitting the input rows of its own taps: nine tap passes over the same two input rows, 3.5 GB through L2
// per launch for 0.8 GB of operands — 0.47 ms, bound by that traffic, not by HBM (1.01x) or the matrix cores.  Here a
// workgroup owns 128 consecutive positions of one row of the PHASE grid and all four phases: a stage is one input row
// (dy) and up to three taps of it — whatever phases they belong to — so the 130-pixel patch of a row is fetched, split
// into bf16 hi / lo and stored once per 32-channel chunk and row (2 loads instead of 9 for the 3x3 layers), the taps read it
// at their column offsets like the row-patch stages of the generic kernel, and each tap's MFMAs go to the accumulators of
// ITS phase (four accumulator sets, the phase a uniform switch around the MFMA block).
//
// Stage table (host, acg_ph4_plan): taps sorted by input row; sub-stages of <= 3 taps; Taps.gpk[i] = first tap | taps << 8 |
// (smallest dx of the ROW & 0xff) << 16 | (first sub-stage of its row) << 24; a tap's weight slab and phase ride in Taps.w
// as slab | phase << 12.  Per accumulator the taps arrive in a fixed order, (dy, dx) ascending.
//
// SUMS (round 6; data gradient of the stride-2 convolution behind a norm + ReLU, networks.py:164-170): dx is the gradient w.r.t.
// that norm's output, so the epilogue also leaves the first pass of the norm's backward — per-channel sum gy and sum gy * xhat,
// gy = dx * act'(y) — for the tile's 512 pixels in g.ns_part (chunk 4 * tile; the tile's other three chunks hold zeros).  The
// norm's input x at the 128 pixels of a phase (32 KB) travels global -> LDS by LDS-DMA into the operand buffers the loop has
// finished with, one phase ahead: no registers (a first version read x by 128 four-byte loads per thread and tile: the loop runs
// at 238 of 256 registers, nothing hid their latency, 0.55-0.60 ms against 0.30 ms — slower than the separate pass).
#include "conv_internal.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int BM = 128, KC = 32, NKC = KC / 16, NK8 = KC / 8, NTX = 3;
// The stage table of the 3x3, pad 1, stride-2 layers (the only ones the model has on 64 output channels; acg_ph4_plan
// builds it from the tap lists and acg_ph4_plan_is_k3 checks it): three sub-stages of three taps per channel chunk —
// input row 0: phase 0 (dx 0), phase 1 (dx 0, 1) | phase 2 (dx 0), phase 3 (dx 0, 1); input row 1: phase 2 (dx 0), phase 3 (dx 0, 1)
constexpr int kSub = 3;
__device__ constexpr int kPhase[kSub][NTX] = {{0, 1, 1}, {2, 3, 3}, {2, 3, 3}};
__device__ constexpr bool kNewRow[kSub] = {true, false, true};
}

template <int BN, int WM, int WN, bool SUMS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void igemm_conv_ph4(const float *__restrict__ in, const __bf16 *__restrict__ wp,
                                                      const float *__restrict__ bias, float *__restrict__ out, Geom g,
                                                      Taps taps, unsigned in_bytes, unsigned w_bytes, unsigned w_lo_bytes)
{
    constexpr int TM = BM / WM, TN = BN / WN, MB = TM / 32, NB = TN / 32;
    constexpr int UPR = KC / 8, RPP = 256 / UPR, AL = BM / RPP, ALX = AL + 1;   // gather passes: BM rows + the 2 extra pixels
    constexpr int BCH = NKC * BN * 2, BL = (BCH + 255) / 256;
    constexpr int APL = (BM + NTX - 1) * 8, BPL = BN * 8 + 32;                  // plane strides (conv_bf16.hip)
    constexpr int A_IMG = NK8 * APL, B_IMG = NK8 * BPL;
    static_assert(WM * WN == 4 && MB >= 1 && NB >= 1, "tile config");
    // ONE operand array (A images, then B images): behind the loop its first 32 KB stage the norm input of a phase (SUMS)
    constexpr int AB_ELEMS = 2 * A_IMG + NTX * 2 * B_IMG;
    static_assert(!SUMS || AB_ELEMS * 2 >= BM * 64 * 4, "the operand buffers hold one phase of x");
    __shared__ __attribute__((aligned(256))) __bf16 lds_ab[AB_ELEMS];
    __bf16 *const As = lds_ab, *const Bs = lds_ab + 2 * A_IMG;
    __shared__ __attribute__((aligned(16))) unsigned out_rel[BM];
    __shared__ float sred[(SUMS ? 2 : 1) * WM * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tiles_n = g.ncols_pad / BN;
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int n0 = tile_n * BN, m0 = tile_m * BM, GHW = g.GH * g.GW;
    const int img = m0 / GHW, gy0 = (m0 - img * GHW) / g.GW, gx0 = m0 - img * GHW - gy0 * g.GW;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)wp, 0, w_bytes, 0x00020000);
    const int u = tid % UPR, rrow = tid / UPR;
    unsigned b_voff[BL];
    int b_lds[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int idx = tid + 256 * i;
        const int kc = idx / (BN * 2);
        const int rem = idx - kc * BN * 2;
        b_voff[i] = (unsigned)(((kc * g.ncols_pad + n0) * 16 + rem * 8) * 2);
        b_lds[i] = (kc * 2 + (rem & 1)) * BPL + (rem >> 1) * 8;
    }
    // phase (0, 0) pixel of grid position gx0 + i, relative to the tile's first; the other phases add a constant
    const long long off0 = (((long long)img * g.Hout + 2 * gy0) * g.Wout + 2 * gx0) * g.Cout;
    if (tid < BM) out_rel[tid] = (unsigned)(2 * tid * g.Cout * 4);

    f32x16 acc[4][MB][NB];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][i][j][r] = 0.f;

    const int nst = taps.ngrp, S = nst * (g.Cin / KC);
    f32x8 ra[ALX];
    u32x4 rb[NTX][BL], rbl[NTX][BL];

    auto load_stage = [&](int s) {
        const int cc = s / nst, c0 = cc * KC;
        const int gd = taps.gpk[s - cc * nst];
        const int first = gd & 0xff, tx0 = (gd << 8) >> 24, newrow = (gd >> 24) & 1;
        if (newrow) { // the row's patch: pixels gx0 + tx0 .. + BM + 1 of input row gy0 + dy (zeros outside the map)
            const int iy = gy0 + ((taps.pk[first] << 24) >> 24);
            const bool rok = (unsigned)iy < (unsigned)g.Hin;
            const int rowbase = (img * g.Hin + iy) * g.Win;
#pragma unroll
            for (int j = 0; j < ALX; ++j) {
                const int ix = gx0 + tx0 + rrow + RPP * j;
                const bool ok = rok && (j < AL || rrow < NTX - 1) && (unsigned)ix < (unsigned)g.Win;
                const unsigned off = (unsigned)((rowbase + ix) * g.Cin + c0 + 8 * u) * 4u;
                const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off, ok), 0, 0);
                const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off + 16u, ok), 0, 0);
                const f32x4 flo = __builtin_bit_cast(f32x4, lo), fhi = __builtin_bit_cast(f32x4, hi);
                ra[j] = (f32x8){flo[0], flo[1], flo[2], flo[3], fhi[0], fhi[1], fhi[2], fhi[3]};
            }
        }
#pragma unroll
        for (int q = 0; q < NTX; ++q) {
            {
                const int tw = (taps.pk[first + q] >> 16) & 0xfff;
                const unsigned soff = (unsigned)(((tw * (g.Cin >> 4) + (c0 >> 4)) * g.ncols_pad) * 16) * 2u;
#pragma unroll
                for (int i = 0; i < BL; ++i)
                    if (tid + 256 * i < BCH) {
                        rb[q][i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff, 0);
                        rbl[q][i] = __builtin_amdgcn_raw_buffer_load_b128(rw, b_voff[i], soff + w_lo_bytes, 0);
                    }
            }
        }
    };

    // The phase of every tap is a compile-time constant (kPhase: the 3x3, pad 1 layout acg_ph4_plan is checked against), so
    // each tap's MFMAs name their accumulator set statically — a run-time switch merges the four sets through VGPRs.
    auto sub_stage = [&](int s, auto ss_c) {
        constexpr int SS = decltype(ss_c)::value;
        const int gd = taps.gpk[SS];
        const int gfirst = gd & 0xff, gtx0 = (gd << 8) >> 24;
        __syncthreads();
        if (kNewRow[SS]) {
#pragma unroll
            for (int j = 0; j < ALX; ++j) {
                if (j == AL && rrow >= NTX - 1) continue;
                const int a_at = u * APL + (rrow + RPP * j) * 8;
                const float v[8] = {ra[j][0], ra[j][1], ra[j][2], ra[j][3], ra[j][4], ra[j][5], ra[j][6], ra[j][7]};
                acg_u32x4 hi, lo;
                acg_split8(v, hi, lo);
                *(acg_u32x4 *)&As[a_at] = hi;
                *(acg_u32x4 *)&As[A_IMG + a_at] = lo;
            }
        }
#pragma unroll
        for (int q = 0; q < NTX; ++q) {
#pragma unroll
            for (int i = 0; i < BL; ++i)
                if (tid + 256 * i < BCH) {
                    *(u32x4 *)&Bs[q * 2 * B_IMG + b_lds[i]] = rb[q][i];
                    *(u32x4 *)&Bs[q * 2 * B_IMG + B_IMG + b_lds[i]] = rbl[q][i];
                }
        }
        __syncthreads();
        if (s + 1 < S) load_stage(s + 1);
#pragma unroll
        for (int q = 0; q < NTX; ++q) {
            constexpr int dummy = 0; (void)dummy;
            const int dxo = ((taps.pk[gfirst + q] << 16) >> 24) - gtx0;
            f32x16 (&ac)[MB][NB] = acc[kPhase[SS][q]];
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                bf16x8 a[MB], b[NB], al[MB], bl[NB];
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const int at = (kc * 2 + (lane >> 5)) * APL + (wm * TM + i * 32 + (lane & 31) + dxo) * 8;
                    a[i] = *(const bf16x8 *)&As[at];
                    al[i] = *(const bf16x8 *)&As[A_IMG + at];
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int bt = q * 2 * B_IMG + (kc * 2 + (lane >> 5)) * BPL + (wn * TN + j * 32 + (lane & 31)) * 8;
                    b[j] = *(const bf16x8 *)&Bs[bt];
                    bl[j] = *(const bf16x8 *)&Bs[B_IMG + bt];
                }
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) { // small cross terms first, the leading term last (conv_bf16.hip)
                        ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], b[j], ac[i][j], 0, 0, 0);
                        ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bl[j], ac[i][j], 0, 0, 0);
                        ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], ac[i][j], 0, 0, 0);
                    }
            }
        }
    };
    load_stage(0);
    for (int s = 0; s < S; s += kSub) {
        sub_stage(s, std::integral_constant<int, 0>{});
        sub_stage(s + 1, std::integral_constant<int, 1>{});
        sub_stage(s + 2, std::integral_constant<int, 2>{});
    }

    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)(out + off0), 0, 0xFFFFFFF0u, 0x00020000);
    // SUMS: this thread's channel (NB == 1), the norm's per-(image, channel) constants, and the LDS-DMA of a phase's x:
    // wave w moves pieces 8 w .. 8 w + 7, a piece = 4 consecutive phase pixels x 64 channels = 1 KB (lane l: pixel l >> 4,
    // 16 bytes at 16 (l & 15)); xs[pixel][64] fp32
    typedef __attribute__((address_space(3))) void lds_void;
    float ns_mu = 0.f, ns_rs = 0.f, ns_ga = 1.f, ns_be = 1.f, ns_s1 = 0.f, ns_s2 = 0.f;
    const int ns_co = n0 + wn * TN + (lane & 31);
    const bool ns_relu = SUMS && g.ns_act == ACG_ACT_RELU;
    const float *const xs = (const float *)lds_ab;
    auto ns_dma = [&](int p) {
        if constexpr (SUMS) {
            const __amdgpu_buffer_rsrc_t rnx = __builtin_amdgcn_make_buffer_rsrc((void *)(g.ns_x + off0), 0, 0xFFFFFFF0u, 0x00020000);
            const unsigned poff = (unsigned)((((p >> 1) * g.Wout + (p & 1)) * g.Cout) * 4);
            const int w0 = __builtin_amdgcn_readfirstlane(wave) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned voff = (unsigned)((4 * (w0 + k) + (lane >> 4)) * 2 * g.Cout * 4 + (lane & 15) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rnx, (lds_void *)((char *)lds_ab + (w0 + k) * 1024), 16, voff, poff, 0, 0);
            }
        }
    };
    if constexpr (SUMS) {
        static_assert(!SUMS || (NB == 1 && BN == 64), "SUMS: one 32-channel block per wave, 64 stored channels");
        if (ns_co < g.Cout) {
            ns_mu = g.ns_mean[img * g.Cout + ns_co];
            ns_rs = g.ns_rstd[img * g.Cout + ns_co];
            if (ns_relu) { ns_ga = g.ns_gamma[g.ns_gstride * img + ns_co]; ns_be = g.ns_beta[g.ns_gstride * img + ns_co]; }
        }
        __syncthreads();   // every wave has read its last fragments: the operand buffers are free
        ns_dma(0);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        __builtin_amdgcn_sched_barrier(0);   // one phase's accumulators in VGPRs at a time (two waves per SIMD: 256 registers)
        if (g.stats != nullptr) { // per-tile (mean, M2) of this phase's 128 pixels (conv_bf16.hip; act == NONE by contract)
            float mu[NB];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int co = n0 + wn * TN + j * 32 + (lane & 31);
                    const float bv = (bias != nullptr && co < g.Cout) ? bias[co] : 0.f;
                    float sm = 0.f;
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[p][i][j][r] + bv;
                            sm += pass == 0 ? v : (v - mu[j]) * (v - mu[j]);
                        }
                    sm += __shfl_xor(sm, 32);
                    if (lane < 32) sred[wm * BN + wn * TN + j * 32 + lane] = sm;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int cl = wn * TN + j * 32 + (lane & 31);
                    float sm = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) sm += sred[w * BN + cl];
                    if (pass == 0) {
                        mu[j] = sm * (1.f / BM);
                    } else if (wm == 0 && lane < 32 && n0 + cl < g.Cout) {
                        const long long tile = m0 / BM, tpi = (long long)GHW / BM;
                        const long long chunk = (tile / tpi) * g.stats_cpi + g.stats_chunk0 + p * tpi + tile % tpi;
                        float *o = g.stats + chunk * 2 * g.Cout + n0 + cl;
                        o[0] = mu[j];
                        o[g.Cout] = sm;
                    }
                }
                __syncthreads();
            }
        }
        const unsigned poff = (unsigned)((((p >> 1) * g.Wout + (p & 1)) * g.Cout) * 4);
        if constexpr (SUMS) {   // (bias == nullptr, act == NONE by contract: the stored value is the accumulator)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of phase p (and the previous phase's stores)
            __syncthreads();                                   // ... and everybody else's
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * TM + i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                    const float xh = (xs[row * 64 + ns_co] - ns_mu) * ns_rs;
                    const float gy = (ns_relu && !(xh * ns_ga + ns_be > 0.f)) ? 0.f : acc[p][i][0][r];   // same expression as norm_apply_kernel
                    ns_s1 += gy;
                    ns_s2 += gy * xh;
                }
            __syncthreads();                                   // phase p is read: the stage is free
            if (p < 3) ns_dma(p + 1);                          // in flight behind this phase's stores
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int co = n0 + wn * TN + j * 32 + (lane & 31);
            const float bv = (bias != nullptr && co < g.Cout) ? bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][i][j][r] = acg_apply_act(acc[p][i][j][r] + bv, g.act);
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const u32x4 rel = *(const u32x4 *)&out_rel[wm * TM + i * 32 + 8 * r4 + 4 * (lane >> 5)];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int co = n0 + wn * TN + j * 32 + (lane & 31);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = acc[p][i][j][4 * r4 + q];
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, acg_masked_off(rel[q] + (unsigned)co * 4u, co < g.Cout), poff, 0);
                    }
                }
            }
    }
    if constexpr (SUMS) {   // 32 rows x 4 phases per thread -> the two lane halves -> the WM waves of a column block, fixed order
        ns_s1 += __shfl_xor(ns_s1, 32);
        ns_s2 += __shfl_xor(ns_s2, 32);
        if (lane < 32) {
            sred[wm * BN + wn * TN + lane] = ns_s1;
            sred[WM * BN + wm * BN + wn * TN + lane] = ns_s2;
        }
        __syncthreads();
        if (tid < BN && n0 + tid < g.Cout) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { a += sred[w * BN + tid]; b += sred[WM * BN + w * BN + tid]; }
            // part[N][4 * GHW / BM chunks][2][Cout]: this tile's four 128-pixel chunks are 4 * t .. 4 * t + 3
            const long long tpi = (long long)GHW / BM, t = (m0 / BM) % tpi;
            float *o = g.ns_part + (((long long)img * 4 * tpi + 4 * t) * 2) * g.Cout + n0 + tid;
            o[0] = a;
            o[g.Cout] = b;
#pragma unroll
            for (int k = 1; k < 4; ++k) { o[(long long)k * 2 * g.Cout] = 0.f; o[(long long)k * 2 * g.Cout + g.Cout] = 0.f; }
        }
    }
}

// the kernel's compile-time stage table (kPhase / kNewRow) is this plan?
static bool acg_ph4_plan_is_k3(const Taps &p)
{
    static const int ph[kSub][NTX] = {{0, 1, 1}, {2, 3, 3}, {2, 3, 3}};
    static const int nr[kSub] = {1, 0, 1};
    if (p.ngrp != kSub || p.n != kSub * NTX) return false;
    for (int s = 0; s < kSub; ++s) {
        if ((p.gpk[s] & 0xff) != s * NTX || ((p.gpk[s] >> 8) & 0xff) != NTX || ((p.gpk[s] >> 24) & 1) != nr[s]) return false;
        for (int q = 0; q < NTX; ++q)
            if ((p.w[s * NTX + q] >> 12) != ph[s][q]) return false;
    }
    return true;
}

// Stage table for the launch above from the per-phase tap lists (t.dy / dx / w at 16 ph + i, counts in nt[4]); false: the
// taps do not fit (a row wider than three columns, more than 16 sub-stages)
bool acg_ph4_plan(const Taps &t, const int nt[4], Taps *out)
{
    struct MT { int dy, dx, w, ph; };
    std::vector<MT> all;
    for (int ph = 0; ph < 4; ++ph)
        for (int i = 0; i < nt[ph]; ++i) all.push_back({t.dy[16 * ph + i], t.dx[16 * ph + i], t.w[16 * ph + i], ph});
    if (all.empty() || all.size() > 64) return false;
    std::stable_sort(all.begin(), all.end(), [](const MT &a, const MT &b) {
        return a.dy != b.dy ? a.dy < b.dy : (a.ph != b.ph ? a.ph < b.ph : a.dx < b.dx);
    });
    Taps r = t;
    r.n = (int)all.size();
    r.ngrp = 0;
    for (int i = 0; i < 64; ++i) { r.dy[i] = 0; r.dx[i] = 0; r.w[i] = 0; r.gpk[i] = 0; }
    for (size_t i = 0; i < all.size(); ++i) {
        if (all[i].w < 0 || all[i].w >= 4096 || all[i].dy < -128 || all[i].dy > 127) return false;
        r.dy[i] = (short)all[i].dy; r.dx[i] = (short)all[i].dx; r.w[i] = (short)(all[i].w | (all[i].ph << 12));
    }
    for (size_t i = 0; i < all.size();) {
        size_t j = i;
        int lo = all[i].dx, hi = all[i].dx;
        while (j < all.size() && all[j].dy == all[i].dy) { lo = std::min(lo, all[j].dx); hi = std::max(hi, all[j].dx); ++j; }
        if (hi - lo > NTX - 1 || lo < -128 || lo > 127) return false;
        for (size_t k = i; k < j; k += NTX) {
            if (r.ngrp == 64) return false;
            const int cnt = (int)std::min<size_t>(NTX, j - k);
            r.gpk[r.ngrp++] = (int)k | (cnt << 8) | ((lo & 0xff) << 16) | ((k == i ? 1 : 0) << 24);
        }
        i = j;
    }
    *out = r;
    return acg_ph4_plan_is_k3(r);   // (other layouts: the four-launch / phased paths of conv_api.hip)
}

// g: the phased Geom of conv_api.hip (GH x GW = the phase grid, os = 2); tp: a plan of acg_ph4_plan
bool acg_igemm_ph4_ok(const Geom &g)
{
    static const bool off = acg_debug_switch("ACG_NO_PH4"); // A/B switch
    return !off && g_acg_precision == ACG_PREC_BF16X3 && g_acg_conv_impl == ACG_IMPL_MFMA && !g.thin && g.is == 1 && g.os == 2 &&
           g.GW % BM == 0 && g.Mtot % BM == 0 && g.Cin % KC == 0 && g.ncols_pad == 64 && g.Hout == 2 * g.GH && g.Wout == 2 * g.GW &&
           !g.reflect;
}

int acg_igemm_ph4_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &plan,
                         long long n_w_elems, hipStream_t st)
{
    const Taps tp = acg_taps_pack(plan);
    const long long nimg = g.Mtot / ((long long)g.GH * g.GW);
    const long long in_bytes = nimg * g.Hin * g.Win * g.Cin * 4, w_bytes = n_w_elems * 2 * 2;
    ACG_REQUIRE(acg_ph4_plan_is_k3(plan), "igemm_conv_ph4: the stage table is not the 3x3 pad-1 one the kernel is built for");
    ACG_REQUIRE(acg_igemm_ph4_ok(g) && in_bytes < (1LL << 32) && w_bytes < (1LL << 32) && 2LL * BM * g.Cout * 4 < (1LL << 31),
                "igemm_conv_ph4: unsupported geometry");
    ACG_REQUIRE(g.stats == nullptr || (((long long)g.GH * g.GW) % BM == 0 && g.act == ACG_ACT_NONE),
                "igemm_conv_ph4: per-tile statistics need whole tiles per image and no activation");
    dim3 grid((unsigned)(g.Mtot / BM) * (g.ncols_pad / 64));
    if (g.ns_part != nullptr) {
        ACG_REQUIRE(g.stats == nullptr && bias == nullptr && g.act == ACG_ACT_NONE && g.ns_x != nullptr && g.ns_mean != nullptr && g.ns_rstd != nullptr &&
                    g.ns_mask == nullptr && (g.ns_act == ACG_ACT_NONE || (g.ns_act == ACG_ACT_RELU && g.ns_gamma != nullptr && g.ns_beta != nullptr)) &&
                    (g.ns_gstride == 0 || g.ns_gstride >= g.Cout) && ((long long)g.GH * g.GW) % BM == 0 && g.ncols_pad == 64 && g.Cout == 64,
                    "igemm_conv_ph4: the norm sums ride on a plain data gradient with 64 stored channels (no bias / activation / statistics; act NONE / RELU recomputed from x)");
        hipLaunchKernelGGL((igemm_conv_ph4<64, 2, 2, true>), grid, dim3(256), 0, st, in, (const __bf16 *)wp, bias, out, g, tp, (unsigned)in_bytes,
                           (unsigned)w_bytes, (unsigned)(n_w_elems * 2));
        ACG_CHECK_LAUNCH("igemm_conv_ph4<SUMS>");
        acg_note_kernel("igemm_conv_ph4<128,64,SUMS=1> (%d sub-stages)", plan.ngrp);
        return ACG_OK;
    }
    hipLaunchKernelGGL((igemm_conv_ph4<64, 2, 2>), grid, dim3(256), 0, st, in, (const __bf16 *)wp, bias, out, g, tp, (unsigned)in_bytes,
                       (unsigned)w_bytes, (unsigned)(n_w_elems * 2));
    ACG_CHECK_LAUNCH("igemm_conv_ph4");
    acg_note_kernel("igemm_conv_ph4<128,64> (%d sub-stages)", plan.ngrp);
    return ACG_OK;
}
