// Persistent row-pipeline convolution for the full-resolution 3x3 stride-1 layers with 32 gathered and 64 written channels
// (networks.py:164: 32 -> 64 forward; the data gradient of networks.py:183's 64 -> 32), bf16x3 arithmetic, zero padding 1.
//
// The generic tile (conv_bf16.hip) spends these layers' time on latency chains, not on bytes or MFMAs: a 128-pixel tile lives
// three short stages, each of which loads its operands (74 KB of packed weights per tile among them), waits, splits, stores,
// barriers and only then multiplies, with three workgroups per CU to hide it all behind: 0.31 ms for 805 MB and 77 GFLOP.
// Here ONE 768-thread workgroup per CU walks down a 128-pixel-wide column band of an image, one output row per step:
//   * the packed weights of all nine taps (74 KB hi + lo) are copied into LDS once and stay there;
//   * consecutive output rows share two of their three input rows: a ring of four row images (130 pixels x 32 channels,
//     bf16 hi / lo planes of 8 channels) keeps them in LDS, so every input row is fetched ONCE per band — by the four
//     producer waves, two rows ahead in registers (loads of row r+4 are in flight while row r+2 is split and stored);
//   * eight consumer waves (32 pixels x 32 channels each, two per SIMD: one alone cannot hide its LDS round trips behind its
//     own MFMAs — 0.155 ms for 0.092 ms of MFMAs without anything else running) only read fragments and issue MFMAs: the
//     weights are the MFMA A operand and the pixels the B operand, so a lane ends up with four consecutive CHANNELS of one
//     pixel — its output leaves as 16-byte stores straight from the accumulators, no staging;
//   * one raw s_barrier per row (lgkmcnt only): global loads and stores stay in flight across it.
#include "common.h"
#include "conv_internal.h"
#include <type_traits>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int RW_PX = 128;                    // output pixels of a row tile
constexpr int RW_WIN = RW_PX + 2;             // input pixels of a row image
constexpr int RW_PLANE = 2304;                // bytes per 8-channel plane: 144 pixel slots of 16 B, a multiple of the 256-B bank row
constexpr int RW_IMG = 4 * RW_PLANE + 64;     // hi (or lo) image of 32 channels; planes 2, 3 sit 64 B further (2-way stores, as x3_ws)
constexpr int RW_SLOT = 2 * RW_IMG;           // hi + lo
constexpr int RW_NSLOT = 4;
constexpr int RW_WIMG = 9 * 4 * 64 * 16;      // weights, hi (or lo): [tap][plane][64 columns][8 bf16]
constexpr int RW_LDS = 2 * RW_WIMG + RW_NSLOT * RW_SLOT;
constexpr int RW_UNITS = RW_WIN * 4;          // 32-byte load units (pixel, plane) of a row
constexpr int RW_UPT = (RW_UNITS + 255) / 256;   // per producer thread: 3
__device__ __forceinline__ int rw_plane_off(int u) { return u * RW_PLANE + (u >> 1) * 64; }
typedef __attribute__((address_space(3))) char lds_char;
}

__global__ __launch_bounds__(768) void conv_rows_x3(const float *__restrict__ in, const __bf16 *__restrict__ wp,
                                                    const float *__restrict__ bias, float *__restrict__ out, Geom g,
                                                    unsigned long long slabs, unsigned in_bytes, long long w_lo_elems, int R, int cpb, int abl)
{
    __shared__ __attribute__((aligned(256))) char lds[RW_LDS];
    __shared__ __attribute__((aligned(16))) float ptab[64 * 4];   // norm sums: (mean, rstd, gamma, beta) of the 64 channels of image n
    char *const Wl = lds, *const Al = lds + 2 * RW_WIMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bands = g.GW / RW_PX;
    int b = blockIdx.x;
    const int chunk = b % cpb; b /= cpb;
    const int band = b % bands;
    const int n = b / bands;
    const int x0 = band * RW_PX, y0 = chunk * R;
    const int H = g.GH, W = g.GW;
    int rows = H - y0;
    rows = rows < R ? rows : R;           // output rows of this workgroup (> 0: the launcher sizes cpb that way)
    auto row_barrier = [&]() {
        if (abl & 8) return;   // (timing ablation: no row barriers — wrong results)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS traffic is done; vector memory stays in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- the weights: [tap = ky * 3 + kx][plane][column][8], from the packed slabs [slab][Cin / 16][64][16] (hi, then lo)
    for (int q = tid; q < 2 * 9 * 4 * 64; q += 768) {
        const int lo = q / (9 * 4 * 64), r = q - lo * (9 * 4 * 64);
        const int tap = r >> 8, plane = (r >> 6) & 3, col = r & 63;
        const int slab = (int)((slabs >> (4 * tap)) & 15ull);
        const long long e = (long long)lo * w_lo_elems + (((long long)slab * 2 + (plane >> 1)) * 64 + col) * 16 + (plane & 1) * 8;
        *(u32x4 *)(Wl + lo * RW_WIMG + ((tap * 4 + plane) * 64 + col) * 16) = *(const u32x4 *)(wp + e);
    }

    if (g.ns_part != nullptr && tid < 64) {
        const bool aff = g.ns_act != ACG_ACT_NONE;
        const int go = g.ns_gstride * n + tid;   // (gstride 0: shared affine parameters)
        *(f32x4 *)&ptab[tid * 4] = (f32x4){g.ns_mean[n * 64 + tid], g.ns_rstd[n * 64 + tid], aff ? g.ns_gamma[go] : 1.f, aff ? g.ns_beta[go] : 1.f};
    }

    if (wave >= 8) {
        // ------------------------------------------------------------------------------------------------ producers
        const int pt = tid - 512;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
        u32x4 rx[2][RW_UPT][2];
        int u_px[RW_UPT], u_pl[RW_UPT];
        bool u_ok[RW_UPT];
#pragma unroll
        for (int i = 0; i < RW_UPT; ++i) {
            const int q = pt + 256 * i;
            u_px[i] = q >> 2; u_pl[i] = q & 3;
            const int ix = x0 - 1 + u_px[i];
            u_ok[i] = q < RW_UNITS && (unsigned)ix < (unsigned)W;
        }
        auto load_row = [&](int i, auto PC) {   // input row y0 + i into register set PC (rows outside the image / the band's range: zeros)
            constexpr int P = decltype(PC)::value;
            const int iy = y0 + i;
            const bool rowok = (unsigned)iy < (unsigned)H && i <= rows;
            if (abl & 2) return;
#pragma unroll
            for (int k = 0; k < RW_UPT; ++k) {
                if (k == RW_UPT - 1 && pt + 256 * k >= RW_UNITS) continue;   // (wave-uniform for whole waves past the end)
                const unsigned off = ((unsigned)((n * H + iy) * W + x0 - 1 + u_px[k]) * 32u + 8u * (unsigned)u_pl[k]) * 4u;   // (in_bytes < 4 GiB: unsigned arithmetic)
                const unsigned o = acg_masked_off(off, rowok && u_ok[k]);
                rx[P][k][0] = __builtin_amdgcn_raw_buffer_load_b128(rin, o, 0, 0);
                rx[P][k][1] = __builtin_amdgcn_raw_buffer_load_b128(rin, o, 16, 0);
            }
        };
        auto store_row = [&](int i, auto PC) {   // ... split and stored into ring slot (i + 1) & 3
            constexpr int P = decltype(PC)::value;
            char *slot = Al + ((i + 1) & (RW_NSLOT - 1)) * RW_SLOT;
            if (abl & 2) return;
#pragma unroll
            for (int k = 0; k < RW_UPT; ++k) {
                if (pt + 256 * k >= RW_UNITS) continue;
                const f32x4 a = __builtin_bit_cast(f32x4, rx[P][k][0]), c = __builtin_bit_cast(f32x4, rx[P][k][1]);
                const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
                acg_u32x4 hi, lo;
                acg_split8(v, hi, lo);
                char *d = slot + rw_plane_off(u_pl[k]) + u_px[k] * 16;
                *(acg_u32x4 *)d = hi;
                *(acg_u32x4 *)(d + RW_IMG) = lo;
            }
        };
        const std::integral_constant<int, 0> c0;
        const std::integral_constant<int, 1> c1;
        // input rows -1 .. rows (relative to y0); row i travels in register set (i + 1) & 1
        load_row(-1, c0);
        load_row(0, c1);
        store_row(-1, c0);
        load_row(1, c0);
        store_row(0, c1);
        load_row(2, c1);
        store_row(1, c0);
        load_row(3, c0);
        row_barrier();                       // rows -1, 0, 1 (and the weights) are in LDS
        for (int j = 0; j < rows; j += 2) {
            // while the consumers compute output row j (input rows j-1 .. j+1): store input row j+2, fetch row j+4
            store_row(j + 2, c1);
            load_row(j + 4, c1);
            row_barrier();
            if (j + 1 < rows) {
                store_row(j + 3, c0);
                load_row(j + 5, c0);
                row_barrier();
            }
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumers
    __builtin_amdgcn_s_setprio(2);
    const int kg = lane >> 4, li = lane & 15;
    const int pw = wave & 3, ch = wave >> 2;     // pixels 32 pw .. + 31, channels 32 ch .. + 31 (column blocks 2 ch, 2 ch + 1)
    const lds_char *wfrag = (const lds_char *)Wl + (kg * 64 + ch * 32 + li) * 16;                        // + tap * 4096 + cb * 256 (+ RW_WIMG)
    const lds_char *pfrag = (const lds_char *)Al + rw_plane_off(kg) + (pw * 32 + li) * 16;               // + slot + pb * 256 + kx * 16 (+ RW_IMG)
    f32x4 bv[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[cb][r] = bias != nullptr ? bias[(ch * 2 + cb) * 16 + 4 * kg + r] : 0.f;
    const int act = __builtin_amdgcn_readfirstlane(g.act);
    // Fragments of tap t + 1 are read while the 12 MFMAs of tap t issue (two register sets, set = parity of the running tap
    // count).  The first tap of the NEXT row reads input row j, which is already in the ring, so the chain carries across
    // the row barrier.
    typedef const __attribute__((address_space(3))) bf16x8 *frag_ptr;
    bf16x8 ph[2][2], pl[2][2], wh[2][2], wl[2][2];
    auto frags = [&](auto SC, int j, int tap) {   // row j, tap = ky * 3 + kx (compile-time after unrolling) into set SC
        constexpr int S = decltype(SC)::value;
        const int ky = tap / 3, kx = tap - ky * 3;
        const lds_char *prow = pfrag + ((j + ky) & (RW_NSLOT - 1)) * RW_SLOT;   // input row j - 1 + ky
        const lds_char *wt = wfrag + tap * 4096;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            ph[S][pb] = *(frag_ptr)(prow + pb * 256 + kx * 16);
            pl[S][pb] = *(frag_ptr)(prow + RW_IMG + pb * 256 + kx * 16);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            wh[S][cb] = *(frag_ptr)(wt + cb * 256);
            wl[S][cb] = *(frag_ptr)(wt + RW_WIMG + cb * 256);
        }
    };
    f32x4 acc[2][2];
    float *const stats = g.stats;
    // norm-backward sums (acg_conv2d_bwd_data_sums): this launch's output is the gradient w.r.t. the OUTPUT of a norm (+ ReLU)
    // whose input is ns_x; the first pass of that norm's backward — sum gy and sum gy * xhat per channel, gy = dx * act'(y) —
    // accumulates beside the statistics' registers (a launch has one or the other) and leaves like them, once per workgroup
    const float *const nsx = g.ns_x;
    const bool sums = g.ns_part != nullptr;
    const int ns_relu = g.ns_act == ACG_ACT_RELU;
    f32x4 xv[2][2];
    f32x4 sm1[2], sm2[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) sm1[cb] = sm2[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mfmas = [&](auto SC) {
        constexpr int S = decltype(SC)::value;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[S][cb], ph[S][pb], acc[cb][pb], 0, 0, 0);
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[S][cb], pl[S][pb], acc[cb][pb], 0, 0, 0);
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[S][cb], ph[S][pb], acc[cb][pb], 0, 0, 0);
            }
    };
    const std::integral_constant<int, 0> s0;
    const std::integral_constant<int, 1> s1;
    auto row = [&](int j, auto PC) {   // PC: parity of the running tap count at the row's first tap
        constexpr int P = decltype(PC)::value;
        const long long oidx = (((long long)n * g.Hout + y0 + j) * g.Wout + x0 + pw * 32 + li) * 64 + ch * 32 + 4 * kg;
        if (sums) {   // the norm input at this lane's output positions: in flight under the row's MFMAs
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) xv[cb][pb] = *(const f32x4 *)(nsx + oidx + pb * 16 * 64 + cb * 16);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) acc[cb][pb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            // the 8 fragment reads of tap t + 1 are ISSUED between the 12 MFMAs of tap t (scheduling groups)
            if (((t + P) & 1) == 0) {
                if (t < 8) frags(s1, j, t + 1); else frags(s1, j + 1, 0);
                mfmas(s0);
            } else {
                if (t < 8) frags(s0, j, t + 1); else frags(s0, j + 1, 0);
                mfmas(s1);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (stats != nullptr) {   // (wave-uniform) running sums of the bias-free outputs and of their squares: see the end
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    sm1[cb] += acc[cb][pb];
                    sm2[cb] += acc[cb][pb] * acc[cb][pb];
                }
        }
        if (sums) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4 pr = *(const f32x4 *)&ptab[((ch * 2 + cb) * 16 + 4 * kg + r) * 4];   // mean, rstd, gamma, beta
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb) {
                        const float xh = (xv[cb][pb][r] - pr[0]) * pr[1];
                        const float v = acc[cb][pb][r] + bv[cb][r];
                        const float gy = (ns_relu && !(xh * pr[2] + pr[3] > 0.f)) ? 0.f : v;   // same expression as norm_apply_kernel
                        sm1[cb][r] += gy;
                        sm2[cb][r] += gy * xh;
                    }
                }
        }
        // lane: pixel li of pixel block pb, channels (2 ch + cb) * 16 + 4 kg .. + 3 (16 bytes; 64 per pixel from the four lanes that share li)
        float *orow = out + oidx;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x4 v = acc[cb][pb] + bv[cb];
                if (act != ACG_ACT_NONE) {   // (wave-uniform; these layers are followed by a norm or are data gradients: rarely taken)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acg_apply_act(v[r], act);
                }
                if (!(abl & 1)) *(f32x4 *)(orow + pb * 16 * 64 + cb * 16) = v;
            }
        row_barrier();
    };
    row_barrier();
    frags(s0, 0, 0);
    for (int j = 0; j < rows; j += 2) {
        row(j, s0);                       // nine taps: the next row starts on the other set
        if (j + 1 < rows) row(j + 1, s1);
    }
    if (stats == nullptr && !sums) return;
    // Statistics for the InstanceNorm behind the layer (acg_conv2d_fwd_stats: one (mean, M2) entry per 128-pixel tile, merged
    // by Chan's formula with 128 pixels each).  The workgroup owns `rows` tiles of every channel: it forms their JOINT mean
    // and M2 — sums of the bias-free outputs (the bias is the pivot) over the lanes' 2 x rows values, the 16 lanes of a
    // channel quad, the four waves of a channel half — and writes the SAME entry (mean, M2 / rows) for each of its tiles: equal
    // means add no between-tile term, so the merge reproduces the joint statistics exactly.  No per-row cross-lane work, no
    // extra barrier in the loop (the producers have left; a barrier now counts the eight consumer waves).
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                sm1[cb][r] += __shfl_xor(sm1[cb][r], m);
                sm2[cb][r] += __shfl_xor(sm2[cb][r], m);
            }
    float *red = (float *)Al;   // [pixel quarter][2][64]: the ring is no longer read for results (the last prefetch is discarded)
    if (li == 0) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[(pw * 2 + 0) * 64 + (ch * 2 + cb) * 16 + 4 * kg + r] = sm1[cb][r];
                red[(pw * 2 + 1) * 64 + (ch * 2 + cb) * 16 + 4 * kg + r] = sm2[cb][r];
            }
    }
    row_barrier();
    if (wave == 0) {
        const int c = lane;
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { a += red[(w * 2 + 0) * 64 + c]; q += red[(w * 2 + 1) * 64 + c]; }
        if (sums) {   // additive: the workgroup's sums go to its first tile's entry, zeros to the others (acg_norm_bwd_partials adds them up)
            for (int j = 0; j < rows; ++j) {
                float *o = g.ns_part + (((long long)n * H * bands + (long long)(y0 + j) * bands + band) * 2) * 64 + c;
                o[0] = j == 0 ? a : 0.f;
                o[64] = j == 0 ? q : 0.f;
            }
            return;
        }
        const float cnt = (float)rows * RW_PX;
        const float mu = a / cnt;
        const float m2 = (q - a * mu) / (float)rows;      // this tile's share of sum (x - mean)^2
        const float mean = mu + (bias != nullptr ? bias[c] : 0.f);
        for (int j = 0; j < rows; ++j) {
            float *o = stats + (((long long)n * g.stats_cpi + g.stats_chunk0 + (long long)(y0 + j) * bands + band) * 2) * 64 + c;
            o[0] = mean;
            o[64] = m2 > 0.f ? m2 : 0.f;
        }
    }
}

// 3x3 window with offsets -1 .. 1 in both directions, every position once: slab of window position (ky, kx) in 4 bits each
static bool rows_slabs(const Taps &t, unsigned long long *slabs)
{
    if (t.n != 9) return false;
    unsigned long long s = 0;
    unsigned seen = 0;
    for (int i = 0; i < 9; ++i) {
        const int ky = t.dy[i] + 1, kx = t.dx[i] + 1;
        if (ky < 0 || ky > 2 || kx < 0 || kx > 2 || t.w[i] < 0 || t.w[i] > 15) return false;
        const int pos = ky * 3 + kx;
        if (seen >> pos & 1u) return false;
        seen |= 1u << pos;
        s |= (unsigned long long)t.w[i] << (4 * pos);
    }
    *slabs = s;
    return true;
}

bool acg_conv_rows_ok(const Geom &g, const Taps &t)
{
    static const bool off = acg_debug_switch("ACG_NO_ROWS"); // A/B switch
    if (off || g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA || g.thin || g.nphase || g.fold_p) return false;
    if (g.stats != nullptr && (g.act != ACG_ACT_NONE || g.stats_chunk0 != 0 || (long long)g.stats_cpi * 128 != (long long)g.GH * g.GW)) return false;
    if (g.ns_part != nullptr && (g.stats != nullptr || g.ns_x == nullptr || g.ns_mask != nullptr || (g.ns_act != ACG_ACT_NONE && g.ns_act != ACG_ACT_RELU) ||
                                 (g.ns_gstride != 0 && g.ns_gstride < 64)))
        return false;
    if (g.reflect || g.addend || g.relu_src || g.out_s16 || g.unpad || g.is != 1 || g.os != 1 || g.oy0 || g.ox0) return false;
    if (g.Cin != 32 || g.Cout != 64 || g.ncols_pad != 64) return false;
    if (g.GH != g.Hin || g.GW != g.Win || g.GH != g.Hout || g.GW != g.Wout || g.GW % RW_PX != 0 || g.GH < 1) return false;
    unsigned long long s;
    return rows_slabs(t, &s);
}

int acg_conv_rows_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &t,
                         long long n_w_elems, hipStream_t st)
{
    unsigned long long slabs = 0;
    ACG_REQUIRE(acg_conv_rows_ok(g, t) && rows_slabs(t, &slabs), "conv_rows_x3: unsupported geometry");
    const long long nimg = g.Mtot / ((long long)g.GH * g.GW);
    const long long in_bytes = nimg * g.Hin * g.Win * g.Cin * 4;
    ACG_REQUIRE(in_bytes < (1LL << 32) && nimg * g.GH * g.GW * 64 * 4 < (1LL << 40), "conv_rows_x3: tensor exceeds the buffer-addressing limit");
    // one workgroup per CU: chunks of rows per (image, band) so that the grid is about one residency wave of 256
    const int bands = g.GW / RW_PX;
    long long cpb = (256 + nimg * bands - 1) / (nimg * bands);
    if (cpb > g.GH / 8) cpb = g.GH / 8;
    if (cpb < 1) cpb = 1;
    const int R = (int)((g.GH + cpb - 1) / cpb);
    cpb = (g.GH + R - 1) / R;
    const long long blocks = nimg * bands * cpb;
    static const int abl = getenv("ACG_ROWS_ABL") && acg_debug_switch("ACG_ROWS_ABL") ? atoi(getenv("ACG_ROWS_ABL")) : 0;   // timing ablations
    hipLaunchKernelGGL(conv_rows_x3, dim3((unsigned)blocks), dim3(768), 0, st, in, (const __bf16 *)wp, bias, out, g, slabs,
                       (unsigned)in_bytes, n_w_elems, R, (int)cpb, abl);
    ACG_CHECK_LAUNCH("conv_rows_x3");
    acg_note_kernel("conv_rows_x3<32,64> (%d rows per workgroup)", R);
    return ACG_OK;
}
