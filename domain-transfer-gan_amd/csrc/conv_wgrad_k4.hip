// Weight gradient of the discriminator's deep 4x4 convolutions (networks.py:321-338: 128 -> 256 on 64 x 64 and 256 -> 256 on
// 63 x 63, stride 1, zero padding 1; bf16x3 arithmetic), one KERNEL ROW per workgroup — the scheme of wgrad_x3_krow
// (conv_wgrad_tr.hip) for four taps, either stride and output rows of any width (63 / 62 here):
//
//   dW[ky][kx][ci][co] = sum over output pixels (n, oy, ox) of  x[n, oy*s + ky - 1, ox*s + kx - 1][ci] * dy[n, oy, ox][co]
//
// The per-tap kernel (wgrad_bf16<128,128>) streams an x tile and a dy tile per tap through L2 and the VGPR split: 16 taps x
// (Cin/128) x (Cout/128) passes over both tensors, 0.49 / 0.99 ms for 133 / 258 GFLOP (0.32 / 0.42 of the bf16x3 ceiling).  Here
// a STAGE is a run of 32 pixels of one output row (a row is ceil(Wo / 32) runs, the pixels past its end zero in the dy image)
// and the four taps of a kernel row share its dy tile and ONE x window, ox0 the run's first pixel:
//   stride 1: window pixels ix = ox0 - 1 .. ox0 + 33 (35 rows of the pixel-major LDS image), tap kx reads rows kx + k;
//   stride 2: window pixels ix = 2 ox0 - 1 .. 2 ox0 + 64, stored DE-INTERLEAVED — odd columns in rows 0..32, even columns in
//             rows 33..65 — so that tap kx again reads 16 CONSECUTIVE rows per K step: rows (kx & 1) * 33 + (kx >> 1) + k.
// Everything else as in wgrad_x3_krow: [pixel][channels] bf16 hi / lo images (rows of 2 C + 64 bytes), both MFMA operands by
// ds_read_b64_tr_b16, 8 waves as 4 (ci) x 2 (co) with a 32 x 64 wave tile per tap (128 accumulator registers: one
// 512-thread workgroup per CU, 256-register budget), two stage buffers, the next stage's loads in flight under the MFMAs.
//
// The same kernel with THREE taps per row and a 64-channel x tile serves the stride-2 3x3 pair of the generators
// (networks.py:168, 178-179: 64 -> 128 downsample, ConvTranspose 128 -> 64) and D_B's 64 -> 128 4x4 stride-2 layer: the
// 64 x 128 channel tile gives the waves only 2 (ci) x 2 (co) positions, so pairs of waves split the two 16-pixel K steps of a
// stage and fold their accumulators through LDS at the end.
#include "common.h"
#include "conv_internal.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int KP = 32;                 // pixels per stage (one run of an output row)
constexpr int BC = 128;                // channel tile on both sides
constexpr int PD = 320;                // bytes per dy pixel row: 128 bf16 + 64 B (rows 0..3 of a transposed block -> 4 bank segments)
constexpr int DIMG = KP * PD;
typedef __attribute__((address_space(3))) char lds_char;

template <int NT, int IS, int BCI> struct KR {
    static constexpr int PX = BCI * 2 + 64;                            // bytes per x pixel row: 320 (128 ch) / 192 (64 ch)
    static constexpr int XW = IS * (KP - 1) + NT;                      // window pixels: 35 / 66 (4 taps), 65 (3 taps, stride 2)
    static constexpr int XIMG = XW * PX;
    static constexpr int BUF = 2 * XIMG + 2 * DIMG;                    // [x hi][x lo][dy hi][dy lo]
    static constexpr int UPP = BCI / 8, PPR = 512 / UPP;               // 8-channel units per x pixel, pixels per load round
    static constexpr int FR = XW / PPR, XTAIL = XW - PPR * FR;         // full load rounds, pixels of the last one
    static constexpr int WKS = BCI == 128 ? 1 : 2;                     // waves along the K steps of a stage
    static_assert(XTAIL * UPP <= 64, "the tail round is loaded by wave 0");
    static constexpr int tap_row(int kx) { return IS == 1 ? kx : (kx & 1) * (KP + 1) + (kx >> 1); }
    static __device__ __forceinline__ int img_row(int j) { return IS == 1 ? j : ((j & 1) ? KP + 1 + (j >> 1) : (j >> 1)); }
};

template <int PITCH> __device__ __forceinline__ bf16x8 tr_frag(const lds_char *p)
{
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + 4 * PITCH));
    const s16x8 v = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
}

// PF: stages of global loads in flight in registers (each stage 24 registers).  One workgroup per CU and one stage ahead
// keeps 33 KB in flight per CU: with the short 64-channel stages (0.5 us of MFMAs) the loop then waits on memory latency
// (0.33 ms for the stride-2 3x3 layer = 1.6 us per stage); two stages ahead cover it.
template <int NT, int IS, int BCI, int PF>
__global__ __launch_bounds__(512) void wgrad_x3_krowg(const float *__restrict__ x, const float *__restrict__ dy,
                                                      float *__restrict__ part, WGeom g, unsigned x_bytes, unsigned d_bytes)
{
    typedef KR<NT, IS, BCI> L;
    constexpr int XIMG = L::XIMG, BUF = L::BUF, FR = L::FR, PX = L::PX, WKS = L::WKS;
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = WKS == 1 ? 0 : (wave & 1);
    const int wi = WKS == 1 ? wave >> 1 : wave >> 2, wj = WKS == 1 ? (wave & 1) : ((wave >> 1) & 1);
    const int tiles_ci = g.CiP / BCI, tiles_co = g.CoP / BC;
    // XCD-aware order: the four kernel rows of a pixel range stream the same dy rows and overlapping x rows — neighbours on one XCD
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    int b = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tco = b % tiles_co; b /= tiles_co;
    const int tci = b % tiles_ci; b /= tiles_ci;
    const int ky = b % NT;
    const int split = b / NT;
    const int ci0 = tci * BCI, co0 = tco * BC;
    const int Hg = g.Hg, Wg = g.Wg;
    const long long mbeg = (long long)split * g.m_per_split;
    long long mend = mbeg + g.m_per_split;
    if (mend > g.Mtot) mend = g.Mtot;
    const int runs = (Wg + KP - 1) / KP;
    const int nst = mbeg < mend ? (int)((mend - mbeg) / Wg) * runs : 0;   // whole output rows per split, `runs` stages each

    f32x16 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd_ = __builtin_amdgcn_make_buffer_rsrc((void *)dy, 0, d_bytes, 0x00020000);

    // the NEXT run to load (wave-uniform): global row number lrow = n * Hg + oy, first pixel lox
    int lrow = (int)(mbeg / Wg);
    int ln = lrow / Hg, loy = lrow - ln * Hg, lox = 0;
    // this thread's units: 8 channels of window pixels pjx + PPR r and of dy pixel pj
    const int c8 = tid & 15, pj = tid >> 4;
    const int c8x = tid % L::UPP, pjx = tid / L::UPP;
    u32x4 rx[PF][FR + 1][2], rda[PF][2];
    const bool do_bias = g.bias_from == 1 && ky == 0 && tci == 0;
    // ConvTranspose bias (column sums of the x side): kernel rows 1 .. IS visit every input row exactly once (iy = oy * IS +
    // ky - 1), and window pixels 1 .. IS * KP of a run every column once
    const bool do_xbias = g.bias_from == 2 && tco == 0 && ky >= 1 && ky <= IS;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    auto load_stage = [&](auto PC) {
        constexpr int P = decltype(PC)::value;
        const int iy = loy * IS + ky - 1;
        const bool rowok = (unsigned)iy < (unsigned)g.Hin;
        const int rowbase = (ln * g.Hin + iy) * g.Win;
        auto xoff = [&](int j, bool use) {
            const int ix = lox * IS - 1 + j;
            const bool ok = use && rowok && (unsigned)ix < (unsigned)g.Win;
            return acg_masked_off((unsigned)((rowbase + ix) * g.Cin + ci0 + 8 * c8x) * 4u, ok);
        };
#pragma unroll
        for (int r = 0; r < FR; ++r) {
            const unsigned o = xoff(pjx + L::PPR * r, true);
            rx[P][r][0] = __builtin_amdgcn_raw_buffer_load_b128(rx_, o, 0, 0);
            rx[P][r][1] = __builtin_amdgcn_raw_buffer_load_b128(rx_, o, 16, 0);
        }
        if (wave == 0) { // the last XTAIL window pixels: the first lanes of wave 0 (the others masked)
            const unsigned o = xoff(pjx + L::PPR * FR, pjx < L::XTAIL);
            rx[P][FR][0] = __builtin_amdgcn_raw_buffer_load_b128(rx_, o, 0, 0);
            rx[P][FR][1] = __builtin_amdgcn_raw_buffer_load_b128(rx_, o, 16, 0);
        }
        const unsigned od = acg_masked_off((unsigned)((lrow * Wg + lox + pj) * g.Cg + co0 + 8 * c8) * 4u, lox + pj < Wg);
        rda[P][0] = __builtin_amdgcn_raw_buffer_load_b128(rd_, od, 0, 0);
        rda[P][1] = __builtin_amdgcn_raw_buffer_load_b128(rd_, od, 16, 0);
        lox += KP;
        if (lox >= Wg) {
            lox = 0;
            ++lrow;
            if (++loy == Hg) { loy = 0; ++ln; }
        }
    };
    auto put = [&](char *img_hi, char *img_lo, int off, const u32x4 (&r)[2]) {
        const f32x4 a = __builtin_bit_cast(f32x4, r[0]), c = __builtin_bit_cast(f32x4, r[1]);
        const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        acg_u32x4 hi, lo;
        acg_split8(v, hi, lo);
        *(acg_u32x4 *)(img_hi + off) = hi;
        *(acg_u32x4 *)(img_lo + off) = lo;
    };
    auto store_stage = [&](int buf, auto PC) {
        constexpr int P = decltype(PC)::value;
        char *base = lds + buf * BUF;
#pragma unroll
        for (int r = 0; r < FR; ++r) put(base, base + XIMG, L::img_row(pjx + L::PPR * r) * PX + c8x * 16, rx[P][r]);
        if (wave == 0 && pjx < L::XTAIL) put(base, base + XIMG, L::img_row(pjx + L::PPR * FR) * PX + c8x * 16, rx[P][FR]);
        put(base + 2 * XIMG, base + 2 * XIMG + DIMG, pj * PD + c8 * 16, rda[P]);
        if (do_xbias) {
#pragma unroll
            for (int r = 0; r <= FR; ++r) {
                const int j = pjx + L::PPR * r;
                if (j >= 1 && j <= IS * KP && (r < FR || (wave == 0 && pjx < L::XTAIL))) {
                    const f32x4 a = __builtin_bit_cast(f32x4, rx[P][r][0]), c = __builtin_bit_cast(f32x4, rx[P][r][1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bsum[e] += a[e]; bsum[4 + e] += c[e]; }
                }
            }
        }
        if (do_bias) {
            const f32x4 a = __builtin_bit_cast(f32x4, rda[P][0]), c = __builtin_bit_cast(f32x4, rda[P][1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bsum[e] += a[e]; bsum[4 + e] += c[e]; }
        }
    };

    // transposed-read lane map (tools/probes/tr_read.hip): lane 4q+p of 16-lane group gq supplies pixel row 8*(gq>>1) + q,
    // channels 16*(gq&1) + 4p .. 4p+3, and receives channel 16*(gq&1) + (lane&15) of the four rows
    const int gq = lane >> 4, li = lane & 15;
    const int frag_row = 8 * (gq >> 1) + (li >> 2), frag_col = 16 * (gq & 1) + 4 * (li & 3);
    const int xlane = frag_row * PX + (wi * 32 + frag_col) * 2;
    const int dlane = frag_row * PD + (wj * 64 + frag_col) * 2;

    // stage k travels in register set k % PF: loaded PF stages ahead, stored into LDS buffer k & 1 during stage k - 1
    static_assert(PF == 1 || PF == 2, "register sets");
    if (nst > 0) load_stage(std::integral_constant<int, 0>());
    if (PF == 2 && nst > 1) load_stage(std::integral_constant<int, PF - 1>());
    if (nst > 0) store_stage(0, std::integral_constant<int, 0>());
    if (PF < nst) load_stage(std::integral_constant<int, 0>());
    __syncthreads();
    // waves 4-7 (the second wave of every SIMD) convert and store the next stage AFTER their MFMAs, waves 0-3 before: one
    // wave of a SIMD is in its VALU / LDS-store phase while its partner feeds the matrix pipe
    const bool late = wave >= 4;
    auto stage = [&](int s, auto NC) { // NC: register set of stage s + 1
        const int cur = s & 1;
        if (!late && s + 1 < nst) {
            store_stage(cur ^ 1, NC);
            if (s + 1 + PF < nst) load_stage(NC);
        }
        const lds_char *xb = (const lds_char *)(lds + cur * BUF) + xlane;
        const lds_char *db = (const lds_char *)(lds + cur * BUF + 2 * XIMG) + dlane;
#pragma unroll
        for (int kq = 0; kq < KP / 16 / WKS; ++kq) {
            const int ks = kq * WKS + wk;   // with two waves along K, each takes one of the two 16-pixel steps
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = tr_frag<PD>(db + ks * 16 * PD + j * 64);
                bl[j] = tr_frag<PD>(db + DIMG + ks * 16 * PD + j * 64);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 ah = tr_frag<PX>(xb + (ks * 16 + L::tap_row(t)) * PX);
                const bf16x8 al = tr_frag<PX>(xb + XIMG + (ks * 16 + L::tap_row(t)) * PX);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[t][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[t][j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[t][j], 0, 0, 0);
            }
        }
        if (late && s + 1 < nst) {
            store_stage(cur ^ 1, NC);
            if (s + 1 + PF < nst) load_stage(NC);
        }
        __syncthreads();
    };
    for (int s = 0; s < nst; s += PF) {
        stage(s, std::integral_constant<int, 1 % PF>());
        if (PF == 2 && s + 1 < nst) stage(s + 1, std::integral_constant<int, 0>());
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

    if (do_xbias) { // the PPR threads that share a channel group of the x tile
        float *red = (float *)lds;
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(pjx * L::UPP + c8x) * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < BCI) {
            float s = 0.f;
            for (int r = 0; r < L::PPR; ++r) s += red[(r * L::UPP + (tid >> 3)) * 8 + (tid & 7)];
            g.bias_part[((long long)split * IS + ky - 1) * g.CiP + ci0 + tid] = s;
        }
    }

    if (WKS == 2) { // the two waves of a (ci, co) position hold partial sums of the same outputs: fold them tap by tap through LDS
        float *red = (float *)lds;
        const int grp = wi * 2 + wj;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            __syncthreads();
            if (wk == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((grp * 2 + j) * 16 + r) * 64 + lane] = acc[t][j][r];
            }
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][j][r] += red[((grp * 2 + j) * 16 + r) * 64 + lane];
            }
        }
        if (wk == 1) return;
    }

#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float *o = part + ((long long)split * (NT * NT) + ky * NT + t) * g.CiP * g.CoP;
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

// the shapes the split plan (wgrad_plan, conv_api.hip) sizes for this kernel: zero padding 1, output rows that fill most of
// their 32-pixel runs, 128-multiple output channels; 4x4 at stride 1 or 2 with 128-multiple (stride 2: or 64) input channels,
// 3x3 at stride 2 with 64 input channels
bool acg_wgrad_krowg_shape_ok(int K, int stride, int pad, int reflect, int Wi, int Wo, int Cx, int Cg)
{
    static const bool off = acg_debug_switch("ACG_NO_KROWG"); // A/B switch
    if (off || g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA) return false;
    if (pad != 1 || reflect || Wo < 16 || (Wo % KP != 0 && Wo % KP < 16) || Cg % BC != 0) return false;
    if ((Wo - 1) * stride + K - 2 > Wi) return false;   // the last tap's column stays inside the padded row (a valid convolution)
    if (K == 4) return (stride == 1 && Cx % BC == 0) || (stride == 2 && (Cx % BC == 0 || Cx == 64));
    return K == 3 && stride == 2 && Cx == 64;
}

bool acg_wgrad_krowg_ok(const WGeom &g, const Taps &t)
{
    const int K = t.n == 16 ? 4 : t.n == 9 ? 3 : 0;
    if (g.thin || K == 0 || g.CiP != g.Cin || g.CoP != g.Cg) return false;
    if (g.bias_from == 2 && (g.Hin != g.is * g.Hg || g.Win != g.is * g.Wg)) return false;   // x-side sums: every input pixel once
    if (!acg_wgrad_krowg_shape_ok(K, g.is, 1, g.reflect, g.Win, g.Wg, g.Cin, g.Cg) || g.m_per_split % g.Wg != 0) return false;
    for (int i = 0; i < t.n; ++i)
        if (t.dy[i] != i / K - 1 || t.dx[i] != i % K - 1) return false;
    return true;
}

int acg_wgrad_krowg_launch(const float *x, const float *dy, float *part, const WGeom &g, const Taps &t, hipStream_t st)
{
    const int K = t.n == 16 ? 4 : 3, bci = g.Cin == 64 ? 64 : BC;
    const int blocks = g.nsplit * K * (g.CiP / bci) * (g.CoP / BC);
    const long long nimg = g.Mtot / ((long long)g.Hg * g.Wg);
    const long long xbytes = nimg * g.Hin * g.Win * g.Cin * 4, dbytes = g.Mtot * g.Cg * 4;
    ACG_REQUIRE(xbytes < (1LL << 32) && dbytes < (1LL << 32), "wgrad_x3_krowg: operand exceeds the 4 GiB buffer-addressing limit");
    static const bool pf2 = acg_debug_switch("ACG_KROWG_PF2"); // A/B switch
#define KROWG(NT, IS, BCI) hipLaunchKernelGGL((wgrad_x3_krowg<NT, IS, BCI, (BCI == 64 ? 2 : 1)>), dim3(blocks), dim3(512), 0, st, x, dy, part, g, (unsigned)xbytes, (unsigned)dbytes)
    if (K == 4 && g.is == 1 && pf2) hipLaunchKernelGGL((wgrad_x3_krowg<4, 1, 128, 2>), dim3(blocks), dim3(512), 0, st, x, dy, part, g, (unsigned)xbytes, (unsigned)dbytes);
    else if (K == 4 && g.is == 1) KROWG(4, 1, 128);
    else if (K == 4 && bci == 128) KROWG(4, 2, 128);
    else if (K == 4) KROWG(4, 2, 64);
    else KROWG(3, 2, 64);
#undef KROWG
    ACG_CHECK_LAUNCH("wgrad_x3_krowg");
    acg_note_kernel("wgrad_x3_krowg<NT=%d,IS=%d,BCI=%d>", K, g.is, bci);
    return ACG_OK;
}
