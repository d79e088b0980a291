// bf16x3 convolutions with a THIN output (stored Cout == 16: the 3-channel image heads and the data gradients into
// image tensors) and 32 gathered channels, stride 1: the 7x7 32->3 head (+tanh) and the data gradient of the 7x7 stem.
//
// On the generic tile (conv_bf16.hip, 128 pixels x 32 columns) these layers gather, split and store the 128 x 32 input
// tile once per TAP — 49 times for a 7x7 — while the MFMA work is tiny (16 useful columns): 0.87 ms for 19.7 GFLOP.
// Here a workgroup owns an 8 x 16 output tile, loads and splits its (8+K-1) x (16+K-1) input PATCH once into LDS and
// walks the taps over it: every tap's A fragment (v_mfma_f32_16x16x32_bf16: 16 pixels of a row x 32 channels) is 16
// consecutive patch pixels.  Weights (2 KB per tap) go straight from L1/L2 into the B fragment registers.
#include "common.h"
#include "conv_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int PT_TH = 8, PT_TW = 16;   // output tile
constexpr int PT_PIX = 320;            // patch pixel capacity per plane: >= (8+6)*(16+6) = 308, and 320*16 B = 5120 B is a
                                       // multiple of the 256-B bank row, so the two planes a ds_read_b128 lane group spans
                                       // line up: 16 consecutive pixels -> 16 distinct slots for any tap offset
constexpr int PT_PLANE = PT_PIX * 8;   // bf16 elements per plane of 8 channels
constexpr int PT_IMG = 4 * PT_PLANE;   // one hi (or lo) image: 4 planes = 32 channels
}

template <bool REFLECT>
__global__ __launch_bounds__(256) void conv_patch16_x3(const float *__restrict__ in, const __bf16 *__restrict__ wp,
                                                       const float *__restrict__ bias, float *__restrict__ out, Geom g,
                                                       Taps taps, int dymin, int dxmin, int PH, int PW, unsigned in_bytes,
                                                       long long w_lo_elems)
{
    __shared__ __attribute__((aligned(16))) __bf16 Ap[2 * PT_IMG]; // [hi|lo][plane][pixel][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (g.GW + PT_TW - 1) / PT_TW, tiles_y = (g.GH + PT_TH - 1) / PT_TH;
    // XCD-aware tile order (bijective for any grid size): workgroups are dealt to the 8 XCDs round-robin, so consecutive tiles
    // — whose patches overlap — are made neighbours on ONE XCD and fetch the shared rows through one L2
    const int nwg_ = gridDim.x, bid_ = blockIdx.x, xq_ = nwg_ >> 3, xr_ = nwg_ & 7, xcd_ = bid_ & 7;
    int b = (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + (bid_ >> 3);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty * PT_TH, gx0 = tx * PT_TW;

    // ---- the patch: pixel pp = py * PW + px  <->  gathered pixel (gy0 + py + dymin, gx0 + px + dxmin); consecutive
    // lanes take consecutive pixels of one 8-channel plane (conflict-free 16-byte LDS stores)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
    const int npp = PH * PW;
    constexpr int NIT = 4 * PT_PIX / 256;   // 5 (plane, pixel) items per thread: all ten loads in flight before the first split
    static_assert(4 * PT_PIX % 256 == 0, "patch items per thread");
    f32x4 plo[NIT], phi[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        const int u = i / PT_PIX, pp = i - u * PT_PIX; // plane, patch pixel
        const int py = pp / PW, px = pp - py * PW;
        int iy = gy0 + py + dymin, ix = gx0 + px + dxmin;
        bool ok = pp < npp;
        if (REFLECT) {
            ok = ok && iy > -g.Hin && iy < 2 * g.Hin - 1 && ix > -g.Win && ix < 2 * g.Win - 1; // pixels of partial tiles
            iy = iy < 0 ? -iy : iy;
            iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
        } else {
            ok = ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
        }
        const unsigned off = (unsigned)(((n * g.Hin + iy) * g.Win + ix) * g.Cin + 8 * u) * 4u;
        plo[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off, ok), 0, 0));
        phi[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off + 16u, ok), 0, 0));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        const int u = i / PT_PIX, pp = i - u * PT_PIX;
        if (pp >= npp) continue;
        const float v[8] = {plo[it][0], plo[it][1], plo[it][2], plo[it][3], phi[it][0], phi[it][1], phi[it][2], phi[it][3]};
        acg_u32x4 hi, lo;
        acg_split8(v, hi, lo);
        *(acg_u32x4 *)&Ap[u * PT_PLANE + pp * 8] = hi;
        *(acg_u32x4 *)&Ap[PT_IMG + u * PT_PLANE + pp * 8] = lo;
    }
    __syncthreads();

    // ---- taps: wave w owns tile rows 2w, 2w+1; lane l: pixel l&15 of the row, channel group l>>4
    const int pl = lane >> 4, lr = lane & 15;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // B fragment of a tap: packed [tap][Cin/16][ncols_pad][16] -> column lr, k = 8*pl .. 8*pl+7 of the 32 channels
    const long long boff = ((long long)(pl >> 1) * g.ncols_pad + lr) * 16 + (pl & 1) * 8;
    const long long tap_stride = (long long)(g.Cin / 16) * g.ncols_pad * 16;
    // The weights of a tap are 2 x 16 bytes per lane straight from L1 / L2, and nothing else in the loop waits on memory: loaded
    // inside the tap's own iteration (with the tap entry fetched by a scalar load in front of them) every tap paid two memory
    // round trips for six MFMAs — 22 VGPRs, the matrix pipe ~12 % busy per wave.  Now the tap table sits in a VGPR (lane t =
    // tap t, read with v_readlane: no memory access in the loop) and the B fragments run PF taps ahead of their MFMAs.
    constexpr int PF = 7;   // a 7x7 window is seven whole groups
    const int tap_v = taps.pk[lane < taps.n ? lane : 0];
    bf16x8 bh[PF], bl[PF];
    auto bload = [&](int t, bf16x8 &h, bf16x8 &l) {   // (a tap index past the end loads tap 0 again: never used)
        const int tw = __builtin_amdgcn_readlane(tap_v, t < taps.n ? t : 0) >> 16;
        const __bf16 *wt = wp + tw * tap_stride + boff;
        h = *(const bf16x8 *)wt;
        l = *(const bf16x8 *)(wt + w_lo_elems);
    };
    auto tap_mma = [&](int t, const bf16x8 &ch, const bf16x8 &cl) {
        const int pk = __builtin_amdgcn_readlane(tap_v, t);
        const int dy = ((pk << 24) >> 24) - dymin, dx = ((pk << 16) >> 24) - dxmin;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int pp = (wave * 2 + r + dy) * PW + dx + lr;
            const bf16x8 ah = *(const bf16x8 *)&Ap[pl * PT_PLANE + pp * 8];
            const bf16x8 al = *(const bf16x8 *)&Ap[PT_IMG + pl * PT_PLANE + pp * 8];
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, ch, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, cl, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ch, acc[r], 0, 0, 0);
        }
    };
    const int nfull = taps.n / PF * PF;
    if (nfull > 0) {
#pragma unroll
        for (int j = 0; j < PF; ++j) bload(j, bh[j], bl[j]);
    }
    for (int t0 = 0; t0 < nfull; t0 += PF) {   // whole groups: no branch in the body, every slot reloaded unconditionally
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            tap_mma(t0 + j, bh[j], bl[j]);
            bload(t0 + j + PF, bh[j], bl[j]);   // this slot's next tap, PF taps ahead, into the registers the MFMAs just read
        }
    }
    for (int t = nfull; t < taps.n; ++t) {   // the taps of a window that is not a multiple of PF
        bf16x8 ch, cl;
        bload(t, ch, cl);
        tap_mma(t, ch, cl);
    }

    // ---- epilogue: lane l holds column l&15 of pixels 4*(l>>4) .. +3 of its rows
    const float bv = (bias != nullptr && lr < g.Cout) ? bias[lr] : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int gy = gy0 + wave * 2 + r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gx = gx0 + 4 * pl + k;
            if (gy < g.GH && gx < g.GW && lr < g.Cout)
                out[(((long long)n * g.Hout + gy) * g.Wout + gx) * g.Cout + lr] = acg_apply_act(acc[r][k] + bv, g.act);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same layers with the image tensor stored C4 (Geom.Cout == 4) and "N-packed" weights: the 16 MFMA columns are
// (4 horizontally adjacent output pixels) x (4 channels) instead of 16 channels of which 13 are padding.  A row of the MFMA
// tile is then a BASE pixel that stands for output pixels 4 b .. 4 b + 3 of its row, and K runs over the window positions
// (ry, u), u = dxo + kw in 0 .. KW + 2: K = KH (KW + 3) 32-channel steps per 64 output pixels (a 4 x 4 block of base
// pixels = 4 rows x 16 pixels) instead of KH KW per 16 — 2.8x fewer MFMAs, fragment reads and weight loads per output
// pixel for a 7x7 (the kernel is bound by those loads: 2 KB of B fragments per tap and wave through the CU's one
// vector-memory pipe).  The B operand of step (ry, u) holds w[ry][u - dxo] in column (dxo, c), zero outside the kernel
// (pack_weight_npack_kernel, conv_api.hip).  A workgroup owns the 8 x 16 output tile of conv_patch16_x3 = two such blocks;
// its four waves split the K steps (step s -> wave s % 4) and fold their partial blocks through LDS, so every B fragment
// is loaded once per workgroup.
template <bool REFLECT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_patchn_x3(const float *__restrict__ in, const __bf16 *__restrict__ wn,
                                                      const float *__restrict__ bias, float *__restrict__ out, Geom g,
                                                      int dymin, int dxmin, int KH, int KW, unsigned in_bytes)
{
    __shared__ __attribute__((aligned(16))) __bf16 Ap[2 * PT_IMG]; // [hi|lo][plane][pixel][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (g.GW + PT_TW - 1) / PT_TW, tiles_y = (g.GH + PT_TH - 1) / PT_TH;
    // XCD-aware tile order (bijective for any grid size): workgroups are dealt to the 8 XCDs round-robin, so consecutive tiles
    // — whose patches overlap — are made neighbours on ONE XCD and fetch the shared rows through one L2
    const int nwg_ = gridDim.x, bid_ = blockIdx.x, xq_ = nwg_ >> 3, xr_ = nwg_ & 7, xcd_ = bid_ & 7;
    int b = (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + (bid_ >> 3);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty * PT_TH, gx0 = tx * PT_TW;
    const int PH = PT_TH + KH - 1, PW = PT_TW + KW - 1;

    // ---- the patch, as in conv_patch16_x3
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
    const int npp = PH * PW;
    constexpr int NIT = 4 * PT_PIX / 256;
    f32x4 plo[NIT], phi[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        const int u = i / PT_PIX, pp = i - u * PT_PIX; // plane, patch pixel
        const int py = pp / PW, px = pp - py * PW;
        int iy = gy0 + py + dymin, ix = gx0 + px + dxmin;
        bool ok = pp < npp;
        if (REFLECT) {
            ok = ok && iy > -g.Hin && iy < 2 * g.Hin - 1 && ix > -g.Win && ix < 2 * g.Win - 1; // pixels of partial tiles
            iy = iy < 0 ? -iy : iy;
            iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
        } else {
            ok = ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
        }
        const unsigned off = (unsigned)(((n * g.Hin + iy) * g.Win + ix) * g.Cin + 8 * u) * 4u;
        plo[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off, ok), 0, 0));
        phi[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off + 16u, ok), 0, 0));
    }
    // this wave's B fragments (steps wave, wave + 4, ...: at most NSW of them), PF steps ahead of their MFMAs (all 18 in
    // registers cost the kernel half its occupancy): the first PF are on their way while the patch is split.  Lane l reads
    // bytes 16 l .. 16 l + 15 of a step's 1 KB hi / lo block — its column l & 15, channels 8 (l >> 4) ..
    const int KU = KW + 3, S = KH * KU;
    constexpr int NSW = (7 * 10 + 3) / 4, PF = 6;   // 18 steps per wave at most
    bf16x8 bh[PF], bl[PF];
    auto bload = [&](int k, bf16x8 &h, bf16x8 &l) {   // step wave + 4 k (past the end: the last step again, unused)
        int s = wave + 4 * k;
        s = s < S ? s : S - 1;
        h = *(const bf16x8 *)(wn + (long long)s * 512 + lane * 8);
        l = *(const bf16x8 *)(wn + (long long)(S + s) * 512 + lane * 8);
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) bload(k, bh[k], bl[k]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = tid + 256 * it;
        const int u = i / PT_PIX, pp = i - u * PT_PIX;
        if (pp >= npp) continue;
        const float v[8] = {plo[it][0], plo[it][1], plo[it][2], plo[it][3], phi[it][0], phi[it][1], phi[it][2], phi[it][3]};
        acg_u32x4 hi, lo;
        acg_split8(v, hi, lo);
        *(acg_u32x4 *)&Ap[u * PT_PLANE + pp * 8] = hi;
        *(acg_u32x4 *)&Ap[PT_IMG + u * PT_PLANE + pp * 8] = lo;
    }
    __syncthreads();

    // ---- K steps: lane l: base pixel m = l & 15 of a block = row m >> 2, output pixels 4 (m & 3) .. + 3; channels 8 (l >> 4) ..
    const int kg = lane >> 4, m = lane & 15, br = m >> 2, bc = m & 3;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int a0 = kg * PT_PLANE + (br * PW + 4 * bc) * 8;   // block 0, window position (0, 0)
#pragma unroll
    for (int k = 0; k < NSW; ++k) {
        const int s = wave + 4 * k;
        if (s < S) {   // (wave-uniform)
            const int ry = s / KU, u = s - ry * KU;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int e = a0 + ((4 * mt + ry) * PW + u) * 8;
                const bf16x8 ah = *(const bf16x8 *)&Ap[e];
                const bf16x8 al = *(const bf16x8 *)&Ap[PT_IMG + e];
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[k % PF], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[k % PF], acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[k % PF], acc[mt], 0, 0, 0);
            }
        }
        if (k + PF < NSW) bload(k + PF, bh[k % PF], bl[k % PF]);   // into the registers the MFMAs just read
    }
    // ---- fold the four waves' partial blocks (fixed order: wave 0 + 1 + 2 + 3), then bias, activation and the C4 store:
    // lane l holds column (dxo, c) = (l & 15) of base pixels (row l >> 4, b = register index)
    __syncthreads();   // the patch is no longer read
    float *red = (float *)Ap;   // [wave][block][reg][lane]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * 2 + mt) * 4 + r) * 64 + lane] = acc[mt][r];
    __syncthreads();
    if (wave < 2) {   // wave w finishes block w
        const int mt = wave, c = lane & 3, dxo = (lane >> 2) & 3, q = lane >> 4;
        const float bv = (bias != nullptr && c < g.Cout) ? bias[c] : 0.f;
        const int gy = gy0 + 4 * mt + q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = red[((0 * 2 + mt) * 4 + r) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) v += red[((w * 2 + mt) * 4 + r) * 64 + lane];
            const int gx = gx0 + 4 * r + dxo;
            if (gy < g.GH && gx < g.GW)   // 16 consecutive lanes: 4 pixels x 4 channels = 64 contiguous bytes
                out[(((long long)n * g.Hout + gy) * g.Wout + gx) * 4 + c] = acg_apply_act(v + bv, g.act);
        }
    }
}

// the taps of the launch are a full KH x KW window (any order: the N-packed weights are indexed by window position)
static bool patchn_window(const Taps &t, int *ymin, int *xmin, int *KH, int *KW)
{
    int y0 = t.dy[0], y1 = t.dy[0], x0 = t.dx[0], x1 = t.dx[0];
    for (int i = 1; i < t.n; ++i) {
        y0 = t.dy[i] < y0 ? t.dy[i] : y0; y1 = t.dy[i] > y1 ? t.dy[i] : y1;
        x0 = t.dx[i] < x0 ? t.dx[i] : x0; x1 = t.dx[i] > x1 ? t.dx[i] : x1;
    }
    const int kh = y1 - y0 + 1, kw = x1 - x0 + 1;
    if (kh != kw || kh * kw != t.n || kh < 2 || kh > 7) return false;
    unsigned long long seen = 0;
    bool all_fwd = true, all_bwd = true;
    for (int i = 0; i < t.n; ++i) {
        const int pos = (t.dy[i] - y0) * kw + (t.dx[i] - x0);
        if (seen >> pos & 1ull) return false;
        seen |= 1ull << pos;
        // the window position must be the kernel position the packing assumed (pack_weight_npack_kernel): a forward tap
        // (kh, kw) = slab kh K + kw sits at window position (kh, kw), a data-gradient tap at (K-1-kh, K-1-kw)
        const int skh = t.w[i] / kw, skw = t.w[i] - skh * kw;
        all_fwd = all_fwd && skh == t.dy[i] - y0 && skw == t.dx[i] - x0;
        all_bwd = all_bwd && skh == kh - 1 - (t.dy[i] - y0) && skw == kw - 1 - (t.dx[i] - x0);
    }
    if (!all_fwd && !all_bwd) return false;
    *ymin = y0; *xmin = x0; *KH = kh; *KW = kw;
    return true;
}

bool acg_conv_patchn_ok(const Geom &g, const Taps &t)
{
    static const bool off = acg_debug_switch("ACG_NO_PATCHN"); // A/B switch
    if (off || g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA || g.thin || g.fold_p || g.stats) return false;
    if (g.Cin != 32 || g.Cout != 4 || g.os != 1 || g.is != 1 || g.oy0 != 0 || g.ox0 != 0 || t.n < 4) return false;
    int a, b, kh, kw;
    if (!patchn_window(t, &a, &b, &kh, &kw)) return false;
    return (PT_TH + kh - 1) * (PT_TW + kw - 1) <= PT_PIX && g.Hin >= 2 && g.Win >= 2;
}

// wn: the N-packed weights (behind the regular packed form: acg_packed_w{f,b}_elems)
int acg_conv_patchn_launch(const float *in, const void *wn, const float *bias, float *out, const Geom &g, const Taps &t, hipStream_t st)
{
    int ymin, xmin, KH, KW;
    ACG_REQUIRE(acg_conv_patchn_ok(g, t) && patchn_window(t, &ymin, &xmin, &KH, &KW), "conv_patchn_x3: unsupported geometry");
    const long long nimg = g.Mtot / ((long long)g.GH * g.GW);
    const long long in_bytes = nimg * g.Hin * g.Win * g.Cin * 4;
    ACG_REQUIRE(in_bytes < (1LL << 32), "conv_patchn_x3: gathered tensor exceeds the 4 GiB buffer-addressing limit");
    const long long blocks = nimg * ((g.GH + PT_TH - 1) / PT_TH) * ((g.GW + PT_TW - 1) / PT_TW);
    if (g.reflect)
        hipLaunchKernelGGL((conv_patchn_x3<true>), dim3((unsigned)blocks), dim3(256), 0, st, in, (const __bf16 *)wn, bias, out, g, ymin, xmin, KH, KW, (unsigned)in_bytes);
    else
        hipLaunchKernelGGL((conv_patchn_x3<false>), dim3((unsigned)blocks), dim3(256), 0, st, in, (const __bf16 *)wn, bias, out, g, ymin, xmin, KH, KW, (unsigned)in_bytes);
    ACG_CHECK_LAUNCH("conv_patchn_x3");
    acg_note_kernel("conv_patchn_x3<REFLECT=%d> (%dx%d window, %d K steps)", g.reflect ? 1 : 0, KH, KW, KH * (KW + 3));
    return ACG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The mirror image: layers that GATHER a C4 image tensor and write 32 channels — the 7x7 reflect stem forward, the data
// gradient of the 7x7 head (networks.py:159-160, 187-188).  The generic thin kernel (conv_igemm.hip, K flattened over (tap,
// 4 channels)) gathers 128 pixels x 8 taps from global memory per 32-deep stage, seven dependent stages per tile: latency,
// not bytes.  Here the 8 x 16 output tile's input patch (14 x 24 pixels x 8 bytes of bf16 hi, the same of lo: 5 KB) is
// loaded and split ONCE, and one K step is one kernel ROW: 8 window columns (the eighth has zero weights) x 4 channels = 32 —
// the two pixels x 4 channels of a lane's A fragment are 16 contiguous bytes of the patch row.  Weights in the row-packed
// form (pack_weight_trow_kernel), B fragments straight from L1 / L2, three rows ahead.
namespace {
constexpr int TR_PW = PT_TW + 8;                  // patch row: 16 output pixels + 8 window columns
constexpr int TR_PHMAX = PT_TH + 6;               // 7-row window
constexpr int TR_PLANE = TR_PHMAX * TR_PW * 4;    // bf16 elements of one hi (or lo) patch
}
template <bool REFLECT>
__global__ __launch_bounds__(256) void conv_thinrow_x3(const float *__restrict__ in, const __bf16 *__restrict__ wr,
                                                       const float *__restrict__ bias, float *__restrict__ out, Geom g,
                                                       int dymin, int dxmin, int KH, unsigned in_bytes)
{
    // the patch [hi|lo][row][pixel][4] bf16 (5.4 KB); afterwards the output tile [128 pixels][32] fp32 + 512 floats of scratch
    __shared__ __attribute__((aligned(16))) float smem[PT_TH * PT_TW * 32 + 512];
    static_assert(sizeof(smem) >= 2 * TR_PLANE * sizeof(__bf16), "patch fits the tile buffer");
    __bf16 *const Ap = (__bf16 *)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (g.GW + PT_TW - 1) / PT_TW, tiles_y = (g.GH + PT_TH - 1) / PT_TH;
    // XCD-aware tile order (bijective for any grid size): workgroups are dealt to the 8 XCDs round-robin, so consecutive tiles
    // — whose patches overlap — are made neighbours on ONE XCD and fetch the shared rows through one L2
    const int nwg_ = gridDim.x, bid_ = blockIdx.x, xq_ = nwg_ >> 3, xr_ = nwg_ & 7, xcd_ = bid_ & 7;
    int b = (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + (bid_ >> 3);
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty * PT_TH, gx0 = tx * PT_TW;
    const int PH = PT_TH + KH - 1;

    // ---- B fragments of the first PF kernel rows (lane l: column l & 15 of column tile ct, k-group l >> 4)
    constexpr int PF = 3;
    const int kg = lane >> 4, lr = lane & 15;
    bf16x8 bh[PF][2], bl[PF][2];
    auto bload = [&](int ry, bf16x8 (&h)[2], bf16x8 (&l)[2]) {   // (a row past the window loads the last row again: unused)
        const int r = ry < KH ? ry : KH - 1;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const long long e = ((long long)(r * 4 + kg) * 32 + ct * 16 + lr) * 8;
            h[ct] = *(const bf16x8 *)(wr + e);
            l[ct] = *(const bf16x8 *)(wr + (long long)KH * 1024 + e);
        }
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) bload(k, bh[k], bl[k]);

    // ---- the patch: pixel (py, px) <-> gathered pixel (gy0 + py + dymin, gx0 + px + dxmin); 16 bytes of fp32 C4 per pixel in,
    // 8 + 8 bytes of bf16 hi / lo out.  Columns past the window only meet zero weights but must hold finite values: they
    // are loaded like the others (zeros outside the image)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
    constexpr int NIT = (TR_PHMAX * TR_PW + 255) / 256;   // 2
    f32x4 pv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pp = tid + 256 * it;
        const int py = pp / TR_PW, px = pp - py * TR_PW;
        int iy = gy0 + py + dymin, ix = gx0 + px + dxmin;
        bool ok = py < PH;
        if (REFLECT) {
            ok = ok && iy > -g.Hin && iy < 2 * g.Hin - 1 && ix > -g.Win && ix < 2 * g.Win - 1;
            iy = iy < 0 ? -iy : iy;
            iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
        } else {
            ok = ok && (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
        }
        const unsigned off = (unsigned)(((n * g.Hin + iy) * g.Win + ix) * 4) * 4u;
        pv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off, ok), 0, 0));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pp = tid + 256 * it;
        if (pp >= PH * TR_PW) continue;
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        uint2 hi, lo;
        unsigned *hp = &hi.x, *lp = &lo.x;
#pragma unroll
        for (int q = 0; q < 2; ++q) { // the arithmetic of acg_split8
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){pv[it][2 * q], pv[it][2 * q + 1]}, bf16x2_t));
            const float ha = __builtin_bit_cast(float, h << 16), hb = __builtin_bit_cast(float, h & 0xffff0000u);
            hp[q] = h;
            lp[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){pv[it][2 * q] - ha, pv[it][2 * q + 1] - hb}, bf16x2_t));
        }
        *(uint2 *)&Ap[pp * 4] = hi;
        *(uint2 *)&Ap[TR_PLANE + pp * 4] = lo;
    }
    __syncthreads();

    // norm-backward sums (the epilogue below): the norm input x at the four pixels this thread stores, requested now — in flight
    // under the kernel rows (whole tiles: the launcher checks)
    f32x4 nsx[PT_TH * PT_TW * 8 / 256];
    if (g.ns_part != nullptr) {   // (uniform)
#pragma unroll
        for (int k = 0; k < PT_TH * PT_TW * 8 / 256; ++k) {
            const int idx = tid + 256 * k, pix = idx >> 3, c4 = idx & 7;
            nsx[k] = *(const f32x4 *)(g.ns_x + (((long long)n * g.Hout + gy0 + pix / PT_TW) * g.Wout + gx0 + pix % PT_TW) * g.Cout + c4 * 4);
        }
    }
    // ---- kernel rows: wave w owns tile rows 2w, 2w+1; lane l: pixel l & 15 of the row, window columns 2 (l >> 4), + 1
    f32x4 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto afrag = [&](int e) {   // 16 bytes at an 8-byte-aligned element offset: two ds_read_b64
        const u32x2 a = *(const u32x2 *)&Ap[e], c = *(const u32x2 *)&Ap[e + 4];
        return __builtin_bit_cast(bf16x8, (u32x4){a[0], a[1], c[0], c[1]});
    };
    const int a0 = ((wave * 2) * TR_PW + lr + 2 * kg) * 4;
#pragma unroll
    for (int ry = 0; ry < 7; ++ry) {
        if (ry < KH) {   // (uniform)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int e = a0 + (r + ry) * TR_PW * 4;
                const bf16x8 ah = afrag(e), al = afrag(TR_PLANE + e);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[ry % PF][ct], acc[r][ct], 0, 0, 0);
                    acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[ry % PF][ct], acc[r][ct], 0, 0, 0);
                    acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[ry % PF][ct], acc[r][ct], 0, 0, 0);
                }
            }
        }
        if (ry + PF < 7) bload(ry + PF, bh[ry % PF], bl[ry % PF]);
    }

    // ---- epilogue through LDS: lane l holds column l & 15 (+ 16 ct) of pixels 4 (l >> 4) .. + 3 of its rows; the tile leaves in
    // whole 128-byte pixel rows (16 bytes per lane), and — for an InstanceNorm behind the layer — with its (mean, M2) per channel
    // over the tile's 128 pixels (Geom.stats: any 128-pixel partition of the image serves acg_norm_stats_from_partials)
    __syncthreads();   // the patch is no longer read
    float *const tile = smem, *const red = smem + PT_TH * PT_TW * 32;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int co = ct * 16 + lr;
        const float bv = (bias != nullptr && co < g.Cout) ? bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                tile[((wave * 2 + r) * PT_TW + 4 * kg + k) * 32 + co] = acg_apply_act(acc[r][ct][k] + bv, g.stats != nullptr ? (int)ACG_ACT_NONE : g.act);
    }
    __syncthreads();
    // Norm-backward sums (acg_conv2d_bwd_data_sums, round 6: the data gradient of the 7x7 head is the gradient w.r.t. the output of
    // the norm + ReLU in front of it, networks.py:184-188): the norm's input ns_x is read at the pixels the tile stores and
    // sum gy, sum gy * xhat (gy = dx * [norm output > 0], recomputed from x) leave per tile — whole tiles, no bias / activation
    const bool nsum = g.ns_part != nullptr;   // (uniform)
    const int nc4 = tid & 7;                  // a thread's channel quad is the same for its four pixels
    f32x4 ns1 = {0.f, 0.f, 0.f, 0.f}, ns2 = ns1, nmu = ns1, nrs = ns1, nga = {1.f, 1.f, 1.f, 1.f}, nbe = nga;
    const bool ns_relu = nsum && g.ns_act == ACG_ACT_RELU;
    if (nsum) {
        nmu = *(const f32x4 *)(g.ns_mean + n * 32 + nc4 * 4);
        nrs = *(const f32x4 *)(g.ns_rstd + n * 32 + nc4 * 4);
        if (ns_relu) {
            nga = *(const f32x4 *)(g.ns_gamma + g.ns_gstride * n + nc4 * 4);
            nbe = *(const f32x4 *)(g.ns_beta + g.ns_gstride * n + nc4 * 4);
        }
    }
#pragma unroll
    for (int k = 0; k < PT_TH * PT_TW * 8 / 256; ++k) {   // 4 float4 per thread: 8 lanes per pixel
        const int idx = tid + 256 * k, pix = idx >> 3, c4 = idx & 7;
        const int gy = gy0 + pix / PT_TW, gx = gx0 + pix % PT_TW;
        if (gy < g.GH && gx < g.GW && c4 * 4 < g.Cout) {
            const long long o = (((long long)n * g.Hout + gy) * g.Wout + gx) * g.Cout + c4 * 4;
            const f32x4 v = *(const f32x4 *)&tile[pix * 32 + c4 * 4];
            *(f32x4 *)(out + o) = v;
            if (nsum) {
                const f32x4 xh = (nsx[k] - nmu) * nrs;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gyv = (ns_relu && !(xh[q] * nga[q] + nbe[q] > 0.f)) ? 0.f : v[q];   // same expression as norm_apply_kernel
                    ns1[q] += gyv;
                    ns2[q] += gyv * xh[q];
                }
            }
        }
    }
    if (nsum) {   // 32 threads share a channel quad: fold through the tile buffer in fixed order
        __syncthreads();
        *(f32x4 *)&tile[(tid >> 3) * 64 + nc4 * 8] = ns1;
        *(f32x4 *)&tile[(tid >> 3) * 64 + nc4 * 8 + 4] = ns2;
        __syncthreads();
        if (tid < 64) {
            const int c = tid & 31, which = tid >> 5;
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) a += tile[r * 64 + (c >> 2) * 8 + which * 4 + (c & 3)];
            g.ns_part[((long long)(n * tiles_y * tiles_x + ty * tiles_x + tx) * 2 + which) * 32 + c] = a;
        }
        return;
    }
    if (g.stats != nullptr) {   // (uniform; whole tiles only: the launcher checks)
        const int c = tid & 31, h = tid >> 5;   // channel c, pixels 16 h .. 16 h + 15
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += tile[(h * 16 + r) * 32 + c];
        red[h * 32 + c] = sum;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot += red[q * 32 + c];
        const float mu = tot * (1.f / (PT_TH * PT_TW));
        float sq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dlt = tile[(h * 16 + r) * 32 + c] - mu;
            sq += dlt * dlt;
        }
        __syncthreads();
        red[h * 32 + c] = sq;
        __syncthreads();
        if (h == 0 && c < g.Cout) {
            float m2 = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) m2 += red[q * 32 + c];
            float *o = g.stats + ((long long)(n * g.stats_cpi + g.stats_chunk0 + ty * tiles_x + tx) * 2) * g.Cout + c;
            o[0] = mu;
            o[g.Cout] = m2;
        }
    }
}

bool acg_conv_thinrow_ok(const Geom &g, const Taps &t)
{
    static const bool off = acg_debug_switch("ACG_NO_THINROW"); // A/B switch
    if (off || g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA || !g.thin || g.fold_p) return false;
    if (g.Cin != 4 || g.Cout != 32 || g.ncols_pad != 32 || g.os != 1 || g.is != 1 || g.oy0 != 0 || g.ox0 != 0 || t.n < 4) return false;
    if (g.stats != nullptr && (g.GH % PT_TH != 0 || g.GW % PT_TW != 0 || g.act != ACG_ACT_NONE)) return false;   // whole 128-pixel tiles
    if (g.ns_part != nullptr && (g.stats != nullptr || g.GH % PT_TH != 0 || g.GW % PT_TW != 0 || g.act != ACG_ACT_NONE || g.ns_x == nullptr ||
                                 g.ns_mean == nullptr || g.ns_rstd == nullptr || g.ns_mask != nullptr ||
                                 (g.ns_act != ACG_ACT_NONE && (g.ns_act != ACG_ACT_RELU || g.ns_gamma == nullptr || g.ns_beta == nullptr)) ||
                                 (g.ns_gstride != 0 && (g.ns_gstride < 32 || g.ns_gstride % 4 != 0))))
        return false;
    int a, b, kh, kw;
    return patchn_window(t, &a, &b, &kh, &kw) && g.Hin >= 2 && g.Win >= 2;
}

// wr: the row-packed weights (behind the regular packed form: acg_packed_w{f,b}_elems)
int acg_conv_thinrow_launch(const float *in, const void *wr, const float *bias, float *out, const Geom &g, const Taps &t, hipStream_t st)
{
    int ymin, xmin, KH, KW;
    ACG_REQUIRE(acg_conv_thinrow_ok(g, t) && patchn_window(t, &ymin, &xmin, &KH, &KW), "conv_thinrow_x3: unsupported geometry");
    ACG_REQUIRE(g.ns_part == nullptr || bias == nullptr, "conv_thinrow_x3: the norm sums ride on a plain data gradient (no bias)");
    const long long nimg = g.Mtot / ((long long)g.GH * g.GW);
    const long long in_bytes = nimg * g.Hin * g.Win * g.Cin * 4;
    ACG_REQUIRE(in_bytes < (1LL << 32), "conv_thinrow_x3: gathered tensor exceeds the 4 GiB buffer-addressing limit");
    const long long blocks = nimg * ((g.GH + PT_TH - 1) / PT_TH) * ((g.GW + PT_TW - 1) / PT_TW);
    if (g.reflect)
        hipLaunchKernelGGL((conv_thinrow_x3<true>), dim3((unsigned)blocks), dim3(256), 0, st, in, (const __bf16 *)wr, bias, out, g, ymin, xmin, KH, (unsigned)in_bytes);
    else
        hipLaunchKernelGGL((conv_thinrow_x3<false>), dim3((unsigned)blocks), dim3(256), 0, st, in, (const __bf16 *)wr, bias, out, g, ymin, xmin, KH, (unsigned)in_bytes);
    ACG_CHECK_LAUNCH("conv_thinrow_x3");
    acg_note_kernel("conv_thinrow_x3<REFLECT=%d%s> (%dx%d window)", g.reflect ? 1 : 0, g.ns_part != nullptr ? ",SUMS=1" : "", KH, KW);
    return ACG_OK;
}

// eligibility: bf16x3, 32 gathered channels, 16 (or, for an image tensor stored C4, 4) stored output channels, unit strides,
// tap window <= 7x7
bool acg_conv_patch16_ok(const Geom &g, const Taps &t)
{
    if (g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA || g.thin || g.fold_p) return false;
    if (g.Cin != 32 || (g.Cout != 16 && g.Cout != 4) || g.os != 1 || g.is != 1 || g.oy0 != 0 || g.ox0 != 0 || t.n < 1) return false;
    int ymin = t.dy[0], ymax = t.dy[0], xmin = t.dx[0], xmax = t.dx[0];
    for (int i = 1; i < t.n; ++i) {
        ymin = t.dy[i] < ymin ? t.dy[i] : ymin; ymax = t.dy[i] > ymax ? t.dy[i] : ymax;
        xmin = t.dx[i] < xmin ? t.dx[i] : xmin; xmax = t.dx[i] > xmax ? t.dx[i] : xmax;
    }
    return (PT_TH + ymax - ymin) * (PT_TW + xmax - xmin) <= PT_PIX && g.Hin >= 2 && g.Win >= 2;
}

int acg_conv_patch16_launch(const float *in, const void *wp, const float *bias, float *out, const Geom &g, const Taps &t0,
                            long long n_w_elems, hipStream_t st)
{
    const Taps t = acg_taps_pack(t0);
    int ymin = t.dy[0], ymax = t.dy[0], xmin = t.dx[0], xmax = t.dx[0];
    for (int i = 1; i < t.n; ++i) {
        ymin = t.dy[i] < ymin ? t.dy[i] : ymin; ymax = t.dy[i] > ymax ? t.dy[i] : ymax;
        xmin = t.dx[i] < xmin ? t.dx[i] : xmin; xmax = t.dx[i] > xmax ? t.dx[i] : xmax;
    }
    const int PH = PT_TH + ymax - ymin, PW = PT_TW + xmax - xmin;
    const long long nimg = g.Mtot / ((long long)g.GH * g.GW);
    const long long in_bytes = nimg * g.Hin * g.Win * g.Cin * 4;
    ACG_REQUIRE(in_bytes < (1LL << 32), "conv_patch16_x3: gathered tensor exceeds the 4 GiB buffer-addressing limit");
    const long long blocks = nimg * ((g.GH + PT_TH - 1) / PT_TH) * ((g.GW + PT_TW - 1) / PT_TW);
    if (g.reflect)
        hipLaunchKernelGGL((conv_patch16_x3<true>), dim3((unsigned)blocks), dim3(256), 0, st, in, (const __bf16 *)wp, bias, out, g, t,
                           ymin, xmin, PH, PW, (unsigned)in_bytes, n_w_elems);
    else
        hipLaunchKernelGGL((conv_patch16_x3<false>), dim3((unsigned)blocks), dim3(256), 0, st, in, (const __bf16 *)wp, bias, out, g, t,
                           ymin, xmin, PH, PW, (unsigned)in_bytes, n_w_elems);
    ACG_CHECK_LAUNCH("conv_patch16_x3");
    acg_note_kernel("conv_patch16_x3<REFLECT=%d>", g.reflect ? 1 : 0);
    return ACG_OK;
}
