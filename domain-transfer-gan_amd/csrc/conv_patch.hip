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
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int gy0 = ty * PT_TH, gx0 = tx * PT_TW;

    // ---- the patch: pixel pp = py * PW + px  <->  gathered pixel (gy0 + py + dymin, gx0 + px + dxmin); consecutive
    // lanes take consecutive pixels of one 8-channel plane (conflict-free 16-byte LDS stores)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, in_bytes, 0x00020000);
    const int npp = PH * PW;
    for (int i = tid; i < 4 * PT_PIX; i += 256) {
        const int u = i / PT_PIX, pp = i - u * PT_PIX; // plane, patch pixel
        if (pp >= npp) continue;
        const int py = pp / PW, px = pp - py * PW;
        int iy = gy0 + py + dymin, ix = gx0 + px + dxmin;
        bool ok = true;
        if (REFLECT) {
            ok = iy > -g.Hin && iy < 2 * g.Hin - 1 && ix > -g.Win && ix < 2 * g.Win - 1; // pixels of partial tiles
            iy = iy < 0 ? -iy : iy;
            iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
        } else {
            ok = (unsigned)iy < (unsigned)g.Hin && (unsigned)ix < (unsigned)g.Win;
        }
        const unsigned off = (unsigned)(((n * g.Hin + iy) * g.Win + ix) * g.Cin + 8 * u) * 4u;
        const f32x4 lo4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off, ok), 0, 0));
        const f32x4 hi4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, acg_masked_off(off + 16u, ok), 0, 0));
        const float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
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
    for (int t = 0; t < taps.n; ++t) {
        const int pk = taps.pk[t];
        const int dy = ((pk << 24) >> 24) - dymin, dx = ((pk << 16) >> 24) - dxmin, tw = pk >> 16;
        const __bf16 *wt = wp + tw * tap_stride + boff;
        const bf16x8 bh = *(const bf16x8 *)wt;
        const bf16x8 bl = *(const bf16x8 *)(wt + w_lo_elems);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int pp = (wave * 2 + r + dy) * PW + dx + lr;
            const bf16x8 ah = *(const bf16x8 *)&Ap[pl * PT_PLANE + pp * 8];
            const bf16x8 al = *(const bf16x8 *)&Ap[PT_IMG + pl * PT_PLANE + pp * 8];
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[r], 0, 0, 0);
            acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[r], 0, 0, 0);
        }
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
