// Convolution entry points of the C ABI: geometry -> tap lists -> kernel launches, plus the
// small support kernels (weight packing, reflection-pad fold, split-K reduction, bias gradient)
// and the naive "direct" kernels kept as an on-device cross-check of the MFMA path.
#include "conv_internal.h"

int g_acg_conv_impl = ACG_IMPL_MFMA;
int g_acg_precision = ACG_PREC_BF16X3;
extern "C" int acg_set_conv_precision(int prec)
{
    ACG_REQUIRE(prec == ACG_PREC_F32 || prec == ACG_PREC_BF16 || prec == ACG_PREC_BF16X3, "acg_set_conv_precision: unknown precision %d", prec);
    g_acg_precision = prec;
    return ACG_OK;
}
// packed weights are bf16 (hi, and for BF16X3 also lo right behind it) whenever the bf16 matrix pipe is used
static bool use_bf16() { return g_acg_precision != ACG_PREC_F32 && g_acg_conv_impl == ACG_IMPL_MFMA; }
// thin-channel K-flattening (fp32 MFMA kernels): the gathered tensor has <= 4 real channels and K > 1
// (thin layers keep the fp32-tile thin kernels — loader, LDS tiles, packed weights — in every mode: they beat the padded bf16
// path; outside the strict fp32 mode their products run as bf16x3, conv_igemm.hip / conv_wgrad.hip X3)
static bool thin_ok(int creal, int K) { return creal >= 1 && creal <= 4 && K > 1 && g_acg_conv_impl == ACG_IMPL_MFMA; }
// a layer is treated as thin on exactly one side (3->3 convolutions do not occur on this path and stay regular)
static bool thin_in(const acg_conv_desc *d) { return thin_ok(d->Cir, d->K) && !thin_ok(d->Cor, d->K); }
static bool thin_out(const acg_conv_desc *d) { return thin_ok(d->Cor, d->K) && !thin_ok(d->Cir, d->K); }
// the VALU thin-output kernel pays off up to 64 gathered channels (>= 4 pixels per wave); wider layers use the MFMA kernel
// widest gathered tensor the VALU thin-output kernel takes.  It beats the padded 32-column MFMA tile only against the
// fp32 matrix pipe (1.7 vs 2.9 ms on the 7x7 32->3 head); on the bf16 pipe the MFMA tile wins 2x (0.87 ms in bf16x3).
static int thin_valu_max() { return g_acg_precision == ACG_PREC_F32 || g_acg_conv_impl != ACG_IMPL_MFMA ? 64 : 0; }
// ... and it needs C/4 lanes per pixel to divide a wave: 16, 32 or 64 stored channels (others take the MFMA tile)
static bool thin_valu_c(int C) { return C <= thin_valu_max() && (C == 16 || C == 32 || C == 64); }
static bool thin_out_valu_fwd(const acg_conv_desc *d) { return thin_out(d) && thin_valu_c(d->Ci); }
static bool thin_in_valu_dgrad(const acg_conv_desc *d) { return thin_in(d) && thin_valu_c(d->Co); }
extern "C" int acg_set_conv_impl(int impl)
{
    ACG_REQUIRE(impl == ACG_IMPL_MFMA || impl == ACG_IMPL_DIRECT, "acg_set_conv_impl: unknown impl %d", impl);
    g_acg_conv_impl = impl;
    return ACG_OK;
}

// ------------------------------------------------------------------------------------------
// weight packing: OIHW (real Or x Ir) -> wf [tap][Ci/8][CoP][8], wb [tap][Co/8][CiP][8]
// ------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float *__restrict__ w, int Or, int Ir, int K, int Ci, int Co, int CoP,
                                   int CiP, float *__restrict__ wf, float *__restrict__ wb)
{
    const int KK = K * K;
    const long long nf = (long long)KK * (Ci / 8) * CoP * 8;
    const long long nb = (long long)KK * (Co / 8) * CiP * 8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nf + nb;
         i += (long long)gridDim.x * blockDim.x) {
        if (i < nf) {
            if (wf == nullptr) continue;
            long long r = i;
            const int c8 = (int)(r % 8); r /= 8;
            const int co = (int)(r % CoP); r /= CoP;
            const int cb = (int)(r % (Ci / 8)); r /= (Ci / 8);
            const int tap = (int)r;
            const int ci = cb * 8 + c8;
            wf[i] = (co < Or && ci < Ir) ? w[((long long)co * Ir + ci) * KK + tap] : 0.f;
        } else {
            if (wb == nullptr) continue;
            long long r = i - nf;
            const int c8 = (int)(r % 8); r /= 8;
            const int ci = (int)(r % CiP); r /= CiP;
            const int cb = (int)(r % (Co / 8)); r /= (Co / 8);
            const int tap = (int)r;
            const int co = cb * 8 + c8;
            wb[i - nf] = (co < Or && ci < Ir) ? w[((long long)co * Ir + ci) * KK + tap] : 0.f;
        }
    }
}

// bf16 packing: wf16 [tap][Ci/16][CoP][16], wb16 [tap][Co/16][CiP][16] (same element counts, half the bytes)
// split != 0: also write lo = bf16(w - float(hi)) at [n_elems ...) of each buffer (the buffers are sized in floats)
// i0 / stride: the calling thread's first element and step over the layer's nf + nb elements
__device__ __forceinline__ void pack_bf16_body(const float *__restrict__ w, int Or, int Ir, int K, int Ci, int Co, int CoP, int CiP,
                                               __bf16 *__restrict__ wf, __bf16 *__restrict__ wb, int split, long long i0, long long stride)
{
    const int KK = K * K;
    const long long nf = (long long)KK * (Ci / 16) * CoP * 16;
    const long long nb = (long long)(KK + (K == 3 ? 3 : 0)) * (Co / 16) * CiP * 16; // wb_slabs(K)
    for (long long i = i0; i < nf + nb; i += stride) {
        if (i < nf) {
            if (wf == nullptr) continue;
            long long r = i;
            const int c16 = (int)(r % 16); r /= 16;
            const int co = (int)(r % CoP); r /= CoP;
            const int cb = (int)(r % (Ci / 16)); r /= (Ci / 16);
            const int tap = (int)r;
            const int ci = cb * 16 + c16;
            const float v = (co < Or && ci < Ir) ? w[((long long)co * Ir + ci) * KK + tap] : 0.f;
            const __bf16 hi = (__bf16)v;
            wf[i] = hi;
            if (split) wf[nf + i] = (__bf16)(v - (float)hi);
        } else {
            if (wb == nullptr) continue;
            long long r = i - nf;
            const int c16 = (int)(r % 16); r /= 16;
            const int ci = (int)(r % CiP); r /= CiP;
            const int cb = (int)(r % (Co / 16)); r /= (Co / 16);
            const int tap = (int)r;
            const int co = cb * 16 + c16;
            float v = 0.f;
            if (co < Or && ci < Ir) {
                const float *wv = w + ((long long)co * Ir + ci) * KK;
                v = tap < KK ? wv[tap] : wv[tap - KK] + wv[6 + tap - KK]; // slab 9 + kw: kernel rows 0 and 2 together
            }
            const __bf16 hi = (__bf16)v;
            wb[i - nf] = hi;
            if (split) wb[nb + i - nf] = (__bf16)(v - (float)hi);
        }
    }
}
__global__ void pack_weight_bf16_kernel(const float *__restrict__ w, int Or, int Ir, int K, int Ci, int Co, int CoP,
                                        int CiP, __bf16 *__restrict__ wf, __bf16 *__restrict__ wb, int split)
{
    pack_bf16_body(w, Or, Ir, K, Ci, Co, CoP, CiP, wf, wb, split, blockIdx.x * (long long)blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}
// every regular (non-thin) layer of a network in ONE launch (acg_pack_conv_weights_multi): the packed copies are refreshed once
// per optimiser step, and one launch per layer was 68 five-microsecond kernels per training step
#define PACK_MAX_ITEMS 48
struct PackTable {
    acg_pack_item it[PACK_MAX_ITEMS];
    int first[PACK_MAX_ITEMS + 1];   // first workgroup of item i
    short cop[PACK_MAX_ITEMS], cip[PACK_MAX_ITEMS];   // acg_ncols_pad of the item's Co / Ci
    int n, split;
};
__global__ __launch_bounds__(256) void pack_weight_bf16_multi_kernel(PackTable T)
{
    int k = 0;
    while (k + 1 < T.n && (int)blockIdx.x >= T.first[k + 1]) ++k;   // (uniform)
    const acg_pack_item q = T.it[k];
    const int nblk = T.first[k + 1] - T.first[k];
    pack_bf16_body(q.w, q.Or, q.Ir, q.K, q.Ci, q.Co, T.cop[k], T.cip[k], (__bf16 *)q.wf, (__bf16 *)q.wb, T.split,
                   ((long long)blockIdx.x - T.first[k]) * 256 + threadIdx.x, (long long)nblk * 256);
}

// thin packing: rows are k = tap*4 + c (c < 4), grouped in 8-chunks: out[kc][col][8].
// mode 0 (forward operand):  c = input channel,  col = output channel -> w[col][c][tap]
// mode 1 (data-gradient):    c = output channel, col = input channel  -> w[c][col][tap]
__global__ void pack_weight_thin_kernel(const float *__restrict__ w, int Or, int Ir, int KK, int ColP, int mode,
                                        float *__restrict__ out)
{
    const int nkc = 4 * ((KK + 7) / 8);
    const long long total = (long long)nkc * ColP * 8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c8 = (int)(r % 8); r /= 8;
        const int col = (int)(r % ColP); r /= ColP;
        const int kflat = (int)r * 8 + c8;
        const int tap = kflat >> 2, c = kflat & 3;
        float v = 0.f;
        if (tap < KK) {
            if (mode == 0) { if (col < Or && c < Ir) v = w[((long long)col * Ir + c) * KK + tap]; }
            else           { if (c < Or && col < Ir) v = w[((long long)c * Ir + col) * KK + tap]; }
        }
        out[i] = v;
    }
}

// thin-OUTPUT packing: out[(tap*Kc + k)*4 + n], n < 4 output columns, k over the Kc gathered channels.
// mode 0 (forward, Cout <= 4): k = input channel, n = output channel -> w[n][k][tap]
// mode 1 (data-gradient, Cin <= 4): k = output channel, n = input channel -> w[k][n][tap]
__global__ void pack_weight_thinN_kernel(const float *__restrict__ w, int Or, int Ir, int KK, int Kc, int mode,
                                         float *__restrict__ out)
{
    const long long total = (long long)KK * Kc * 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i & 3);
        const int k = (int)((i >> 2) % Kc);
        const int tap = (int)((i >> 2) / Kc);
        float v = 0.f;
        if (mode == 0) { if (n < Or && k < Ir) v = w[((long long)n * Ir + k) * KK + tap]; }
        else           { if (k < Or && n < Ir) v = w[((long long)k * Ir + n) * KK + tap]; }
        out[i] = v;
    }
}

// Convolutions whose OUTPUT has <= 4 real channels (7x7 32->3 head + tanh, PatchGAN heads, data gradients into
// image tensors).  N = 4 cannot feed a 32-wide MFMA tile, so this is a VALU kernel (HBM/L1-friendly form):
// LPP = Cin/4 lanes cooperate on one output pixel — lane j owns channels 4j..4j+3, so a wave reads whole
// contiguous pixel rows (coalesced) — each lane keeps 4 partial sums, the weights [tap][ci][4] are staged once
// per block in LDS, and the LPP partials are folded with wave shuffles.  Same Geom/Taps formulation as the MFMA
// kernel; lane 0 of each pixel writes the full C16 row (pad channels = 0).
#define THIN_PIX_ITERS 32
template <int LPP>
__global__ __launch_bounds__(256) void thin_out_conv_kernel(const float *__restrict__ in, const float *__restrict__ wn,
                                                            const float *__restrict__ bias, float *__restrict__ out,
                                                            Geom g, Taps taps)
{
    extern __shared__ __attribute__((aligned(16))) float wsm[]; // [taps.n][Cin][4]
    constexpr int PPB = 256 / LPP; // pixels per block pass
    const int tid = threadIdx.x, j = tid % LPP, pl = tid / LPP;
    const int wtot = taps.n * g.Cin; // float4 entries; slab order follows the tap LIST (taps.w indexes global slabs)
    for (int i = tid; i < wtot; i += 256) {
        const int t = i / g.Cin, k = i - t * g.Cin;
        *(f32x4 *)&wsm[i * 4] = *(const f32x4 *)(wn + ((long long)taps.w[t] * g.Cin + k) * 4);
    }
    __shared__ int tdy[64], tdx[64];
    if (tid < 64) { tdy[tid] = tid < taps.n ? taps.dy[tid] : 0; tdx[tid] = tid < taps.n ? taps.dx[tid] : 0; }
    __syncthreads();
    const int GHW = g.GH * g.GW;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 bv = z;
    if (bias != nullptr) bv = *(const f32x4 *)bias;
    for (int it = 0; it < THIN_PIX_ITERS; ++it) {
        const long long m = ((long long)blockIdx.x * THIN_PIX_ITERS + it) * PPB + pl;
        const bool mok = m < g.Mtot; // uniform across the LPP lanes of a pixel
        const long long mm = mok ? m : 0;
        const int n = (int)(mm / GHW);
        const int r = (int)(mm - (long long)n * GHW);
        const int gy = r / g.GW, gx = r - gy * g.GW;
        f32x4 acc = z;
        const float *img = in + (long long)n * g.Hin * g.Win * g.Cin + 4 * j;
        for (int t0 = 0; t0 < taps.n; t0 += 4) { // 4 taps per trip: their gathers are issued together
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u;
                const int tt = t < taps.n ? t : 0;
                int iy = gy * g.is + tdy[tt], ix = gx * g.is + tdx[tt];
                bool ok = mok && t < taps.n;
                if (g.reflect) {
                    iy = iy < 0 ? -iy : iy;
                    iy = iy >= g.Hin ? 2 * (g.Hin - 1) - iy : iy;
                    ix = ix < 0 ? -ix : ix;
                    ix = ix >= g.Win ? 2 * (g.Win - 1) - ix : ix;
                } else {
                    ok = ok && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win;
                }
                v[u] = z;
                if (ok) v[u] = *(const f32x4 *)(img + ((long long)iy * g.Win + ix) * g.Cin);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + u < taps.n ? t0 + u : 0; // v[u] is zero for the tail taps
                const float *wt = &wsm[((long long)t * g.Cin + 4 * j) * 4];
                acc += v[u][0] * *(const f32x4 *)(wt) + v[u][1] * *(const f32x4 *)(wt + 4) +
                       v[u][2] * *(const f32x4 *)(wt + 8) + v[u][3] * *(const f32x4 *)(wt + 12);
            }
        }
#pragma unroll
        for (int sft = LPP / 2; sft >= 1; sft >>= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += __shfl_xor(acc[k], sft, 64);
        }
        if (j == 0 && mok) {
            acc += bv;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] = acg_apply_act(acc[k], g.act);
            float *o = out + (((long long)n * g.Hout + (gy * g.os + g.oy0)) * g.Wout + (gx * g.os + g.ox0)) * g.Cout;
            *(f32x4 *)o = acc;
            for (int c = 4; c < g.Cout; c += 4) *(f32x4 *)(o + c) = z;
        }
    }
}

static int thin_out_launch(const float *in, const float *wn, const float *bias, float *out, const Geom &g, const Taps &t,
                           hipStream_t st)
{
    if (g.Mtot <= 0 || t.n <= 0) return ACG_OK;
    const int lpp = g.Cin / 4;
    ACG_REQUIRE(lpp == 4 || lpp == 8 || lpp == 16 || lpp == 32 || lpp == 64, "thin_out_conv: Cin=%d unsupported", g.Cin);
    const size_t lds = (size_t)t.n * g.Cin * 4 * sizeof(float);
    ACG_REQUIRE(lds <= 160 * 1024, "thin_out_conv: weights (%zu B) exceed LDS", lds);
    const int ppb = (256 / lpp) * THIN_PIX_ITERS;
    dim3 grid(acg_cdiv(g.Mtot, ppb)), block(256);
#define THIN_LAUNCH(L) hipLaunchKernelGGL((thin_out_conv_kernel<L>), grid, block, lds, st, in, wn, bias, out, g, t)
    switch (lpp) {
    case 4: THIN_LAUNCH(4); break;
    case 8: THIN_LAUNCH(8); break;
    case 16: THIN_LAUNCH(16); break;
    case 32: THIN_LAUNCH(32); break;
    default: THIN_LAUNCH(64); break;
    }
#undef THIN_LAUNCH
    ACG_CHECK_LAUNCH("thin_out_conv_kernel");
    return ACG_OK;
}

// Packed weights are laid out on channel counts padded to 16 whatever the stored width of the activation tensors: an image
// tensor (<= 4 real channels) may be stored with 4 channels ("C4", acg_conv_desc), its layer's weights are packed as before.
static inline int c16(int c) { return (c + 15) / 16 * 16; }
// The thin-OUTPUT layers with 32 gathered channels (7x7 32 -> nc head, data gradient of the nc -> 32 stem) carry a second,
// "N-packed" weight form behind the regular one (conv_patch.hip conv_patchn_x3): K rows x (K + 3) window columns of 1 KB
// hi + 1 KB lo, where the 16 MFMA columns are (4 horizontally adjacent output pixels) x (4 channels).
static inline size_t npack_elems(int K) { return (size_t)K * (K + 3) * 512; }   // floats
static inline bool npack_wf(int K, int Ci, int Co) { return K > 1 && K <= 7 && c16(Co) == 16 && c16(Ci) == 32; }   // thin-out forward
static inline bool npack_wb(int K, int Ci, int Co) { return K > 1 && K <= 7 && c16(Ci) == 16 && c16(Co) == 32; }   // thin-in data gradient
// ... and the thin-INPUT layers with 32 output channels (7x7 nc -> 32 stem forward, data gradient of the 32 -> nc head) a
// "row-packed" form (conv_patch.hip conv_thinrow_x3): per kernel row one 32-deep K step = 8 window columns (the eighth zero)
// x 4 channels, [row][k-group 4][32 columns][8] bf16 hi + lo = 4 KB per row.
static inline size_t trow_elems(int K) { return (size_t)K * 1024; }   // floats
static inline bool trow_wf(int K, int Ci, int Co) { return K > 1 && K <= 7 && c16(Ci) == 16 && c16(Co) == 32; }   // thin-in forward
static inline bool trow_wb(int K, int Ci, int Co) { return K > 1 && K <= 7 && c16(Co) == 16 && c16(Ci) == 32; }   // thin-out data gradient
static size_t wf_regular_elems(int K, int Ci, int Co) { return (size_t)K * K * (c16(Ci) / 8) * acg_ncols_pad(c16(Co)) * 8; }
extern "C" size_t acg_packed_wf_elems(int K, int Ci, int Co)
{
    return wf_regular_elems(K, Ci, Co) + (npack_wf(K, Ci, Co) ? npack_elems(K) : 0) + (trow_wf(K, Ci, Co) ? trow_elems(K) : 0);
}
// K == 3: three more slabs behind the nine taps, 9 + kw = w[0][kw] + w[2][kw] (bf16 packings only): what the kernel row that
// reads a mirrored row uses in the un-padded data gradient of a reflection-padded layer (Geom.unpad)
static inline int wb_slabs(int K) { return K * K + (K == 3 ? 3 : 0); }
static size_t wb_regular_elems(int K, int Ci, int Co) { return (size_t)wb_slabs(K) * (c16(Co) / 8) * acg_ncols_pad(c16(Ci)) * 8; }
extern "C" size_t acg_packed_wb_elems(int K, int Ci, int Co)
{
    return wb_regular_elems(K, Ci, Co) + (npack_wb(K, Ci, Co) ? npack_elems(K) : 0) + (trow_wb(K, Ci, Co) ? trow_elems(K) : 0);
}

// Row-packed thin-K weights: out[hi | lo][ry][kg (4)][col (32)][8]: k = 8 kg + j = window column kw = 2 kg + (j >> 2), gathered
// channel ch = j & 3.  mode 0 (forward of a thin-input layer): value w[col][ch][ry][kw]; mode 1 (data gradient of a
// thin-output layer; gathered channel = its output channel, col = its input channel): the flipped kernel,
// w[ch][col][K-1-ry][K-1-kw].  Zero for kw >= K (the eighth column of a 7-wide row).
__global__ void pack_weight_trow_kernel(const float *__restrict__ w, int Or, int Ir, int K, int mode, __bf16 *__restrict__ out)
{
    const int total = K * 1024;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, col = (i >> 3) & 31, kg = (i >> 8) & 3, ry = i >> 10;
        const int kw = 2 * kg + (j >> 2), ch = j & 3;
        float v = 0.f;
        if (kw < K) {
            if (mode == 0) { if (col < Or && ch < Ir) v = w[(((long long)col * Ir + ch) * K + ry) * K + kw]; }
            else           { if (ch < Or && col < Ir) v = w[(((long long)ch * Ir + col) * K + (K - 1 - ry)) * K + (K - 1 - kw)]; }
        }
        const __bf16 hi = (__bf16)v;
        out[i] = hi;
        out[total + i] = (__bf16)(v - (float)hi);
    }
}

// N-packed weights: out[hi | lo][(ry * (K + 3) + u)][kg (4)][col (16)][8]: k = 8 kg + j is the gathered channel, col = 4 dxo + c
// the output pixel offset dxo and channel c.  mode 0 (forward of a thin-output layer): k = input channel, c = output channel,
// value w[c][k][ry][u - dxo]; mode 1 (data gradient of a thin-input layer): k = output channel, c = input channel, the
// window walks the flipped kernel: value w[k][c][K-1-ry][K-1-(u - dxo)].  Zero where u - dxo falls outside the kernel.
__global__ void pack_weight_npack_kernel(const float *__restrict__ w, int Or, int Ir, int K, int mode, __bf16 *__restrict__ out)
{
    const int KU = K + 3, total = K * KU * 512;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 7, col = (i >> 3) & 15, kg = (i >> 7) & 3, slab = i >> 9;
        const int ry = slab / KU, u = slab - ry * KU, dxo = col >> 2, c = col & 3, k = kg * 8 + j, kw = u - dxo;
        float v = 0.f;
        if (kw >= 0 && kw < K) {
            if (mode == 0) { if (c < Or && k < Ir) v = w[(((long long)c * Ir + k) * K + ry) * K + kw]; }
            else           { if (k < Or && c < Ir) v = w[(((long long)k * Ir + c) * K + (K - 1 - ry)) * K + (K - 1 - kw)]; }
        }
        const __bf16 hi = (__bf16)v;
        out[i] = hi;
        out[total + i] = (__bf16)(v - (float)hi);
    }
}

extern "C" int acg_pack_conv_weight(const float *w, int Or, int Ir, int K, int Ci, int Co, float *wf, float *wb,
                                    void *stream)
{
    ACG_REQUIRE(Ci % 16 == 0 && Co % 16 == 0 && Or <= Co && Ir <= Ci && K >= 1 && K <= 7,
                "acg_pack_conv_weight: bad dims Or=%d Ir=%d K=%d Ci=%d Co=%d", Or, Ir, K, Ci, Co);
    const long long n = (long long)acg_packed_wf_elems(K, Ci, Co) + (long long)acg_packed_wb_elems(K, Ci, Co);
    const int blocks = acg_cdiv(n, 256) > 2048 ? 2048 : acg_cdiv(n, 256);
    const bool thin_i = thin_ok(Ir, K), thin_o = thin_ok(Or, K);
    if ((thin_i || thin_o) && !(thin_i && thin_o)) {
        // thin layers: each operand in the layout its kernel wants (all fit in the regular-size buffers)
        //   Cin <= 4 : wf = thin-K (flattened taps, MFMA fwd)   wb = thin-N (VALU data gradient into the image)
        //   Cout <= 4: wf = thin-N (VALU forward)               wb = thin-K (MFMA data gradient gathers thin dy)
        hipStream_t st = (hipStream_t)stream;
        // the NON-thin operand of a thin layer (only when its kernel is the regular MFMA one) follows the precision mode
        auto regular = [&](float *of, float *ob) {
            if (use_bf16())
                hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(blocks), dim3(256), 0, st, w, Or, Ir, K, Ci, Co, acg_ncols_pad(Co),
                                   acg_ncols_pad(Ci), (__bf16 *)of, (__bf16 *)ob, (int)(g_acg_precision == ACG_PREC_BF16X3));
            else
                hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, st, w, Or, Ir, K, Ci, Co, acg_ncols_pad(Co),
                                   acg_ncols_pad(Ci), of, ob);
        };
        if (thin_i) {
            if (wf) hipLaunchKernelGGL(pack_weight_thin_kernel, dim3(64), dim3(256), 0, st, w, Or, Ir, K * K, acg_ncols_pad(Co), 0, wf);
            if (wb && thin_valu_c(Co)) hipLaunchKernelGGL(pack_weight_thinN_kernel, dim3(64), dim3(256), 0, st, w, Or, Ir, K * K, Co, 1, wb);
            else if (wb) regular(nullptr, wb);
            if (wb && npack_wb(K, Ci, Co) && g_acg_precision == ACG_PREC_BF16X3 && use_bf16())
                hipLaunchKernelGGL(pack_weight_npack_kernel, dim3(64), dim3(256), 0, st, w, Or, Ir, K, 1, (__bf16 *)(wb + wb_regular_elems(K, Ci, Co)));
            if (wf && trow_wf(K, Ci, Co) && g_acg_precision == ACG_PREC_BF16X3 && use_bf16())
                hipLaunchKernelGGL(pack_weight_trow_kernel, dim3(28), dim3(256), 0, st, w, Or, Ir, K, 0, (__bf16 *)(wf + wf_regular_elems(K, Ci, Co)));
        } else {
            if (wf && thin_valu_c(Ci)) hipLaunchKernelGGL(pack_weight_thinN_kernel, dim3(64), dim3(256), 0, st, w, Or, Ir, K * K, Ci, 0, wf);
            else if (wf) regular(wf, nullptr);
            if (wb) hipLaunchKernelGGL(pack_weight_thin_kernel, dim3(64), dim3(256), 0, st, w, Or, Ir, K * K, acg_ncols_pad(Ci), 1, wb);
            if (wf && npack_wf(K, Ci, Co) && g_acg_precision == ACG_PREC_BF16X3 && use_bf16())
                hipLaunchKernelGGL(pack_weight_npack_kernel, dim3(64), dim3(256), 0, st, w, Or, Ir, K, 0, (__bf16 *)(wf + wf_regular_elems(K, Ci, Co)));
            if (wb && trow_wb(K, Ci, Co) && g_acg_precision == ACG_PREC_BF16X3 && use_bf16())
                hipLaunchKernelGGL(pack_weight_trow_kernel, dim3(28), dim3(256), 0, st, w, Or, Ir, K, 1, (__bf16 *)(wb + wb_regular_elems(K, Ci, Co)));
        }
        ACG_CHECK_LAUNCH("pack_weight_thin_kernel");
        return ACG_OK;
    }
    if (use_bf16())
        hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Or, Ir, K, Ci, Co,
                           acg_ncols_pad(Co), acg_ncols_pad(Ci), (__bf16 *)wf, (__bf16 *)wb,
                           (int)(g_acg_precision == ACG_PREC_BF16X3));
    else
        hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Or, Ir, K, Ci, Co,
                           acg_ncols_pad(Co), acg_ncols_pad(Ci), wf, wb);
    ACG_CHECK_LAUNCH("pack_weight_kernel");
    return ACG_OK;
}

// the same for the regular layers of a whole network at once (bf16 / bf16x3 arithmetic; thin layers keep acg_pack_conv_weight)
extern "C" int acg_pack_conv_weights_multi_supported(int Or, int Ir, int K)
{
    const bool thin_i = thin_ok(Ir, K), thin_o = thin_ok(Or, K);
    return use_bf16() && !((thin_i || thin_o) && !(thin_i && thin_o)) ? 1 : 0;
}
extern "C" int acg_pack_conv_weights_multi(const acg_pack_item *items, int n, void *stream)
{
    ACG_REQUIRE(items != nullptr && n >= 1 && use_bf16(), "acg_pack_conv_weights_multi: bf16 / bf16x3 arithmetic only (see acg_pack_conv_weights_multi_supported)");
    for (int base = 0; base < n; base += PACK_MAX_ITEMS) {
        PackTable T;
        T.n = n - base < PACK_MAX_ITEMS ? n - base : PACK_MAX_ITEMS;
        T.split = (int)(g_acg_precision == ACG_PREC_BF16X3);
        T.first[0] = 0;
        for (int i = 0; i < T.n; ++i) {
            const acg_pack_item &q = items[base + i];
            ACG_REQUIRE(q.w != nullptr && q.wf != nullptr && q.wb != nullptr && q.Ci % 16 == 0 && q.Co % 16 == 0 && q.Or <= q.Co && q.Ir <= q.Ci && q.K >= 1 && q.K <= 7 &&
                        acg_pack_conv_weights_multi_supported(q.Or, q.Ir, q.K),
                        "acg_pack_conv_weights_multi: item %d: bad dims or a thin layer (Or=%d Ir=%d K=%d Ci=%d Co=%d)", base + i, q.Or, q.Ir, q.K, q.Ci, q.Co);
            T.it[i] = q;
            T.cop[i] = (short)acg_ncols_pad(q.Co); T.cip[i] = (short)acg_ncols_pad(q.Ci);
            const long long ne = (long long)acg_packed_wf_elems(q.K, q.Ci, q.Co) + (long long)acg_packed_wb_elems(q.K, q.Ci, q.Co);
            const int nb = acg_cdiv(ne, 256 * 8) > 256 ? 256 : acg_cdiv(ne, 256 * 8);   // ~8 elements per thread
            T.first[i + 1] = T.first[i] + (nb < 1 ? 1 : nb);
        }
        hipLaunchKernelGGL(pack_weight_bf16_multi_kernel, dim3(T.first[T.n]), dim3(256), 0, (hipStream_t)stream, T);
        ACG_CHECK_LAUNCH("pack_weight_bf16_multi_kernel");
    }
    return ACG_OK;
}

__global__ void pad_vector_kernel(const float *__restrict__ s, int n, float *__restrict__ d, int np)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) d[i] = i < n ? s[i] : 0.f;
}
extern "C" int acg_pad_vector(const float *src, int n, float *dst, int np, void *stream)
{
    hipLaunchKernelGGL(pad_vector_kernel, dim3(acg_cdiv(np, 256)), dim3(256), 0, (hipStream_t)stream, src, n, dst, np);
    ACG_CHECK_LAUNCH("pad_vector_kernel");
    return ACG_OK;
}

// ------------------------------------------------------------------------------------------
// reflection-pad adjoint: dxp[N][H+2p][W+2p][C] -> dx[N][H][W][C], folding mirrored borders
// (torch reflection_pad2d_backward).  float4 over channels.
// ------------------------------------------------------------------------------------------
__global__ void reflect_fold_kernel(const float *__restrict__ dxp, float *__restrict__ dx, int N, int H, int W, int C,
                                    int p)
{
    const int C4 = C / 4;
    const long long total = (long long)N * H * W * C4;
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c4 = (int)(r % C4); r /= C4;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); r /= H;
        const int n = (int)r;
        // padded rows that read input row y: y+p, plus mirrors p-y (1<=y<=p) and 2(H-1)-y+p (H-1-p<=y<=H-2)
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + p;
        if (y >= 1 && y <= p) ys[ny++] = p - y;
        if (y >= H - 1 - p && y <= H - 2) ys[ny++] = 2 * (H - 1) - y + p;
        xs[nx++] = x + p;
        if (x >= 1 && x <= p) xs[nx++] = p - x;
        if (x >= W - 1 - p && x <= W - 2) xs[nx++] = 2 * (W - 1) - x + p;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b)
                acc += *(const f32x4 *)(dxp + (((long long)n * Hp + ys[a]) * Wp + xs[b]) * C + c4 * 4);
        *(f32x4 *)(dx + i * 4) = acc;
    }
}

// The same fold restricted to the FRAME: the input pixels the pad ring mirrors onto (rows / columns 1..p and
// H-1-p..H-2).  Used when the data-gradient kernel already stored every other pixel straight into dx (Geom.fold_p):
// 2p rows x W plus 2p columns x (H - 2p) pixels per image instead of all H x W.
__global__ void reflect_fold_frame_kernel(const float *__restrict__ dxp, float *__restrict__ dx, int N, int H, int W, int C,
                                          int p, const float *__restrict__ addend, const float *__restrict__ relu_src,
                                          const unsigned *__restrict__ addend_mask, int out_s16, int relu_s16)
{
    const int C4 = C / 4;
    const int nrow = 2 * p * W, ncol = 2 * p * (H - 2 * p); // frame pixels per image: dirty rows, then dirty columns
    const long long total = (long long)N * (nrow + ncol) * C4;
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c4 = (int)(r % C4); r /= C4;
        const int f = (int)(r % (nrow + ncol));
        const int n = (int)(r / (nrow + ncol));
        int y, x;
        if (f < nrow) { // dirty row k: rows 1..p then H-1-p..H-2
            const int k = f / W;
            x = f - k * W;
            y = k < p ? 1 + k : H - 1 - p + (k - p);
        } else {        // dirty column k of a clean row
            const int q = f - nrow, k = q / (H - 2 * p), yy = q - k * (H - 2 * p);
            x = k < p ? 1 + k : W - 1 - p + (k - p);
            y = yy == 0 ? 0 : (yy <= H - 2 - 2 * p ? p + yy : H - 1); // clean rows: 0, p+1..H-2-p, H-1
        }
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + p;
        if (y >= 1 && y <= p) ys[ny++] = p - y;
        if (y >= H - 1 - p && y <= H - 2) ys[ny++] = 2 * (H - 1) - y + p;
        xs[nx++] = x + p;
        if (x >= 1 && x <= p) xs[nx++] = p - x;
        if (x >= W - 1 - p && x <= W - 2) xs[nx++] = 2 * (W - 1) - x + p;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b)
                acc += *(const f32x4 *)(dxp + (((long long)n * Hp + ys[a]) * Wp + xs[b]) * C + c4 * 4);
        const long long o = (((long long)n * H + y) * W + x) * C + c4 * 4;
        // pre-split (S16) tensors: the 8-channel group of element o starts at byte 4 * (o & ~7); hi halves at +0, lo at +16
        const long long sb = 4 * (o & ~7LL) + 2 * (o & 7);
        if (relu_src != nullptr) { // same order as the convolution epilogue: mask, then addend
            if (relu_s16) {
                const uint2 sv = *(const uint2 *)((const char *)relu_src + sb);
                const unsigned h[4] = {sv.x & 0xffffu, sv.x >> 16, sv.y & 0xffffu, sv.y >> 16};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = (h[q] - 1u) < 0x7fffu ? acc[q] : 0.f;
            } else {
                const f32x4 mv = *(const f32x4 *)(relu_src + o);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mv[q] > 0.f ? acc[q] : 0.f;
            }
        }
        if (addend != nullptr) {
            f32x4 av = *(const f32x4 *)(addend + o);
            if (addend_mask != nullptr) {
                const long long f = o >> 2;
                const unsigned nb = (addend_mask[f >> 3] >> (4 * (int)(f & 7))) & 15u;
#pragma unroll
                for (int q = 0; q < 4; ++q) av[q] = (nb >> q) & 1u ? av[q] : 0.f;
            }
            acc += av;
        }
        if (out_s16) {
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            uint2 hi, lo;
            unsigned *hp = &hi.x, *lp = &lo.x;
#pragma unroll
            for (int q = 0; q < 2; ++q) { // the arithmetic of acg_split8
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[2 * q], acc[2 * q + 1]}, bf16x2_t));
                const float ha = __builtin_bit_cast(float, h << 16), hb = __builtin_bit_cast(float, h & 0xffff0000u);
                hp[q] = h;
                lp[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[2 * q] - ha, acc[2 * q + 1] - hb}, bf16x2_t));
            }
            *(uint2 *)((char *)dx + sb) = hi;
            *(uint2 *)((char *)dx + sb + 16) = lo;
        } else {
            *(f32x4 *)(dx + o) = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------
// column sums (bias gradient): dy[M][C] -> db[c] (first Cr columns), two deterministic stages
// ------------------------------------------------------------------------------------------
#define COLSUM_ROWS 2048
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ dy, long long M, int C,
                                                             float *__restrict__ part)
{
    __shared__ float red[256 * 4];
    const int C4 = C / 4;             // <= 256
    const int lanes_per_row = C4;     // threads covering one row
    const int rows_par = 256 / lanes_per_row;
    const int c4 = threadIdx.x % lanes_per_row, rl = threadIdx.x / lanes_per_row;
    const long long r0 = (long long)blockIdx.x * COLSUM_ROWS;
    long long r1 = r0 + COLSUM_ROWS;
    if (r1 > M) r1 = M;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (rl < rows_par)
        for (long long r = r0 + rl; r < r1; r += rows_par) acc += *(const f32x4 *)(dy + r * C + c4 * 4);
    *(f32x4 *)&red[threadIdx.x * 4] = acc;
    __syncthreads();
    if (threadIdx.x < lanes_per_row) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < rows_par; ++k) s += *(const f32x4 *)&red[(k * lanes_per_row + threadIdx.x) * 4];
        *(f32x4 *)(part + (long long)blockIdx.x * C + threadIdx.x * 4) = s;
    }
}
// one block per 16 channels: 16 partial-block lanes per channel, fixed-order LDS tree (deterministic)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ part, int nblk, int C, int Cr,
                                                           float *__restrict__ db, int accumulate)
{
    __shared__ float red[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15), k = threadIdx.x >> 4;
    float s = 0.f;
    if (c < Cr)
        for (int b = k; b < nblk; b += 16) s += part[(long long)b * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st >= 16; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 16 && c < Cr) db[c] = (accumulate ? db[c] : 0.f) + red[threadIdx.x];
}
static size_t colsum_ws_bytes(long long M, int C) { return (size_t)acg_cdiv(M, COLSUM_ROWS) * C * sizeof(float); }
static int colsum_launch(const float *dy, long long M, int C, int Cr, float *db, float *ws, hipStream_t st, int accumulate)
{
    ACG_REQUIRE(C % 4 == 0 && C / 4 <= 256, "colsum: C=%d unsupported", C);
    const int nblk = acg_cdiv(M, COLSUM_ROWS);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, dy, M, C, ws);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(acg_cdiv(Cr, 16)), dim3(256), 0, st, ws, nblk, C, Cr, db, accumulate);
    ACG_CHECK_LAUNCH("colsum");
    return ACG_OK;
}

// ------------------------------------------------------------------------------------------
// split-K reduction of weight-gradient partials -> torch OIHW (real Or x Ir)
// part[nsplit][KK][CiP][CoP]
// ------------------------------------------------------------------------------------------
// thin: part[nsplit][1][CiP][CoP] with row = tap*4 + ci
// accumulate != 0: dw (and db) are ADDED to — the caller passes the parameter's .grad itself, so no separate accumulation
// kernel runs per parameter (torch's AccumulateGrad launched 564 five-microsecond adds per training step).
// The bias reduction rides in the same launch: blocks [wblocks, wblocks + ceil(Cr/16)) reduce bias_part[nsplit][Cp] -> db
// (16 channels per block, 16 split-lanes each, fixed-order tree: deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int nsplit, int KK, int CiP, int CoP, int Or,
                                    int Ir, float *__restrict__ dw, int thin, int accumulate, int wblocks,
                                    const float *__restrict__ bias_part, int Cp, int Cr, float *__restrict__ db, int bias_slots,
                                    int el_log2)
{
    if ((int)blockIdx.x >= wblocks) {
        __shared__ float red[256];
        const int c = ((int)blockIdx.x - wblocks) * 16 + (threadIdx.x & 15), k0 = threadIdx.x >> 4;
        float s = 0.f;
        if (c < Cr)
            for (int k = k0; k < bias_slots; k += 16) s += bias_part[(long long)k * Cp + c];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int st = 128; st >= 16; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x < 16 && c < Cr) db[c] = (accumulate ? db[c] : 0.f) + red[threadIdx.x];
        return;
    }
    // Weight part: a block is EL elements x (256 / EL) split-lanes; lane group kl sums the slabs k = kl, kl + KL, ... of its
    // element (coalesced over the EL elements), the groups fold through LDS in fixed order: deterministic.  An element is four
    // adjacent output channels where the layout allows (16-byte loads; the one-float-per-thread sequential version read a
    // 50 MB trunk slab set at 3.8 TB/s, 128 launches = 1.7 ms per step), else one.  `el_log2` = 6 (64 elements x 4 lanes), or 4
    // (16 x 16) when there are few elements and many slabs (the persistent thin-patch kernel leaves 768 of them).
    __shared__ f32x4 red4[256];
    const int EL = 1 << el_log2, KL = 256 >> el_log2;
    const int el = threadIdx.x & (EL - 1), kl = threadIdx.x >> el_log2;
    const bool quad = thin != 2 && (Or & 3) == 0 && (CoP & 3) == 0;
    const int On = quad ? Or >> 2 : Or;
    const long long total = (long long)KK * Ir * On;
    const long long i = (long long)blockIdx.x * EL + el;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    int o = 0, ci = 0, tap = 0;
    if (i < total) {
        long long r = i;
        o = (int)(r % On); r /= On;
        ci = (int)(r % Ir); r /= Ir;
        tap = (int)r;
        const long long stride = thin ? (long long)CiP * CoP : (long long)KK * CiP * CoP;
        const int oo = quad ? o * 4 : o;
        const float *p = thin == 2 ? part + ((long long)(tap * 4 + oo)) * CoP + ci   // rows (tap, co), columns ci
                       : thin == 1 ? part + ((long long)(tap * 4 + ci)) * CoP + oo  // rows (tap, ci), columns co
                                   : part + ((long long)tap * CiP + ci) * CoP + oo;
        if (quad) {
#pragma unroll 4
            for (int k = kl; k < nsplit; k += KL) sum += *(const f32x4 *)(p + k * stride);
        } else {
#pragma unroll 4
            for (int k = kl; k < nsplit; k += KL) sum[0] += p[k * stride];
        }
    }
    red4[threadIdx.x] = sum;
    __syncthreads();
    if (kl != 0 || i >= total) return;
    for (int k = 1; k < KL; ++k) sum += red4[k * EL + el];
    const int ne = quad ? 4 : 1;
    for (int e = 0; e < ne; ++e) {
        float *dst = dw + ((long long)((quad ? o * 4 : o) + e) * Ir + ci) * KK + tap;
        *dst = (accumulate ? *dst : 0.f) + sum[e];
    }
}

// ------------------------------------------------------------------------------------------
// naive direct kernels (cross-check path, ACG_IMPL_DIRECT): one thread per output element,
// geometry taken straight from the descriptor (independent of the tap-list machinery).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int c16d(int c) { return (c + 15) / 16 * 16; }   // packed-weight channel count of a stored width
__device__ __forceinline__ int reflect_idx(int i, int n)
{
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__global__ void direct_fwd_kernel(acg_conv_desc d, const float *__restrict__ x, const float *__restrict__ wf,
                                  const float *__restrict__ bias, float *__restrict__ y, int act, int CoP)
{
    const long long total = (long long)d.N * d.Ho * d.Wo * d.Co;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    long long r = i;
    const int co = (int)(r % d.Co); r /= d.Co;
    const int ox = (int)(r % d.Wo); r /= d.Wo;
    const int oy = (int)(r % d.Ho); r /= d.Ho;
    const int n = (int)r;
    float acc = bias ? bias[co] : 0.f;
    for (int kh = 0; kh < d.K; ++kh)
        for (int kw = 0; kw < d.K; ++kw) {
            int iy = oy * d.stride + kh - d.pad, ix = ox * d.stride + kw - d.pad;
            if (d.pad_mode == ACG_PAD_REFLECT) {
                iy = reflect_idx(iy, d.Hi);
                ix = reflect_idx(ix, d.Wi);
            } else if (iy < 0 || iy >= d.Hi || ix < 0 || ix >= d.Wi)
                continue;
            const float *xp = x + (((long long)n * d.Hi + iy) * d.Wi + ix) * d.Ci;
            const int tap = kh * d.K + kw;
            for (int ci = 0; ci < d.Ci; ++ci)
                acc += xp[ci] * wf[(((long long)tap * (c16d(d.Ci) / 8) + ci / 8) * CoP + co) * 8 + (ci & 7)];
        }
    y[i] = acg_apply_act(acc, act);
}

// dx[n,iy,ix,ci] = sum over padded preimages (py,px), taps, co.  Also used (with bias/act) as the
// ConvTranspose2d forward.
__global__ void direct_dgrad_kernel(acg_conv_desc d, const float *__restrict__ dy, const float *__restrict__ wb,
                                    const float *__restrict__ bias, float *__restrict__ dx, int act, int CiP)
{
    const long long total = (long long)d.N * d.Hi * d.Wi * d.Ci;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    long long r = i;
    const int ci = (int)(r % d.Ci); r /= d.Ci;
    const int ix = (int)(r % d.Wi); r /= d.Wi;
    const int iy = (int)(r % d.Hi); r /= d.Hi;
    const int n = (int)r;
    const int p = d.pad;
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = iy + p;
    xs[nx++] = ix + p;
    if (d.pad_mode == ACG_PAD_REFLECT) {
        if (iy >= 1 && iy <= p) ys[ny++] = p - iy;
        if (iy >= d.Hi - 1 - p && iy <= d.Hi - 2) ys[ny++] = 2 * (d.Hi - 1) - iy + p;
        if (ix >= 1 && ix <= p) xs[nx++] = p - ix;
        if (ix >= d.Wi - 1 - p && ix <= d.Wi - 2) xs[nx++] = 2 * (d.Wi - 1) - ix + p;
    }
    float acc = bias ? bias[ci] : 0.f;
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b)
            for (int kh = 0; kh < d.K; ++kh)
                for (int kw = 0; kw < d.K; ++kw) {
                    const int ty = ys[a] - kh, tx = xs[b] - kw;
                    if (ty < 0 || tx < 0 || ty % d.stride || tx % d.stride) continue;
                    const int oy = ty / d.stride, ox = tx / d.stride;
                    if (oy >= d.Ho || ox >= d.Wo) continue;
                    const float *gp = dy + (((long long)n * d.Ho + oy) * d.Wo + ox) * d.Co;
                    const int tap = kh * d.K + kw;
                    for (int co = 0; co < d.Co; ++co)
                        acc += gp[co] * wb[(((long long)tap * (c16d(d.Co) / 8) + co / 8) * CiP + ci) * 8 + (co & 7)];
                }
    dx[i] = acg_apply_act(acc, act);
}

// dw[o][i][kh][kw] (real Or x Ir), one thread per weight, serial over all pixels (tests only)
__global__ void direct_wgrad_kernel(acg_conv_desc d, const float *__restrict__ x, const float *__restrict__ dy,
                                    float *__restrict__ dw, int Or, int Ir, int accumulate)
{
    const int KK = d.K * d.K;
    const long long total = (long long)Or * Ir * KK;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= total) return;
    long long r = i;
    const int tap = (int)(r % KK); r /= KK;
    const int ci = (int)(r % Ir); r /= Ir;
    const int co = (int)r;
    const int kh = tap / d.K, kw = tap % d.K;
    float acc = 0.f;
    for (int n = 0; n < d.N; ++n)
        for (int oy = 0; oy < d.Ho; ++oy)
            for (int ox = 0; ox < d.Wo; ++ox) {
                int iy = oy * d.stride + kh - d.pad, ix = ox * d.stride + kw - d.pad;
                if (d.pad_mode == ACG_PAD_REFLECT) {
                    iy = reflect_idx(iy, d.Hi);
                    ix = reflect_idx(ix, d.Wi);
                } else if (iy < 0 || iy >= d.Hi || ix < 0 || ix >= d.Wi)
                    continue;
                acc += x[(((long long)n * d.Hi + iy) * d.Wi + ix) * d.Ci + ci] *
                       dy[(((long long)n * d.Ho + oy) * d.Wo + ox) * d.Co + co];
            }
    dw[i] = (accumulate ? dw[i] : 0.f) + acc;
}

// ------------------------------------------------------------------------------------------
// descriptor checks + tap-list builders
// ------------------------------------------------------------------------------------------
static int check_desc(const acg_conv_desc *d, const char *who)
{
    ACG_REQUIRE(d != nullptr, "%s: null descriptor", who);
    ACG_REQUIRE(d->N > 0 && d->Hi > 0 && d->Wi > 0 && d->Ho > 0 && d->Wo > 0, "%s: empty tensor", who);
    // stored channel counts: multiples of 16, or 4 for a tensor of <= 4 real channels on the thin side of a thin layer (K > 1,
    // the other side wider: what thin_in / thin_out select) — every other kernel gathers 16-channel chunks
    ACG_REQUIRE(d->Ci > 0 && d->Co > 0 && (d->Ci % 16 == 0 || d->Ci == 4) && (d->Co % 16 == 0 || d->Co == 4),
                "%s: channels must be padded to 16, or to 4 for image tensors (Ci=%d Co=%d)", who, d->Ci, d->Co);
    ACG_REQUIRE(d->Ci != 4 || (d->Cir >= 1 && d->Cir <= 4 && d->K > 1 && !(d->Cor >= 1 && d->Cor <= 4)),
                "%s: a 4-channel input needs a thin-input layer (Cir=%d Cor=%d K=%d)", who, d->Cir, d->Cor, d->K);
    ACG_REQUIRE(d->Co != 4 || (d->Cor >= 1 && d->Cor <= 4 && d->K > 1 && !(d->Cir >= 1 && d->Cir <= 4)),
                "%s: a 4-channel output needs a thin-output layer (Cir=%d Cor=%d K=%d)", who, d->Cir, d->Cor, d->K);
    ACG_REQUIRE(d->K >= 1 && d->K <= 7 && (d->stride == 1 || d->stride == 2), "%s: K=%d stride=%d unsupported", who,
                d->K, d->stride);
    ACG_REQUIRE(d->pad >= 0 && d->pad < d->K, "%s: pad=%d", who, d->pad);
    ACG_REQUIRE(d->Ho == (d->Hi + 2 * d->pad - d->K) / d->stride + 1 && d->Wo == (d->Wi + 2 * d->pad - d->K) / d->stride + 1,
                "%s: output size %dx%d inconsistent with input %dx%d K=%d s=%d p=%d", who, d->Ho, d->Wo, d->Hi, d->Wi,
                d->K, d->stride, d->pad);
    if (d->pad_mode == ACG_PAD_REFLECT)
        ACG_REQUIRE(d->stride == 1 && d->pad < d->Hi && d->pad < d->Wi, "%s: reflect pad needs stride 1 and pad < size", who);
    return ACG_OK;
}

static void fwd_geom(const acg_conv_desc *d, Geom *g, Taps *t, int act)
{
    g->Hin = d->Hi; g->Win = d->Wi; g->Cin = d->Ci;
    g->Hout = d->Ho; g->Wout = d->Wo; g->Cout = d->Co;
    g->GH = d->Ho; g->GW = d->Wo; g->os = 1; g->oy0 = 0; g->ox0 = 0; g->is = d->stride;
    g->reflect = d->pad_mode == ACG_PAD_REFLECT; g->act = act; g->ncols_pad = acg_ncols_pad(d->Co);
    g->Mtot = (long long)d->N * d->Ho * d->Wo;
    g->thin = thin_in(d) ? 1 : 0;
    g->w_elems = (long long)wf_regular_elems(d->K, d->Ci, d->Co);
    t->n = 0;
    for (int kh = 0; kh < d->K; ++kh)
        for (int kw = 0; kw < d->K; ++kw) {
            t->dy[t->n] = (short)(kh - d->pad); t->dx[t->n] = (short)(kw - d->pad); t->w[t->n] = (short)(kh * d->K + kw);
            t->n++;
        }
}

// Column part of the reflect adjoint for the un-padded data gradient (Geom.unpad): pad column -1 mirrors onto column 1, pad
// column W onto column W-2, i.e. dx[y][1] += sum_kh dy[y + 1 - kh][0] . w[kh][0] and dx[y][W-2] += sum_kh dy[y + 1 - kh][W-1] .
// w[kh][2] (rows outside the map are zero; rows 1 and H-2 also receive the corner terms dy[0] . w[0][.] / dy[H-1] . w[2][.]
// their own mirrored rows carry).  A (N H 2) x (3 C) x C GEMM, 0.4 % of the layer: one 32x32x16 MFMA tile per wave, both
// operands read straight into fragment layout — a pre-split pixel's 8-channel group IS an A fragment, 8 consecutive output
// channels of a packed-wb row ARE a B fragment — same bf16x3 products as the main kernel.
// grid (N * H / 32, 2, CiP / 128) x 256 threads: 32 rows of one image, one side, wave w = dx channels 32w .. 32w+31 of 128.
typedef __bf16 cf_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void dgrad_colfix_kernel(const char *__restrict__ dy, const __bf16 *__restrict__ wb,
                                                           long long w_lo_elems, float *__restrict__ colfix, int H, int W,
                                                           int C, int CiP, int Cdx)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles = H / 32, n = blockIdx.x / tiles, qy0 = (blockIdx.x - n * tiles) * 32, side = blockIdx.y;
    const int lr = lane & 31, kg = lane >> 5;
    const int qy = qy0 + lr, ci = blockIdx.z * 128 + wave * 32 + lr;
    const int col = side ? W - 1 : 0, kw = side ? 2 : 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const cf_bf16x8 zero = {};
    // kernel rows 0..2, then the corner term of the tile that holds row 1 (kh 0, dy row 0) or row H-2 (kh 2, dy row H-1).
    // A step is 128 channels = eight 16-channel chunks whose 32 fragment loads are issued together, the next step's before this
    // step's MFMAs (one wave per SIMD: registers are free, memory latency is the whole cost of this kernel).
    const int extra = qy0 == 0 ? 0 : (qy0 + 32 == H ? 2 : -1);
    const int nsteps = (3 + (extra >= 0 ? 1 : 0)) * (C / 128);
    cf_bf16x8 ah[2][8], al[2][8], bh[2][8], bl[2][8];
    auto load = [&](int s, int b) {
        const int step = s / (C / 128), c128 = s - step * (C / 128);
        const int kh = step < 3 ? step : extra;
        int ry;
        bool ok;
        if (step < 3) { ry = qy + 1 - kh; ok = (unsigned)ry < (unsigned)H; }
        else { ry = extra == 0 ? 0 : H - 1; ok = qy == (extra == 0 ? 1 : H - 2); }
        const char *ap = dy + (((long long)n * H + (ok ? ry : 0)) * W + col) * C * 4 + c128 * 512 + kg * 32;
        const __bf16 *bp = wb + (((long long)(kh * 3 + kw) * (C / 16) + c128 * 8) * CiP + ci) * 16 + kg * 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            ah[b][u] = zero; al[b][u] = zero;
            if (ok) { ah[b][u] = *(const cf_bf16x8 *)(ap + u * 64); al[b][u] = *(const cf_bf16x8 *)(ap + u * 64 + 16); }
            bh[b][u] = *(const cf_bf16x8 *)(bp + (long long)u * CiP * 16);
            bl[b][u] = *(const cf_bf16x8 *)(bp + w_lo_elems + (long long)u * CiP * 16);
        }
    };
    auto mma = [&](int b) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[b][u], bh[b][u], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[b][u], bl[b][u], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[b][u], bh[b][u], acc, 0, 0, 0);
        }
    };
    load(0, 0);
    for (int s = 0; s < nsteps; s += 2) {   // two steps per trip: the buffer index stays a compile-time constant
        if (s + 1 < nsteps) load(s + 1, 1);
        mma(0);
        if (s + 2 < nsteps) load(s + 2, 0);
        if (s + 1 < nsteps) mma(1);
    }
    if (ci < Cdx) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
            colfix[(((long long)n * H + qy0 + row) * 2 + side) * Cdx + ci] = acc[r];
        }
    }
}

// data gradient (and ConvTranspose forward): gathers from the conv-OUTPUT side tensor `src`
// (N,Ho,Wo,Co) with packed wb, writes the conv-INPUT side tensor `dst` (N,Hi,Wi,Ci).
// addend (optional): tensor of dst's shape added to the result; only the frame path below implements it
static bool dgrad_frame_ok(const acg_conv_desc *d, const Geom &g)
{
    const int p = d->pad;
    return d->stride == 1 && d->pad_mode == ACG_PAD_REFLECT && p > 0 && !thin_in_valu_dgrad(d) && acg_igemm_uses_ws(g) &&
           d->Hi > 4 * p + 1 && d->Wi > 4 * p + 1;
}

// the un-padded grid of the pre-split reflect data gradient (Geom.unpad): 3x3, pad 1, rows that are whole 128-pixel tiles
static bool dgrad_unpad_ok(const acg_conv_desc *d)
{
    static const bool no_unpad = acg_debug_switch("ACG_NO_UNPAD");   // A/B switch
    return !no_unpad && d->stride == 1 && d->pad_mode == ACG_PAD_REFLECT && d->K == 3 && d->pad == 1 && d->Wi % 128 == 0 &&
           d->Hi % 32 == 0 && d->Hi >= 64 && d->Co % 128 == 0;
}

static int dgrad_igemm(const acg_conv_desc *d, const float *src, const float *wb, const float *bias, float *dst,
                       int act, void *ws, size_t ws_bytes, hipStream_t st, const float *addend = nullptr,
                       const float *relu_src = nullptr, const unsigned *addend_mask = nullptr, int in_s16 = 0, int out_s16 = 0,
                       int relu_s16 = 0, float *stats = nullptr, const acg_norm_sums *ns = nullptr,
                       const unsigned *relu_mask = nullptr)
{
    Geom g; Taps t;
    g.Hin = d->Ho; g.Win = d->Wo; g.Cin = d->Co;
    g.Cout = d->Ci; g.reflect = 0; g.act = act; g.ncols_pad = acg_ncols_pad(d->Ci); g.is = 1;
    g.thin = (d->stride == 1 && thin_out(d)) ? 1 : 0;
    g.w_elems = (long long)wb_regular_elems(d->K, d->Ci, d->Co);
    const int p = d->pad, K = d->K;
    if (d->stride == 1) {
        const bool refl = d->pad_mode == ACG_PAD_REFLECT && p > 0;
        const int e = refl ? p : 0; // compute on the padded grid, then fold
        float *out = dst;
        if (refl) {
            const size_t need = (size_t)d->N * (d->Hi + 2 * p) * (d->Wi + 2 * p) * d->Ci * sizeof(float);
            if (ws == nullptr || ws_bytes < need) {
                acg_set_error("acg_conv2d_bwd_data: workspace %zu < %zu", ws_bytes, need);
                return ACG_ERR_WORKSPACE;
            }
            ACG_REQUIRE(bias == nullptr && act == ACG_ACT_NONE, "dgrad: reflect with epilogue unsupported");
            out = (float *)ws;
        }
        g.Hout = d->Hi + 2 * e; g.Wout = d->Wi + 2 * e; g.GH = g.Hout; g.GW = g.Wout;
        g.os = 1; g.oy0 = 0; g.ox0 = 0;
        g.Mtot = (long long)d->N * g.GH * g.GW;
        // Pre-split operands, 3x3, pad 1, rows that are whole tiles: the un-padded grid (Geom.unpad) — 3 % fewer tiles than the
        // padded grid, one row segment per tile, and the fold pass over the frame goes away
        if (refl && in_s16 && dgrad_unpad_ok(d) && dgrad_frame_ok(d, g)) {
            g.Hout = d->Hi; g.Wout = d->Wi; g.GH = d->Hi; g.GW = d->Wi;
            g.Mtot = (long long)d->N * g.GH * g.GW;
            t.n = 0;
            for (int kh = 0; kh < K; ++kh)
                for (int kw = 0; kw < K; ++kw) { t.dy[t.n] = (short)(p - kh); t.dx[t.n] = (short)(p - kw); t.w[t.n] = (short)(kh * K + kw); t.n++; }
            ACG_REQUIRE(acg_igemm_x3_pre_ok(g, t) && (relu_s16 == 0 || out_s16) && (out_s16 == 0 || addend == nullptr) &&
                        (relu_src == nullptr || relu_s16 == out_s16) && d->Co % 128 == 0,
                        "dgrad: unsupported pre-split combination (query acg_conv2d_s16_supported)");
            const int CiP = acg_ncols_pad(d->Ci);
            hipLaunchKernelGGL(dgrad_colfix_kernel, dim3(d->N * (d->Hi / 32), 2, CiP / 128), dim3(256), 0, st, (const char *)src,
                               (const __bf16 *)wb, g.w_elems, (float *)ws, d->Hi, d->Wi, d->Co, CiP, d->Ci);
            ACG_CHECK_LAUNCH("dgrad_colfix_kernel");
            g.unpad = 1; g.colfix = (const float *)ws; g.out2 = dst; g.addend = addend; g.relu_src = relu_src; g.addend_mask = addend_mask;
            g.out_s16 = out_s16; g.relu_s16 = relu_s16; g.relu_mask = relu_mask;
            if (ns != nullptr) {
                g.ns_x = ns->x; g.ns_mean = ns->mean; g.ns_rstd = ns->rstd; g.ns_gamma = ns->gamma; g.ns_beta = ns->beta;
                g.ns_gstride = ns->gstride; g.ns_mask = ns->sign_mask; g.ns_act = ns->act; g.ns_part = ns->part;
                ACG_REQUIRE(ns->part != nullptr && (ns->gstride == 0 || (ns->gstride >= d->Ci && ns->gstride % 4 == 0)), "dgrad: bad acg_norm_sums");
            }
            return acg_igemm_x3_pre_launch(src, wb, bias, dst, g, t, g.w_elems, st);
        }
        ACG_REQUIRE((ns == nullptr || (!refl && !in_s16 && !out_s16 && addend == nullptr && relu_src == nullptr)) && relu_mask == nullptr,
                    "dgrad: norm sums / a sign bitmask as the ReLU source need the un-padded pre-split path or the row pipeline (query acg_conv2d_bwd_data_s16_sums_supported / acg_conv2d_bwd_data_sums_supported)");
        t.n = 0;
        // zero pad: dy row = iy + p - kh ; reflect (padded grid): dy row = py - kh
        const int base = refl ? 0 : p;
        for (int kh = 0; kh < K; ++kh)
            for (int kw = 0; kw < K; ++kw) {
                t.dy[t.n] = (short)(base - kh); t.dx[t.n] = (short)(base - kw); t.w[t.n] = (short)(kh * K + kw);
                t.n++;
            }
        // the wave-specialised kernel stores the pixels nothing is mirrored onto straight into dst: only the frame is folded
        const bool frame = refl && dgrad_frame_ok(d, g);
        ACG_REQUIRE((addend == nullptr && relu_src == nullptr) || frame,
                    "dgrad: the fused addend / ReLU mask need the frame path (query acg_conv2d_bwd_data_add_supported)");
        if (frame) { g.fold_p = p; g.fold_H = d->Hi; g.fold_W = d->Wi; g.out2 = dst; g.addend = addend; g.relu_src = relu_src; g.addend_mask = addend_mask; }
        int rc;
        if (in_s16 || out_s16 || relu_s16) { // pre-split operands: the frame path of the pre-split kernel only
            ACG_REQUIRE(in_s16 && frame && acg_igemm_x3_pre_ok(g, t) && (relu_s16 == 0 || out_s16) && (out_s16 == 0 || addend == nullptr) &&
                        (relu_src == nullptr || relu_s16 == out_s16),
                        "dgrad: unsupported pre-split combination (query acg_conv2d_s16_supported)");
            g.out_s16 = out_s16; g.relu_s16 = relu_s16;
            rc = acg_igemm_x3_pre_launch(src, wb, bias, out, g, t, g.w_elems, st);
        } else {
            if (ns != nullptr) {   // fp32 operands: only the persistent row pipeline (conv_rows.hip) emits the norm-backward sums
                g.ns_x = ns->x; g.ns_mean = ns->mean; g.ns_rstd = ns->rstd; g.ns_gamma = ns->gamma; g.ns_beta = ns->beta;
                g.ns_gstride = ns->gstride; g.ns_mask = ns->sign_mask; g.ns_act = ns->act; g.ns_part = ns->part;
                // (which kernel takes them is checked where the launch is dispatched: conv_bf16.hip / conv_igemm.hip refuse a
                // geometry that would land on a kernel without the sums epilogue)
                ACG_REQUIRE(ns->part != nullptr && !thin_in_valu_dgrad(d) && !frame && acg_conv2d_bwd_data_sums_supported(d),
                            "dgrad: norm sums on fp32 operands: unsupported geometry (query acg_conv2d_bwd_data_sums_supported)");
            }
            rc = thin_in_valu_dgrad(d) ? thin_out_launch(src, wb, bias, out, g, t, st) : acg_igemm_launch(src, wb, bias, out, g, t, st);
        }
        if (rc != ACG_OK) return rc;
        if (frame) {
            const long long total = (long long)d->N * (2 * p * d->Wi + 2 * p * (d->Hi - 2 * p)) * (d->Ci / 4);
            const int blocks = acg_cdiv(total, 256) > 4096 ? 4096 : acg_cdiv(total, 256);
            hipLaunchKernelGGL(reflect_fold_frame_kernel, dim3(blocks), dim3(256), 0, st, (const float *)ws, dst, d->N, d->Hi,
                               d->Wi, d->Ci, p, addend, relu_src, addend_mask, out_s16, relu_s16);
            ACG_CHECK_LAUNCH("reflect_fold_frame_kernel");
        } else if (refl) {
            const long long total = (long long)d->N * d->Hi * d->Wi * (d->Ci / 4);
            const int blocks = acg_cdiv(total, 256) > 4096 ? 4096 : acg_cdiv(total, 256);
            hipLaunchKernelGGL(reflect_fold_kernel, dim3(blocks), dim3(256), 0, st, (const float *)ws, dst, d->N, d->Hi,
                               d->Wi, d->Ci, p);
            ACG_CHECK_LAUNCH("reflect_fold_kernel");
        }
        return ACG_OK;
    }
    ACG_REQUIRE(!thin_out(d), "dgrad: stride 2 with <= 4 output channels is not supported by the thin packing");
    // stride 2: four sub-pixel phases, each a dense small-tap convolution (no zero insertion)
    ACG_REQUIRE(d->pad_mode == ACG_PAD_ZERO, "dgrad: stride 2 needs zero padding");
    ACG_REQUIRE(addend == nullptr && relu_src == nullptr && relu_mask == nullptr && !in_s16 && !out_s16, "dgrad: stride 2 takes no fused side inputs");
    ACG_REQUIRE(ns == nullptr || acg_conv2d_bwd_data_sums_supported(d), "dgrad: norm sums on this stride-2 geometry (query acg_conv2d_bwd_data_sums_supported)");
    g.Hout = d->Hi; g.Wout = d->Wi; g.os = 2;
    auto phase_taps = [&](int py, int px, Taps &tt, int base) {
        int n = 0;
        for (int kh = 0; kh < K; ++kh) {
            if ((py + p - kh) & 1) continue;
            for (int kw = 0; kw < K; ++kw) {
                if ((px + p - kw) & 1) continue;
                tt.dy[base + n] = (short)((py + p - kh) / 2); tt.dx[base + n] = (short)((px + p - kw) / 2);
                tt.w[base + n] = (short)(kh * K + kw);
                n++;
            }
        }
        return n;
    };
    // One launch for the four phases where the generic bf16 tile runs them anyway: even output sizes (equal phase grids of
    // whole 128-pixel tiles), at most 16 taps per phase.  The phases of a tile read the same rows of `src`: side by side on
    // one XCD they fetch them from HBM once (four launches: 2.0x the algorithmic traffic, profiles/r02_a_layer_traffic).
    static const bool no_phased = acg_debug_switch("ACG_NO_PHASED");   // A/B switch
    g.GH = d->Hi / 2; g.GW = d->Wi / 2; g.oy0 = 0; g.ox0 = 0;
    g.Mtot = (long long)d->N * g.GH * g.GW;
    if (!no_phased && d->Hi % 2 == 0 && d->Wi % 2 == 0 && g.Mtot % 128 == 0 && (K + 1) / 2 * ((K + 1) / 2) <= 16 &&
        g_acg_precision != ACG_PREC_F32 && g_acg_conv_impl == ACG_IMPL_MFMA && !g.thin && !thin_in_valu_dgrad(d) && !thin_out(d) &&
        !acg_igemm_uses_ws(g)) {
        t.n = 64;
        for (int i = 0; i < 64; ++i) { t.dy[i] = 0; t.dx[i] = 0; t.w[i] = 0; }
        g.nphase = 4;
        g.ph_ntaps = 0;
        int ntp[4];
        for (int ph = 0; ph < 4; ++ph) {
            const int nt = phase_taps(ph >> 1, ph & 1, t, 16 * ph);
            ACG_REQUIRE(nt > 0, "dgrad: empty phase (K=%d p=%d)", K, p);
            g.ph_ntaps |= nt << (8 * ph);
            ntp[ph] = nt;
        }
        if (stats != nullptr) {
            const int per = (int)(((long long)g.GH * g.GW) / 128);
            g.stats = stats; g.stats_cpi = 4 * per; g.stats_chunk0 = 0;
        }
        // 64 output channels in the bf16x3 arithmetic: all four phases in one tile, the input rows fetched once (conv_ph4.hip)
        Geom g4 = g;
        g4.nphase = 0; g4.ph_ntaps = 0;
        Taps plan;
        if (ns != nullptr) {   // the first backward pass of the norm in front of the stride-2 convolution rides on the four-phase tile
            g4.ns_x = ns->x; g4.ns_mean = ns->mean; g4.ns_rstd = ns->rstd; g4.ns_gamma = ns->gamma; g4.ns_beta = ns->beta;
            g4.ns_gstride = ns->gstride; g4.ns_mask = ns->sign_mask; g4.ns_act = ns->act; g4.ns_part = ns->part;
            ACG_REQUIRE(ns->part != nullptr && stats == nullptr && acg_igemm_ph4_ok(g4) && acg_ph4_plan(t, ntp, &plan),
                        "dgrad: norm sums on a stride-2 data gradient need the four-phase tile (query acg_conv2d_bwd_data_sums_supported)");
            return acg_igemm_ph4_launch(src, wb, bias, dst, g4, plan, g.w_elems, st);
        }
        if (acg_igemm_ph4_ok(g4) && acg_ph4_plan(t, ntp, &plan)) return acg_igemm_ph4_launch(src, wb, bias, dst, g4, plan, g.w_elems, st);
        return acg_igemm_launch(src, wb, bias, dst, g, t, st);
    }

    ACG_REQUIRE(ns == nullptr, "dgrad: norm sums on a stride-2 data gradient need the four-phase tile (query acg_conv2d_bwd_data_sums_supported)");
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            g.oy0 = py; g.ox0 = px;
            g.GH = (d->Hi - py + 1) / 2; g.GW = (d->Wi - px + 1) / 2;
            g.Mtot = (long long)d->N * g.GH * g.GW;
            t.n = phase_taps(py, px, t, 0);
            ACG_REQUIRE(t.n > 0, "dgrad: empty phase (K=%d p=%d)", K, p);
            if (stats != nullptr) { // each phase owns a quarter of every image's 128-pixel chunks
                const int per = (int)(((long long)g.GH * g.GW) / 128);
                g.stats = stats; g.stats_cpi = 4 * per; g.stats_chunk0 = (py * 2 + px) * per;
            }
            int rc = thin_in_valu_dgrad(d) ? thin_out_launch(src, wb, bias, dst, g, t, st) : acg_igemm_launch(src, wb, bias, dst, g, t, st);
            if (rc != ACG_OK) return rc;
        }
    return ACG_OK;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int acg_conv2d_fwd(const acg_conv_desc *d, const float *x, const float *wf, const float *bias, float *y,
                              int act, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_fwd");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (g_acg_conv_impl == ACG_IMPL_DIRECT) {
        const long long total = (long long)d->N * d->Ho * d->Wo * d->Co;
        hipLaunchKernelGGL(direct_fwd_kernel, dim3(acg_cdiv(total, 256)), dim3(256), 0, st, *d, x, wf, bias, y, act,
                           acg_ncols_pad(d->Co));
        ACG_CHECK_LAUNCH("direct_fwd_kernel");
        return ACG_OK;
    }
    Geom g; Taps t;
    fwd_geom(d, &g, &t, act);
    if (thin_out_valu_fwd(d)) return thin_out_launch(x, wf, bias, y, g, t, st);
    return acg_igemm_launch(x, wf, bias, y, g, t, st);
}

// Forward convolution that ALSO emits per-128-pixel-tile (mean, M2) of its output for the InstanceNorm behind it
// (modules.py:24-31 computes those statistics from the conv output in a separate pass).  Only the wave-specialised
// bf16x3 kernel implements it: 128-column tiles, Cin % 32 == 0, Ho*Wo % 128 == 0.  stats: [N][Ho*Wo/128][2][Co].
extern "C" int acg_conv2d_fwd_stats_supported(const acg_conv_desc *d)
{
    if (d == nullptr || g_acg_precision == ACG_PREC_F32 || g_acg_conv_impl != ACG_IMPL_MFMA) return 0;
    // the C4 image -> 32 channel stem (conv_thinrow_x3): statistics over its 8 x 16 pixel tiles
    if (thin_in(d) && d->Ci == 4 && d->Co == 32 && d->stride == 1 && d->K <= 7 && g_acg_precision == ACG_PREC_BF16X3 && d->Ho % 8 == 0 &&
        d->Wo % 16 == 0 && !acg_debug_switch("ACG_NO_THINROW"))
        return 1;
    if (thin_in(d) || thin_out(d) || d->Co < 32 || d->Ci % 16 != 0) return 0;    // MFMA tiles of the bf16 kernels only
    if (d->Co < 128 && acg_debug_switch("ACG_NO_GENERIC_STATS")) return 0;   // A/B switch: statistics pass for the narrow layers
    return ((long long)d->Ho * d->Wo) % 128 == 0 ? 1 : 0;
}

extern "C" int acg_conv2d_fwd_stats(const acg_conv_desc *d, const float *x, const float *wf, const float *bias, float *y,
                                    float *stats, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_fwd_stats");
    if (rc) return rc;
    ACG_REQUIRE(acg_conv2d_fwd_stats_supported(d) && stats != nullptr, "acg_conv2d_fwd_stats: unsupported shape or mode");
    Geom g; Taps t;
    fwd_geom(d, &g, &t, ACG_ACT_NONE);
    g.stats = stats; g.stats_chunk0 = 0; g.stats_cpi = (int)(((long long)d->Ho * d->Wo) / 128);
    return acg_igemm_launch(x, wf, bias, y, g, t, (hipStream_t)stream);
}

// ConvTranspose2d forward that also emits the per-tile statistics: its four sub-pixel phase launches each cover a quarter
// of every image's pixels and write their own chunks.  stats: [N][Hi*Wi/128][2][Ci] (Hi x Wi = the transposed
// convolution's OUTPUT, Ci its output channels).
extern "C" int acg_conv_transpose2d_fwd_stats_supported(const acg_conv_desc *d)
{
    if (d == nullptr || g_acg_precision == ACG_PREC_F32 || g_acg_conv_impl != ACG_IMPL_MFMA) return 0;
    if (d->stride != 2 || thin_in(d) || thin_out(d) || d->Ci < 32 || d->Co % 16 != 0 || d->Ci >= 128) return 0;
    if (acg_debug_switch("ACG_NO_GENERIC_STATS")) return 0;
    if (d->Hi % 2 || d->Wi % 2) return 0;
    return ((long long)(d->Hi / 2) * (d->Wi / 2)) % 128 == 0 ? 1 : 0;
}

extern "C" size_t acg_conv2d_bwd_data_workspace_bytes(const acg_conv_desc *d)
{
    if (d == nullptr || d->pad_mode != ACG_PAD_REFLECT || d->pad == 0) return 0;
    return (size_t)d->N * (d->Hi + 2 * d->pad) * (d->Wi + 2 * d->pad) * d->Ci * sizeof(float);
}

extern "C" int acg_conv2d_bwd_data(const acg_conv_desc *d, const float *dy, const float *wb, float *dx, void *ws,
                                   size_t ws_bytes, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_bwd_data");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (g_acg_conv_impl == ACG_IMPL_DIRECT) {
        const long long total = (long long)d->N * d->Hi * d->Wi * d->Ci;
        hipLaunchKernelGGL(direct_dgrad_kernel, dim3(acg_cdiv(total, 256)), dim3(256), 0, st, *d, dy, wb,
                           (const float *)nullptr, dx, (int)ACG_ACT_NONE, acg_ncols_pad(d->Ci));
        ACG_CHECK_LAUNCH("direct_dgrad_kernel");
        return ACG_OK;
    }
    return dgrad_igemm(d, dy, wb, nullptr, dx, ACG_ACT_NONE, ws, ws_bytes, st);
}

// dx = data gradient + addend (a tensor of dx's shape): the residual-path gradient of a ResnetBlock joins the gradient
// of the block's first convolution (modules.py:185-188, 232-235: out = x + conv_block(x)) inside the convolution's
// epilogue instead of in a separate element-wise pass.  Implemented by the frame path of the reflect data gradient.
extern "C" int acg_conv2d_bwd_data_add_supported(const acg_conv_desc *d)
{
    if (d == nullptr || g_acg_conv_impl != ACG_IMPL_MFMA) return 0;
    Geom g;
    g.Cin = d->Co; g.Cout = d->Ci; g.thin = (d->stride == 1 && thin_out(d)) ? 1 : 0;
    return dgrad_frame_ok(d, g) ? 1 : 0;
}

extern "C" int acg_conv2d_bwd_data_add(const acg_conv_desc *d, const float *dy, const float *wb, const float *addend,
                                       const unsigned *addend_mask, float *dx, void *ws, size_t ws_bytes, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_bwd_data_add");
    if (rc) return rc;
    ACG_REQUIRE(addend != nullptr && acg_conv2d_bwd_data_add_supported(d), "acg_conv2d_bwd_data_add: unsupported shape or mode");
    ACG_REQUIRE(addend_mask == nullptr || ((long long)d->Hi * d->Wi * (d->Ci / 4)) % 8 == 0,
                "acg_conv2d_bwd_data_add: the sign bitmask layout needs Hi*Wi*Ci/4 %% 8 == 0");
    return dgrad_igemm(d, dy, wb, nullptr, dx, ACG_ACT_NONE, ws, ws_bytes, (hipStream_t)stream, addend, nullptr, addend_mask);
}

extern "C" int acg_conv2d_bwd_data_relu(const acg_conv_desc *d, const float *dy, const float *wb, const float *x,
                                        float *dx, void *ws, size_t ws_bytes, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_bwd_data_relu");
    if (rc) return rc;
    ACG_REQUIRE(x != nullptr && acg_conv2d_bwd_data_add_supported(d), "acg_conv2d_bwd_data_relu: unsupported shape or mode");
    return dgrad_igemm(d, dy, wb, nullptr, dx, ACG_ACT_NONE, ws, ws_bytes, (hipStream_t)stream, nullptr, x);
}

// ---- pre-split ("S16") activation storage for the MFMA-bound 3x3 layers (conv_x3_pre.hip) ---------------------------------
// An S16 tensor has the shape and byte size of its fp32 NHWC twin; per pixel and 8-channel group it holds 16 bytes of bf16
// hi followed by 16 bytes of bf16 lo (x = hi + lo up to 2^-17 |x|: exactly the operand the bf16x3 convolutions consume).
// Replaces the per-launch split of `modules.py:205-227`'s activations inside the convolution loaders.
static bool acg_wgrad_krow_s16_ok(const acg_conv_desc *d) // the conditions under which wgrad_plan sizes the kernel-row split
{
    return g_acg_precision == ACG_PREC_BF16X3 && g_acg_conv_impl == ACG_IMPL_MFMA && !thin_in(d) && d->K == 3 && d->stride == 1 &&
           d->pad == 1 && d->Hi == d->Ho && d->Wi == d->Wo && d->Wo % 32 == 0 && d->Ci % 128 == 0 && d->Co % 128 == 0 &&
           !acg_debug_switch("ACG_NO_KROW");
}

static bool s16_dgrad_geom_ok(const acg_conv_desc *d)
{
    if (d->stride != 1 || d->pad_mode != ACG_PAD_REFLECT || d->pad <= 0) return false;
    Geom g; Taps t;
    g.Hin = d->Ho; g.Win = d->Wo; g.Cin = d->Co; g.Cout = d->Ci; g.reflect = 0; g.act = 0; g.ncols_pad = acg_ncols_pad(d->Ci);
    g.is = 1; g.thin = (d->stride == 1 && thin_out(d)) ? 1 : 0;
    const int p = d->pad, K = d->K;
    g.Hout = d->Hi + 2 * p; g.Wout = d->Wi + 2 * p; g.GH = g.Hout; g.GW = g.Wout; g.os = 1; g.oy0 = 0; g.ox0 = 0;
    g.Mtot = (long long)d->N * g.GH * g.GW;
    t.n = 0;
    for (int kh = 0; kh < K; ++kh)
        for (int kw = 0; kw < K; ++kw) { t.dy[t.n] = (short)(-kh); t.dx[t.n] = (short)(-kw); t.w[t.n] = (short)(kh * K + kw); t.n++; }
    return dgrad_frame_ok(d, g) && acg_igemm_x3_pre_ok(g, t);
}

extern "C" int acg_conv2d_s16_supported(const acg_conv_desc *d)
{
    if (d == nullptr || g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA) return 0;
    if (check_desc(d, "acg_conv2d_s16_supported") != ACG_OK || thin_in(d) || thin_out(d)) return 0;
    Geom g; Taps t;
    fwd_geom(d, &g, &t, 0);
    if (!acg_igemm_x3_pre_ok(g, t) || !s16_dgrad_geom_ok(d)) return 0;
    return acg_wgrad_krow_s16_ok(d) ? 1 : 0;
}

extern "C" int acg_conv2d_fwd_s16(const acg_conv_desc *d, const void *x, const float *wf, const float *bias, void *y, int act,
                                  float *stats, int out_s16, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_fwd_s16");
    if (rc) return rc;
    ACG_REQUIRE(g_acg_precision == ACG_PREC_BF16X3 && g_acg_conv_impl == ACG_IMPL_MFMA, "acg_conv2d_fwd_s16: bf16x3 MFMA mode only");
    Geom g; Taps t;
    fwd_geom(d, &g, &t, stats != nullptr ? (int)ACG_ACT_NONE : act);
    ACG_REQUIRE(stats == nullptr || act == ACG_ACT_NONE, "acg_conv2d_fwd_s16: statistics with an activation");
    g.out_s16 = out_s16;
    return acg_igemm_x3_pre_launch(x, wf, bias, (float *)y, g, t, g.w_elems, (hipStream_t)stream, stats);
}

// dy pre-split; addend (fp32, + optional sign bitmask) only with fp32 output; relu_src (pre-split) only with pre-split output
extern "C" int acg_conv2d_bwd_data_s16(const acg_conv_desc *d, const void *dy, const float *wb, void *dx, void *ws,
                                       size_t ws_bytes, const float *addend, const unsigned *addend_mask, const void *relu_src,
                                       int out_s16, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_bwd_data_s16");
    if (rc) return rc;
    ACG_REQUIRE(g_acg_precision == ACG_PREC_BF16X3 && g_acg_conv_impl == ACG_IMPL_MFMA && s16_dgrad_geom_ok(d),
                "acg_conv2d_bwd_data_s16: unsupported shape or mode");
    ACG_REQUIRE(addend_mask == nullptr || (addend != nullptr && ((long long)d->Hi * d->Wi * (d->Ci / 4)) % 8 == 0),
                "acg_conv2d_bwd_data_s16: the sign bitmask needs an addend and Hi*Wi*Ci/4 %% 8 == 0");
    return dgrad_igemm(d, (const float *)dy, wb, nullptr, (float *)dx, ACG_ACT_NONE, ws, ws_bytes, (hipStream_t)stream, addend,
                       (const float *)relu_src, addend_mask, 1, out_s16, relu_src != nullptr ? 1 : 0);
}

// conv + ReLU with pre-split output that also leaves the sign bitmask of that output, and the data gradient of the NEXT
// convolution masked by it (instead of reading the pre-split activation for its sign: 1/32 of the bytes)
extern "C" int acg_conv2d_fwd_s16_mask(const acg_conv_desc *d, const void *x, const float *wf, const float *bias, void *y,
                                       unsigned *sign_mask, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_fwd_s16_mask");
    if (rc) return rc;
    ACG_REQUIRE(g_acg_precision == ACG_PREC_BF16X3 && g_acg_conv_impl == ACG_IMPL_MFMA && sign_mask != nullptr && d->Co % 32 == 0,
                "acg_conv2d_fwd_s16_mask: bf16x3 MFMA mode, 32-multiple output channels");
    Geom g; Taps t;
    fwd_geom(d, &g, &t, ACG_ACT_RELU);
    g.out_s16 = 1; g.mask_out = sign_mask;
    return acg_igemm_x3_pre_launch(x, wf, bias, (float *)y, g, t, g.w_elems, (hipStream_t)stream, nullptr);
}

extern "C" int acg_conv2d_bwd_data_s16_mask(const acg_conv_desc *d, const void *dy, const float *wb, void *dx, void *ws,
                                            size_t ws_bytes, const unsigned *relu_sign_mask, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_bwd_data_s16_mask");
    if (rc) return rc;
    ACG_REQUIRE(relu_sign_mask != nullptr && acg_conv2d_bwd_data_s16_sums_supported(d) && d->Ci % 32 == 0,
                "acg_conv2d_bwd_data_s16_mask: unsupported shape or mode (query acg_conv2d_bwd_data_s16_sums_supported)");
    return dgrad_igemm(d, (const float *)dy, wb, nullptr, (float *)dx, ACG_ACT_NONE, ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr,
                       nullptr, 1, 1, 0, nullptr, nullptr, relu_sign_mask);
}

extern "C" int acg_conv2d_bwd_data_s16_sums_supported(const acg_conv_desc *d)
{
    return acg_conv2d_s16_supported(d) && dgrad_unpad_ok(d) && d->Ci % 4 == 0 && ((long long)d->Hi * d->Wi) % 128 == 0 ? 1 : 0;
}

extern "C" int acg_conv2d_bwd_data_s16_sums(const acg_conv_desc *d, const void *dy, const float *wb, float *dx, void *ws,
                                            size_t ws_bytes, const float *addend, const unsigned *addend_mask,
                                            const acg_norm_sums *ns, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_bwd_data_s16_sums");
    if (rc) return rc;
    ACG_REQUIRE(ns != nullptr && acg_conv2d_bwd_data_s16_sums_supported(d), "acg_conv2d_bwd_data_s16_sums: unsupported shape or mode");
    ACG_REQUIRE(addend_mask == nullptr || (addend != nullptr && ((long long)d->Hi * d->Wi * (d->Ci / 4)) % 8 == 0),
                "acg_conv2d_bwd_data_s16_sums: the sign bitmask needs an addend and Hi*Wi*Ci/4 %% 8 == 0");
    ACG_REQUIRE(ns->sign_mask == nullptr || ((long long)d->Hi * d->Wi * (d->Ci / 4)) % 8 == 0, "acg_conv2d_bwd_data_s16_sums: bitmask layout");
    return dgrad_igemm(d, (const float *)dy, wb, nullptr, dx, ACG_ACT_NONE, ws, ws_bytes, (hipStream_t)stream, addend, nullptr,
                       addend_mask, 1, 0, 0, nullptr, ns);
}

// The same on fp32 operands, where the data gradient runs on the persistent row pipeline (conv_rows_x3: zero-padded 3x3 stride 1,
// 32 output and 64 input channels of the convolution, width a multiple of 128), on the generic tile (the 32 -> 64 layer's data
// gradient) or on conv_thinrow_x3 (the head's): part[N][Hi*Wi/128][2][Ci], summed over the chunks by acg_norm_bwd_partials like the
// pre-split kernel's (where a workgroup owns several chunks its sums sit in the first, zeros in the others)
extern "C" int acg_conv2d_bwd_data_sums_supported(const acg_conv_desc *d)
{
    if (d == nullptr || g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA || acg_debug_switch("ACG_NO_ROWS")) return 0;
    if (check_desc(d, "acg_conv2d_bwd_data_sums_supported") != ACG_OK) return 0;
    static const bool no_tile_sums = acg_debug_switch("ACG_NO_TILE_SUMS");   // A/B switch: the three producers of round 6
    // the four-phase tile of the stride-2 3x3 data gradient (igemm_conv_ph4<SUMS>): 64 input channels of the convolution, phase
    // grid rows that are whole 128-pixel tiles
    if (!no_tile_sums && !acg_debug_switch("ACG_NO_PH4_SUMS") && d->K == 3 && d->stride == 2 && d->pad == 1 && d->pad_mode != ACG_PAD_REFLECT && d->Ci == 64 &&
        d->Co % 32 == 0 && d->Hi == 2 * d->Ho && d->Wi == 2 * d->Wo && d->Wo % 128 == 0 && !acg_debug_switch("ACG_NO_PH4") && !acg_debug_switch("ACG_NO_PHASED"))
        return 1;
    // the 7x7 (K <= 7) stride-1 zero-padded layer with a C4 image on its output side (the head, networks.py:187-188):
    // conv_thinrow_x3's whole 8 x 16 tiles
    if (!no_tile_sums && d->stride == 1 && d->pad_mode != ACG_PAD_REFLECT && thin_out(d) && d->Co == 4 && d->Ci == 32 && d->K >= 2 && d->K <= 7 &&
        d->Hi == d->Ho && d->Wi == d->Wo && d->Hi % 8 == 0 && d->Wi % 16 == 0 && !thin_in_valu_dgrad(d) && !acg_debug_switch("ACG_NO_THINROW"))
        return 1;
    // the generic 128-pixel tile on the data gradient of the zero-padded 3x3 stride-1 32 -> 64 layer (networks.py:164: 64 gathered,
    // 32 written channels — the mirror shape the row pipeline does not take): row-patch tiles, i.e. rows of whole 128-pixel tiles
    if (!no_tile_sums && d->K == 3 && d->stride == 1 && d->pad == 1 && d->pad_mode != ACG_PAD_REFLECT && d->Ci == 32 && d->Co == 64 &&
        d->Hi == d->Ho && d->Wi == d->Wo && d->Wi % 128 == 0 && !acg_debug_switch("ACG_NO_RP"))
        return 1;
    return d->K == 3 && d->stride == 1 && d->pad == 1 && d->pad_mode != ACG_PAD_REFLECT && d->Co == 32 && d->Ci == 64 && d->Hi == d->Ho &&
           d->Wi == d->Wo && d->Wi % 128 == 0 ? 1 : 0;
}

extern "C" int acg_conv2d_bwd_data_sums(const acg_conv_desc *d, const float *dy, const float *wb, float *dx, void *ws, size_t ws_bytes,
                                        const acg_norm_sums *ns, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_bwd_data_sums");
    if (rc) return rc;
    ACG_REQUIRE(ns != nullptr && ns->sign_mask == nullptr && acg_conv2d_bwd_data_sums_supported(d), "acg_conv2d_bwd_data_sums: unsupported shape or mode");
    return dgrad_igemm(d, dy, wb, nullptr, dx, ACG_ACT_NONE, ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr, nullptr, 0, 0, 0, nullptr, ns);
}

// x and dy pre-split; dw / db fp32 as in acg_conv2d_bwd_weight (db = column sums of dy, produced by the same launch)
extern "C" int acg_conv2d_bwd_weight_s16(const acg_conv_desc *d, const void *x, const void *dy, float *dw, float *db, int Or,
                                         int Ir, void *ws, size_t ws_bytes, int accumulate, void *stream);

// split-K plan shared by the workspace query and the launch
static bool wgrad_thin(const acg_conv_desc *d) { return thin_in(d); }
// the persistent patch kernel of the 7x7 image layers (conv_wgrad_thin.hip): workgroups = partial slabs, or 0 where the layer
// does not take it (same conditions as acg_wgrad_thin_patch_ok, on the descriptor)
static int wgrad_thin_patch_splits(const acg_conv_desc *d)
{
    if (g_acg_precision != ACG_PREC_BF16X3 || g_acg_conv_impl != ACG_IMPL_MFMA || d->stride != 1 || d->K < 2 || d->K > 7 ||
        d->Hi != d->Ho || d->Wi != d->Wo || acg_debug_switch("ACG_NO_WGRAD_THIN_PATCH"))
        return 0;
    const bool stem = thin_in(d) && d->Ci == 4 && d->Co == 32;
    const bool head = thin_out(d) && d->Co == 4 && d->Ci == 32 && !(d->pad_mode == ACG_PAD_REFLECT && d->pad > 0);
    if (!stem && !head) return 0;
    const long long ntiles = (long long)d->N * ((d->Ho + 7) / 8) * ((d->Wo + 15) / 16);
    return (int)(ntiles < 768 ? ntiles : 768);
}

static void wgrad_plan(const acg_conv_desc *d, int Cx, int Cg, long long Mtot, int *CiP, int *CoP, int *nsplit,
                       long long *mps)
{
    int bci, bco;
    acg_wgrad_tiles(Cx, Cg, &bci, &bco, wgrad_thin(d) ? 0 : d->K * d->K);
    *CiP = (Cx + bci - 1) / bci * bci;
    *CoP = (Cg + bco - 1) / bco * bco;
    if (wgrad_thin(d)) { // gathered columns = (tap, 4 channels): 32 per 8 taps, ONE tap-block
        bci = bco = 32;
        *CiP = 32 * ((d->K * d->K + 7) / 8);
        *CoP = (Cg + 31) / 32 * 32;
    }
    const int KP = 256; // multiple of every kernel variant's pixels-per-stage (fp32: 32/128, bf16: 64/256)
    const int nt = acg_wgrad_taps_per_wg(Cx, Cg, d->K * d->K, wgrad_thin(d) ? 1 : 0);   // taps per workgroup (bf16 kernels)
    const long long base = (wgrad_thin(d) ? 1LL : (long long)d->K * d->K / nt) * (*CiP / bci) * (*CoP / bco);
    // workgroups per launch: a whole number of residency waves.  The bf16 128x128 kernel holds 2 workgroups per CU:
    // 512 = exactly one wave (vs 1536: -6..-10 %, and a third of the partial-sum traffic); 768 = 1.5 waves is the worst
    // choice (+15 %).  The smaller tiles hold 3-4 per CU and keep more, shorter workgroups.
    long long target = (g_acg_precision != ACG_PREC_F32 && g_acg_conv_impl == ACG_IMPL_MFMA && ((bci == 128 && bco == 128) || nt == 3) && !wgrad_thin(d)) ? 512 : 1024;
    long long nblk = base;
    int gran = KP;
    // kernel-row weight gradient (conv_wgrad_tr.hip, same conditions as acg_wgrad_krow_ok): three taps per workgroup, one
    // 512-thread workgroup per CU -> one residency wave of 256
    static const bool no_krow = acg_debug_switch("ACG_NO_KROW");
    if (!no_krow && g_acg_precision == ACG_PREC_BF16X3 && g_acg_conv_impl == ACG_IMPL_MFMA && !wgrad_thin(d) && d->K == 3 &&
        d->stride == 1 && d->pad == 1 && d->Hi == d->Ho && d->Wi == d->Wo && d->Wo % 32 == 0 && Cx % 128 == 0 && Cg % 128 == 0) {
        nblk = 3LL * (*CiP / 128) * (*CoP / 128);
        target = 256;
        gran = 32; // its stage is a 32-pixel run: splits this fine fill 255 of the 256 CUs at batch 32 (256-pixel splits: 246)
    } else if (!no_krow && g_acg_precision == ACG_PREC_BF16X3 && g_acg_conv_impl == ACG_IMPL_MFMA && !wgrad_thin(d) && d->K == 3 &&
               d->stride == 1 && d->pad == 1 && d->Hi == d->Ho && d->Wi == d->Wo && d->Wo % 128 == 0 &&
               ((Cx == 32 && Cg == 64) || (Cx == 64 && Cg == 32))) {
        nblk = 3; // the 32 <-> 64 channel variant (acg_wgrad_krow_s_ok): 256 threads, two workgroups per CU
        target = 512;
        gran = 128;
    }
    if (!wgrad_thin(d) && acg_wgrad_krowg_shape_ok(d->K, d->stride, d->pad, d->pad_mode == ACG_PAD_REFLECT, d->Wi, d->Wo, Cx, Cg)) {
        nblk = (long long)d->K * (*CiP / (Cx == 64 ? 64 : 128)) * (*CoP / 128);   // wgrad_x3_krowg (conv_wgrad_k4.hip): a kernel row per
        target = 256;                                                             // workgroup, one workgroup per CU, whole output rows
        gran = d->Wo;                                                             // per split
    }
    if (wgrad_thin(d) && wgrad_thin_patch_splits(d) > 0) {   // one slab per persistent workgroup
        *nsplit = wgrad_thin_patch_splits(d);
        *mps = (Mtot + *nsplit - 1) / *nsplit;
        return;
    }
    long long ns = target / nblk;
    const long long cap = Mtot / (KP * 4);
    if (ns > cap) ns = cap;
    if (ns > 512) ns = 512;
    if (ns < 1) ns = 1;
    long long per = (Mtot + ns - 1) / ns;
    per = (per + gran - 1) / gran * gran;
    ns = (Mtot + per - 1) / per;
    *nsplit = (int)ns;
    *mps = per;
}

static size_t wgrad_ws_bytes(const acg_conv_desc *d, int Cx, int Cg, long long Mtot)
{
    int CiP, CoP, ns; long long mps;
    wgrad_plan(d, Cx, Cg, Mtot, &CiP, &CoP, &ns, &mps);
    const size_t part = (size_t)ns * d->K * d->K * CiP * CoP * sizeof(float);
    const int Cmax = d->Ci > d->Co ? d->Ci : d->Co;
    const long long Mbig = (long long)d->N * (d->Hi > d->Ho ? d->Hi : d->Ho) * (d->Wi > d->Wo ? d->Wi : d->Wo);
    const size_t bias_part = (size_t)ns * 2 * (CiP > CoP ? CiP : CoP) * sizeof(float);   // x-side sums: `stride` slots per split
    size_t total = acg_round_up(part, 256) + acg_round_up(colsum_ws_bytes(Mbig, Cmax) + bias_part, 256);
    if (thin_out(d) && d->stride == 1) { // wgrad_thin_out's partial buffer (its own split plan, <= 512 splits)
        const long long Mx = (long long)d->N * d->Hi * d->Wi;
        long long ns2 = 1536 / ((long long)((d->K * d->K + 7) / 8) * ((d->Ci + 31) / 32)), cap = Mx / 1024;
        if (ns2 > cap) ns2 = cap;
        if (ns2 > 512) ns2 = 512;
        if (ns2 < 1) ns2 = 1;
        if (wgrad_thin_patch_splits(d) > ns2) ns2 = wgrad_thin_patch_splits(d);
        const size_t t2 = (size_t)(ns2 + 1) * 32 * ((d->K * d->K + 7) / 8) * ((d->Ci + 31) / 32 * 32) * sizeof(float);
        if (t2 > total) total = t2;
    }
    return total;
}

extern "C" size_t acg_conv2d_bwd_weight_workspace_bytes(const acg_conv_desc *d)
{
    if (d == nullptr) return 0;
    // covers both orientations (Conv2d and ConvTranspose2d use of the same descriptor)
    const size_t a = wgrad_ws_bytes(d, d->Ci, d->Co, (long long)d->N * d->Ho * d->Wo);
    return a;
}

// x_side: conv-input-side tensor (N,Hi,Wi,Ci); g_side: conv-output-side tensor (N,Ho,Wo,Co)
// bias_from: 0 none; 1 db[c] = column sums of g_side (Conv2d bias, Or entries); 2 of x_side (ConvTranspose bias, Ir entries)
static int wgrad_common(const acg_conv_desc *d, const float *x_side, const float *g_side, float *dw, int Or, int Ir,
                        void *ws, size_t ws_bytes, hipStream_t st, int accumulate, int bias_from = 0, float *db = nullptr,
                        bool thin_conv = false, bool s16 = false)
{
    ACG_REQUIRE(Or <= d->Co && Ir <= d->Ci, "wgrad: Or=%d Ir=%d exceed padded dims", Or, Ir);
    if (g_acg_conv_impl == ACG_IMPL_DIRECT) {
        const long long total = (long long)Or * Ir * d->K * d->K;
        hipLaunchKernelGGL(direct_wgrad_kernel, dim3(acg_cdiv(total, 64)), dim3(64), 0, st, *d, x_side, g_side, dw, Or, Ir, accumulate);
        ACG_CHECK_LAUNCH("direct_wgrad_kernel");
        return ACG_OK;
    }
    WGeom g; Taps t; Geom gf;
    fwd_geom(d, &gf, &t, 0);
    g.Hin = d->Hi; g.Win = d->Wi; g.Cin = d->Ci; g.Hg = d->Ho; g.Wg = d->Wo; g.Cg = d->Co;
    g.is = d->stride; g.reflect = d->pad_mode == ACG_PAD_REFLECT;
    g.thin = (thin_conv && wgrad_thin(d)) ? 1 : 0;
    g.Mtot = (long long)d->N * d->Ho * d->Wo;
    wgrad_plan(d, d->Ci, d->Co, g.Mtot, &g.CiP, &g.CoP, &g.nsplit, &g.m_per_split);
    const int ntb = g.thin ? 1 : t.n;
    const size_t need = (size_t)g.nsplit * ntb * g.CiP * g.CoP * sizeof(float);
    if (ws == nullptr || ws_bytes < need) {
        acg_set_error("acg_conv2d_bwd_weight: workspace %zu < %zu", ws_bytes, need);
        return ACG_ERR_WORKSPACE;
    }
    g.bias_from = db != nullptr ? bias_from : 0;
    g.bias_part = (float *)((char *)ws + acg_round_up(need, 256));
    int rc;
    if (s16) {
        ACG_REQUIRE(acg_wgrad_krow_s16_ok(d) && g.CiP == d->Ci && g.CoP == d->Co && g.m_per_split % 32 == 0 && g.bias_from != 2,
                    "wgrad: pre-split operands need the kernel-row geometry (query acg_conv2d_s16_supported)");
        rc = acg_wgrad_krow_s16_launch(x_side, g_side, (float *)ws, g, st);
    } else {
        ACG_REQUIRE(g.bias_from != 2 || acg_wgrad_krowg_ok(g, t), "wgrad: x-side bias sums only in the kernel-row kernel");
        rc = acg_wgrad_launch(x_side, g_side, (float *)ws, g, t, st);
    }
    if (rc) return rc;
    acg_record_mid_event(st);
    const int Cp = bias_from == 1 ? g.CoP : g.CiP, Cr = g.bias_from ? (bias_from == 1 ? Or : Ir) : 0;
    const bool quad = g.thin != 2 && Or % 4 == 0 && g.CoP % 4 == 0;   // the kernel's four-channels-per-element path
    const long long total = (long long)t.n * Ir * (quad ? Or / 4 : Or);
    const int el_log2 = (total < 8192 && g.nsplit >= 64) ? 4 : 6;
    const int wblocks = acg_cdiv(total, 1 << el_log2);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wblocks + acg_cdiv(Cr, 16)), dim3(256), 0, st, (const float *)ws, g.nsplit,
                       t.n, g.CiP, g.CoP, Or, Ir, dw, g.thin, accumulate, wblocks, (const float *)g.bias_part, Cp, Cr, db,
                       bias_from == 2 ? g.nsplit * g.is : g.nsplit, el_log2);
    ACG_CHECK_LAUNCH("wgrad_reduce_kernel");
    return ACG_OK;
}

// Weight gradient of a convolution with <= 4 OUTPUT channels (stride 1): mirror image of the thin-Cin case.
//   dW[co][ci][kh,kw] = sum_{iy,ix} x[iy,ix][ci] * dy[iy-kh+p, ix-kw+p][co]
// GEMM rows (gathered, thin) = (tap, co<4) from dy, columns = ci from the plain x rows, K = input pixels.
static int wgrad_thin_out(const acg_conv_desc *d, const float *x, const float *dy, float *dw, int Or, int Ir, void *ws,
                          size_t ws_bytes, hipStream_t st, int accumulate)
{
    WGeom g; Taps t;
    const int K = d->K, p = d->pad;
    t.n = 0;
    for (int kh = 0; kh < K; ++kh)
        for (int kw = 0; kw < K; ++kw) { t.dy[t.n] = (short)(p - kh); t.dx[t.n] = (short)(p - kw); t.w[t.n] = (short)(kh * K + kw); t.n++; }
    g.Hin = d->Ho; g.Win = d->Wo; g.Cin = d->Co;      // gathered side: dy
    g.Hg = d->Hi; g.Wg = d->Wi; g.Cg = d->Ci;         // plain rows: x
    g.is = 1; g.reflect = 0; g.thin = 1; g.bias_from = 0; g.bias_part = nullptr;
    g.Mtot = (long long)d->N * d->Hi * d->Wi;
    g.CiP = 32 * ((K * K + 7) / 8);
    g.CoP = (d->Ci + 31) / 32 * 32;
    const long long base = (long long)(g.CiP / 32) * (g.CoP / 32);
    long long ns = 1536 / base, cap = g.Mtot / 1024;
    if (ns > cap) ns = cap;
    if (ns > 512) ns = 512;
    if (ns < 1) ns = 1;
    long long per = (g.Mtot + ns - 1) / ns;
    per = (per + 255) / 256 * 256;
    g.nsplit = (int)((g.Mtot + per - 1) / per);
    g.m_per_split = per;
    if (wgrad_thin_patch_splits(d) > 0) g.nsplit = wgrad_thin_patch_splits(d);   // one slab per persistent workgroup
    const size_t need = (size_t)g.nsplit * g.CiP * g.CoP * sizeof(float);
    if (ws == nullptr || ws_bytes < need) {
        acg_set_error("acg_conv2d_bwd_weight(thin out): workspace %zu < %zu", ws_bytes, need);
        return ACG_ERR_WORKSPACE;
    }
    int rc = acg_wgrad_launch(dy, x, (float *)ws, g, t, st);
    if (rc) return rc;
    acg_record_mid_event(st);
    const long long total = (long long)t.n * Ir * Or;
    const int el_log2 = (total < 8192 && g.nsplit >= 64) ? 4 : 6;
    const int wblocks = acg_cdiv(total, 1 << el_log2);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(wblocks), dim3(256), 0, st, (const float *)ws, g.nsplit, t.n,
                       g.CiP, g.CoP, Or, Ir, dw, 2, accumulate, wblocks, (const float *)nullptr, 0, 0, (float *)nullptr, 0, el_log2);
    ACG_CHECK_LAUNCH("wgrad_reduce_kernel");
    return ACG_OK;
}

static float *colsum_area(const acg_conv_desc *d, void *ws, size_t ws_bytes, size_t *avail)
{
    int CiP, CoP, ns; long long mps;
    wgrad_plan(d, d->Ci, d->Co, (long long)d->N * d->Ho * d->Wo, &CiP, &CoP, &ns, &mps);
    const size_t part = acg_round_up((size_t)ns * d->K * d->K * CiP * CoP * sizeof(float), 256);
    *avail = ws_bytes > part ? ws_bytes - part : 0;
    return (float *)((char *)ws + part);
}

extern "C" int acg_conv2d_bwd_weight(const acg_conv_desc *d, const float *x, const float *dy, float *dw, float *db,
                                     int Or, int Ir, void *ws, size_t ws_bytes, int accumulate, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_bwd_weight");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ACG_REQUIRE(ws != nullptr && ws_bytes >= acg_conv2d_bwd_weight_workspace_bytes(d), "acg_conv2d_bwd_weight: workspace too small");
    // the mirrored thin formulation walks the UNPADDED input pixels: right for zero padding only (a reflected border
    // pairs x[refl(q)] with dy of pixels outside that walk); reflect-padded thin-output layers take the general path
    const bool tout = thin_out(d) && d->stride == 1 && !(d->pad_mode == ACG_PAD_REFLECT && d->pad > 0);
    const bool fused = dw != nullptr && db != nullptr && g_acg_conv_impl == ACG_IMPL_MFMA && !tout;
    if (dw != nullptr) {
        rc = tout ? wgrad_thin_out(d, x, dy, dw, Or, Ir, ws, ws_bytes, st, accumulate)
                  : wgrad_common(d, x, dy, dw, Or, Ir, ws, ws_bytes, st, accumulate, 1, fused ? db : nullptr, true);
        if (rc) return rc;
    }
    if (db != nullptr && !fused) {
        size_t avail; float *cw = colsum_area(d, ws, ws_bytes, &avail);
        const long long M = (long long)d->N * d->Ho * d->Wo;
        ACG_REQUIRE(avail >= colsum_ws_bytes(M, d->Co), "acg_conv2d_bwd_weight: colsum workspace");
        rc = colsum_launch(dy, M, d->Co, Or, db, cw, st, accumulate);
    }
    return rc;
}

extern "C" int acg_conv2d_bwd_weight_s16(const acg_conv_desc *d, const void *x, const void *dy, float *dw, float *db, int Or,
                                         int Ir, void *ws, size_t ws_bytes, int accumulate, void *stream)
{
    int rc = check_desc(d, "acg_conv2d_bwd_weight_s16");
    if (rc) return rc;
    ACG_REQUIRE(dw != nullptr && ws != nullptr && ws_bytes >= acg_conv2d_bwd_weight_workspace_bytes(d),
                "acg_conv2d_bwd_weight_s16: workspace too small");
    return wgrad_common(d, (const float *)x, (const float *)dy, dw, Or, Ir, ws, ws_bytes, (hipStream_t)stream, accumulate, 1, db, false, true);
}

extern "C" int acg_conv_transpose2d_fwd(const acg_conv_desc *d, const float *x, const float *wb, const float *bias,
                                        float *y, int act, void *stream)
{
    int rc = check_desc(d, "acg_conv_transpose2d_fwd");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (g_acg_conv_impl == ACG_IMPL_DIRECT) {
        const long long total = (long long)d->N * d->Hi * d->Wi * d->Ci;
        hipLaunchKernelGGL(direct_dgrad_kernel, dim3(acg_cdiv(total, 256)), dim3(256), 0, st, *d, x, wb, bias, y, act,
                           acg_ncols_pad(d->Ci));
        ACG_CHECK_LAUNCH("direct_dgrad_kernel");
        return ACG_OK;
    }
    return dgrad_igemm(d, x, wb, bias, y, act, nullptr, 0, st);
}

extern "C" int acg_conv_transpose2d_fwd_stats(const acg_conv_desc *d, const float *x, const float *wb, const float *bias,
                                             float *y, float *stats, void *stream)
{
    int rc = check_desc(d, "acg_conv_transpose2d_fwd_stats");
    if (rc) return rc;
    ACG_REQUIRE(acg_conv_transpose2d_fwd_stats_supported(d) && stats != nullptr, "acg_conv_transpose2d_fwd_stats: unsupported shape or mode");
    return dgrad_igemm(d, x, wb, bias, y, ACG_ACT_NONE, nullptr, 0, (hipStream_t)stream, nullptr, nullptr, nullptr, 0, 0, 0, stats);
}

extern "C" int acg_conv_transpose2d_bwd_data(const acg_conv_desc *d, const float *dy, const float *wf, float *dx,
                                             void *stream)
{
    // adjoint of the adjoint: the plain forward convolution, no bias / activation
    return acg_conv2d_fwd(d, dy, wf, nullptr, dx, ACG_ACT_NONE, stream);
}

extern "C" int acg_conv_transpose2d_bwd_weight(const acg_conv_desc *d, const float *x, const float *dy, float *dw,
                                               float *db, int Or, int Ir, void *ws, size_t ws_bytes, int accumulate,
                                               void *stream)
{
    // underlying Conv2d: input side = ConvTranspose OUTPUT gradient dy (N,Hi,Wi,Ci), output side = x (N,Ho,Wo,Co);
    // weight (Cin_T, Cout_T, k, k) == OIHW of that Conv2d.  Bias gradient sums dy over pixels (Cout_T = Ir).
    int rc = check_desc(d, "acg_conv_transpose2d_bwd_weight");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ACG_REQUIRE(ws != nullptr && ws_bytes >= acg_conv2d_bwd_weight_workspace_bytes(d), "acg_conv_transpose2d_bwd_weight: workspace too small");
    // the bias sums the GATHERED-side operand (dy of the ConvTranspose), whose tap-0 gather visits only a strided subset of
    // its pixels: fused only in the kernel-row kernel (conv_wgrad_k4.hip), where kernel rows 1 .. stride visit every row
    // once; otherwise one separate column-sum pass
    const bool fused = dw != nullptr && db != nullptr && g_acg_conv_impl == ACG_IMPL_MFMA && d->stride <= 2 &&
                       d->Hi == d->stride * d->Ho && d->Wi == d->stride * d->Wo &&
                       acg_wgrad_krowg_shape_ok(d->K, d->stride, d->pad, d->pad_mode == ACG_PAD_REFLECT, d->Wi, d->Wo, d->Ci, d->Co);
    if (dw != nullptr) {
        rc = wgrad_common(d, dy, x, dw, Or, Ir, ws, ws_bytes, st, accumulate, fused ? 2 : 0, fused ? db : nullptr);
        if (rc) return rc;
    }
    if (db != nullptr && !fused) {
        size_t avail; float *cw = colsum_area(d, ws, ws_bytes, &avail);
        const long long M = (long long)d->N * d->Hi * d->Wi;
        ACG_REQUIRE(avail >= colsum_ws_bytes(M, d->Ci), "acg_conv_transpose2d_bwd_weight: colsum workspace");
        rc = colsum_launch(dy, M, d->Ci, Ir, db, cw, st, accumulate);
    }
    return rc;
}
