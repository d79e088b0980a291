// Elementwise / reduction / small-dense kernels (all HBM- or latency-bound) and the optimiser.
#include <math.h>
#include "common.h"
#include <cstdlib>

static int ew_blocks(long long total, int per = 256)
{
    long long b = (total + per - 1) / per;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}
#define GRID_STRIDE(i, total) \
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

// ---------------------------------------------------------------- error plumbing / version
static thread_local char g_err[512] = "";
void acg_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *acg_last_error(void) { return g_err; }
// name (template arguments included) of the convolution kernel the calling thread dispatched last — bench.py labels its
// roofline with what actually ran instead of a hard-coded string
static thread_local char g_kern[160] = "";
void acg_note_kernel(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kern, sizeof(g_kern), fmt, ap);
    va_end(ap);
}
extern "C" const char *acg_last_kernel(void) { return g_kern; }
bool acg_debug_switch(const char *name)
{
    static const bool enabled = getenv("ACG_DEBUG_SWITCHES") != nullptr;
    return enabled && getenv(name) != nullptr;
}
extern "C" int acg_version(void) { return ACG_VERSION; }

// bench.py's timing hook: an event the NEXT weight-gradient entry point of this thread records on its stream between its
// main kernel and its split-K reduction, so that the two are timed apart (then cleared)
static thread_local void *g_mid_event = nullptr;
extern "C" int acg_debug_mid_event(void *event) { g_mid_event = event; return ACG_OK; }
void acg_record_mid_event(hipStream_t st)
{
    if (g_mid_event != nullptr) {
        (void)hipEventRecord((hipEvent_t)g_mid_event, st);
        g_mid_event = nullptr;
    }
}

// What the matrix pipe HOLDS under load on this device, now: a register-only loop of v_mfma_f32_16x16x32_bf16 on random bf16
// operands (no LDS, no memory), two waves per SIMD on every CU — tools/probes/mfma_rate.hip inside the library, so that
// bench.py can print the sustained ceiling it measured itself next to the spec peak.  The caller times the launch;
// *flops_out = the FLOPs it executes.
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned probe_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mfma_rate_probe_kernel(float *__restrict__ out, int iters)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    probe_bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // two bf16 per word: hashed sign / mantissa, exponents near 1.0
        probe_u32x4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned h = (tid * 8u + i) * 2654435761u + k * 40503u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const unsigned lo = (h & 0x807fu) | ((120u + (h >> 20) % 8u) << 7), hi = ((h >> 8) & 0x807fu) | ((120u + (h >> 24) % 8u) << 7);
            w[k] = lo | (hi << 16);
        }
        if (i < 4) a[i] = __builtin_bit_cast(probe_bf16x8, w); else b[i - 4] = __builtin_bit_cast(probe_bf16x8, w);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) keep += acc[i][j][0] + acc[i][j][3];
    if (keep == 12345.678f) out[tid] = keep;
}
extern "C" int acg_probe_mfma_rate(float *scratch, size_t scratch_floats, int iters, double *flops_out, void *stream)
{
    int dev = 0, cus = 0;
    ACG_REQUIRE(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0,
                "acg_probe_mfma_rate: no device");
    ACG_REQUIRE(scratch != nullptr && scratch_floats >= (size_t)cus * 512 && iters > 0 && flops_out != nullptr, "acg_probe_mfma_rate: scratch of >= 512 floats per CU");
    hipLaunchKernelGGL(mfma_rate_probe_kernel, dim3(cus), dim3(512), 0, (hipStream_t)stream, scratch, iters);
    ACG_CHECK_LAUNCH("mfma_rate_probe_kernel");
    *flops_out = (double)cus * 8 * iters * 16 * 2.0 * 16 * 16 * 32;
    return ACG_OK;
}

// ---------------------------------------------------------------- activation backward
__global__ void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, float *__restrict__ dx,
                               long long n4, int act)
{
    GRID_STRIDE(i, n4) {
        f32x4 g = *(const f32x4 *)(dy + i * 4);
        const f32x4 yy = *(const f32x4 *)(y + i * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] *= acg_act_grad_from_y(yy[k], act);
        *(f32x4 *)(dx + i * 4) = g;
    }
}
extern "C" int acg_act_bwd(const float *dy, const float *y, float *dx, size_t n, int act, void *stream)
{
    ACG_REQUIRE(n % 4 == 0, "acg_act_bwd: n %% 4 != 0");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, dx,
                       (long long)(n / 4), act);
    ACG_CHECK_LAUNCH("act_bwd_kernel");
    return ACG_OK;
}

// ---------------------------------------------------------------- pre-split ("S16") storage: fp32 <-> (bf16 hi, bf16 lo)
// per 8 consecutive elements: 16 bytes of hi then 16 bytes of lo (conv_internal.h, acg_split8): the operand form of the
// bf16x3 convolutions, written once by the producer of an activation instead of by every loader that gathers it
#include "conv_internal.h"
__global__ void s16_encode_kernel(const float *__restrict__ x, char *__restrict__ y, long long n8)
{
    GRID_STRIDE(i, n8) {
        const f32x4 a = *(const f32x4 *)(x + i * 8), c = *(const f32x4 *)(x + i * 8 + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        acg_u32x4 hi, lo;
        acg_split8(v, hi, lo);
        *(acg_u32x4 *)(y + i * 32) = hi;
        *(acg_u32x4 *)(y + i * 32 + 16) = lo;
    }
}
__global__ void s16_decode_kernel(const char *__restrict__ x, float *__restrict__ y, long long n8)
{
    GRID_STRIDE(i, n8) {
        const acg_u32x4 hi = *(const acg_u32x4 *)(x + i * 32), lo = *(const acg_u32x4 *)(x + i * 32 + 16);
        f32x4 a, c;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a[2 * q] = __builtin_bit_cast(float, hi[q] << 16) + __builtin_bit_cast(float, lo[q] << 16);
            a[2 * q + 1] = __builtin_bit_cast(float, hi[q] & 0xffff0000u) + __builtin_bit_cast(float, lo[q] & 0xffff0000u);
            c[2 * q] = __builtin_bit_cast(float, hi[2 + q] << 16) + __builtin_bit_cast(float, lo[2 + q] << 16);
            c[2 * q + 1] = __builtin_bit_cast(float, hi[2 + q] & 0xffff0000u) + __builtin_bit_cast(float, lo[2 + q] & 0xffff0000u);
        }
        *(f32x4 *)(y + i * 8) = a;
        *(f32x4 *)(y + i * 8 + 4) = c;
    }
}
extern "C" int acg_s16_encode(const float *x, void *y, size_t n, void *stream)
{
    ACG_REQUIRE(n % 8 == 0, "acg_s16_encode: n %% 8 != 0");
    hipLaunchKernelGGL(s16_encode_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, x, (char *)y, (long long)(n / 8));
    ACG_CHECK_LAUNCH("s16_encode_kernel");
    return ACG_OK;
}
extern "C" int acg_s16_decode(const void *x, float *y, size_t n, void *stream)
{
    ACG_REQUIRE(n % 8 == 0, "acg_s16_decode: n %% 8 != 0");
    hipLaunchKernelGGL(s16_decode_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream, (const char *)x, y, (long long)(n / 8));
    ACG_CHECK_LAUNCH("s16_decode_kernel");
    return ACG_OK;
}

// ---------------------------------------------------------------- layout at the API edge
// NCHW (C real) -> NHWC with Cp channels (zeros beyond C).  One thread per (n, h, w) pixel-channel-quad.
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ s, float *__restrict__ d, int N, int C, int H, int W, int Cp)
{
    const long long HW = (long long)H * W;
    const long long total = (long long)N * HW * Cp;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % Cp);
        const long long p = i / Cp;
        const long long n = p / HW, hw = p - n * HW;
        d[i] = c < C ? s[(n * C + c) * HW + hw] : 0.f;
    }
}
__global__ void nhwc_to_nchw_kernel(const float *__restrict__ s, float *__restrict__ d, int N, int C, int H, int W, int Cp)
{
    const long long HW = (long long)H * W;
    const long long total = (long long)N * C * HW;
    GRID_STRIDE(i, total) {
        const long long hw = i % HW;
        const long long nc = i / HW;
        const long long n = nc / C;
        const int c = (int)(nc - n * C);
        d[i] = s[(n * HW + hw) * Cp + c];
    }
}
extern "C" int acg_nchw_to_nhwc16(const float *src, float *dst, int N, int C, int H, int W, int Cp, void *stream)
{
    ACG_REQUIRE(Cp >= C && Cp % 4 == 0, "acg_nchw_to_nhwc16: Cp=%d C=%d", Cp, C);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_blocks((long long)N * H * W * Cp)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, N, C, H, W, Cp);
    ACG_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return ACG_OK;
}
extern "C" int acg_nhwc16_to_nchw(const float *src, float *dst, int N, int C, int H, int W, int Cp, void *stream)
{
    ACG_REQUIRE(Cp >= C, "acg_nhwc16_to_nchw: Cp=%d C=%d", Cp, C);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_blocks((long long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, N, C, H, W, Cp);
    ACG_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return ACG_OK;
}

// ---- data path (dataloader.py:17-35): raw fields (N,H,W,Craw) -> first C channels, NaN -> 0, per-sample / per-channel
// min-max to [-1, 1] (a constant plane -> 0), NCHW.  One workgroup per (sample, channel) plane: min/max over the plane
// (wave shuffles + LDS), then the scaled plane.  Strided reads (channel c of NHWC rows): this runs once per data set.
__global__ __launch_bounds__(256) void minmax_scale_kernel(const float *__restrict__ raw, float *__restrict__ out, int C,
                                                           int Craw, long long HW)
{
    __shared__ float smn[4], smx[4];
    const int n = blockIdx.x / C, c = blockIdx.x % C;
    const float *src = raw + (long long)n * HW * Craw + c;
    float mn = INFINITY, mx = -INFINITY;
    for (long long i = threadIdx.x; i < HW; i += 256) {
        float v = src[i * Craw];
        v = (v != v) ? 0.f : v;            // np.nan_to_num: NaN -> 0 (+-inf -> +-FLT_MAX below)
        v = fminf(fmaxf(v, -3.4028234664e38f), 3.4028234664e38f);
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    const float inv = mx > mn ? 2.f / (mx - mn) : 0.f;
    float *dst = out + ((long long)n * C + c) * HW;
    for (long long i = threadIdx.x; i < HW; i += 256) {
        float v = src[i * Craw];
        v = (v != v) ? 0.f : v;
        v = fminf(fmaxf(v, -3.4028234664e38f), 3.4028234664e38f);
        dst[i] = mx > mn ? -1.f + (v - mn) * inv : 0.f;
    }
}
extern "C" int acg_minmax_scale_nhwc_to_nchw(const float *raw, float *out, int N, int H, int W, int Craw, int C, void *stream)
{
    ACG_REQUIRE(raw != nullptr && out != nullptr && N > 0 && H > 0 && W > 0 && C > 0 && C <= Craw,
                "acg_minmax_scale_nhwc_to_nchw: bad arguments (N=%d H=%d W=%d Craw=%d C=%d)", N, H, W, Craw, C);
    hipLaunchKernelGGL(minmax_scale_kernel, dim3((unsigned)(N * C)), dim3(256), 0, (hipStream_t)stream, raw, out, C, Craw,
                       (long long)H * W);
    ACG_CHECK_LAUNCH("minmax_scale_kernel");
    return ACG_OK;
}

__global__ void concat_kernel(const float *__restrict__ a, int Ca, int Cap, const float *__restrict__ b, int Cb, int Cbp,
                              float *__restrict__ d, int Cdp, long long npix)
{
    const long long total = npix * Cdp;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % Cdp);
        const long long p = i / Cdp;
        float v = 0.f;
        if (c < Ca) v = a[p * Cap + c];
        else if (c < Ca + Cb) v = b[p * Cbp + (c - Ca)];
        d[i] = v;
    }
}
extern "C" int acg_concat_channels(const float *a, int Ca, int Cap, const float *b, int Cb, int Cbp, float *dst, int Cdp,
                                   size_t npix, void *stream)
{
    ACG_REQUIRE(Ca + Cb <= Cdp, "acg_concat_channels: %d+%d > %d", Ca, Cb, Cdp);
    hipLaunchKernelGGL(concat_kernel, dim3(ew_blocks((long long)npix * Cdp)), dim3(256), 0, (hipStream_t)stream, a, Ca,
                       Cap, b, Cb, Cbp, dst, Cdp, (long long)npix);
    ACG_CHECK_LAUNCH("concat_kernel");
    return ACG_OK;
}
__global__ void split_kernel(const float *__restrict__ g, int Cdp, float *__restrict__ ga, int Ca, int Cap,
                             float *__restrict__ gb, int Cb, int Cbp, long long npix)
{
    const int Cm = Cap > Cbp ? Cap : Cbp;
    const long long total = npix * Cm;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % Cm);
        const long long p = i / Cm;
        if (ga != nullptr && c < Cap) ga[p * Cap + c] = c < Ca ? g[p * Cdp + c] : 0.f;
        if (gb != nullptr && c < Cbp) gb[p * Cbp + c] = c < Cb ? g[p * Cdp + Ca + c] : 0.f;
    }
}
extern "C" int acg_split_channels(const float *gdst, int Cdp, float *ga, int Ca, int Cap, float *gb, int Cb, int Cbp,
                                  size_t npix, void *stream)
{
    ACG_REQUIRE(Ca + Cb <= Cdp, "acg_split_channels: %d+%d > %d", Ca, Cb, Cdp);
    const int Cm = Cap > Cbp ? Cap : Cbp;
    hipLaunchKernelGGL(split_kernel, dim3(ew_blocks((long long)npix * Cm)), dim3(256), 0, (hipStream_t)stream, gdst, Cdp,
                       ga, Ca, Cap, gb, Cb, Cbp, (long long)npix);
    ACG_CHECK_LAUNCH("split_kernel");
    return ACG_OK;
}

// ---------------------------------------------------------------- small dense layers
__global__ void linear_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,
                                  float *__restrict__ y, int N, int I, int ldx, int O, int Op, int act)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Op) return;
    const int n = i / Op, o = i - n * Op;
    float acc = 0.f;
    if (o < O) {
        acc = b ? b[o] : 0.f;
        for (int k = 0; k < I; ++k) acc += x[(long long)n * ldx + k] * w[(long long)o * I + k];
        acc = acg_apply_act(acc, act);
    }
    y[i] = acc;
}
extern "C" int acg_linear_fwd(const float *x, const float *w, const float *b, float *y, int N, int I, int ldx, int O,
                              int Op, int act, void *stream)
{
    ACG_REQUIRE(N > 0 && I > 0 && O > 0 && Op >= O && ldx >= I, "acg_linear_fwd: bad dims");
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(acg_cdiv((long)N * Op, 128)), dim3(128), 0, (hipStream_t)stream, x, w, b, y,
                       N, I, ldx, O, Op, act);
    ACG_CHECK_LAUNCH("linear_fwd_kernel");
    return ACG_OK;
}
// mode 0: dx[n][i] = sum_o g*w[o][i]   (threads over N*I)
// mode 1: dw[o][i] = sum_n g*x[n][i]   (threads over O*I) ; db[o] = sum_n g (threads with i == 0)
__global__ void linear_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, const float *__restrict__ x,
                                  const float *__restrict__ w, float *__restrict__ dx, float *__restrict__ dw,
                                  float *__restrict__ db, int N, int I, int ldx, int O, int Op, int act, int mode)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == 0) {
        if (t >= N * I) return;
        const int n = t / I, i = t - n * I;
        float acc = 0.f;
#pragma unroll 8 // independent loads in flight: the rolled loop paid one L2 latency per output channel (19 us for 128)
        for (int o = 0; o < O; ++o) {
            const float g = dy[(long long)n * Op + o] * acg_act_grad_from_y(y[(long long)n * Op + o], act);
            acc += g * w[(long long)o * I + i];
        }
        dx[(long long)n * ldx + i] = acc;
    } else {
        if (t >= O * I) return;
        const int o = t / I, i = t - o * I;
        float acc = 0.f, bs = 0.f;
#pragma unroll 8
        for (int n = 0; n < N; ++n) {
            const float g = dy[(long long)n * Op + o] * acg_act_grad_from_y(y[(long long)n * Op + o], act);
            acc += g * x[(long long)n * ldx + i];
            bs += g;
        }
        if (dw) dw[t] = acc;
        if (db && i == 0) db[o] = bs;
    }
}
// dx[n][i] = sum_o g[n][o] * w[o][i] with the sum over o split across the workgroup: one workgroup per row n, thread
// (chunk, i) adds its chunk of output channels, the chunks meet in LDS in fixed order (deterministic).  The one-thread-per-
// output loop above needs O dependent-latency steps (19 us for O = 128 at 4 workgroups); this one O / chunks.
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                            const float *__restrict__ w, float *__restrict__ dx, int I, int ldx,
                                                            int O, int Op, int act)
{
    __shared__ float red[256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int CH = 256 / I, OC = (O + CH - 1) / CH; // launcher guarantees I <= 256
    const int i = tid % I, ch = tid / I;
    float acc = 0.f;
    if (ch < CH) {
        const int o1 = (ch + 1) * OC < O ? (ch + 1) * OC : O;
#pragma unroll 4
        for (int o = ch * OC; o < o1; ++o) {
            const float g = dy[(long long)n * Op + o] * acg_act_grad_from_y(y[(long long)n * Op + o], act);
            acc += g * w[(long long)o * I + i];
        }
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < I) {
        float s = 0.f;
        for (int c = 0; c < CH; ++c) s += red[c * I + tid];
        dx[(long long)n * ldx + tid] = s;
    }
}
extern "C" int acg_linear_bwd(const float *dy, const float *y, const float *x, const float *w, float *dx, float *dw,
                              float *db, int N, int I, int ldx, int O, int Op, int act, void *stream)
{
    ACG_REQUIRE(N > 0 && I > 0 && O > 0 && Op >= O && ldx >= I, "acg_linear_bwd: bad dims");
    hipStream_t st = (hipStream_t)stream;
    if (dx != nullptr && I <= 256 && O >= 32)
        hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(N), dim3(256), 0, st, dy, y, w, dx, I, ldx, O, Op, act);
    else if (dx != nullptr)
        hipLaunchKernelGGL(linear_bwd_kernel, dim3(acg_cdiv((long)N * I, 128)), dim3(128), 0, st, dy, y, x, w, dx, dw, db, N,
                           I, ldx, O, Op, act, 0);
    if (dw != nullptr || db != nullptr)
        hipLaunchKernelGGL(linear_bwd_kernel, dim3(acg_cdiv((long)O * I, 128)), dim3(128), 0, st, dy, y, x, w, dx, dw, db, N,
                           I, ldx, O, Op, act, 1);
    ACG_CHECK_LAUNCH("linear_bwd_kernel");
    return ACG_OK;
}

// ---------------------------------------------------------------- one launch for many small copies / accumulations
// dst[s][i] (+)= src[off[s] + i]: the per-layer slices of a concatenated gradient into the layers' own .grad tensors
__global__ void segments_accumulate_kernel(const float *__restrict__ src, acg_segments sg, int accumulate)
{
    const int s = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sg.len[s]) return;
    float *d = (float *)sg.dst[s];
    const float v = src[(long long)sg.off[s] + i];
    d[i] = accumulate ? d[i] + v : v;
}
extern "C" int acg_segments_accumulate(const float *src, const acg_segments *segs, int accumulate, void *stream)
{
    ACG_REQUIRE(segs != nullptr && segs->n >= 0 && segs->n <= ACG_MAX_SEGMENTS, "acg_segments_accumulate: bad segment count");
    int mx = 0;
    for (int s = 0; s < segs->n; ++s) {
        ACG_REQUIRE(segs->dst[s] != nullptr && segs->len[s] >= 0 && segs->off[s] >= 0, "acg_segments_accumulate: bad segment %d", s);
        mx = segs->len[s] > mx ? segs->len[s] : mx;
    }
    if (segs->n == 0 || mx == 0) return ACG_OK;
    hipLaunchKernelGGL(segments_accumulate_kernel, dim3(acg_cdiv(mx, 256), segs->n), dim3(256), 0, (hipStream_t)stream, src, *segs,
                       accumulate);
    ACG_CHECK_LAUNCH("segments_accumulate_kernel");
    return ACG_OK;
}

// ---------------------------------------------------------------- spatial mean [N][P][Cp] -> [N][Cp]
__global__ void spatial_mean_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, long long P, int Cp)
{
    __shared__ float red[256];
    const int n = blockIdx.x, c = blockIdx.y;
    float s = 0.f;
    for (long long p = threadIdx.x; p < P; p += blockDim.x) s += x[((long long)n * P + p) * Cp + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) y[(long long)n * Cp + c] = red[0] / (float)P;
}
__global__ void spatial_mean_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx, long long P, int Cp, long long total)
{
    GRID_STRIDE(i, total) {
        const int c = (int)(i % Cp);
        const long long n = i / ((long long)P * Cp);
        dx[i] = dy[n * Cp + c] / (float)P;
    }
}
extern "C" int acg_spatial_mean_fwd(const float *x, float *y, int N, size_t P, int Cp, void *stream)
{
    hipLaunchKernelGGL(spatial_mean_fwd_kernel, dim3(N, Cp), dim3(256), 0, (hipStream_t)stream, x, y, (long long)P, Cp);
    ACG_CHECK_LAUNCH("spatial_mean_fwd_kernel");
    return ACG_OK;
}
extern "C" int acg_spatial_mean_bwd(const float *dy, float *dx, int N, size_t P, int Cp, void *stream)
{
    const long long total = (long long)N * P * Cp;
    hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, dx,
                       (long long)P, Cp, total);
    ACG_CHECK_LAUNCH("spatial_mean_bwd_kernel");
    return ACG_OK;
}

// ---------------------------------------------------------------- reductions to a device scalar
// mode: 0 = sum((p-t)^2) ; 1 = sum(|a-b|) ; 2 = sum(p) ; 3 = sum(p^2) (no channel mask)
#define RED_BLOCKS 1024
template <int MODE>
__global__ __launch_bounds__(256) void reduce_partial_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                             long long total, int C, int Cp, float target,
                                                             float *__restrict__ part)
{
    __shared__ float red[256];
    float s = 0.f;
    GRID_STRIDE(i, total) {
        if (MODE != 3 && (int)(i % Cp) >= C) continue;
        const float v = a[i];
        if (MODE == 0) { const float d = v - target; s += d * d; }
        else if (MODE == 1) s += fabsf(v - b[i]);
        else if (MODE == 2) s += v;
        else s += v * v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void reduce_final_kernel(const float *__restrict__ part, int nb, float scale,
                                                           float *__restrict__ out)
{
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * scale;
}
extern "C" size_t acg_reduce_workspace_bytes(size_t n) { (void)n; return RED_BLOCKS * sizeof(float); }

template <int MODE>
static int reduce_launch(const float *a, const float *b, long long total, int C, int Cp, float target, float scale,
                         float *out, void *ws, size_t ws_bytes, hipStream_t st, const char *who)
{
    if (ws == nullptr || ws_bytes < RED_BLOCKS * sizeof(float)) {
        acg_set_error("%s: workspace too small", who);
        return ACG_ERR_WORKSPACE;
    }
    long long nbl = (total + 2047) / 2048;
    const int nb = (int)(nbl > RED_BLOCKS ? RED_BLOCKS : (nbl < 1 ? 1 : nbl));
    hipLaunchKernelGGL((reduce_partial_kernel<MODE>), dim3(nb), dim3(256), 0, st, a, b, total, C, Cp, target, (float *)ws);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, st, (const float *)ws, nb, scale, out);
    ACG_CHECK_LAUNCH(who);
    return ACG_OK;
}
extern "C" int acg_mse_const_fwd(const float *p, size_t npix, int C, int Cp, float target, float *out, void *ws,
                                 size_t ws_bytes, void *stream)
{
    return reduce_launch<0>(p, nullptr, (long long)npix * Cp, C, Cp, target, 1.f / ((float)npix * C), out, ws, ws_bytes,
                            (hipStream_t)stream, "acg_mse_const_fwd");
}
extern "C" int acg_l1_fwd(const float *a, const float *b, size_t npix, int C, int Cp, float *out, void *ws,
                          size_t ws_bytes, void *stream)
{
    return reduce_launch<1>(a, b, (long long)npix * Cp, C, Cp, 0.f, 1.f / ((float)npix * C), out, ws, ws_bytes,
                            (hipStream_t)stream, "acg_l1_fwd");
}
extern "C" int acg_mean_fwd(const float *x, size_t npix, int C, int Cp, float *out, void *ws, size_t ws_bytes,
                            void *stream)
{
    return reduce_launch<2>(x, nullptr, (long long)npix * Cp, C, Cp, 0.f, 1.f / ((float)npix * C), out, ws, ws_bytes,
                            (hipStream_t)stream, "acg_mean_fwd");
}
extern "C" int acg_sumsq(const float *g, size_t n, float *out, void *ws, size_t ws_bytes, void *stream)
{
    return reduce_launch<3>(g, nullptr, (long long)n, 1, 1, 0.f, 1.f, out, ws, ws_bytes, (hipStream_t)stream, "acg_sumsq");
}

__global__ void mse_const_bwd_kernel(const float *__restrict__ p, long long total, int C, int Cp, float target,
                                     float k, const float *__restrict__ gout, float *__restrict__ dp)
{
    const float g = gout[0] * k;
    GRID_STRIDE(i, total) { dp[i] = (int)(i % Cp) < C ? g * (p[i] - target) : 0.f; }
}
extern "C" int acg_mse_const_bwd(const float *p, size_t npix, int C, int Cp, float target, const float *gout, float *dp,
                                 void *stream)
{
    const long long total = (long long)npix * Cp;
    hipLaunchKernelGGL(mse_const_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, p, total, C, Cp,
                       target, 2.f / ((float)npix * C), gout, dp);
    ACG_CHECK_LAUNCH("mse_const_bwd_kernel");
    return ACG_OK;
}
__global__ void l1_bwd_kernel(const float *__restrict__ a, const float *__restrict__ b, long long total, int C, int Cp,
                              float k, const float *__restrict__ gout, float *__restrict__ da, float *__restrict__ db)
{
    const float g = gout[0] * k;
    GRID_STRIDE(i, total) {
        float v = 0.f;
        if ((int)(i % Cp) < C) {
            const float d = a[i] - b[i];
            v = d > 0.f ? g : (d < 0.f ? -g : 0.f); // sign(0) = 0, as torch
        }
        if (da) da[i] = v;
        if (db) db[i] = -v;
    }
}
extern "C" int acg_l1_bwd(const float *a, const float *b, size_t npix, int C, int Cp, const float *gout, float *da,
                          float *db, void *stream)
{
    const long long total = (long long)npix * Cp;
    hipLaunchKernelGGL(l1_bwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, a, b, total, C, Cp,
                       1.f / ((float)npix * C), gout, da, db);
    ACG_CHECK_LAUNCH("l1_bwd_kernel");
    return ACG_OK;
}

// ---------------------------------------------------------------- clip + Adam on a flat buffer
__global__ void adam_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                            long long n, const float *__restrict__ sumsq, float max_norm, float step_size, float beta1,
                            float beta2, float inv_bc2_sqrt, float eps, int scale_grads)
{
#pragma clang fp contract(off) // the same roundings as adam_multi_kernel whatever the compiler would fuse in either
    float coef = 1.f;
    if (sumsq != nullptr) {
        coef = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
        coef = coef < 1.f ? coef : 1.f;
    }
    GRID_STRIDE(i, n) {
        const float gi = g[i] * coef;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_bc2_sqrt + eps);
        if (scale_grads) g[i] = gi;
    }
}
// ---------------------------------------------------------------- the same for all networks of an optimiser phase
// (multi-tensor: one partial-sums launch, one final launch, one Adam launch for up to ACG_ADAM_MAX_GROUPS flat buffers;
// every group keeps the block count, the per-thread order of sums and the arithmetic of acg_sumsq + acg_adam_step, so the
// results are bit-identical to calling those per network)
struct AdamGroups {
    acg_adam_group gr[ACG_ADAM_MAX_GROUPS];
    int nb[ACG_ADAM_MAX_GROUPS], first[ACG_ADAM_MAX_GROUPS + 1]; // blocks of group i: [first[i], first[i] + nb[i])
    int n;
};
__device__ __forceinline__ int adam_group_of(const AdamGroups &G, int block)
{
    int gi = 0;
#pragma unroll
    for (int i = 1; i < ACG_ADAM_MAX_GROUPS; ++i) gi += (i < G.n && block >= G.first[i]) ? 1 : 0;
    return gi;
}
__global__ __launch_bounds__(256) void sumsq_multi_partial_kernel(AdamGroups G, float *__restrict__ part)
{
    __shared__ float red[256];
    const int gi = adam_group_of(G, blockIdx.x), lb = blockIdx.x - G.first[gi], nb = G.nb[gi];
    const float *__restrict__ a = G.gr[gi].g;
    const long long total = (long long)G.gr[gi].n;
    float s = 0.f;
    for (long long i = lb * 256LL + threadIdx.x; i < total; i += nb * 256LL) { const float v = a[i]; s += v * v; }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[gi * RED_BLOCKS + lb] = red[0];
}
__global__ __launch_bounds__(256) void sumsq_multi_final_kernel(AdamGroups G, const float *__restrict__ part)
{
    __shared__ float red[256];
    const int gi = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < G.nb[gi]; i += 256) s += part[gi * RED_BLOCKS + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) G.gr[gi].sumsq[0] = red[0] * 1.f;
}
__global__ void adam_multi_kernel(AdamGroups G, float max_norm, float step_size, float beta1, float beta2, float inv_bc2_sqrt,
                                  float eps, float lr, const int *__restrict__ step_dev)
{
#pragma clang fp contract(off) // bit-identical to adam_kernel (see there)
    const int gi = adam_group_of(G, blockIdx.x), lb = blockIdx.x - G.first[gi], nb = G.nb[gi];
    const acg_adam_group q = G.gr[gi];
    if (step_dev != nullptr) { // the step number lives on the device (a captured graph replays this launch with fixed arguments)
        const int step = step_dev[0] + 1;
        step_size = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
        inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    }
    float coef = max_norm / (sqrtf(q.sumsq[0]) + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
    for (long long i = lb * 256LL + threadIdx.x; i < (long long)q.n; i += nb * 256LL) {
        const float gi_ = q.g[i] * coef;
        const float mi = beta1 * q.m[i] + (1.f - beta1) * gi_;
        const float vi = beta2 * q.v[i] + (1.f - beta2) * gi_ * gi_;
        q.m[i] = mi;
        q.v[i] = vi;
        q.p[i] -= step_size * mi / (sqrtf(vi) * inv_bc2_sqrt + eps);
        q.g[i] = gi_;
    }
}
extern "C" size_t acg_clip_adam_multi_workspace_bytes(int ngroups)
{
    return (size_t)(ngroups > 0 ? ngroups : 0) * RED_BLOCKS * sizeof(float);
}
extern "C" int acg_clip_adam_multi(const acg_adam_group *groups, int ngroups, float max_norm, float lr, float beta1, float beta2,
                                   float eps, int step, const int *step_dev, void *ws, size_t ws_bytes, void *stream)
{
    ACG_REQUIRE(groups != nullptr && ngroups >= 1 && ngroups <= ACG_ADAM_MAX_GROUPS, "acg_clip_adam_multi: 1..%d groups, got %d",
                ACG_ADAM_MAX_GROUPS, ngroups);
    ACG_REQUIRE(step >= 1 || step_dev != nullptr, "acg_clip_adam_multi: step must be >= 1");
    if (step < 1) step = 1;
    if (ws == nullptr || ws_bytes < acg_clip_adam_multi_workspace_bytes(ngroups)) {
        acg_set_error("acg_clip_adam_multi: workspace too small");
        return ACG_ERR_WORKSPACE;
    }
    AdamGroups S, A; // block tables of the sums pass (as acg_sumsq) and of the update pass (as acg_adam_step)
    S.n = A.n = ngroups;
    S.first[0] = A.first[0] = 0;
    for (int i = 0; i < ngroups; ++i) {
        const acg_adam_group &q = groups[i];
        ACG_REQUIRE(q.p && q.g && q.m && q.v && q.sumsq && q.n > 0, "acg_clip_adam_multi: group %d has a null pointer or no elements", i);
        S.gr[i] = A.gr[i] = q;
        const long long nbl = ((long long)q.n + 2047) / 2048;
        S.nb[i] = (int)(nbl > RED_BLOCKS ? RED_BLOCKS : (nbl < 1 ? 1 : nbl));
        A.nb[i] = ew_blocks((long long)q.n);
        S.first[i + 1] = S.first[i] + S.nb[i];
        A.first[i + 1] = A.first[i] + A.nb[i];
    }
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_multi_partial_kernel, dim3(S.first[ngroups]), dim3(256), 0, st, S, (float *)ws);
    hipLaunchKernelGGL(sumsq_multi_final_kernel, dim3(ngroups), dim3(256), 0, st, S, (const float *)ws);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(A.first[ngroups]), dim3(256), 0, st, A, max_norm, (float)(lr / bc1), beta1, beta2,
                       (float)(1.0 / sqrt(bc2)), eps, lr, step_dev);
    ACG_CHECK_LAUNCH("acg_clip_adam_multi");
    return ACG_OK;
}

extern "C" int acg_adam_step(float *p, float *g, float *m, float *v, size_t n, const float *sumsq, float max_norm,
                             float lr, float beta1, float beta2, float eps, int step, int scale_grads, void *stream)
{
    ACG_REQUIRE(step >= 1, "acg_adam_step: step must be >= 1");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks((long long)n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (long long)n, sumsq, max_norm, (float)(lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps,
                       scale_grads);
    ACG_CHECK_LAUNCH("adam_kernel");
    return ACG_OK;
}
