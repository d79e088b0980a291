// Normalisation kernels (HBM-bound): InstanceNorm / CondInstanceNorm / BatchNorm(train) over an
// NHWC tensor viewed as [G groups][P pixels][C channels] (IN/CIN: G=N, P=H*W; BN: G=1, P=N*H*W).
//
// Statistics: one read of x.  Each thread keeps plain sums over <=256 rows, converts them to
// (mean, M2) and all further merging (threads -> block -> chunks) uses Chan's parallel update,
// which is well conditioned — the reference computes mean((x-mean)^2) in two passes
// (modules.py:86-88); this matches it to fp32 rounding without the second read.
// All access is float4 over channels, consecutive lanes on consecutive channels (coalesced).
#include "common.h"

#define EW_UNROLL 4   // independent 16-byte loads per stream and thread in the element-wise passes
#define NORM_ROWS 256 // pixels per block of the reduction passes: 32 images x 64 chunks = 2048 blocks at 128x128
                      // (1024 gave 512 blocks = 2 per CU with one 16-byte load in flight per thread: ~2 TB/s)

__device__ __forceinline__ void chan_merge(float &na, float &ma, float &sa, float nb, float mb, float sb)
{
    if (nb == 0.f) return;
    const float n = na + nb;
    const float d = mb - ma;
    ma += d * (nb / n);
    sa += sb + d * d * (na * nb / n);
    na = n;
}

// partial layout: part[((g*nchunks + chunk)*2 + {0:mean,1:M2})*C + c]; the chunk's row count is implied
__global__ __launch_bounds__(256) void norm_stats_partial(const float *__restrict__ x, long long P, int C,
                                                          int nchunks, float *__restrict__ part)
{
    __shared__ float sm[256 * 4], sq[256 * 4], sn[256];
    const int C4 = C / 4;
    const int rows_par = 256 / C4;
    const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4;
    const int g = blockIdx.y, chunk = blockIdx.x;
    const long long r0 = (long long)chunk * NORM_ROWS;
    long long r1 = r0 + NORM_ROWS;
    if (r1 > P) r1 = P;
    const float *xg = x + (long long)g * P * C;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    float cnt = 0.f;
    if (rl < rows_par)
#pragma unroll 4
        for (long long r = r0 + rl; r < r1; r += rows_par) {
            const f32x4 v = *(const f32x4 *)(xg + r * C + c4 * 4);
            s1 += v;
            s2 += v * v;
            cnt += 1.f;
        }
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
    if (cnt > 0.f) {
        mean = s1 / cnt;
        m2 = s2 - s1 * mean;
#pragma unroll
        for (int k = 0; k < 4; ++k) m2[k] = m2[k] < 0.f ? 0.f : m2[k];
    }
    *(f32x4 *)&sm[threadIdx.x * 4] = mean;
    *(f32x4 *)&sq[threadIdx.x * 4] = m2;
    sn[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x < C4) {
        float n[4], m[4], s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { n[k] = 0.f; m[k] = 0.f; s[k] = 0.f; }
        for (int j = 0; j < rows_par; ++j) {
            const int t = j * C4 + threadIdx.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) chan_merge(n[k], m[k], s[k], sn[t], sm[t * 4 + k], sq[t * 4 + k]);
        }
        float *o = part + ((long long)(g * nchunks + chunk) * 2) * C + threadIdx.x * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[k] = m[k]; o[C + k] = s[k]; }
    }
}

// One workgroup per (group, 32 channels): 8 lanes share a channel and merge every 8th chunk each, then the 8 partial
// (n, mean, M2) triples are merged in fixed order through LDS.  (One thread per channel walking all chunks serially
// took 52 us per call once the chunks shrank to 128-256 pixels: 128 dependent round trips to L2.)
#define FIN_CH 32
#define FIN_KQ 8
__global__ __launch_bounds__(256) void norm_stats_final(const float *__restrict__ part, int G, long long P, int C,
                                                        int nchunks, float eps, int unbiased, float *__restrict__ mean,
                                                        float *__restrict__ rstd, float *__restrict__ run_mean,
                                                        float *__restrict__ run_var, float momentum, int rows_per_chunk)
{
    __shared__ float sn[FIN_KQ][FIN_CH], sm[FIN_KQ][FIN_CH], ss[FIN_KQ][FIN_CH];
    const int cc = threadIdx.x % FIN_CH, kq = threadIdx.x / FIN_CH;
    const int cblocks = (C + FIN_CH - 1) / FIN_CH;
    const int g = blockIdx.x / cblocks, c = (blockIdx.x % cblocks) * FIN_CH + cc;
    float n = 0.f, m = 0.f, s = 0.f;
    if (P == (long long)nchunks * rows_per_chunk) {
        // equal chunks (every convolution-epilogue partial): mean = average of the chunk means, M2 = sum M2_k + n_k (mean_k -
        // mean)^2 — two sweeps of independent loads instead of a chain of pairwise merges with a division each (13.5 -> ~6 us;
        // 100 calls per step)
        float a = 0.f;
        if (c < C) {
#pragma unroll 8
            for (int k = kq; k < nchunks; k += FIN_KQ) a += part[((long long)(g * nchunks + k) * 2) * C + c];
        }
        sm[kq][cc] = a;
        __syncthreads();
        float mu = 0.f;
#pragma unroll
        for (int q = 0; q < FIN_KQ; ++q) mu += sm[q][cc];
        mu *= 1.f / (float)nchunks;
        float b = 0.f;
        if (c < C) {
#pragma unroll 8
            for (int k = kq; k < nchunks; k += FIN_KQ) {
                const float *p = part + ((long long)(g * nchunks + k) * 2) * C + c;
                const float d = p[0] - mu;
                b += p[C] + (float)rows_per_chunk * d * d;
            }
        }
        ss[kq][cc] = b;
        __syncthreads();
        if (kq != 0 || c >= C) return;
        s = 0.f;
#pragma unroll
        for (int q = 0; q < FIN_KQ; ++q) s += ss[q][cc];
        const int i = g * C + c;
        const float denom = unbiased ? (float)(P - 1) : (float)P;
        mean[i] = mu;
        rstd[i] = rsqrtf(s / denom + eps);
        if (run_mean != nullptr) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * (s / (float)(P > 1 ? P - 1 : 1));
        }
        return;
    }
    if (c < C) {
#pragma unroll 4
        for (int k = kq; k < nchunks; k += FIN_KQ) {
            long long rows = P - (long long)k * rows_per_chunk;
            if (rows > rows_per_chunk) rows = rows_per_chunk;
            const float *p = part + ((long long)(g * nchunks + k) * 2) * C + c;
            chan_merge(n, m, s, (float)rows, p[0], p[C]);
        }
    }
    sn[kq][cc] = n; sm[kq][cc] = m; ss[kq][cc] = s;
    __syncthreads();
    if (kq != 0 || c >= C) return;
    n = 0.f; m = 0.f; s = 0.f;
#pragma unroll
    for (int q = 0; q < FIN_KQ; ++q) chan_merge(n, m, s, sn[q][cc], sm[q][cc], ss[q][cc]);
    const int i = g * C + c;
    const float denom = unbiased ? (float)(P - 1) : (float)P;
    mean[i] = m;
    rstd[i] = rsqrtf(s / denom + eps);
    if (run_mean != nullptr) { // BatchNorm (G == 1): running_var takes the unbiased estimate
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * m;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (s / (float)(P > 1 ? P - 1 : 1));
    }
}

// Pre-split ("S16", conv_internal.h) tensors in the float4-per-thread element-wise kernels: the 8-channel group of a
// lane PAIR occupies 32 contiguous bytes, 16 of hi then 16 of lo.  Lane 2k moves the hi half and lane 2k+1 the lo half as one
// 16-byte access each (fully contiguous across the wave), and the pair swaps 8 bytes through DPP so that each lane ends up
// with (stores: starts from) hi and lo of its own four channels.  `o` is the lane's float index (a multiple of 4, parity
// of o/4 == parity of the lane).
typedef unsigned norm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 s16_load4(const float *__restrict__ t, long long o)
{
    const int odd = (int)(o >> 2) & 1;
    const norm_u32x4 own = *(const norm_u32x4 *)((const char *)t + 4 * (o & ~7LL) + 16 * odd);   // even lane: hi[0..7], odd lane: lo[0..7]
    const unsigned s0 = odd ? own[0] : own[2], s1 = odd ? own[1] : own[3];                      // what the partner needs
    const unsigned r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
    const unsigned h0 = odd ? r0 : own[0], h1 = odd ? r1 : own[1], l0 = odd ? own[2] : r0, l1 = odd ? own[3] : r1;
    f32x4 v;
    v[0] = __builtin_bit_cast(float, h0 << 16) + __builtin_bit_cast(float, l0 << 16);
    v[1] = __builtin_bit_cast(float, h0 & 0xffff0000u) + __builtin_bit_cast(float, l0 & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, h1 << 16) + __builtin_bit_cast(float, l1 << 16);
    v[3] = __builtin_bit_cast(float, h1 & 0xffff0000u) + __builtin_bit_cast(float, l1 & 0xffff0000u);
    return v;
}
// every lane of a pair must call this (the exchange), `store` says whether the lane's own element is in range
__device__ __forceinline__ void s16_store4(float *__restrict__ t, long long o, const f32x4 v, bool store)
{
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    unsigned hi[2], lo[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { // the arithmetic of acg_split8
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2 * q], v[2 * q + 1]}, bf16x2_t));
        const float ha = __builtin_bit_cast(float, h << 16), hb = __builtin_bit_cast(float, h & 0xffff0000u);
        hi[q] = h;
        lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2 * q] - ha, v[2 * q + 1] - hb}, bf16x2_t));
    }
    const int odd = (int)(o >> 2) & 1;
    const unsigned s0 = odd ? hi[0] : lo[0], s1 = odd ? hi[1] : lo[1];
    const unsigned r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
    const norm_u32x4 w = odd ? (norm_u32x4){r0, r1, lo[0], lo[1]} : (norm_u32x4){hi[0], hi[1], r0, r1};
    if (store) *(norm_u32x4 *)((char *)t + 4 * (o & ~7LL) + 16 * odd) = w;
}

// ACT / HAS_RES are template parameters: with run-time switches the body is a web of scalar branches with
// s_waitcnt vmcnt(0) between them and the loads of a thread are issued one at a time (2.7-3.6 TB/s); specialised, all
// loads of a thread go out back to back like a streaming copy.
//
// MASK: also store one bit per output element, (y > 0) — all the backward pass needs of y when a residual was added
// before the activation (without a residual it recomputes the sign from x).  Bit e%32 of word e/32 for the element at
// float index e; a thread's float4 is one nibble, 8 consecutive lanes own one word (OR-butterfly over lanes ^1, ^2, ^4).
// The backward passes then read 1/32 of a tensor instead of y itself: -2 of the 8 tensor streams of a block-output norm.
// FMT bit 0: `res` is pre-split (S16), bit 1: y is written pre-split (both need C % 8 == 0 and whole lane pairs in range)
template <int ACT, bool HAS_RES, bool MASK, int FMT = 0>
__global__ __launch_bounds__(256) void norm_apply_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                         const float *__restrict__ rstd,
                                                         const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, int gstride,
                                                         const float *__restrict__ res, float *__restrict__ y,
                                                         long long P, int C, unsigned *__restrict__ mask)
{
    const int C4 = C / 4;
    const int g = blockIdx.y;
    const long long total = P * C4;
    const long long base = (long long)g * P * C;
    // one contiguous run of EW_UNROLL x 256 float4 per workgroup, a thread's elements 256 float4 apart
    const long long i0 = (long long)blockIdx.x * (256 * EW_UNROLL) + threadIdx.x;
    const bool inv = 256 % C4 == 0; // then the thread's channel quad is the same for all its elements
    f32x4 v[EW_UNROLL], rv[EW_UNROLL];
    bool ok[EW_UNROLL];
#pragma unroll
    for (int u = 0; u < EW_UNROLL; ++u) {
        ok[u] = i0 + u * 256 < total;
        const long long o = base + (ok[u] ? i0 + u * 256 : 0) * 4;
        v[u] = *(const f32x4 *)(x + o);
        if (HAS_RES) rv[u] = (FMT & 1) ? s16_load4(res, o) : *(const f32x4 *)(res + o);
    }
    int c = (int)(i0 % C4) * 4;
    f32x4 mu = *(const f32x4 *)(mean + g * C + c), rs = *(const f32x4 *)(rstd + g * C + c);
    f32x4 ga = *(const f32x4 *)(gamma + g * gstride + c), be = *(const f32x4 *)(beta + g * gstride + c);
#pragma unroll
    for (int u = 0; u < EW_UNROLL; ++u) {
        if (!inv && u > 0) {
            c = (int)((i0 + u * 256) % C4) * 4;
            mu = *(const f32x4 *)(mean + g * C + c); rs = *(const f32x4 *)(rstd + g * C + c);
            ga = *(const f32x4 *)(gamma + g * gstride + c); be = *(const f32x4 *)(beta + g * gstride + c);
        }
        f32x4 o = (v[u] - mu) * rs * ga + be;
        if (HAS_RES) o += rv[u];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = acg_apply_act(o[k], ACT);
        if (FMT & 2) s16_store4(y, base + (i0 + u * 256) * 4, o, ok[u]);
        else if (ok[u]) *(f32x4 *)(y + base + (i0 + u * 256) * 4) = o;
        if (MASK) {
            const long long f = base / 4 + i0 + u * 256;   // float4 index; (f & 7) == (lane & 7): the launcher checks P*C/4 % 8 == 0
            unsigned v = ok[u] ? ((o[0] > 0.f ? 1u : 0u) | (o[1] > 0.f ? 2u : 0u) | (o[2] > 0.f ? 4u : 0u) | (o[3] > 0.f ? 8u : 0u)) : 0u;
            v <<= 4 * (threadIdx.x & 7);
            v |= __shfl_xor(v, 1);
            v |= __shfl_xor(v, 2);
            v |= __shfl_xor(v, 4);
            if ((threadIdx.x & 7) == 0 && ok[u]) mask[f >> 3] = v;
        }
    }
}

// the nibble of the float4 at float4-index f
__device__ __forceinline__ unsigned norm_mask_nibble(const unsigned *__restrict__ mask, long long f)
{
    return (mask[f >> 3] >> (4 * (int)(f & 7))) & 15u;
}

// backward pass 1: per (group, chunk) partial sums of gy and gy*xhat, gy = dy*act'(y)
// part[((g*nchunks + chunk)*2 + {0:S1,1:S2})*C + c]
// y == nullptr: the activation mask is recomputed from x (pre = xhat*gamma + beta) instead of read — valid when no
// residual was added before the activation; saves one tensor stream per pass.
// mask != nullptr: the sign bits norm_apply_kernel<.., MASK> stored replace y
template <int ACT, int MSRC> // activation mask: MSRC 0 = read y, 1 = recompute from x (y == nullptr), 2 = sign bitmask
__global__ __launch_bounds__(256) void norm_bwd_partial(const float *__restrict__ dy, const float *__restrict__ y,
                                                        const float *__restrict__ x, const float *__restrict__ mean,
                                                        const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, int gstride, long long P, int C,
                                                        int nchunks, float *__restrict__ part,
                                                        const unsigned *__restrict__ mask)
{
    __shared__ float sa[256 * 4], sb[256 * 4];
    const int C4 = C / 4;
    const int rows_par = 256 / C4;
    const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4;
    const int g = blockIdx.y, chunk = blockIdx.x;
    const long long r0 = (long long)chunk * NORM_ROWS;
    long long r1 = r0 + NORM_ROWS;
    if (r1 > P) r1 = P;
    const long long base = (long long)g * P * C;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (rl < rows_par) {
        const f32x4 mu = *(const f32x4 *)(mean + g * C + c4 * 4);
        const f32x4 rs = *(const f32x4 *)(rstd + g * C + c4 * 4);
        f32x4 ga = {0.f, 0.f, 0.f, 0.f}, be = ga;
        if (ACT != ACG_ACT_NONE && MSRC == 1) {
            ga = *(const f32x4 *)(gamma + g * gstride + c4 * 4);
            be = *(const f32x4 *)(beta + g * gstride + c4 * 4);
        }
        // EW_UNROLL rows per trip, every load issued before the first use (with the activation and its mask source as run-time
        // branches the loop was not unrolled and drained its two loads every row); rows are still summed in ascending order
        auto masked = [&](f32x4 gy, const f32x4 &xv, const f32x4 &yv) {
            if (ACT != ACG_ACT_NONE) {
                const f32x4 yy = MSRC == 1 ? (xv - mu) * rs * ga + be : yv; // same expression as norm_apply_kernel (sign is all we need)
#pragma unroll
                for (int k = 0; k < 4; ++k) gy[k] *= acg_act_grad_from_y(yy[k], ACT);
            }
            return gy;
        };
        auto side = [&](long long o) {
            f32x4 yv = {0.f, 0.f, 0.f, 0.f};
            if (ACT != ACG_ACT_NONE && MSRC == 0) yv = *(const f32x4 *)(y + o);
            if (ACT != ACG_ACT_NONE && MSRC == 2) {
                const unsigned nb = norm_mask_nibble(mask, o >> 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) yv[k] = (nb >> k) & 1u ? 1.f : -1.f;
            }
            return yv;
        };
        long long r = r0 + rl;
        for (; r + (long long)rows_par * (EW_UNROLL - 1) < r1; r += (long long)rows_par * EW_UNROLL) {
            f32x4 gv[EW_UNROLL], xv[EW_UNROLL], yv[EW_UNROLL];
#pragma unroll
            for (int u = 0; u < EW_UNROLL; ++u) {
                const long long o = base + (r + (long long)u * rows_par) * C + c4 * 4;
                gv[u] = *(const f32x4 *)(dy + o);
                xv[u] = *(const f32x4 *)(x + o);
                yv[u] = side(o);
            }
#pragma unroll
            for (int u = 0; u < EW_UNROLL; ++u) {
                const f32x4 gy = masked(gv[u], xv[u], yv[u]);
                s1 += gy;
                s2 += gy * ((xv[u] - mu) * rs);
            }
        }
        for (; r < r1; r += rows_par) {
            const long long o = base + r * C + c4 * 4;
            const f32x4 xv = *(const f32x4 *)(x + o);
            const f32x4 gy = masked(*(const f32x4 *)(dy + o), xv, side(o));
            s1 += gy;
            s2 += gy * ((xv - mu) * rs);
        }
    }
    *(f32x4 *)&sa[threadIdx.x * 4] = s1;
    *(f32x4 *)&sb[threadIdx.x * 4] = s2;
    __syncthreads();
    if (threadIdx.x < C4) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < rows_par; ++j) {
            a += *(const f32x4 *)&sa[(j * C4 + threadIdx.x) * 4];
            b += *(const f32x4 *)&sb[(j * C4 + threadIdx.x) * 4];
        }
        float *o = part + ((long long)(g * nchunks + chunk) * 2) * C + threadIdx.x * 4;
        *(f32x4 *)o = a;
        *(f32x4 *)(o + C) = b;
    }
}

// sums[(g*2 + {0,1})*C + c] = sum over chunks (fixed order)
// dgamma_gc / dbeta_gc (CondInstanceNorm: per-sample affine, gstride == C): the sums ARE the parameter gradients
__global__ __launch_bounds__(256) void norm_bwd_final(const float *__restrict__ part, int G, int C, int nchunks,
                                                      float *__restrict__ sums, float *__restrict__ dgamma_gc,
                                                      float *__restrict__ dbeta_gc)
{
    __shared__ float sa[FIN_KQ][FIN_CH], sb[FIN_KQ][FIN_CH];
    const int cc = threadIdx.x % FIN_CH, kq = threadIdx.x / FIN_CH;
    const int cblocks = (C + FIN_CH - 1) / FIN_CH;
    const int g = blockIdx.x / cblocks, c = (blockIdx.x % cblocks) * FIN_CH + cc;
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int k = kq; k < nchunks; k += FIN_KQ) {
            const float *p = part + ((long long)(g * nchunks + k) * 2) * C + c;
            a += p[0];
            b += p[C];
        }
    }
    sa[kq][cc] = a; sb[kq][cc] = b;
    __syncthreads();
    if (kq != 0 || c >= C) return;
    a = 0.f; b = 0.f;
#pragma unroll
    for (int q = 0; q < FIN_KQ; ++q) { a += sa[q][cc]; b += sb[q][cc]; }
    sums[(g * 2) * C + c] = a;
    sums[(g * 2 + 1) * C + c] = b;
    if (dbeta_gc) dbeta_gc[g * C + c] = a;
    if (dgamma_gc) dgamma_gc[g * C + c] = b;
}

// parameter gradients of a shared affine (InstanceNorm / BatchNorm, gstride == 0): dgamma[c] = sum_g S2, dbeta[c] = sum_g S1
// for the first `nparam` (real, unpadded) channels; accumulate != 0 adds to the destination (the parameter's .grad)
__global__ void norm_bwd_params(const float *__restrict__ sums, int G, int C, int nparam, int accumulate,
                                float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nparam) return;
    float a = 0.f, b = 0.f;
    for (int g = 0; g < G; ++g) { a += sums[(g * 2) * C + i]; b += sums[(g * 2 + 1) * C + i]; }
    if (dbeta) dbeta[i] = (accumulate ? dbeta[i] : 0.f) + a;
    if (dgamma) dgamma[i] = (accumulate ? dgamma[i] : 0.f) + b;
}

// backward pass 2: dx = gamma*rstd*(gy - S1/P - xhat*S2/D) ; dres = gy
// DXS16: dx is written pre-split (S16)
template <int ACT, int MSRC, bool HAS_DRES, bool DXS16 = false> // activation mask: MSRC 0 = read y, 1 = recompute from x (y == nullptr), 2 = sign bitmask
__global__ __launch_bounds__(256) void norm_bwd_apply(const float *__restrict__ dy, const float *__restrict__ y,
                                                      const float *__restrict__ x, const float *__restrict__ mean,
                                                      const float *__restrict__ rstd,
                                                      const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, int gstride,
                                                      const float *__restrict__ sums, float *__restrict__ dx,
                                                      float *__restrict__ dres, long long P, int C, float invP,
                                                      float invD, const unsigned *__restrict__ mask, int pG, int nparam,
                                                      int accumulate, float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    constexpr bool RECOMPUTE = MSRC == 1;
    // shared affine parameters (InstanceNorm / BatchNorm): dgamma[c] = sum_g S2, dbeta[c] = sum_g S1 for the first nparam
    // (real) channels, by the first workgroup on the side — it used to be a launch of its own (norm_bwd_params: one
    // workgroup, 9 us, 72 times per training step)
    if (nparam > 0 && blockIdx.x == 0 && blockIdx.y == 0) {
        for (int i = threadIdx.x; i < nparam; i += 256) {
            float a = 0.f, b = 0.f;
            for (int gg = 0; gg < pG; ++gg) { a += sums[(gg * 2) * C + i]; b += sums[(gg * 2 + 1) * C + i]; }
            if (dbeta) dbeta[i] = (accumulate ? dbeta[i] : 0.f) + a;
            if (dgamma) dgamma[i] = (accumulate ? dgamma[i] : 0.f) + b;
        }
    }
    const int C4 = C / 4;
    const int g = blockIdx.y;
    const long long total = P * C4;
    const long long base = (long long)g * P * C;
    const long long i0 = (long long)blockIdx.x * (256 * EW_UNROLL) + threadIdx.x;
    const bool inv = 256 % C4 == 0;
    f32x4 gv[EW_UNROLL], xv[EW_UNROLL], yv[EW_UNROLL];
    bool ok[EW_UNROLL];
#pragma unroll
    for (int u = 0; u < EW_UNROLL; ++u) {
        ok[u] = i0 + u * 256 < total;
        const long long o = base + (ok[u] ? i0 + u * 256 : 0) * 4;
        gv[u] = *(const f32x4 *)(dy + o);
        xv[u] = *(const f32x4 *)(x + o);
        if (ACT != ACG_ACT_NONE && MSRC == 0) yv[u] = *(const f32x4 *)(y + o);
        if (ACT != ACG_ACT_NONE && MSRC == 2) {
            const unsigned nb = norm_mask_nibble(mask, o >> 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) yv[u][k] = (nb >> k) & 1u ? 1.f : -1.f;
        }
    }
    int c = (int)(i0 % C4) * 4;
    f32x4 mu, rs, ga, be = {0.f, 0.f, 0.f, 0.f}, s1, s2;
    auto params = [&]() {
        mu = *(const f32x4 *)(mean + g * C + c);
        rs = *(const f32x4 *)(rstd + g * C + c);
        ga = *(const f32x4 *)(gamma + g * gstride + c);
        if (ACT != ACG_ACT_NONE && RECOMPUTE) be = *(const f32x4 *)(beta + g * gstride + c);
        s1 = *(const f32x4 *)(sums + (g * 2) * C + c) * invP;
        s2 = *(const f32x4 *)(sums + (g * 2 + 1) * C + c) * invD;
    };
    params();
#pragma unroll
    for (int u = 0; u < EW_UNROLL; ++u) {
        if (!inv && u > 0) {
            c = (int)((i0 + u * 256) % C4) * 4;
            params();
        }
        f32x4 gy = gv[u];
        const f32x4 xh = (xv[u] - mu) * rs;
        if (ACT != ACG_ACT_NONE) {
            const f32x4 yy = RECOMPUTE ? (xv[u] - mu) * rs * ga + be : yv[u]; // same expression as norm_apply_kernel
#pragma unroll
            for (int k = 0; k < 4; ++k) gy[k] *= acg_act_grad_from_y(yy[k], ACT);
        }
        {
            const long long o = base + (i0 + u * 256) * 4;
            const f32x4 d = ga * rs * (gy - s1 - xh * s2);
            if (DXS16) s16_store4(dx, o, d, ok[u]);
            else if (ok[u]) *(f32x4 *)(dx + o) = d;
            if (HAS_DRES && ok[u]) *(f32x4 *)(dres + o) = gy;
        }
    }
}

__global__ __launch_bounds__(256) void mask_apply_kernel(const float *__restrict__ x, const unsigned *__restrict__ mask,
                                                         float *__restrict__ out, long long n4, float scale)
{
    for (long long f = blockIdx.x * (long long)blockDim.x + threadIdx.x; f < n4; f += (long long)gridDim.x * blockDim.x) {
        f32x4 v = *(const f32x4 *)(x + f * 4);
        const unsigned nb = norm_mask_nibble(mask, f);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (nb >> k) & 1u ? v[k] * scale : 0.f;   // scale 1: exact
        *(f32x4 *)(out + f * 4) = v;
    }
}
extern "C" int acg_mask_apply(const float *x, const unsigned *sign_mask, float *out, size_t n, void *stream)
{
    ACG_REQUIRE(x != nullptr && sign_mask != nullptr && out != nullptr && n % 4 == 0, "acg_mask_apply: bad arguments");
    const long long n4 = (long long)(n / 4);
    const int blocks = acg_cdiv(n4, 256) > 8192 ? 8192 : acg_cdiv(n4, 256);
    hipLaunchKernelGGL(mask_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, sign_mask, out, n4, 1.f);
    ACG_CHECK_LAUNCH("mask_apply_kernel");
    return ACG_OK;
}
// nn.Dropout(p) in training mode with the keep draw GIVEN as a bitmask (bit e % 32 of word e / 32 for the float at index e):
// out = keep ? x * scale : 0 with scale = 1 / (1 - p); the backward is the same map on the gradient
extern "C" int acg_dropout_apply(const float *x, const unsigned *keep_bits, float scale, float *out, size_t n, void *stream)
{
    ACG_REQUIRE(x != nullptr && keep_bits != nullptr && out != nullptr && n % 4 == 0 && scale > 0.f, "acg_dropout_apply: bad arguments");
    const long long n4 = (long long)(n / 4);
    const int blocks = acg_cdiv(n4, 256) > 8192 ? 8192 : acg_cdiv(n4, 256);
    hipLaunchKernelGGL(mask_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, keep_bits, out, n4, scale);
    ACG_CHECK_LAUNCH("mask_apply_kernel(dropout)");
    return ACG_OK;
}

// run-time (act, flags) -> template instance
#define NORM_ACT_SWITCH(act, M)                                                  \
    switch (act) {                                                               \
    case ACG_ACT_RELU: M(ACG_ACT_RELU); break;                                   \
    case ACG_ACT_LRELU: M(ACG_ACT_LRELU); break;                                 \
    case ACG_ACT_TANH: M(ACG_ACT_TANH); break;                                   \
    default: M(ACG_ACT_NONE); break;                                             \
    }

static void launch_norm_apply(dim3 grid, hipStream_t st, const float *x, const float *mean, const float *rstd,
                              const float *gamma, const float *beta, int gstride, const float *res, float *y, long long P,
                              int C, int act, unsigned *mask, int fmt)
{
#define LA(A, R, K, F) hipLaunchKernelGGL((norm_apply_kernel<A, R, K, F>), grid, dim3(256), 0, st, x, mean, rstd, gamma, beta, gstride, res, y, P, C, mask)
    if (fmt) { // pre-split I/O: the combinations the residual trunk uses (ReLU; residual + bitmask, or neither)
        if (res && mask && fmt == 3) LA(ACG_ACT_RELU, true, true, 3);
        else if (res && fmt == 3) LA(ACG_ACT_RELU, true, false, 3);
        else if (res && mask && fmt == 1) LA(ACG_ACT_RELU, true, true, 1);   // the LAST block of a trunk: residual pre-split, y fp32
        else if (res && fmt == 1) LA(ACG_ACT_RELU, true, false, 1);
        else LA(ACG_ACT_RELU, false, false, 2);
        return;
    }
#define M(A)                                                                                                              \
    do {                                                                                                                  \
        if (res && mask) hipLaunchKernelGGL((norm_apply_kernel<A, true, true>), grid, dim3(256), 0, st, x, mean, rstd, gamma, beta, gstride, res, y, P, C, mask); \
        else if (res) hipLaunchKernelGGL((norm_apply_kernel<A, true, false>), grid, dim3(256), 0, st, x, mean, rstd, gamma, beta, gstride, res, y, P, C, mask); \
        else hipLaunchKernelGGL((norm_apply_kernel<A, false, false>), grid, dim3(256), 0, st, x, mean, rstd, gamma, beta, gstride, res, y, P, C, mask);   \
    } while (0)
    NORM_ACT_SWITCH(act, M)
#undef M
#undef LA
}

static void launch_norm_bwd_partial(dim3 grid, hipStream_t st, const float *dy, const float *y, const float *x, const float *mean,
                                    const float *rstd, const float *gamma, const float *beta, int gstride, long long P, int C,
                                    int nch, int act, float *part, const unsigned *mask)
{
#define L(A, R) hipLaunchKernelGGL((norm_bwd_partial<A, R>), grid, dim3(256), 0, st, dy, y, x, mean, rstd, gamma, beta, gstride, P, C, nch, part, mask)
#define M(A)                                                                         \
    do {                                                                             \
        if (mask) L(A, 2);                                                           \
        else if (y == nullptr) L(A, 1);                                              \
        else L(A, 0);                                                                \
    } while (0)
    NORM_ACT_SWITCH(act, M)
#undef M
#undef L
}

static void launch_norm_bwd_apply(dim3 grid, hipStream_t st, const float *dy, const float *y, const float *x,
                                  const float *mean, const float *rstd, const float *gamma, const float *beta, int gstride,
                                  const float *sums, float *dx, float *dres, long long P, int C, int act, float invP,
                                  float invD, const unsigned *mask, int dx_s16 = 0, int pG = 0, int nparam = 0, int accumulate = 0,
                                  float *dgamma = nullptr, float *dbeta = nullptr)
{
#define LS(A, R) hipLaunchKernelGGL((norm_bwd_apply<A, R, false, true>), grid, dim3(256), 0, st, dy, y, x, mean, rstd, gamma, beta, gstride, sums, dx, dres, P, C, invP, invD, mask, pG, nparam, accumulate, dgamma, dbeta)
    if (dx_s16) { // pre-split dx: ReLU with the sign bitmask (block-output norm) or recomputed from x (CondInstanceNorm)
        if (mask) LS(ACG_ACT_RELU, 2);
        else LS(ACG_ACT_RELU, 1);
        return;
    }
#define L(A, R, D) hipLaunchKernelGGL((norm_bwd_apply<A, R, D>), grid, dim3(256), 0, st, dy, y, x, mean, rstd, gamma, beta, gstride, sums, dx, dres, P, C, invP, invD, mask, pG, nparam, accumulate, dgamma, dbeta)
#define M(A)                                                                         \
    do {                                                                             \
        const bool rc = y == nullptr && mask == nullptr;                             \
        if (mask && dres) L(A, 2, true);                                             \
        else if (mask) L(A, 2, false);                                               \
        else if (rc && dres) L(A, 1, true);                                          \
        else if (rc) L(A, 1, false);                                                 \
        else if (dres) L(A, 0, true);                                                \
        else L(A, 0, false);                                                         \
    } while (0)
    NORM_ACT_SWITCH(act, M)
#undef M
#undef L
#undef LS
}

// BatchNorm eval mode: statistics come from the running buffers
__global__ void bn_eval_stats_kernel(const float *__restrict__ rm, const float *__restrict__ rv, int C, int Cp, float eps,
                                     float *__restrict__ mean, float *__restrict__ rstd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cp) return;
    mean[c] = c < C ? rm[c] : 0.f;
    rstd[c] = rsqrtf((c < C ? rv[c] : 0.f) + eps);
}
extern "C" int acg_bn_eval_stats(const float *run_mean, const float *run_var, int C, int Cp, float eps, float *mean,
                                 float *rstd, void *stream)
{
    ACG_REQUIRE(C > 0 && Cp >= C, "acg_bn_eval_stats: bad dims");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(acg_cdiv(Cp, 256)), dim3(256), 0, (hipStream_t)stream, run_mean, run_var, C,
                       Cp, eps, mean, rstd);
    ACG_CHECK_LAUNCH("bn_eval_stats_kernel");
    return ACG_OK;
}

static int nchunks_of(size_t P) { return (int)((P + NORM_ROWS - 1) / NORM_ROWS); }

extern "C" size_t acg_norm_workspace_bytes(int G, size_t P, int C)
{
    return ((size_t)G * nchunks_of(P) * 2 * C + (size_t)G * 2 * C) * sizeof(float);
}

static int check_norm(int G, size_t P, int C, const char *who)
{
    ACG_REQUIRE(G > 0 && P > 0 && C > 0 && C % 4 == 0 && C <= 1024, "%s: bad shape G=%d P=%zu C=%d", who, G, P, C);
    ACG_REQUIRE(G <= 65535, "%s: G=%d exceeds grid.y", who, G);
    return ACG_OK;
}

extern "C" int acg_norm_stats(const float *x, int G, size_t P, int C, float eps, int unbiased, float *mean,
                              float *rstd, float *run_mean, float *run_var, float momentum, void *ws, size_t ws_bytes,
                              void *stream)
{
    int rc = check_norm(G, P, C, "acg_norm_stats");
    if (rc) return rc;
    ACG_REQUIRE(!(unbiased && P < 2), "acg_norm_stats: unbiased variance needs P >= 2");
    ACG_REQUIRE(run_mean == nullptr || G == 1, "acg_norm_stats: running stats need G == 1");
    if (ws == nullptr || ws_bytes < acg_norm_workspace_bytes(G, P, C)) {
        acg_set_error("acg_norm_stats: workspace too small");
        return ACG_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = nchunks_of(P);
    hipLaunchKernelGGL(norm_stats_partial, dim3(nch, G), dim3(256), 0, st, x, (long long)P, C, nch, (float *)ws);
    hipLaunchKernelGGL(norm_stats_final, dim3(G * acg_cdiv(C, FIN_CH)), dim3(256), 0, st, (const float *)ws, G,
                       (long long)P, C, nch, eps, unbiased, mean, rstd, run_mean, run_var, momentum, NORM_ROWS);
    ACG_CHECK_LAUNCH("norm_stats");
    return ACG_OK;
}

// Statistics from per-chunk (mean, M2) partials that a producer already holds — acg_conv2d_fwd_stats writes them from
// its epilogue, one chunk per 128-pixel output tile — merged with the same Chan formula: the x tensor is not re-read.
// part[((g*nchunks + chunk)*2 + {0:mean,1:M2})*C + c], nchunks = ceil(P / rows_per_chunk).
extern "C" int acg_norm_stats_from_partials(const float *part, int G, size_t P, int C, int rows_per_chunk, float eps,
                                            int unbiased, float *mean, float *rstd, void *stream)
{
    int rc = check_norm(G, P, C, "acg_norm_stats_from_partials");
    if (rc) return rc;
    ACG_REQUIRE(rows_per_chunk > 0 && part != nullptr, "acg_norm_stats_from_partials: bad partials");
    ACG_REQUIRE(!(unbiased && P < 2), "acg_norm_stats_from_partials: unbiased variance needs P >= 2");
    const int nch = (int)((P + rows_per_chunk - 1) / rows_per_chunk);
    hipLaunchKernelGGL(norm_stats_final, dim3(G * acg_cdiv(C, FIN_CH)), dim3(256), 0, (hipStream_t)stream, part, G,
                       (long long)P, C, nch, eps, unbiased, mean, rstd, (float *)nullptr, (float *)nullptr, 0.f, rows_per_chunk);
    ACG_CHECK_LAUNCH("norm_stats_final");
    return ACG_OK;
}

static int ew_blocks(long long total)
{
    // exactly one run of EW_UNROLL x 256 float4 per workgroup (the generic loops are grid-stride and take any grid)
    const long long b = (total + 256 * EW_UNROLL - 1) / (256 * EW_UNROLL);
    return (int)(b < 1 ? 1 : b);
}

extern "C" int acg_norm_apply(const float *x, const float *mean, const float *rstd, const float *gamma,
                              const float *beta, int gstride, const float *res, float *y, unsigned *mask, int G, size_t P,
                              int C, int act, int fmt, void *stream)
{
    int rc = check_norm(G, P, C, "acg_norm_apply");
    if (rc) return rc;
    ACG_REQUIRE(fmt == 0 || ((fmt == 2 || ((fmt == 3 || fmt == 1) && res != nullptr)) && act == ACG_ACT_RELU && C % 8 == 0 && (mask == nullptr || res != nullptr)),
                "acg_norm_apply: pre-split I/O (fmt %d) is implemented for ReLU: y pre-split (2), y and residual (3), residual only (1)", fmt);
    ACG_REQUIRE(gstride == 0 || (gstride >= C && gstride % 4 == 0), "acg_norm_apply: gstride must be 0 or a row stride >= C");
    ACG_REQUIRE(mask == nullptr || (res != nullptr && ((long long)P * (C / 4)) % 8 == 0 && (act == ACG_ACT_RELU || act == ACG_ACT_LRELU)),
                "acg_norm_apply: the sign bitmask needs a residual, ReLU / LeakyReLU and P*C/4 %% 8 == 0");
    launch_norm_apply(dim3(ew_blocks((long long)P * (C / 4)), G), (hipStream_t)stream, x, mean, rstd, gamma, beta, gstride, res,
                      y, (long long)P, C, act, mask, fmt);
    ACG_CHECK_LAUNCH("norm_apply_kernel");
    return ACG_OK;
}

// local reduction only: sums[(g*2+{0,1})*C + c] = sum_p gy, sum_p gy*xhat (what SyncBN all-reduces across ranks)
extern "C" int acg_norm_bwd_sums(const float *dy, const float *y, const float *x, const float *mean, const float *rstd,
                                 float *sums, int G, size_t P, int C, int act, void *ws, size_t ws_bytes, void *stream)
// (SyncBN path: always reads y)
{
    int rc = check_norm(G, P, C, "acg_norm_bwd_sums");
    if (rc) return rc;
    if (ws == nullptr || ws_bytes < acg_norm_workspace_bytes(G, P, C)) {
        acg_set_error("acg_norm_bwd_sums: workspace too small");
        return ACG_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = nchunks_of(P);
    ACG_REQUIRE(act == ACG_ACT_NONE || y != nullptr, "acg_norm_bwd_sums: y required");
    launch_norm_bwd_partial(dim3(nch, G), st, dy, y, x, mean, rstd, (const float *)nullptr, (const float *)nullptr, 0,
                            (long long)P, C, nch, act, (float *)ws, (const unsigned *)nullptr);
    hipLaunchKernelGGL(norm_bwd_final, dim3(G * acg_cdiv(C, FIN_CH)), dim3(256), 0, st, (const float *)ws, G, C, nch, sums,
                       (float *)nullptr, (float *)nullptr);
    ACG_CHECK_LAUNCH("norm_bwd_sums");
    return ACG_OK;
}

// apply with externally supplied (e.g. all-reduced) sums and the matching total pixel count Ptot
extern "C" int acg_norm_bwd_apply(const float *dy, const float *y, const float *x, const float *mean, const float *rstd,
                                  const float *gamma, int gstride, const float *sums, float *dx, float *dres, int G,
                                  size_t P, size_t Ptot, int C, int act, int unbiased, void *stream)
{
    int rc = check_norm(G, P, C, "acg_norm_bwd_apply");
    if (rc) return rc;
    const float invP = unbiased == 2 ? 0.f : 1.f / (float)Ptot;
    const float invD = unbiased == 2 ? 0.f : (unbiased ? 1.f / (float)(Ptot - 1) : invP);
    ACG_REQUIRE(act == ACG_ACT_NONE || y != nullptr, "acg_norm_bwd_apply: y required");
    launch_norm_bwd_apply(dim3(ew_blocks((long long)P * (C / 4)), G), (hipStream_t)stream, dy, y, x, mean, rstd, gamma,
                          (const float *)nullptr, gstride, sums, dx, dres, (long long)P, C, act, invP, invD, (const unsigned *)nullptr);
    ACG_CHECK_LAUNCH("norm_bwd_apply");
    return ACG_OK;
}

static int norm_bwd_impl(const float *dy, const float *y, const unsigned *mask, const float *x, const float *mean,
                         const float *rstd, const float *gamma, const float *beta, int gstride, float *dx, float *dres,
                         float *dgamma, float *dbeta, int nparam, int accumulate, int G, size_t P, int C, int act, int unbiased,
                         int dx_s16, const float *part_in, int nch_in, void *ws, size_t ws_bytes, void *stream);

extern "C" int acg_norm_bwd(const float *dy, const float *y, const unsigned *mask, const float *x, const float *mean,
                            const float *rstd,
                            const float *gamma, const float *beta, int gstride, float *dx, float *dres, float *dgamma,
                            float *dbeta, int nparam, int accumulate, int G, size_t P, int C, int act, int unbiased,
                            int dx_s16, void *ws, size_t ws_bytes, void *stream)
{
    return norm_bwd_impl(dy, y, mask, x, mean, rstd, gamma, beta, gstride, dx, dres, dgamma, dbeta, nparam, accumulate, G, P, C,
                         act, unbiased, dx_s16, nullptr, 0, ws, ws_bytes, stream);
}

extern "C" int acg_norm_bwd_partials(const float *dy, const float *y, const unsigned *mask, const float *x, const float *mean,
                                     const float *rstd, const float *gamma, const float *beta, int gstride, float *dx,
                                     float *dres, float *dgamma, float *dbeta, int nparam, int accumulate, int G, size_t P,
                                     int C, int act, int unbiased, int dx_s16, const float *part, int nchunks, void *ws,
                                     size_t ws_bytes, void *stream)
{
    ACG_REQUIRE(part != nullptr && nchunks > 0, "acg_norm_bwd_partials: no partial sums");
    return norm_bwd_impl(dy, y, mask, x, mean, rstd, gamma, beta, gstride, dx, dres, dgamma, dbeta, nparam, accumulate, G, P, C,
                         act, unbiased, dx_s16, part, nchunks, ws, ws_bytes, stream);
}

static int norm_bwd_impl(const float *dy, const float *y, const unsigned *mask, const float *x, const float *mean,
                         const float *rstd, const float *gamma, const float *beta, int gstride, float *dx, float *dres,
                         float *dgamma, float *dbeta, int nparam, int accumulate, int G, size_t P, int C, int act, int unbiased,
                         int dx_s16, const float *part_in, int nch_in, void *ws, size_t ws_bytes, void *stream)
{
    int rc = check_norm(G, P, C, "acg_norm_bwd");
    if (rc) return rc;
    ACG_REQUIRE(dx_s16 == 0 || (act == ACG_ACT_RELU && dres == nullptr && C % 8 == 0 && (mask != nullptr || y == nullptr)),
                "acg_norm_bwd: pre-split dx is implemented for ReLU with the sign bitmask or the mask recomputed from x, without dres");
    ACG_REQUIRE(gstride != 0 || (nparam >= 0 && nparam <= C), "acg_norm_bwd: nparam=%d exceeds C=%d", nparam, C);
    ACG_REQUIRE(gstride == 0 || accumulate == 0, "acg_norm_bwd: accumulate is for shared (gstride == 0) parameters");
    ACG_REQUIRE(gstride == 0 || (gstride >= C && gstride % 4 == 0), "acg_norm_bwd: gstride must be 0 or a row stride >= C");
    ACG_REQUIRE(act == ACG_ACT_NONE || act == ACG_ACT_RELU || act == ACG_ACT_LRELU, "acg_norm_bwd: act %d", act);
    if (ws == nullptr || ws_bytes < acg_norm_workspace_bytes(G, P, C)) {
        acg_set_error("acg_norm_bwd: workspace too small");
        return ACG_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nch = nchunks_of(P);
    float *part = (float *)ws;
    float *sums = part + (size_t)G * nch * 2 * C;
    ACG_REQUIRE(act == ACG_ACT_NONE || y != nullptr || mask != nullptr || beta != nullptr,
                "acg_norm_bwd: y, the sign bitmask or beta required for the activation mask");
    ACG_REQUIRE(mask == nullptr || ((long long)P * (C / 4)) % 8 == 0, "acg_norm_bwd: bitmask layout needs P*C/4 %% 8 == 0");
    if (part_in == nullptr)
        launch_norm_bwd_partial(dim3(nch, G), st, dy, y, x, mean, rstd, gamma, beta, gstride, (long long)P, C, nch, act, part, mask);
    hipLaunchKernelGGL(norm_bwd_final, dim3(G * acg_cdiv(C, FIN_CH)), dim3(256), 0, st, part_in != nullptr ? part_in : (const float *)part,
                       G, C, part_in != nullptr ? nch_in : nch, sums, gstride ? dgamma : (float *)nullptr, gstride ? dbeta : (float *)nullptr);
    const bool params = gstride == 0 && nparam > 0 && (dgamma != nullptr || dbeta != nullptr);
    // unbiased == 2: statistics are constants (BatchNorm eval mode) -> dx = gamma * rstd * gy
    const float invP = unbiased == 2 ? 0.f : 1.f / (float)P;
    const float invD = unbiased == 2 ? 0.f : (unbiased ? 1.f / (float)(P - 1) : invP);
    launch_norm_bwd_apply(dim3(ew_blocks((long long)P * (C / 4)), G), st, dy, y, x, mean, rstd, gamma, beta, gstride,
                          (const float *)sums, dx, dres, (long long)P, C, act, invP, invD, mask, dx_s16, G, params ? nparam : 0,
                          accumulate, dgamma, dbeta);
    ACG_CHECK_LAUNCH("norm_bwd");
    return ACG_OK;
}
