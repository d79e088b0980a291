// DiscriminatorLatent (networks.py:396-433) as ONE kernel per direction: Linear(I->H) BN1d LReLU(0.2), twice more H->H,
// then Linear(H->1).  The whole batch (N x H activations) lives in one workgroup's LDS; BatchNorm1d runs in train mode
// (batch statistics, biased variance for the normalisation, running buffers updated with the unbiased one, networks.py:407-415).
// Layer by layer this took 4 linear + 3 x (statistics, final, apply) launches forward and about twice that backward, three
// times per training step: launch latency only (the arithmetic is 0.4 MFLOP).
#include "common.h"

struct MlpParams {   // device pointers (mirrors acg_latent_mlp_params)
    const float *w[4], *b[4], *gamma[3], *beta[3];
    float *run_mean[3], *run_var[3];
};
struct MlpGrads {    // mirrors acg_latent_mlp_grads
    float *dw[4], *db[4], *dgamma[3], *dbeta[3];
};

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.2f * v; }

// Thread mapping of both kernels: element i = tid + 256 j of an [N][H] activation, j < J.  256 % H == 0 (launcher), so a
// thread's feature o = tid % H is the same for all its elements and its samples are n = tid / H + (256 / H) j: the features
// of a layer lie across the lanes, the batch across the waves and the registers.  A dot product reads its weight row from
// an LDS copy with rows padded to K + 1 floats (lanes o apart by K + 1 words: conflict-free; the transposed access of the
// backward pass, lanes k consecutive, too) and its input as a wave-uniform broadcast; BatchNorm1d's sums over the batch are
// a register partial per thread plus one LDS fold over the 256 / H thread groups — no loop over the batch anywhere, no
// scalar global loads in any inner loop (the first version ran every dot product on strided global loads of the weights
// and the statistics as serial loops of one thread per feature: 143 / 149 us per launch for 0.4 MFLOP).
constexpr int MLP_JMAX = 16;   // N * H <= 4096 elements

struct MlpLds {               // offsets (floats) into the dynamic LDS block
    int xs, ws, red, hs, ds, total;
    __host__ __device__ MlpLds(int N, int H)
    {
        xs = 0;                        // [N][H + 1] activations / gradient input
        ws = xs + N * (H + 1);         // [H][H + 1] weight rows
        red = ws + H * (H + 1);        // [2][256 / H][H] fold buffer
        hs = red + 2 * 256;            // [N][H + 1] (backward) input of the linear layer
        ds = hs + N * (H + 1);         // [N][H + 1] (backward) da
        total = ds + N * (H + 1);
    }
};

// sum over the batch of a per-thread partial: fold the 256 / H thread groups through LDS (which = 0 / 1: two buffers, so
// that two sums can share a barrier).  Every thread returns the total of ITS feature.
__device__ __forceinline__ void mlp_fold_put(float *red, int which, int tid, float v) { red[which * 256 + tid] = v; }
__device__ __forceinline__ float mlp_fold_get(const float *red, int which, int o, int H)
{
    float s = 0.f;
    for (int w = 0; w < 256 / H; ++w) s += red[which * 256 + w * H + o];
    return s;
}

template <int J>
__global__ __launch_bounds__(256) void latent_mlp_fwd_kernel(MlpParams p, const float *__restrict__ z, int ldz, int N, int I,
                                                             int H, float eps, float momentum, float *__restrict__ a_save,
                                                             float *__restrict__ stats_save, float *__restrict__ out)
{
    extern __shared__ float lds[];
    const MlpLds L(N, H);
    float *xs = lds + L.xs, *ws = lds + L.ws, *red = lds + L.red;
    const int tid = threadIdx.x, o = tid % H, n0 = tid / H, nstep = 256 / H, P = H + 1;
    for (int i = tid; i < N * H; i += 256) {
        const int n = i / H, k = i - n * H;
        xs[n * P + k] = k < I ? z[(long long)n * ldz + k] : 0.f;
    }
    int K = I;
    for (int l = 0; l < 3; ++l) {
        const int KP = K + 1;
        for (int i = tid; i < H * K; i += 256) ws[(i / K) * KP + (i % K)] = p.w[l][i];   // coalesced read, padded rows
        const float bo = p.b[l][o], go = p.gamma[l][o], beo = p.beta[l][o];
        __syncthreads();
        float acc[J];
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] = bo;
        for (int k = 0; k < K; ++k) {
            const float w = ws[o * KP + k];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int n = n0 + nstep * j;
                acc[j] += (n < N ? xs[n * P + k] : 0.f) * w;
            }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int n = n0 + nstep * j;
            if (n < N) {
                a_save[(long long)l * N * H + n * H + o] = acc[j];
                s += acc[j];
            }
        }
        mlp_fold_put(red, 0, tid, s);
        __syncthreads();   // (also: every thread is done reading xs / ws of this layer)
        const float mean = mlp_fold_get(red, 0, o, H) / (float)N;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float d = acc[j] - mean;
            if (n0 + nstep * j < N) q += d * d;
        }
        mlp_fold_put(red, 1, tid, q);
        __syncthreads();
        q = mlp_fold_get(red, 1, o, H);
        const float rstd = rsqrtf(q / (float)N + eps);
        if (tid < H) {
            stats_save[(l * 2) * H + o] = mean;
            stats_save[(l * 2 + 1) * H + o] = rstd;
            if (p.run_mean[l] != nullptr) {
                p.run_mean[l][o] = (1.f - momentum) * p.run_mean[l][o] + momentum * mean;
                p.run_var[l][o] = (1.f - momentum) * p.run_var[l][o] + momentum * (q / (float)(N > 1 ? N - 1 : 1));
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int n = n0 + nstep * j;
            if (n < N) xs[n * P + o] = lrelu((acc[j] - mean) * rstd * go + beo);
        }
        __syncthreads();
        K = H;
    }
    // head: one lane group of 16 per sample, shuffle fold
    for (int n = tid >> 4; n < N; n += 16) {
        float acc = 0.f;
        for (int k = tid & 15; k < H; k += 16) acc += xs[n * P + k] * p.w[3][k];
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 8);
        if ((tid & 15) == 0) *(f32x4 *)(out + (long long)n * 4) = (f32x4){acc + p.b[3][0], 0.f, 0.f, 0.f};
    }
}

template <int J>
__global__ __launch_bounds__(256) void latent_mlp_bwd_kernel(MlpParams p, MlpGrads gr, const float *__restrict__ z, int ldz, int N,
                                                             int I, int H, const float *__restrict__ a_save,
                                                             const float *__restrict__ stats_save, const float *__restrict__ dout,
                                                             float *__restrict__ dz, int accumulate)
{
    extern __shared__ float lds[];
    const MlpLds L(N, H);
    float *ws = lds + L.ws, *red = lds + L.red, *hs = lds + L.hs, *ds = lds + L.ds;
    const int tid = threadIdx.x, o = tid % H, n0 = tid / H, nstep = 256 / H, P = H + 1;
    auto put = [&](float *dst, float v) { if (dst) *dst = (accumulate ? *dst : 0.f) + v; };
    const float invN = 1.f / (float)N;
    // ---- head: p[n] = b4 + h3[n] . w4  (h3 recomputed from the saved pre-norm values of layer 2)
    float g[J];   // gradient w.r.t. the output of the current layer's LeakyReLU, element (n0 + nstep j, o)
    {
        const float *a = a_save + (long long)2 * N * H, *st = stats_save + 2 * 2 * H;
        const float mean = st[o], rstd = st[H + o], go = p.gamma[2][o], beo = p.beta[2][o], w4 = p.w[3][o];
        float sw = 0.f, sb = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int n = n0 + nstep * j;
            g[j] = 0.f;
            if (n < N) {
                const float d = dout[(long long)n * 4];
                g[j] = d * w4;
                sw += d * lrelu((a[n * H + o] - mean) * rstd * go + beo);
                if (o == 0) sb += d;
            }
        }
        mlp_fold_put(red, 0, tid, sw);
        mlp_fold_put(red, 1, tid, sb);
        __syncthreads();
        if (tid < H) put(gr.dw[3] ? gr.dw[3] + o : nullptr, mlp_fold_get(red, 0, o, H));
        if (tid == 0) put(gr.db[3], mlp_fold_get(red, 1, 0, H));
        __syncthreads();
    }
    for (int l = 2; l >= 0; --l) {
        const float *a = a_save + (long long)l * N * H, *st = stats_save + l * 2 * H;
        const int K = l == 0 ? I : H, KP = K + 1;
        const float mean = st[o], rstd = st[H + o], go = p.gamma[l][o], beo = p.beta[l][o];
        float xh[J], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int n = n0 + nstep * j;
            xh[j] = 0.f;
            if (n < N) {
                xh[j] = (a[n * H + o] - mean) * rstd;
                g[j] *= (xh[j] * go + beo) > 0.f ? 1.f : 0.2f;      // dy = g * lrelu'(pre)
                s1 += g[j];
                s2 += g[j] * xh[j];
            }
        }
        mlp_fold_put(red, 0, tid, s1);
        mlp_fold_put(red, 1, tid, s2);
        // meanwhile: the input of linear l into hs, the weights into ws (nothing below reads them before the barrier)
        if (l == 0) {
            for (int i = tid; i < N * H; i += 256) {
                const int n = i / H, k = i - n * H;
                hs[n * P + k] = k < I ? z[(long long)n * ldz + k] : 0.f;
            }
        } else {
            const float *ap = a_save + (long long)(l - 1) * N * H, *sp = stats_save + (l - 1) * 2 * H;
            const float m1 = sp[o], r1 = sp[H + o], g1 = p.gamma[l - 1][o], b1 = p.beta[l - 1][o];
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int n = n0 + nstep * j;
                if (n < N) hs[n * P + o] = lrelu((ap[n * H + o] - m1) * r1 * g1 + b1);
            }
        }
        for (int i = tid; i < H * K; i += 256) ws[(i / K) * KP + (i % K)] = p.w[l][i];
        __syncthreads();
        s1 = mlp_fold_get(red, 0, o, H);
        s2 = mlp_fold_get(red, 1, o, H);
        if (tid < H) {
            put(gr.dbeta[l] ? gr.dbeta[l] + o : nullptr, s1);
            put(gr.dgamma[l] ? gr.dgamma[l] + o : nullptr, s2);
        }
        float sdb = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int n = n0 + nstep * j;
            if (n < N) {
                const float da = go * rstd * (g[j] - s1 * invN - xh[j] * s2 * invN);
                ds[n * P + o] = da;
                sdb += da;
            }
        }
        __syncthreads();   // ds complete; the fold buffers have been read
        mlp_fold_put(red, 0, tid, sdb);
        // dW[o][k] = sum_n da[n][o] * h[n][k]: thread (o, k = n0 + nstep m): da lane-consecutive, h a wave-uniform broadcast
        {
            float dw[MLP_JMAX];
            const int nk = (K + nstep - 1) / nstep;          // <= H / nstep <= 64 for H <= 256 ... bounded by MLP_JMAX below
#pragma unroll
            for (int m = 0; m < MLP_JMAX; ++m) dw[m] = 0.f;
            for (int kb = 0; kb < nk; kb += MLP_JMAX) {
                for (int n = 0; n < N; ++n) {
                    const float dv = ds[n * P + o];
#pragma unroll
                    for (int m = 0; m < MLP_JMAX; ++m) {
                        const int k = n0 + nstep * (kb + m);
                        dw[m] += dv * (k < K ? hs[n * P + k] : 0.f);
                    }
                }
#pragma unroll
                for (int m = 0; m < MLP_JMAX; ++m) {
                    const int k = n0 + nstep * (kb + m);
                    if (k < K) put(gr.dw[l] ? gr.dw[l] + o * K + k : nullptr, dw[m]);
                    dw[m] = 0.f;
                }
            }
        }
        // gradient w.r.t. the input of linear l, element (n, k = o): sum_o' da[n][o'] * W[o'][k]
        if (l > 0 || dz != nullptr) {
            float gp[J];
#pragma unroll
            for (int j = 0; j < J; ++j) gp[j] = 0.f;
            if (o < K) {
                for (int oo = 0; oo < H; ++oo) {
                    const float w = ws[oo * KP + o];
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const int n = n0 + nstep * j;
                        gp[j] += (n < N ? ds[n * P + oo] : 0.f) * w;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                g[j] = gp[j];
                const int n = n0 + nstep * j;
                if (l == 0 && o < K && n < N) dz[(long long)n * ldz + o] = gp[j];
            }
        }
        __syncthreads();   // hs / ds / ws are rewritten by the next layer; the db partials are in place
        if (tid < H) put(gr.db[l] ? gr.db[l] + o : nullptr, mlp_fold_get(red, 0, o, H));
        __syncthreads();
    }
}

extern "C" int acg_latent_mlp_supported(int N, int I, int H);
static int mlp_check(const MlpParams *p, int N, int I, int H, const char *who)
{
    ACG_REQUIRE(p != nullptr && acg_latent_mlp_supported(N, I, H),
                "%s: N=%d I=%d H=%d outside the fused kernel (256 %% H == 0, N * H <= %d; use the layer-by-layer path)", who, N, I, H,
                256 * MLP_JMAX);
    return ACG_OK;
}

extern "C" int acg_latent_mlp_supported(int N, int I, int H)
{
    return N > 0 && I > 0 && H >= 16 && I <= H && H <= 256 && 256 % H == 0 && (long long)N * H <= 256 * MLP_JMAX &&
           (size_t)MlpLds(N, H).total * sizeof(float) <= 160 * 1024 - 1024;
}

extern "C" int acg_latent_mlp_fwd(const acg_latent_mlp_params *params, const float *z, int ldz, int N, int I, int H, float eps,
                                  float momentum, float *a_save, float *stats_save, float *out, void *stream)
{
    const MlpParams *p = (const MlpParams *)params;
    int rc = mlp_check(p, N, I, H, "acg_latent_mlp_fwd");
    if (rc) return rc;
    ACG_REQUIRE(z != nullptr && ldz >= I && a_save != nullptr && stats_save != nullptr && out != nullptr, "acg_latent_mlp_fwd: null");
    const size_t sh = (size_t)MlpLds(N, H).hs * sizeof(float);   // the forward uses the first three blocks
    const int J = (N * H + 255) / 256;
#define MLP_FWD(JJ)                                                                                                              \
    do {                                                                                                                         \
        if (sh > 64 * 1024)                                                                                                      \
            (void)hipFuncSetAttribute((const void *)latent_mlp_fwd_kernel<JJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        hipLaunchKernelGGL(latent_mlp_fwd_kernel<JJ>, dim3(1), dim3(256), sh, (hipStream_t)stream, *p, z, ldz, N, I, H, eps,     \
                           momentum, a_save, stats_save, out);                                                                   \
    } while (0)
    if (J <= 1) MLP_FWD(1); else if (J <= 2) MLP_FWD(2); else if (J <= 4) MLP_FWD(4); else if (J <= 8) MLP_FWD(8); else MLP_FWD(16);
#undef MLP_FWD
    ACG_CHECK_LAUNCH("latent_mlp_fwd_kernel");
    return ACG_OK;
}

extern "C" int acg_latent_mlp_bwd(const acg_latent_mlp_params *params, const acg_latent_mlp_grads *grads, const float *z, int ldz,
                                  int N, int I, int H, const float *a_save, const float *stats_save, const float *dout, float *dz,
                                  int accumulate, void *stream)
{
    const MlpParams *p = (const MlpParams *)params;
    int rc = mlp_check(p, N, I, H, "acg_latent_mlp_bwd");
    if (rc) return rc;
    ACG_REQUIRE(grads != nullptr && z != nullptr && a_save != nullptr && stats_save != nullptr && dout != nullptr, "acg_latent_mlp_bwd: null");
    const size_t sh = (size_t)MlpLds(N, H).total * sizeof(float);
    const int J = (N * H + 255) / 256;
#define MLP_BWD(JJ)                                                                                                              \
    do {                                                                                                                         \
        if (sh > 64 * 1024)                                                                                                      \
            (void)hipFuncSetAttribute((const void *)latent_mlp_bwd_kernel<JJ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
        hipLaunchKernelGGL(latent_mlp_bwd_kernel<JJ>, dim3(1), dim3(256), sh, (hipStream_t)stream, *p, *(const MlpGrads *)grads, z,  \
                           ldz, N, I, H, a_save, stats_save, dout, dz, accumulate);                                              \
    } while (0)
    if (J <= 1) MLP_BWD(1); else if (J <= 2) MLP_BWD(2); else if (J <= 4) MLP_BWD(4); else if (J <= 8) MLP_BWD(8); else MLP_BWD(16);
#undef MLP_BWD
    ACG_CHECK_LAUNCH("latent_mlp_bwd_kernel");
    return ACG_OK;
}
