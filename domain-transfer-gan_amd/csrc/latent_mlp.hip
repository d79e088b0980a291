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

// dynamic LDS: xin[N*H] | act[N*H] | col[2*H]
__global__ __launch_bounds__(256) void latent_mlp_fwd_kernel(MlpParams p, const float *__restrict__ z, int ldz, int N, int I,
                                                             int H, float eps, float momentum, float *__restrict__ a_save,
                                                             float *__restrict__ stats_save, float *__restrict__ out)
{
    extern __shared__ float lds[];
    float *xin = lds, *act = lds + N * H, *col = lds + 2 * N * H;
    const int tid = threadIdx.x;
    for (int i = tid; i < N * I; i += 256) xin[(i / I) * H + (i % I)] = z[(long long)(i / I) * ldz + (i % I)];
    __syncthreads();
    int K = I;
    for (int l = 0; l < 3; ++l) {
        const float *W = p.w[l], *B = p.b[l];
        for (int i = tid; i < N * H; i += 256) {
            const int n = i / H, o = i - n * H;
            float acc = B[o];
            for (int k = 0; k < K; ++k) acc += xin[n * H + k] * W[o * K + k];
            act[i] = acc;
            a_save[(long long)l * N * H + i] = acc;
        }
        __syncthreads();
        if (tid < H) {   // batch statistics of column tid (two passes over <= a few hundred rows)
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += act[n * H + tid];
            const float mean = s / (float)N;
            float q = 0.f;
            for (int n = 0; n < N; ++n) { const float d = act[n * H + tid] - mean; q += d * d; }
            const float var = q / (float)N, rstd = rsqrtf(var + eps);
            col[tid] = mean; col[H + tid] = rstd;
            stats_save[(l * 2) * H + tid] = mean;
            stats_save[(l * 2 + 1) * H + tid] = rstd;
            if (p.run_mean[l] != nullptr) {
                p.run_mean[l][tid] = (1.f - momentum) * p.run_mean[l][tid] + momentum * mean;
                p.run_var[l][tid] = (1.f - momentum) * p.run_var[l][tid] + momentum * (q / (float)(N > 1 ? N - 1 : 1));
            }
        }
        __syncthreads();
        const float *G = p.gamma[l], *Be = p.beta[l];
        for (int i = tid; i < N * H; i += 256) {
            const int o = i % H;
            xin[i] = lrelu((act[i] - col[o]) * col[H + o] * G[o] + Be[o]);
        }
        __syncthreads();
        K = H;
    }
    for (int n = tid; n < N; n += 256) {
        float acc = p.b[3][0];
        for (int k = 0; k < H; ++k) acc += xin[n * H + k] * p.w[3][k];
        *(f32x4 *)(out + (long long)n * 4) = (f32x4){acc, 0.f, 0.f, 0.f};
    }
}

// dynamic LDS: g[N*H] | da[N*H] | hprev[N*H] | col[2*H]
__global__ __launch_bounds__(256) void latent_mlp_bwd_kernel(MlpParams p, MlpGrads gr, const float *__restrict__ z, int ldz, int N,
                                                             int I, int H, const float *__restrict__ a_save,
                                                             const float *__restrict__ stats_save, const float *__restrict__ dout,
                                                             float *__restrict__ dz, int accumulate)
{
    extern __shared__ float lds[];
    float *g = lds, *da = lds + N * H, *hprev = lds + 2 * N * H, *col = lds + 3 * N * H;
    const int tid = threadIdx.x;
    auto put = [&](float *dst, float v) { if (dst) *dst = (accumulate ? *dst : 0.f) + v; };
    // activations of layer l (1..3) recomputed from the saved pre-norm values; l == 0: the input
    auto load_h = [&](int l) {
        if (l == 0) {
            for (int i = tid; i < N * H; i += 256) {
                const int n = i / H, k = i - n * H;
                hprev[i] = k < I ? z[(long long)n * ldz + k] : 0.f;
            }
        } else {
            const float *a = a_save + (long long)(l - 1) * N * H, *st = stats_save + (l - 1) * 2 * H;
            const float *G = p.gamma[l - 1], *Be = p.beta[l - 1];
            for (int i = tid; i < N * H; i += 256) {
                const int o = i % H;
                hprev[i] = lrelu((a[i] - st[o]) * st[H + o] * G[o] + Be[o]);
            }
        }
    };
    // ---- head: p[n] = b4 + h3[n] . w4
    load_h(3);
    __syncthreads();
    for (int i = tid; i < N * H; i += 256) g[i] = dout[(long long)(i / H) * 4] * p.w[3][i % H];
    if (tid < H) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dout[(long long)n * 4] * hprev[n * H + tid];
        put(gr.dw[3] ? gr.dw[3] + tid : nullptr, s);
    }
    if (tid == 0) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dout[(long long)n * 4];
        put(gr.db[3], s);
    }
    __syncthreads();
    for (int l = 2; l >= 0; --l) {
        const float *a = a_save + (long long)l * N * H, *st = stats_save + l * 2 * H;
        const float *G = p.gamma[l], *Be = p.beta[l], *W = p.w[l];
        const int K = l == 0 ? I : H;
        // dy = g * lrelu'(pre), in place
        for (int i = tid; i < N * H; i += 256) {
            const int o = i % H;
            const float pre = (a[i] - st[o]) * st[H + o] * G[o] + Be[o];
            g[i] *= pre > 0.f ? 1.f : 0.2f;
        }
        __syncthreads();
        if (tid < H) {
            float s1 = 0.f, s2 = 0.f;
            for (int n = 0; n < N; ++n) {
                const float dy = g[n * H + tid], xh = (a[n * H + tid] - st[tid]) * st[H + tid];
                s1 += dy; s2 += dy * xh;
            }
            col[tid] = s1; col[H + tid] = s2;
            put(gr.dbeta[l] ? gr.dbeta[l] + tid : nullptr, s1);
            put(gr.dgamma[l] ? gr.dgamma[l] + tid : nullptr, s2);
        }
        load_h(l);   // input of linear l (independent of the sums)
        __syncthreads();
        const float invN = 1.f / (float)N;
        for (int i = tid; i < N * H; i += 256) {
            const int o = i % H;
            const float xh = (a[i] - st[o]) * st[H + o];
            da[i] = G[o] * st[H + o] * (g[i] - col[o] * invN - xh * col[H + o] * invN);
        }
        __syncthreads();
        // dW[o][k] = sum_n da[n][o] * hprev[n][k]; db[o] = sum_n da[n][o]
        for (int i = tid; i < H * K; i += 256) {
            const int o = i / K, k = i - o * K;
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += da[n * H + o] * hprev[n * H + k];
            put(gr.dw[l] ? gr.dw[l] + i : nullptr, s);
        }
        if (tid < H) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += da[n * H + tid];
            put(gr.db[l] ? gr.db[l] + tid : nullptr, s);
        }
        // gradient w.r.t. the input of linear l
        if (l > 0 || dz != nullptr) {
            for (int i = tid; i < N * K; i += 256) {
                const int n = i / K, k = i - n * K;
                float s = 0.f;
                for (int o = 0; o < H; ++o) s += da[n * H + o] * W[o * K + k];
                if (l > 0) g[n * H + k] = s;        // g was consumed when da was formed (barrier above); hprev is still being read
                else dz[(long long)n * ldz + k] = s;
            }
        }
        __syncthreads();
    }
}

static int mlp_check(const MlpParams *p, int N, int I, int H, const char *who)
{
    ACG_REQUIRE(p != nullptr && N > 0 && I > 0 && H > 0 && I <= H && H <= 256, "%s: bad dims N=%d I=%d H=%d", who, N, I, H);
    ACG_REQUIRE((size_t)N * H * 3 * sizeof(float) + 2 * H * sizeof(float) <= 160 * 1024 - 1024,
                "%s: N x H = %d x %d does not fit one workgroup's LDS (use the layer-by-layer path)", who, N, H);
    return ACG_OK;
}

extern "C" int acg_latent_mlp_supported(int N, int I, int H)
{
    return N > 0 && I > 0 && H > 0 && I <= H && H <= 256 &&
           (size_t)N * H * 3 * sizeof(float) + 2 * H * sizeof(float) <= 160 * 1024 - 1024;
}

extern "C" int acg_latent_mlp_fwd(const acg_latent_mlp_params *params, const float *z, int ldz, int N, int I, int H, float eps,
                                  float momentum, float *a_save, float *stats_save, float *out, void *stream)
{
    const MlpParams *p = (const MlpParams *)params;
    int rc = mlp_check(p, N, I, H, "acg_latent_mlp_fwd");
    if (rc) return rc;
    ACG_REQUIRE(z != nullptr && ldz >= I && a_save != nullptr && stats_save != nullptr && out != nullptr, "acg_latent_mlp_fwd: null");
    const size_t sh = ((size_t)2 * N * H + 2 * H) * sizeof(float);
    if (sh > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)latent_mlp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(latent_mlp_fwd_kernel, dim3(1), dim3(256), sh, (hipStream_t)stream, *p, z, ldz, N, I, H, eps, momentum,
                       a_save, stats_save, out);
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
    const size_t sh = ((size_t)3 * N * H + 2 * H) * sizeof(float);
    if (sh > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)latent_mlp_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(latent_mlp_bwd_kernel, dim3(1), dim3(256), sh, (hipStream_t)stream, *p, *(const MlpGrads *)grads, z, ldz, N,
                       I, H, a_save, stats_save, dout, dz, accumulate);
    ACG_CHECK_LAUNCH("latent_mlp_bwd_kernel");
    return ACG_OK;
}
