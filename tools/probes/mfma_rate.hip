// Probe: the bf16 MFMA rate the chip SUSTAINS on register operands (no LDS, no memory), as a ceiling for the convolution
// kernels' executed-MFMA rate.  hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o tools/probes/mfma_rate
//   ./mfma_rate [waves_per_simd=2] [random=1] [shape=16|32] [dep=0|1: three dependent MFMAs per accumulator, as the kernels issue them]
// Every wave issues ITER x 16 MFMAs on 16 independent accumulators; operands are random bf16 (or zeros).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(const u32x4 *__restrict__ src, float *__restrict__ out, int iters, int dep)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + i) & 65535]);
        b[i] = __builtin_bit_cast(bf16x8, src[(tid * 8 + 4 + i) & 65535]);
    }
    float keep = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (dep) { // the convolution kernels' order: three products back to back on the same accumulator
            for (int it = 0; it < iters / 3; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + 1) & 3], b[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(j + 1) & 3], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) keep += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2 * k], b[j + 2 * k], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) keep += acc[i][j][0] + acc[i][j][15];
    }
    if (keep == 12345.678f) out[tid] = keep;
}

int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 2, rnd = argc > 2 ? atoi(argv[2]) : 1, shape = argc > 3 ? atoi(argv[3]) : 16;
    const int iters = 19998, dep = argc > 4 ? atoi(argv[4]) : 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int threads = 64 * 4 * wps > 512 ? 512 : 64 * 4 * wps;        // one workgroup per CU (two when wps == 4 is asked with 512-thread groups)
    const int wgs = cus * (64 * 4 * wps) / threads;
    u32x4 *src;
    float *out;
    hipMalloc(&src, 65536 * sizeof(u32x4));
    hipMalloc(&out, (size_t)wgs * threads * sizeof(float));
    unsigned *h = (unsigned *)malloc(65536 * 16);
    srand(7);
    for (int i = 0; i < 65536 * 4; ++i) {
        // two bf16 per word: random sign / mantissa, exponent near 1.0 (as normalised activations and small weights have)
        const unsigned lo = (rand() & 0x807f) | ((120 + (rand() % 8)) << 7), hi = (rand() & 0x807f) | ((120 + (rand() % 8)) << 7);
        h[i] = rnd ? (lo | (hi << 16)) : 0u;
    }
    hipMemcpy(src, h, 65536 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(wgs), dim3(threads), 0, 0, src, out, iters, dep);
        else hipLaunchKernelGGL(mfma_loop<32>, dim3(wgs), dim3(threads), 0, 0, src, out, iters, dep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double nm = (double)wgs * (threads / 64) * iters * (shape == 16 ? 16 : 8);
        const double fl = nm * 2.0 * (shape == 16 ? 16 * 16 * 32 : 32 * 32 * 16);
        printf("shape %dx  waves/SIMD %d  %s: %.2f ms  %.0f TFLOP/s  (%.2f GHz-equivalent of 2.4 GHz peak)\n", shape, wps,
               rnd ? "random" : "zeros", ms, fl / ms * 1e-9, fl / ms * 1e-9 / 2516.6 * 2.4);
    }
    return 0;
}
