// MFMA + LDS-read co-issue probe (GPU box): a wave issues 24 v_mfma_f32_16x16x32_bf16 (8 accumulators x 3, the bf16x3 pattern) and
// R conflict-free ds_read_b128 per iteration, interleaved one read per 24 / R MFMAs; 4 or 8 waves per CU (1 or 2 per SIMD).  Prints
// the MFMA rate as a fraction of 16 cycles per MFMA and SIMD at the measured time — how much of the matrix pipe a kernel keeps
// when its fragment reads ride beside the MFMAs (conv_rows_x3: R = 12, 4 waves; the 64 x 64 wave tile of the trunk: R = 8, 8 waves).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds_mix tools/probes/mfma_lds_mix.hip && ./mfma_lds_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R, int DEP>
__global__ __launch_bounds__(512) void mix(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0x3f803f80u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned addr = (unsigned)(size_t)(const void *)((const u32x4 *)lds + lane + (wave & 7) * 64);
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 f0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, f1 = f0, f2 = f0, f3 = f0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, (m & 1) ? f0 : f1), b = __builtin_bit_cast(bf16x8, (m & 2) ? f2 : f3);
            // DEP = 1: three consecutive MFMAs per accumulator (the order a bf16x3 product is usually written in); 0: rotating
            const int k = DEP ? m / 3 : (m & 7);
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
            if (R > 0 && (m % (24 / (R > 24 ? 24 : R))) == 0) {
                // the read's result replaces a fragment register a few MFMAs later (in-order return; waited for at the loop end)
                if ((m / (24 / (R > 24 ? 24 : R))) & 1) asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(f0) : "v"(addr) : "memory");
                else asm volatile("ds_read_b128 %0, %1 offset:9216" : "=v"(f2) : "v"(addr) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == 12345.f) out[blockIdx.x] = s;
}

template <int R, int DEP> static void run(float *out, int threads)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    mix<R, DEP><<<256, threads>>>(out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    mix<R, DEP><<<256, threads>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)(threads / 64) / 4.0 * iters * 24;
    printf("%s, R = %2d reads per 24 MFMAs, %d waves per CU: %.3f ms, %.1f ns per MFMA and SIMD (16 cycles = 6.7-7.6 ns at 2.4-2.1 GHz), LDS %.0f B/ns per CU\n",
           DEP ? "3 dependent MFMAs in a row" : "rotating accumulators     ", R, threads / 64, ms, ms * 1e6 / mfma_per_simd, (double)(threads / 64) * iters * R * 1024.0 / (ms * 1e6));
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 4096);
    for (int threads : {256, 512}) {
        run<0, 0>(out, threads); run<8, 0>(out, threads); run<12, 0>(out, threads); run<24, 0>(out, threads);
        run<0, 1>(out, threads); run<8, 1>(out, threads); run<12, 1>(out, threads);
    }
    return 0;
}
