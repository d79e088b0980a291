// LDS read bandwidth probe (GPU box): every workgroup's waves stream conflict-free ds_read_b128 from a 32 KB LDS image, no other
// work.  Prints the aggregate rate and bytes per CU and nanosecond for 1, 2, 4, 8, 16 waves per CU (256 workgroups = one per CU;
// 512 = two per CU for the 16-wave point) — the ceiling a kernel's fragment reads can approach.
//   hipcc --offload-arch=gfx950 -O3 -o lds_read_bw tools/probes/lds_read_bw.hip && ./lds_read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void lds_read(unsigned *out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[8192];   // 32 KB
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4 *p = (const u32x4 *)lds + lane + (wave & 7) * 64;   // 64 lanes x 16 B = 1 KB per read, consecutive slots (+ offsets < 18 KB: inside 32 KB)
    // inline asm: the compiler folds a plain C++ read loop (repeating addresses, xor of equal values)
    const unsigned addr = (unsigned)(size_t)(const void *)p;   // LDS byte address (low 32 bits of the generic pointer = LDS offset)
    u32x4 v0, v1, v2, v3, v4, v5, v6, v7;
    for (int it = 0; it < iters; ++it) {
        asm volatile("ds_read_b128 %0, %8 offset:0\n\t"
                     "ds_read_b128 %1, %8 offset:8192\n\t"
                     "ds_read_b128 %2, %8 offset:16384\n\t"
                     "ds_read_b128 %3, %8 offset:1024\n\t"
                     "ds_read_b128 %4, %8 offset:9216\n\t"
                     "ds_read_b128 %5, %8 offset:17408\n\t"
                     "ds_read_b128 %6, %8 offset:2048\n\t"
                     "ds_read_b128 %7, %8 offset:10240\n\t"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(addr) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const u32x4 acc = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = acc[0];
}

int main()
{
    unsigned *out;
    hipMalloc(&out, 4096 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const int cfgs[][2] = {{256, 64}, {256, 128}, {256, 256}, {256, 512}, {512, 512}, {256, 1024}};
    for (auto &c : cfgs) {
        lds_read<8><<<c[0], c[1]>>>(out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        lds_read<8><<<c[0], c[1]>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)c[0] * (c[1] / 64) * iters * 8 * 1024.0;
        printf("%4d workgroups x %4d threads (%2d waves per CU): %7.1f TB/s aggregate, %6.1f B/ns per CU (= B/cycle at 1 GHz; divide by the clock in GHz)\n",
               c[0], c[1], c[0] * (c[1] / 64) / 256, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    }
    return 0;
}
