// ds_read_b64_tr_b16 operand map check (gfx950): a [pixel][channel] 16-bit LDS image read as the K-contiguous MFMA operand.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read.hip -o tools/probes/tr_read   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 160;
__global__ void k(short *out)
{
    __shared__ __attribute__((aligned(16))) short img[64 * PITCH];
    for (int i = threadIdx.x; i < 64 * PITCH; i += 64) img[i] = (short)((i / PITCH) * 256 + (i % PITCH)); // row*256 + col
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    // group g reads the block of pixels 8*(g>>1) + 0..3, channels 16*(g&1) + 0..15: lane 4q+p supplies row q, columns 4p..4p+3
    __attribute__((address_space(3))) s16x4 *ptr =
        (__attribute__((address_space(3))) s16x4 *)&img[(8 * (g >> 1) + q) * PITCH + 16 * (g & 1) + 4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main()
{
    short *d, h[256];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            const int row = h[l * 4 + e] >> 8, col = h[l * 4 + e] & 255;
            printf(" (p%d,c%d)", row, col);
            bad += !(row == 8 * (g >> 1) + e && col == 16 * (g & 1) + i);
        }
        printf("\n");
    }
    printf("expected lane (g,i) element e = (pixel 8*(g>>1)+e, channel 16*(g&1)+i): %s\n", bad ? "MISMATCH" : "ok");
    return bad != 0;
}
