// Streaming-copy experiment: achieved GB/s of a [N,64] fp32 copy with (a) one pixel half per lane (each lane
// moves 128 contiguous bytes with eight 16-byte accesses) and (b) lane-linear 16-byte accesses.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void __launch_bounds__(256) copy_lane_private(const float4* __restrict__ in, float4* __restrict__ out, long ntok) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tok = lane & 31, h = lane >> 5;
    for (long tile = blockIdx.x; tile * 128 < ntok; tile += gridDim.x) {
        const long n = tile * 128 + wv * 32 + tok;
        if (n >= ntok) continue;
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = in[n * 16 + 8 * h + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) out[n * 16 + 8 * h + q] = v[q];
    }
}

__global__ void __launch_bounds__(256) copy_linear(const float4* __restrict__ in, float4* __restrict__ out, long ntok) {
    for (long tile = blockIdx.x; tile * 128 < ntok; tile += gridDim.x) {
        const long base = tile * 128 * 16;
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const long i = base + q * 256 + threadIdx.x;
            v[q] = i < ntok * 16 ? in[i] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const long i = base + q * 256 + threadIdx.x;
            if (i < ntok * 16) out[i] = v[q];
        }
    }
}

int main() {
    const long ntok = 64L * 128 * 128;
    float4 *in, *out;
    hipMalloc(&in, ntok * 256);
    hipMalloc(&out, ntok * 256);
    hipMemset(in, 0, ntok * 256);
    hipEvent_t s, e;
    hipEventCreate(&s);
    hipEventCreate(&e);
    for (int variant = 0; variant < 2; ++variant)
        for (int grid : {512, 1024, 2048, 8192}) {
            for (int it = 0; it < 3; ++it) {
                if (variant == 0) hipLaunchKernelGGL(copy_lane_private, dim3(grid), dim3(256), 0, 0, in, out, ntok);
                else hipLaunchKernelGGL(copy_linear, dim3(grid), dim3(256), 0, 0, in, out, ntok);
            }
            hipEventRecord(s);
            for (int it = 0; it < 20; ++it) {
                if (variant == 0) hipLaunchKernelGGL(copy_lane_private, dim3(grid), dim3(256), 0, 0, in, out, ntok);
                else hipLaunchKernelGGL(copy_linear, dim3(grid), dim3(256), 0, 0, in, out, ntok);
            }
            hipEventRecord(e);
            hipEventSynchronize(e);
            float ms;
            hipEventElapsedTime(&ms, s, e);
            ms /= 20;
            printf("%s grid=%d: %.4f ms  %.0f GB/s (read+write)\n", variant == 0 ? "lane_private" : "linear      ", grid, ms,
                   2.0 * ntok * 256 / ms / 1e6);
        }
    return 0;
}
