// Streaming probe: what a [N,64] bf16 / fp32 pass can reach on this GPU with different access shapes.
//   copy1      one float4 per thread, one pass (grid = elements / 256)
//   copyU<U>   persistent, U float4 per thread in flight (lane-linear), grid sweep
//   readU<U>   read-only (sum), writeU: write-only
//   nt         the same with non-temporal loads / stores
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* p) { const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void nt_store(float4 x, float4* p) { const v4f v = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); }

__global__ void __launch_bounds__(256) copy1(const float4* __restrict__ in, float4* __restrict__ out, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) out[i] = in[i];
}

template <int U, bool NT>
__global__ void __launch_bounds__(256) copyU(const float4* __restrict__ in, float4* __restrict__ out, long n4) {
    const long per = (long)U * 256;
    for (long base = (long)blockIdx.x * per; base < n4; base += (long)gridDim.x * per) {
        float4 v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const long i = base + q * 256 + threadIdx.x;
            if (NT) v[q] = i < n4 ? nt_load(in + i) : make_float4(0, 0, 0, 0);
            else v[q] = i < n4 ? in[i] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const long i = base + q * 256 + threadIdx.x;
            if (i < n4) { if (NT) nt_store(v[q], out + i); else out[i] = v[q]; }
        }
    }
}

template <int U, bool NT>
__global__ void __launch_bounds__(256) readU(const float4* __restrict__ in, float* __restrict__ out, long n4) {
    const long per = (long)U * 256;
    float s = 0.f;
    for (long base = (long)blockIdx.x * per; base < n4; base += (long)gridDim.x * per) {
        float4 v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const long i = base + q * 256 + threadIdx.x;
            if (NT) v[q] = i < n4 ? nt_load(in + i) : make_float4(0, 0, 0, 0);
            else v[q] = i < n4 ? in[i] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < U; ++q) s += v[q].x + v[q].y + v[q].z + v[q].w;
    }
    if (s == 123.456f) out[0] = s;
}

template <int U, bool NT>
__global__ void __launch_bounds__(256) writeU(float4* __restrict__ out, long n4) {
    const long per = (long)U * 256;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
    for (long base = (long)blockIdx.x * per; base < n4; base += (long)gridDim.x * per) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const long i = base + q * 256 + threadIdx.x;
            if (i < n4) { if (NT) nt_store(v, out + i); else out[i] = v; }
        }
    }
}

template <class F>
float timeit(F f) {
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int it = 0; it < 3; ++it) f();
    hipEventRecord(s);
    for (int it = 0; it < 20; ++it) f();
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    return ms / 20;
}

int main() {
    for (long mb : {128L, 512L}) {
        const long bytes = mb << 20, n4 = bytes / 16;
        float4 *in, *out; float* sink;
        hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMalloc(&sink, 64);
        hipMemset(in, 0, bytes);
        printf("---- %ld MB per tensor\n", mb);
        float ms = timeit([&] { hipLaunchKernelGGL(copy1, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, in, out, n4); });
        printf("copy1                : %.4f ms %6.0f GB/s r+w\n", ms, 2.0 * bytes / ms / 1e6);
        for (int grid : {512, 1024, 2048, 4096}) {
            ms = timeit([&] { hipLaunchKernelGGL((copyU<4, false>), dim3(grid), dim3(256), 0, 0, in, out, n4); });
            printf("copyU<4>    grid=%4d: %.4f ms %6.0f GB/s r+w\n", grid, ms, 2.0 * bytes / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((copyU<8, false>), dim3(grid), dim3(256), 0, 0, in, out, n4); });
            printf("copyU<8>    grid=%4d: %.4f ms %6.0f GB/s r+w\n", grid, ms, 2.0 * bytes / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((copyU<8, true>), dim3(grid), dim3(256), 0, 0, in, out, n4); });
            printf("copyU<8,nt> grid=%4d: %.4f ms %6.0f GB/s r+w\n", grid, ms, 2.0 * bytes / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((copyU<16, false>), dim3(grid), dim3(256), 0, 0, in, out, n4); });
            printf("copyU<16>   grid=%4d: %.4f ms %6.0f GB/s r+w\n", grid, ms, 2.0 * bytes / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((readU<8, false>), dim3(grid), dim3(256), 0, 0, in, sink, n4); });
            printf("readU<8>    grid=%4d: %.4f ms %6.0f GB/s r\n", grid, ms, 1.0 * bytes / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((readU<8, true>), dim3(grid), dim3(256), 0, 0, in, sink, n4); });
            printf("readU<8,nt> grid=%4d: %.4f ms %6.0f GB/s r\n", grid, ms, 1.0 * bytes / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((writeU<8, false>), dim3(grid), dim3(256), 0, 0, out, n4); });
            printf("writeU<8>   grid=%4d: %.4f ms %6.0f GB/s w\n", grid, ms, 1.0 * bytes / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((writeU<8, true>), dim3(grid), dim3(256), 0, 0, out, n4); });
            printf("writeU<8,nt>grid=%4d: %.4f ms %6.0f GB/s w\n", grid, ms, 1.0 * bytes / ms / 1e6);
        }
        hipFree(in); hipFree(out); hipFree(sink);
    }
    return 0;
}
