// Issue rate of a few integer vector instructions on one GPU: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate
// Every thread runs 8 independent chains of one instruction; 4 waves per SIMD on every CU.  Prints wave-instructions per
// SIMD and microsecond; the ratio between the rows is what matters.
// MI355X: the integer instructions within 15 % of each other (≈ 420-500 per us: one per ≈ 5 cycles; the 32-bit multiply is a
// full-rate instruction); v_fma_f32 670 per us; v_pk_fma_f32 347 per us (two multiply-adds per lane at half the instruction rate:
// the same flops as v_fma_f32, fewer instructions to issue); v_rcp_f32 / v_exp_f32 219 per us; v_cvt_pk_bf16_f32 458 per us.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(unsigned* out, unsigned seed, int iters) {
    unsigned x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 9u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
            if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
            if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
            if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(seed));
            if (OP == 4) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
            if (OP == 5) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(x[i]));
            if (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
            if (OP == 7) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(seed));
            if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(seed));
            if (OP == 9 && (i & 1) == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*reinterpret_cast<unsigned long long*>(&x[i])) : "v"(*reinterpret_cast<unsigned long long*>(&x[(i + 2) & 7])));
            if (OP == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
            if (OP == 11) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            if (OP == 12) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s ^= x[i];
    if (s == 0x12345u) out[threadIdx.x] = s;
}

template <int OP>
void run(const char* name, unsigned* out) {
    const int iters = 4096, grid = 256 * 4;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), 0, 0, out, 12345u, 16);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), 0, 0, out, 12345u, iters);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double per_simd = (double)iters * 8 * (grid * 4 / 1024);     // wave-instructions per SIMD (grid x 4 waves over 1024 SIMDs)
    printf("%-16s %.3f ms  %.1f wave-instructions per SIMD and us\n", name, ms, per_simd / (ms * 1e3));
}

int main() {
    unsigned* out;
    if (hipMalloc(&out, 4096) != hipSuccess) return 1;
    run<2>("v_xor_b32", out);
    run<6>("v_add_u32", out);
    run<0>("v_mul_lo_u32", out);
    run<4>("v_mul_hi_u32", out);
    run<1>("v_mul_u32_u24", out);
    run<3>("v_mad_u32_u24", out);
    run<5>("v_alignbit_b32", out);
    run<7>("v_perm_b32", out);
    run<8>("v_fma_f32", out);
    run<9>("v_pk_fma_f32 x1/2", out);     // half as many instructions per iteration: a rate equal to v_fma_f32's shows as half
    run<10>("v_rcp_f32", out);
    run<11>("v_exp_f32", out);
    run<12>("v_cvt_pk_bf16_f32", out);
    hipFree(out);
    return 0;
}
