// Depthwise 3x3 over an LDS halo tile, one pixel half (32 channels) per lane.  Shared by the forward
// trunk kernels and by the backward kernels (the transposed convolution is the same stencil with the
// taps flipped).
#pragma once
#include "ksr_gemm.h"

#if defined(__HIPCC__)

constexpr int HALO_W = 34, HALO_H = 6, HALO_LD = 68;   // 4 x 32 pixel tile + 1 pixel rim; 64 ch + 4 pad
constexpr int HALO_FLOATS = HALO_H * HALO_W * HALO_LD;

// Load channels [c0, c0+64) of the (6 x 34) pixel window whose interior starts at (y0, x0) of image b into an
// fp32 LDS tile.  src is NHWC with `ld` channels per pixel, stored as T; pixels outside the image are zero.
// All 256 threads call.
template <class T>
__device__ __forceinline__ void ksr_stage_halo(float* __restrict__ halo, const void* __restrict__ src, int ld,
                                               int c0, int b, int y0, int x0, int H, int W, int tid) {
    const long HW = (long)H * W;
    if constexpr (sizeof(T) == 4) {
        for (int c = tid; c < HALO_H * HALO_W * 16; c += 256) {
            const int q = c & 15, pix = c >> 4;
            const int hx = pix % HALO_W, hy = pix / HALO_W;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + ((long)b * HW + (long)gy * W + gx) * ld + c0 + 4 * q);
            *reinterpret_cast<float4*>(&halo[pix * HALO_LD + 4 * q]) = v;
        }
    } else {
        for (int c = tid; c < HALO_H * HALO_W * 8; c += 256) {
            const int q = c & 7, pix = c >> 3;
            const int hx = pix % HALO_W, hy = pix / HALO_W;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            uint4 w = make_uint4(0u, 0u, 0u, 0u);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                w = *reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(src) + ((long)b * HW + (long)gy * W + gx) * ld + c0 + 8 * q);
            float x[8];
            ksr_unpack8(w, x);
            *reinterpret_cast<float4*>(&halo[pix * HALO_LD + 8 * q]) = make_float4(x[0], x[1], x[2], x[3]);
            *reinterpret_cast<float4*>(&halo[pix * HALO_LD + 8 * q + 4]) = make_float4(x[4], x[5], x[6], x[7]);
        }
    }
}

// Stage depthwise weights [64][9] as [tap][channel]; FLIP stores tap 8-t (transposed convolution).
template <bool FLIP>
__device__ __forceinline__ void ksr_stage_dw(float* __restrict__ dww, const float* __restrict__ w, int tid) {
    for (int i = tid; i < 9 * 64; i += 256) {
        const int c = i & 63, tap = i >> 6;
        dww[i] = w[c * 9 + (FLIP ? 8 - tap : tap)];
    }
}

// g[i] += sum_taps dww[tap][32h+i] * halo[(row+dy, col+dx)][32h+i]   (row = wave index, col = tok)
__device__ __forceinline__ void ksr_dwconv_lane(const float* __restrict__ halo, const float* __restrict__ dww,
                                                int row, int col, int h, float (&g)[32]) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const float* hp = &halo[((row + dy) * HALO_W + (col + dx)) * HALO_LD + 32 * h];
            const float* wp = &dww[(dy * 3 + dx) * 64 + 32 * h];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 tv = *reinterpret_cast<const f32x4*>(hp + 4 * q);
                const f32x4 wq = *reinterpret_cast<const f32x4*>(wp + 4 * q);
                g[4 * q + 0] += wq[0] * tv[0]; g[4 * q + 1] += wq[1] * tv[1];
                g[4 * q + 2] += wq[2] * tv[2]; g[4 * q + 3] += wq[3] * tv[3];
            }
            // Pin the partial sums here: without it the compiler may sink a tap's FMAs far below its LDS
            // reads (seen when the stencil result is also stored to memory) and spill hundreds of VGPRs.
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(g[i]));
            __builtin_amdgcn_sched_barrier(0);   // one tap at a time: bounds live registers
        }
}

#endif
