// Depthwise 3x3 over an LDS halo tile, one pixel half (32 channels) per lane.  Shared by the forward
// trunk kernels and by the backward kernels (the transposed convolution is the same stencil with the
// taps flipped).
#pragma once
#include "ksr_gemm.h"

#if defined(__HIPCC__)

constexpr int HALO_W = 34, HALO_H = 6, HALO_LD = 68;   // 4 x 32 pixel tile + 1 pixel rim; 64 ch + 4 pad
constexpr int HALO_FLOATS = HALO_H * HALO_W * HALO_LD;

// Channels [c0, c0+64) of the (6 x 34) pixel window whose interior starts at (y0, x0) of image b, as an fp32 LDS
// tile.  src is NHWC with `ld` channels per pixel, stored as T; pixels outside the image are zero.  Two phases, both
// called by all 256 threads: ksr_halo_fetch issues every global load of the thread (they are in flight together, and
// may stay in flight across other work), ksr_halo_commit converts and writes them to LDS.
template <class T> struct KsrHalo { float4 v[(HALO_H * HALO_W * 16 + 255) / 256]; };
template <> struct KsrHalo<__bf16> { uint4 v[(HALO_H * HALO_W * 8 + 255) / 256]; };

template <class T>
__device__ __forceinline__ void ksr_halo_fetch(const void* __restrict__ src, int ld, int c0, int b, int y0, int x0, int H,
                                               int W, int tid, KsrHalo<T>& r) {
    const long HW = (long)H * W;
    constexpr int PER = sizeof(T) == 4 ? 16 : 8;                 // 16-byte pieces per pixel
    constexpr int NIT = (HALO_H * HALO_W * PER + 255) / 256;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = tid + 256 * it;
        const int q = c % PER, pix = c / PER;
        const int hx = pix % HALO_W, hy = pix / HALO_W;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool in = c < HALO_H * HALO_W * PER && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const long e = ((long)b * HW + (long)gy * W + gx) * ld + c0;
        if constexpr (sizeof(T) == 4) {
            r.v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in) r.v[it] = ksr_ld_stream(reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + e + 4 * q));
        } else {
            r.v[it] = make_uint4(0u, 0u, 0u, 0u);
            if (in) r.v[it] = ksr_ld_stream(reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(src) + e + 8 * q));
        }
    }
}

template <class T>
__device__ __forceinline__ void ksr_halo_commit(float* __restrict__ halo, int tid, const KsrHalo<T>& r) {
    constexpr int PER = sizeof(T) == 4 ? 16 : 8;
    constexpr int NIT = (HALO_H * HALO_W * PER + 255) / 256;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = tid + 256 * it;
        const int q = c % PER, pix = c / PER;
        if (c >= HALO_H * HALO_W * PER) continue;
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(&halo[pix * HALO_LD + 4 * q]) = r.v[it];
        } else {
            float x[8];
            ksr_unpack8(r.v[it], x);
            *reinterpret_cast<float4*>(&halo[pix * HALO_LD + 8 * q]) = make_float4(x[0], x[1], x[2], x[3]);
            *reinterpret_cast<float4*>(&halo[pix * HALO_LD + 8 * q + 4]) = make_float4(x[4], x[5], x[6], x[7]);
        }
    }
}

template <class T>
__device__ __forceinline__ void ksr_stage_halo(float* __restrict__ halo, const void* __restrict__ src, int ld,
                                               int c0, int b, int y0, int x0, int H, int W, int tid) {
    KsrHalo<T> r;
    ksr_halo_fetch<T>(src, ld, c0, b, y0, x0, H, W, tid, r);
    ksr_halo_commit<T>(halo, tid, r);
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

// The same with the depthwise weights kept in the PADDING of the halo tile (4 floats behind each pixel's 64 channels: element
// k of the [tap][channel] table sits at halo[(k >> 2) * HALO_LD + 64 + (k & 3)]; 816 slots for 576 weights).  Nothing that
// writes pixels or uses the tile as transposers touches the padding, so the table is staged once per workgroup.  For the
// split-fp32 instances, whose three weight planes leave no room for a table of its own.
template <bool FLIP>
__device__ __forceinline__ void ksr_stage_dw_pad(float* __restrict__ halo, const float* __restrict__ w, int tid) {
    static_assert(HALO_H * HALO_W * 4 >= 9 * 64, "the padding holds the table");
    for (int i = tid; i < 9 * 64; i += 256) {
        const int c = i & 63, tap = i >> 6;
        halo[(i >> 2) * HALO_LD + 64 + (i & 3)] = w[c * 9 + (FLIP ? 8 - tap : tap)];
    }
}
__device__ __forceinline__ void ksr_dwconv_lane_pad(const float* __restrict__ halo, int row, int col, int h, float (&g)[32]) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const float* hp = &halo[((row + dy) * HALO_W + (col + dx)) * HALO_LD + 32 * h];
            const float* wp = &halo[(((dy * 3 + dx) * 64 + 32 * h) >> 2) * HALO_LD + 64];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 tv = *reinterpret_cast<const f32x4*>(hp + 4 * q);
                const f32x4 wq = *reinterpret_cast<const f32x4*>(wp + q * HALO_LD);
                g[4 * q + 0] += wq[0] * tv[0]; g[4 * q + 1] += wq[1] * tv[1];
                g[4 * q + 2] += wq[2] * tv[2]; g[4 * q + 3] += wq[3] * tv[3];
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(g[i]));
            __builtin_amdgcn_sched_barrier(0);
        }
}

#endif
