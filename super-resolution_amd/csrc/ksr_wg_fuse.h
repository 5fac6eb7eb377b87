// Weight gradient of a 64 -> 64 1x1 conv accumulated inside the data-gradient kernel that already holds both of its
// operands in registers: dW[j][k] += sum_tok dz[tok][j] * a[tok][k], db[j] += sum_tok dz[tok][j].
// A wave owns 32 tokens in half layout (ksr_gemm.h).  The contraction runs over tokens, so both operands have to be
// read "transposed" (8 tokens of one channel per lane): the wave writes them as bf16 row-major tiles [32 tokens][64
// channels] into its private LDS transposer area (2 x 4 KB) and reads them back with ds_read_b64_tr_b16; four
// v_mfma_f32_32x32x16_bf16 per 16 tokens keep the 64 x 64 sums of the wave in 64 accumulator registers.  At the end of
// the kernel the four waves' sums meet in LDS in a fixed order and the workgroup writes one partial row; a fixed-order
// reduction over the workgroups finishes the gradient (deterministic, no atomics).
#pragma once
#include "ksr_gemm.h"

#if defined(__HIPCC__)

constexpr int KSR_WGF_ROW = 64 * 64 + 64;        // floats per partial row: dW [64][64] | db [64]
constexpr int KSR_WGF_MAX_GRID = 768;            // most workgroups a data-gradient kernel launches

// Reduction of G partial rows [J*K + J] to dW [J][K] and db [J] (ksr_wgrad.hip).
int ksr_slab_reduce(const float* slab, int G, int J, int K, float* dW, float* db, hipStream_t stream);
// The same for partial rows of `stride` floats that hold [J*K | J] at their start (several gradients per row).
int ksr_slab_reduce_strided(const float* slab, int G, int J, int K, long stride, float* dW, float* db, hipStream_t stream);
// out[j * out_row_stride + k] = the same sum of a [J][K] block at the start of every partial row (no bias part).
int ksr_slab_reduce_rows(const float* slab, int G, int J, int K, long stride, float* out, int out_row_stride, hipStream_t stream);

// 16-byte chunk c (8 bf16) of row r sits at chunk position c ^ sw(r): row writes (a lane writes 4 chunks of its own row),
// the transposing reads (4 rows x 16 channels per 16-lane group) and the column reads are then free of bank conflicts.
__device__ __forceinline__ int ksr_wgf_off(int r, int c) { return r * 64 + 8 * (c ^ ((((r >> 1) & 1) << 2) | ((r >> 2) & 3))); }

struct KsrWgFused {
    f32x16 acc[2][2];       // [row block jb][column block kb]: rows j = 32 jb + (q&3) + 8 (q>>2) + 4 (lane>>5), column k = 32 kb + (lane&31)
    float bsum;             // lane c: sum over tokens of dz[tok][c]
};

__device__ __forceinline__ void ksr_wgf_init(KsrWgFused& w) {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int q = 0; q < 16; ++q) w.acc[jb][kb][q] = 0.f;
    w.bsum = 0.f;
}

// dz, av: this lane's 32 channels (32 (lane>>5) ..) of token lane&31; rows past the end of the tensor must hold zeros.
// xb: the wave's private LDS area of KSR_XPOSE_FLOATS floats.
__device__ __forceinline__ void ksr_wgf_accumulate(float* __restrict__ xb, int lane, const float (&dz)[32],
                                                   const float (&av)[32], KsrWgFused& w) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    __bf16* tz = reinterpret_cast<__bf16*>(xb);
    __bf16* ta = tz + 32 * 64;
    const int tok = lane & 31, h = lane >> 5;
    ksr_wave_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<uint4*>(tz + ksr_wgf_off(tok, 4 * h + q)) = ksr_pack8(dz + 8 * q);
        *reinterpret_cast<uint4*>(ta + ksr_wgf_off(tok, 4 * h + q)) = ksr_pack8(av + 8 * q);
    }
    ksr_wave_fence();
    // operand element j of lane half g in 16-token step s: token 16 s + 8 (j>>2) + 4 g + (j&3), the same for both operands
    const int g = lane >> 5, db = (lane >> 4) & 1, qq = (lane & 15) >> 2, p = lane & 3;
    auto tr = [&](const __bf16* tile, int row, int chunk) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(tile + ksr_wgf_off(row, chunk) + 4 * (p & 1)));
    };
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r0 = 16 * s + 4 * g + qq;
        union { s16x4 hv[2]; bf16x8 v; } A[2], Bv[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int chunk = 4 * blk + 2 * db + (p >> 1);
            A[blk].hv[0] = tr(tz, r0, chunk); A[blk].hv[1] = tr(tz, r0 + 8, chunk);
            Bv[blk].hv[0] = tr(ta, r0, chunk); Bv[blk].hv[1] = tr(ta, r0 + 8, chunk);
        }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                w.acc[jb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[jb].v, Bv[kb].v, w.acc[jb][kb], 0, 0, 0);
    }
    // bias gradient: lane c adds up column c of the dz tile (the values the products saw)
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 32; r += 2) {
        s0 += (float)tz[ksr_wgf_off(r, lane >> 3) + (lane & 7)];
        s1 += (float)tz[ksr_wgf_off(r + 1, lane >> 3) + (lane & 7)];
    }
    w.bsum += s0 + s1;
    ksr_wave_fence();
}

// Called by all 256 threads after the tile loop.  area: KSR_WGF_ROW floats of LDS that every wave may overwrite now.
__device__ __forceinline__ void ksr_wgf_finish(float* __restrict__ area, float* __restrict__ slab_row, int tid,
                                               const KsrWgFused& w) {
    const int lane = tid & 63, wv = tid >> 6, col = lane & 31, g = lane >> 5;
    for (int turn = 0; turn < 4; ++turn) {
        __syncthreads();
        if (wv == turn) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int idx = (32 * jb + (q & 3) + 8 * (q >> 2) + 4 * g) * 64 + 32 * kb + col;
                        area[idx] = (turn == 0 ? 0.f : area[idx]) + w.acc[jb][kb][q];
                    }
            area[64 * 64 + lane] = (turn == 0 ? 0.f : area[64 * 64 + lane]) + w.bsum;
        }
    }
    __syncthreads();
    for (int i = tid; i < KSR_WGF_ROW; i += 256) slab_row[i] = area[i];
}

// The same scheme for products with a 32-channel operand (the transformer's linear1: 32 x 64, linear2: 64 x 32): JB / KB =
// number of 32-channel blocks of dz (rows of dW) / of av (columns).  A 32-channel tensor sits in the half layout of its own:
// lane half h holds channels 16 h .. 16 h + 15 of token lane & 31 (16 values per lane), and fills chunks 0..3 of its tile rows.
template <int JB, int KB>
struct KsrWgFusedT {
    f32x16 acc[JB][KB];     // rows j = 32 jb + (q&3) + 8 (q>>2) + 4 (lane>>5), column k = 32 kb + (lane&31)
    float bsum;             // lane c < 32 JB: sum over tokens of dz[tok][c]
};

template <int JB, int KB>
__device__ __forceinline__ void ksr_wgft_init(KsrWgFusedT<JB, KB>& w) {
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int q = 0; q < 16; ++q) w.acc[jb][kb][q] = 0.f;
    w.bsum = 0.f;
}

// one operand (NB blocks of 32 channels, 16 NB values per lane) -> the wave's bf16 tile
template <int NB>
__device__ __forceinline__ void ksr_wgft_put(__bf16* __restrict__ tile, int lane, const float (&v)[16 * NB]) {
    const int tok = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 2 * NB; ++q) *reinterpret_cast<uint4*>(tile + ksr_wgf_off(tok, 2 * NB * h + q)) = ksr_pack8(v + 8 * q);
}

template <int JB, int KB>
__device__ __forceinline__ void ksr_wgft_accumulate(float* __restrict__ xb, int lane, const float (&dz)[16 * JB],
                                                    const float (&av)[16 * KB], KsrWgFusedT<JB, KB>& w) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    __bf16* tz = reinterpret_cast<__bf16*>(xb);
    __bf16* ta = tz + 32 * 64;
    ksr_wave_fence();
    ksr_wgft_put<JB>(tz, lane, dz);
    ksr_wgft_put<KB>(ta, lane, av);
    ksr_wave_fence();
    const int g = lane >> 5, db = (lane >> 4) & 1, qq = (lane & 15) >> 2, p = lane & 3;
    auto tr = [&](const __bf16* tile, int row, int chunk) -> s16x4 {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(tile + ksr_wgf_off(row, chunk) + 4 * (p & 1)));
    };
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r0 = 16 * s + 4 * g + qq;
        union { s16x4 hv[2]; bf16x8 v; } A[JB], Bv[KB];
#pragma unroll
        for (int blk = 0; blk < JB; ++blk) {
            const int chunk = 4 * blk + 2 * db + (p >> 1);
            A[blk].hv[0] = tr(tz, r0, chunk); A[blk].hv[1] = tr(tz, r0 + 8, chunk);
        }
#pragma unroll
        for (int blk = 0; blk < KB; ++blk) {
            const int chunk = 4 * blk + 2 * db + (p >> 1);
            Bv[blk].hv[0] = tr(ta, r0, chunk); Bv[blk].hv[1] = tr(ta, r0 + 8, chunk);
        }
#pragma unroll
        for (int jb = 0; jb < JB; ++jb)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                w.acc[jb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[jb].v, Bv[kb].v, w.acc[jb][kb], 0, 0, 0);
    }
    // bias gradient: lane c adds up column c of the dz tile (the values the products saw)
    if (lane < 32 * JB) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 32; r += 2) {
            s0 += (float)tz[ksr_wgf_off(r, lane >> 3) + (lane & 7)];
            s1 += (float)tz[ksr_wgf_off(r + 1, lane >> 3) + (lane & 7)];
        }
        w.bsum += s0 + s1;
    }
    ksr_wave_fence();
}

// Called by all 256 threads after the tile loop.  area: (32 JB) * (32 KB) + 32 JB floats of LDS that every wave may
// overwrite now; slab_row receives dW [32 JB][32 KB] | db [32 JB].
template <int JB, int KB>
__device__ __forceinline__ void ksr_wgft_finish(float* __restrict__ area, float* __restrict__ slab_row, int tid,
                                                const KsrWgFusedT<JB, KB>& w) {
    constexpr int J = 32 * JB, K = 32 * KB;
    const int lane = tid & 63, wv = tid >> 6, col = lane & 31, g = lane >> 5;
    for (int turn = 0; turn < 4; ++turn) {
        __syncthreads();
        if (wv == turn) {
#pragma unroll
            for (int jb = 0; jb < JB; ++jb)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int idx = (32 * jb + (q & 3) + 8 * (q >> 2) + 4 * g) * K + 32 * kb + col;
                        area[idx] = (turn == 0 ? 0.f : area[idx]) + w.acc[jb][kb][q];
                    }
            if (lane < J) area[J * K + lane] = (turn == 0 ? 0.f : area[J * K + lane]) + w.bsum;
        }
    }
    __syncthreads();
    for (int i = tid; i < J * K + J; i += 256) slab_row[i] = area[i];
}

#endif
