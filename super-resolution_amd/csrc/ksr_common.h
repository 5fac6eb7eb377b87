// Shared device/host helpers for libksr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define KSR_OK 0
#define KSR_ERR_ARG (-1)
#define KSR_ERR_UNSUPPORTED (-2)
#define KSR_ERR_HIP (-3)
#define KSR_ERR_COMM (-4)

// Dropout stream id that changes without a new kernel argument (a recorded HIP graph replays its arguments): when the
// calling thread has named a device word (ksr_set_dropout_seed_source), every kernel that draws dropout masks adds
// word * odd constant to its by-value seed before use.  Forward and backward kernels of one step read the same word.
const unsigned long long* ksr_seed_dev();
#define KSR_MIX_SEED(a)                                                        \
    do {                                                                       \
        if ((a).seed_dev) (a).seed += *(a).seed_dev * 0x9E3779B97F4A7C15ULL;   \
    } while (0)

#define KSR_MAX_KNOTS 64
#define KSR_MAX_ORDER 3

// thread-local last-error text (read through ksr_last_error())
void ksr_set_error(const char* fmt, ...);

#define KSR_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ksr_set_error(__VA_ARGS__);          \
            return KSR_ERR_ARG;                  \
        }                                        \
    } while (0)

#define KSR_UNSUPPORTED(cond, ...)               \
    do {                                         \
        if (cond) {                              \
            ksr_set_error(__VA_ARGS__);          \
            return KSR_ERR_UNSUPPORTED;          \
        }                                        \
    } while (0)

#define KSR_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            ksr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return KSR_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)

// Workgroups of `kernel` (with `block` threads) that the current device holds at once: workgroups per CU by registers and
// LDS (hipOccupancyMaxActiveBlocksPerMultiprocessor) x CUs, looked up once per kernel (ksr_api.hip).  A persistent
// (grid-stride) kernel launched with more workgroups than that runs in rounds, and a last round that is not full leaves
// part of the chip idle for its whole length (768 workgroups at two per CU on 256 CUs: 512 + 256).
long ksr_wg_slots(const void* kernel, int block);
// Grid of a persistent kernel over ntiles tiles: every tile its own workgroup while they all fit at once, else exactly one
// full round (never more than cap, the size a caller's per-workgroup workspace allows).
template <class K>
inline unsigned ksr_persistent_grid(K kernel, long ntiles, long cap, int block = 256) {
    long slots = ksr_wg_slots(reinterpret_cast<const void*>(kernel), block);
    if (slots <= 0 || slots > cap) slots = cap;
    return (unsigned)(ntiles < slots ? ntiles : slots);
}

// Deferred slab reductions (ksr_reduce_defer / ksr_reduce_flush, ksr_wgrad.hip): whether the calling thread records its
// reductions, and the UNet weight-gradient kernels' share of the recorded list (ksr_unet_bwd.hip).
bool ksr_reduce_deferring();
int ksr_unet_reduce_flush(hipStream_t stream);
int ksr_unet_reduce_pending();
void ksr_unet_reduce_clear();

struct KsrKnots {
    float t[KSR_MAX_KNOTS];
    int M;  // number of knots = G + 2k + 1
    int k;  // spline order
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#if defined(__HIPCC__)

// 16-byte loads of activation / gradient tensors that a kernel reads once ("streams"): non-temporal loads.  Measured on
// MI355X with 512-MB tensors and the persistent grids these kernels use (512 workgroups): 7.1 TB/s against 4.2 TB/s for
// plain loads (6.5 at 1024 workgroups); they are still served by the L2, so neighbouring tiles share their halo rows.
typedef float ksr_f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned ksr_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ksr_ld_stream(const float4* p) {
    const ksr_f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const ksr_f32x4_t*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint4 ksr_ld_stream(const uint4* p) {
    const ksr_u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const ksr_u32x4_t*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------------------------
// B-spline basis, local form.  For x in [t_m, t_{m+1}) only the k+1 functions B_{m-k..m} are
// non-zero.  N[a] (a = 0..K) receives B_{m-K+a, K}(x); entries whose index falls outside
// [0, M-2-K] are 0 and must not be used.  The arithmetic follows the textbook recursion
//   B_{j,p} = ((x - t_j) / (t_{j+p} - t_j)) * B_{j,p-1} + ((t_{j+p+1} - x) / (t_{j+p+1} - t_{j+1})) * B_{j+1,p-1}
// with separately rounded divide, multiply and add (no FMA contraction), so the values are
// bit-identical to an fp32 evaluation of the dense recursion.  dN (optional) is d/dx with the
// order-0 indicator held constant.  Returns m, or -1 when x lies in no interval (all zero).
template <int K, bool WITH_D>
__device__ __forceinline__ int ksr_bspline_local(float x, const float* __restrict__ t, int M,
                                                 float (&N)[K + 1], float (&dN)[K + 1]) {
#pragma clang fp contract(off)
    int cnt = 0;
    for (int j = 0; j < M; ++j) cnt += (x >= t[j]) ? 1 : 0;
    const int m = cnt - 1;
#pragma unroll
    for (int a = 0; a <= K; ++a) { N[a] = 0.f; dN[a] = 0.f; }
    if (m < 0 || m >= M - 1 || !(x < t[m + 1])) return -1;
    // order 0: single non-zero at local slot 0 (j = m); keep values right-aligned per order:
    // at order p the live slots are [0..p] with j = m - p + a.
    N[0] = 1.f;
#pragma unroll
    for (int p = 1; p <= K; ++p) {
#pragma unroll
        for (int a = p; a >= 0; --a) {
            const int j = m - p + a;
            const bool valid = (j >= 0) && (j <= M - 2 - p);
            const int jc = valid ? j : 0;
            const float tj = t[jc], tjp = t[jc + p], tjp1 = t[jc + p + 1], tj1 = t[jc + 1];
            const float oldL = (a >= 1) ? N[a - 1] : 0.f;   // B_{j,p-1}
            const float oldR = (a <= p - 1) ? N[a] : 0.f;   // B_{j+1,p-1}
            const float d1 = __fsub_rn(tjp, tj), d2 = __fsub_rn(tjp1, tj1);
            const float L = __fdiv_rn(__fsub_rn(x, tj), d1);
            const float R = __fdiv_rn(__fsub_rn(tjp1, x), d2);
            float v = __fadd_rn(__fmul_rn(L, oldL), __fmul_rn(R, oldR));
            if (WITH_D) {
                const float doldL = (a >= 1) ? dN[a - 1] : 0.f;
                const float doldR = (a <= p - 1) ? dN[a] : 0.f;
                float dv = __fdiv_rn(oldL, d1) + L * doldL - __fdiv_rn(oldR, d2) + R * doldR;
                dN[a] = valid ? dv : 0.f;
            }
            N[a] = valid ? v : 0.f;
        }
    }
    return m;
}

// Dropout keep decision for element base + i of a stream (base even, i a small unrolled index): one hash serves the
// two elements of a pair, 16 bits each, so the keep probability is quantised to 1/65536.  Counter-based
// (a 32-bit avalanche hash of seed, stream and element index) and stateless, so the backward regenerates exactly
// the masks of the forward.
__device__ __forceinline__ bool ksr_keep(unsigned long long seed, unsigned stream, unsigned long long base, int i, float keep) {
    const unsigned s32 = (unsigned)seed * 0x9E3779B1u ^ (unsigned)(seed >> 32) * 0x85EBCA77u ^ (stream * 0xC2B2AE3Du + 0x27D4EB2Fu);
    const unsigned long long pair = (base >> 1) + (unsigned)(i >> 1);
    unsigned x = (unsigned)pair * 0x9E3779B1u ^ (unsigned)(pair >> 32) * 0x85EBCA77u ^ s32;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    const unsigned half = (i & 1) ? (x >> 16) : (x & 0xffffu);
    return half < (unsigned)(keep * 65536.0f);
}

__device__ __forceinline__ unsigned ksr_hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
// Dropout on the attention probabilities: one 32-bit base per (query token, head), mixed with the key's
// image index.  Forward and both backward roles (query side, key side) derive the same bit from it.
__device__ __forceinline__ unsigned ksr_attn_drop_base(unsigned long long seed, unsigned long long token, int head) {
    const unsigned long long idx = token * 2ULL + (unsigned)head;
    const unsigned s32 = (unsigned)seed * 0x9E3779B1u ^ (unsigned)(seed >> 32) * 0x85EBCA77u ^ 0x27D4EB2Fu;
    return ksr_hash32((unsigned)idx * 0x9E3779B1u ^ (unsigned)(idx >> 32) * 0x85EBCA77u ^ s32);
}
// Workgroups are handed to the 8 XCDs round-robin and every XCD has its own L2.  Maps the hardware workgroup index to
// a logical index such that each XCD works through one contiguous range of [0, total): neighbouring tiles, which
// share halo rows and cache lines, then share an L2.
__device__ __forceinline__ unsigned ksr_xcd_contiguous(unsigned wg, unsigned total) {
    const unsigned per = total >> 3, rem = total & 7u, xcd = wg & 7u, slot = wg >> 3;
    return xcd < rem ? xcd * (per + 1) + slot : rem * (per + 1) + (xcd - rem) * per + slot;
}
// Tile walk of a persistent stencil kernel: for (t = first; t < end; t += step).  XCD x works through the contiguous eighth
// [x per, (x+1) per) of the tiles, and its resident workgroups take NEIGHBOURING tiles at any one time, so the halo rows
// and columns that adjacent tiles share are fetched by one L2 once (a plain grid-stride walk hands neighbouring tiles to
// different XCDs: every L2 fetches the halo again).  A grid that is no multiple of 8 falls back to the grid-stride walk.
struct KsrTileWalk { long first, end, step; };
__device__ __forceinline__ KsrTileWalk ksr_xcd_tile_walk(long ntiles) {
    const unsigned G = gridDim.x;
    if (G & 7u) return {(long)blockIdx.x, ntiles, (long)G};
    const long per = (ntiles + 7) >> 3;
    const long lo = (long)(blockIdx.x & 7u) * per, hi = lo + per < ntiles ? lo + per : ntiles;
    return {lo + (long)(blockIdx.x >> 3), hi, (long)(G >> 3)};
}
// The same walk behind a compile-time switch, for kernels whose template instances are partly stencils: with XCD = false
// first() / step() are the hardware indices themselves, read where they are used (keeping them in registers of their own
// made a 254-register instance spill).
template <bool XCD>
struct KsrTileWalkT {
    long first_, end_, step_;
    __device__ __forceinline__ explicit KsrTileWalkT(long ntiles) : first_(0), end_(ntiles), step_(0) {
        if constexpr (XCD) { const KsrTileWalk w = ksr_xcd_tile_walk(ntiles); first_ = w.first; end_ = w.end; step_ = w.step; }
    }
    __device__ __forceinline__ long first() const { if constexpr (XCD) return first_; else return (long)blockIdx.x; }
    __device__ __forceinline__ long end() const { return end_; }
    __device__ __forceinline__ long step() const { if constexpr (XCD) return step_; else return (long)gridDim.x; }
};

// One hash serves two neighbouring keys (16 bits each): the keep probability is quantised to 1/65536.  `base` is
// already a full hash of (seed, token, head); the key index enters as a Weyl step followed by ONE xorshift-multiply
// round (the attention kernels are bound by instruction issue: five instructions per pair of keys instead of nine).  Measured
// on 10^6 bases x 64 keys: mean keep 0.89999, variance of the per-query count 5.75-5.76 (binomial 5.76), largest
// pairwise correlation between keys 0.005 (sampling noise 0.004).
__device__ __forceinline__ unsigned ksr_attn_keep_word(unsigned base, int key_b) {   // low half: key_b & ~1, high half: key_b | 1
    unsigned x = base + (unsigned)(key_b >> 1) * 0x9E3779B1u;
    x ^= x >> 15; x *= 0xD168AAADu; x ^= x >> 15;
    return x;
}
__device__ __forceinline__ unsigned ksr_attn_keep_threshold(float keep) { return (unsigned)(keep * 65536.0f); }
__device__ __forceinline__ bool ksr_attn_keep(unsigned base, int key_b, float keep) {
    const unsigned x = ksr_attn_keep_word(base, key_b);
    const unsigned half = (key_b & 1) ? (x >> 16) : (x & 0xffffu);
    return half < ksr_attn_keep_threshold(keep);
}

// One LDS-DMA instruction: lane l moves 16 bytes from its own global address to lds_wave_base + 16 l.  Issued from inline
// assembly (M0 saved and restored around it): the compiler then keeps no record of transfers in flight, where it would
// otherwise make every read of an LDS array that no transfer ever targets (statistics, transposer tiles) wait for ALL of
// them.  Every wait for such a transfer is therefore explicit (ksr_dma_barrier, ksr_dma_barrier_but).
template <bool NT>
__device__ __forceinline__ void ksr_dma16(const void* gsrc, void* lds_wave_base) {
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    unsigned keep_m0;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep_m0) : "s"(base), "v"(gsrc) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep_m0) : "s"(base), "v"(gsrc) : "memory");
}

// The 4-byte form: lane l moves one dword from its own global address to lds_wave_base + 4 l.
__device__ __forceinline__ void ksr_dma4(const void* gsrc, void* lds_wave_base) {
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0) : "s"(base), "v"(gsrc) : "memory");
}
// This wave's transfers have landed, except that its YOUNGER most recent vector-memory instructions (stores issued after
// the transfers) may still be in flight; no workgroup barrier (wave-private staging).  See ksr_dma_barrier_but.
template <int YOUNGER>
__device__ __forceinline__ void ksr_dma_wait_but() {
    static_assert(YOUNGER >= 0 && YOUNGER < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
}

// every wave's LDS writes are complete and visible, every wave has arrived
__device__ __forceinline__ void ksr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// as ksr_lds_barrier, and this wave's transfers (and stores) have completed before it arrives
__device__ __forceinline__ void ksr_dma_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// as ksr_dma_barrier, but the wave's YOUNGEST vector-memory instructions may still be in flight: vmcnt counts loads and
// stores of a wave in issue order on this target (the compiler itself waits for a load past younger stores with
// vmcnt(n > 0)), so "at most YOUNGER outstanding" means every transfer issued before them has landed.  Only usable
// where the number of instructions issued after the transfers is exact (no store skipped by a branch).
template <int YOUNGER>
__device__ __forceinline__ void ksr_dma_barrier_but() {
    static_assert(YOUNGER >= 0 && YOUNGER < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(YOUNGER) : "memory");
}

// Stores of tensors that are read again only much later (the backward's saved tensors): non-temporal, so that they do not
// displace what the next kernels are about to read.  Measured on the in-projection + attention forward: 403 MB of saved
// q, k, v cost 70 us as plain stores and 31 us non-temporal.
template <bool NT>
__device__ __forceinline__ void ksr_st16(float4* p, const float4& v) {
    if constexpr (NT) {
        typedef float f32x4_nt __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_nt*>(p));
    } else {
        *p = v;
    }
}
template <bool NT>
__device__ __forceinline__ void ksr_st16(uint4* p, const uint4& v) {
    if constexpr (NT) {
        typedef unsigned u32x4_nt __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(u32x4_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_nt*>(p));
    } else {
        *p = v;
    }
}

// 1 / (1 + exp(-x)) with the hardware exp2 and reciprocal (1 ulp each) instead of expf and an IEEE division: about 5
// instructions instead of 15 per value, in kernels that are bound by instruction issue
__device__ __forceinline__ float ksr_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float ksr_silu(float x) { return x * ksr_sigmoid(x); }

#endif  // __HIPCC__
