// Wave-level "token per lane" GEMM primitive for gfx950 (fp32 in / fp32 accumulate MFMA).
//
// Every 1x1 convolution / linear layer of the hyper-encoder has 16..192 channels, so one
// pixel's ("token's") whole activation vector fits in registers.  A wave owns 32 tokens; lane
// l = (tok = l & 31, h = l >> 5) holds HALF of its token's channels ("half layout", HL):
//   64-channel tensor : lane half h holds channels 32h .. 32h+31, register s <-> channel 32h+s
//   64*g-channel tensor: g groups of that shape (register 32*grp + s <-> channel 64*grp+32h+s)
//
// The product is computed transposed, D[row = out channel][col = token] = W * X^T, with
// v_mfma_f32_32x32x2_f32: the weight matrix is the A operand (rows on lanes, read from LDS in a
// pre-arranged fragment order), the activations are the B operand (one VGPR per k-step, straight
// from the registers above).  D has the token on the lane and 16 rows per register block; the
// row->channel assignment is chosen when the fragments are staged so that the OUTPUT is in half
// layout again: a chain of layers never leaves the register file and never transposes.
//
// Exact fp32: the MFMA is a k-ordered fmaf chain (one rounding per product).
#pragma once
#include "ksr_common.h"
#include <type_traits>

#if defined(__HIPCC__)

#define KSR_WAVE 64

// MFMA operand precision of ksr_wave_gemm: exact fp32 (v_mfma_f32_32x32x2_f32) or bf16 operands with fp32
// accumulation (v_mfma_f32_32x32x16_bf16, 16x the rate).  Storage in HBM and all other arithmetic stay fp32.
enum { KSR_F32 = 0, KSR_BF16 = 1, KSR_BF16S = 2,      // KSR_BF16S: bf16 MFMA operands AND bf16 activation storage
       // fp32 products on the bf16 matrix cores: both operands are split exactly into three bf16 terms (8 + 8 + 8 = 24
       // significand bits), six of the nine partial products are formed (those below 2^-24 of the product are dropped), each
       // exact in the fp32 accumulator.  v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate and its time adds to the vector
       // instructions' of a kernel; the six bf16 instructions take 3/8 of its time on a pipe of their own.  A template
       // argument of the forward kernels that have an instance for it, never a value of ksr_prec() (ksr_f32_split()).
       KSR_F32X3 = 3 };
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
int ksr_prec();   // the calling thread's mode (ksr_set_precision; thread-local, a new thread starts in KSR_F32)
// Runs CALL(PREC) for the current mode; CALL is a macro taking the mode constant.
#define KSR_DISPATCH_PREC(CALL)                                   \
    do {                                                          \
        switch (ksr_prec()) {                                     \
            case KSR_BF16S: CALL(KSR_BF16S); break;               \
            case KSR_BF16: CALL(KSR_BF16); break;                 \
            default: CALL(KSR_F32); break;                        \
        }                                                         \
    } while (0)

// D row of a 32x32 MFMA block held by (lane half h, register q): (q&3) + 8*(q>>2) + 4*h.
// Inverse: for A-operand row rho, the result lands on half (rho>>2)&1, register (rho&3)+4*(rho>>3).
__device__ __forceinline__ int ksr_row_half(int rho) { return (rho >> 2) & 1; }
__device__ __forceinline__ int ksr_row_reg(int rho) { return (rho & 3) + 4 * (rho >> 3); }

// ---- channel maps: (lane half h, per-half index e) -> channel, or -1 for padding -------------
struct MapHL {       // channels multiple of 64, nh = C/2 entries per half
    int nh;
    __device__ __forceinline__ int operator()(int h, int e) const {
        return e < nh ? 64 * (e >> 5) + 32 * h + (e & 31) : -1;
    }
};
struct MapSplit {    // C channels, half h holds [h*C/2, (h+1)*C/2)
    int nh;
    __device__ __forceinline__ int operator()(int h, int e) const { return e < nh ? h * nh + e : -1; }
};
// Per-pixel KAN parameter vector: half h evaluates outputs o in [h*Oh, (h+1)*Oh), Oh = O/2, and holds
// for each of them (coef[0..Kc-1], uw, rw) at e = ol*(Kc+2) + gi.  Channel numbering of the
// reference: coef (o*Kc + g) | uw (O*Kc + o) | rw (O*Kc + O + o)   (in_dim = 1).
struct MapKan {
    int O, Kc;
    __device__ __forceinline__ int operator()(int h, int e) const {
        const int Oh = O >> 1, per = Kc + 2;
        if (e >= Oh * per) return -1;
        const int ol = e / per, gi = e - ol * per, o = h * Oh + ol;
        return gi < Kc ? o * Kc + gi : (gi == Kc ? O * Kc + o : O * Kc + O + o);
    }
};

// Number of floats of LDS a staged GEMM needs: NB blocks x KH steps x 64 lanes (+ NB*32 bias).
__host__ __device__ constexpr int ksr_frag_floats(int KH, int NB) { return NB * KH * 64; }
// ... for a given mode: fp32 fragments, one bf16 plane, or the three planes of the split form
template <int PREC> __host__ __device__ constexpr int ksr_frag_floats_p(int KH, int NB) {
    return PREC == KSR_F32 ? NB * KH * 64 : (PREC == KSR_F32X3 ? NB * KH * 96 : NB * KH * 32);
}
bool ksr_f32_split();   // the calling thread's choice for fp32 forward kernels that have a split instance (ksr_set_f32_split)

// x = t[0] + t[1] + t[2] with three bf16 terms (exact to 24 significand bits)
__device__ __forceinline__ void ksr_split3(float x, __bf16 (&t)[3]) {
    t[0] = (__bf16)x;
    float r = x - (float)t[0];
    t[1] = (__bf16)r;
    r -= (float)t[1];
    t[2] = (__bf16)r;
}
__host__ __device__ constexpr int ksr_bias_floats(int NB) { return NB * 32; }

// Stage W (row-major [rows][ldw]; element (o, i) at W[o*ldw+i], or W[i*ldw+o] when TRANS) into
// fragment order.  frag[((blk*(KH/4) + s/4)*64 + lane)*4 + s%4] = W(out(h(rho), 16*blk+q(rho)), in(hh, s))
// for lane = rho + 32*hh.  biasfrag[(blk*2 + h)*16 + q] = bias[out(h, 16*blk+q)].
// Called by all threads of the workgroup; caller synchronises afterwards.
template <int PREC, int KH, int NB, bool TRANS, int CH = 16, class OutMap, class InMap>
__device__ __forceinline__ void ksr_stage_weights(float* __restrict__ frag, float* __restrict__ biasfrag,
                                                  const float* __restrict__ W, int ldw,
                                                  const float* __restrict__ bias, OutMap om, InMap im,
                                                  int tid, int nthreads) {
    static_assert(KH % 8 == 0, "KH must be a multiple of 8");
    constexpr int total = NB * KH * 64;
    __bf16* frag16 = reinterpret_cast<__bf16*>(frag);
    // Sixteen elements per thread are requested before the first is converted and stored: the gather is one 4-byte load
    // per element, and element by element (the compiler kept four in flight) a 64 x 64 weight took eight dependent trips
    // to the L2 at the head of every launch of every kernel with weights - 5 of the ~9 us a launch costs before its first
    // tile, whatever the batch.  (CH = 1 is the element-by-element loop: for a kernel at the register limit whose allocation
    // the wider prologue upsets.)
    for (int base = tid; base < total; base += CH * nthreads) {
        float v[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int idx = base + u * nthreads;
            v[u] = 0.f;
            if (idx < total) {
                // fp32: idx = ((blk*(KH/4) + s/4)*64 + lane)*4 + s%4 ; bf16: ((blk*(KH/8) + s/8)*64 + lane)*8 + s%8
                constexpr int G = (PREC != KSR_F32) ? 8 : 4;
                const int s_lo = idx % G;
                const int lane = (idx / G) & 63;
                const int rest = idx / (G * 64);
                const int s = (rest % (KH / G)) * G + s_lo;
                const int blk = rest / (KH / G);
                const int rho = lane & 31, hh = lane >> 5;
                const int o = om(ksr_row_half(rho), 16 * blk + ksr_row_reg(rho));
                const int i = im(hh, s);
                if (o >= 0 && i >= 0) v[u] = TRANS ? W[(long)i * ldw + o] : W[(long)o * ldw + i];
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int idx = base + u * nthreads;
            if (idx >= total) continue;
            if (PREC == KSR_F32X3) {
                __bf16 t[3];
                ksr_split3(v[u], t);
                frag16[idx] = t[0]; frag16[total + idx] = t[1]; frag16[2 * total + idx] = t[2];
            } else if (PREC != KSR_F32) frag16[idx] = (__bf16)v[u];
            else frag[idx] = v[u];
        }
    }
    if (biasfrag) {
        for (int idx = tid; idx < NB * 32; idx += nthreads) {
            const int q = idx & 15, h = (idx >> 4) & 1, blk = idx >> 5;
            const int o = om(h, 16 * blk + q);
            biasfrag[idx] = (bias && o >= 0) ? bias[o] : 0.f;
        }
    }
}

// acc[blk] = bias + sum_s W_frag * x.  x[s] is this lane's input for k-step s.
template <int PREC, int KH, int NB>
__device__ __forceinline__ void ksr_wave_gemm(const float* __restrict__ frag, const float* __restrict__ biasfrag,
                                              const float (&x)[KH], f32x16 (&acc)[NB], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        if (biasfrag) {
            const f32x4* b4 = reinterpret_cast<const f32x4*>(biasfrag + (blk * 2 + h) * 16);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 b = b4[q4];
                acc[blk][4 * q4 + 0] = b[0]; acc[blk][4 * q4 + 1] = b[1];
                acc[blk][4 * q4 + 2] = b[2]; acc[blk][4 * q4 + 3] = b[3];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[blk][q] = 0.f;
        }
    }
    if (PREC == KSR_F32X3) {
        // three bf16 planes of the weights (fragment order of the bf16 modes, NB * KH * 64 elements apart); the activations
        // are split once; per 8 k-steps six products, the small ones first
        bf16x8 x1[KH / 8], x2[KH / 8], x3[KH / 8];
#pragma unroll
        for (int s8 = 0; s8 < KH / 8; ++s8)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 t[3];
                ksr_split3(x[8 * s8 + j], t);
                x1[s8][j] = t[0]; x2[s8][j] = t[1]; x3[s8][j] = t[2];
            }
        constexpr int PLANE = NB * (KH / 8) * 64;          // in 16-byte pieces
        const bf16x8* w8 = reinterpret_cast<const bf16x8*>(frag) + lane;
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
            for (int s8 = 0; s8 < KH / 8; ++s8) {
                const int cur = (blk * (KH / 8) + s8) * 64;
                const bf16x8 w1 = w8[cur], w2 = w8[PLANE + cur], w3 = w8[2 * PLANE + cur];
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, x1[s8], acc[blk], 0, 0, 0);
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x3[s8], acc[blk], 0, 0, 0);
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x2[s8], acc[blk], 0, 0, 0);
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x1[s8], acc[blk], 0, 0, 0);
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x2[s8], acc[blk], 0, 0, 0);
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1[s8], acc[blk], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    if (PREC != KSR_F32) {
        // activations -> bf16 once; one 16-byte LDS read (8 bf16 of weights) per MFMA
        bf16x8 xb[KH / 8];
#pragma unroll
        for (int s8 = 0; s8 < KH / 8; ++s8)
#pragma unroll
            for (int j = 0; j < 8; ++j) xb[s8][j] = (__bf16)x[8 * s8 + j];
        const bf16x8* w8 = reinterpret_cast<const bf16x8*>(frag) + lane;
        bf16x8 w = w8[0];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
            for (int s8 = 0; s8 < KH / 8; ++s8) {
                constexpr int last = NB * (KH / 8) - 1;
                const int cur = blk * (KH / 8) + s8;
                const bf16x8 wn = w8[(cur < last ? cur + 1 : last) * 64];
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xb[s8], acc[blk], 0, 0, 0);
                w = wn;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
    // One 16-byte LDS read feeds four MFMAs (256 cycles of matrix pipe); the next read is issued
    // one group ahead.  The scheduling barrier keeps the compiler from hoisting every fragment read
    // of the unrolled loop to the top (which costs > 256 VGPRs).
    const f32x4* w4 = reinterpret_cast<const f32x4*>(frag) + lane;
    f32x4 w = w4[0];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
        for (int s4 = 0; s4 < KH / 4; ++s4) {
            constexpr int last = NB * (KH / 4) - 1;
            const int cur = blk * (KH / 4) + s4;
            const f32x4 wn = w4[(cur < last ? cur + 1 : last) * 64];
            acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0], x[4 * s4 + 0], acc[blk], 0, 0, 0);
            acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1], x[4 * s4 + 1], acc[blk], 0, 0, 0);
            acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[2], x[4 * s4 + 2], acc[blk], 0, 0, 0);
            acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[3], x[4 * s4 + 3], acc[blk], 0, 0, 0);
            w = wn;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- storage type of the [N, C] activation / gradient tensors exchanged between kernels -------------
// fp32 in modes KSR_F32 and KSR_BF16, bf16 in mode KSR_BF16S.  The C ABI keeps `float*` parameters for these
// tensors in every mode; the kernels reinterpret them according to their PREC template argument.
template <int PREC> struct KsrStore { typedef float T; };
template <> struct KsrStore<KSR_BF16S> { typedef __bf16 T; };

__device__ __forceinline__ void ksr_unpack8(const uint4& w, float* __restrict__ x) {   // 8 bf16 -> 8 floats
    x[0] = __uint_as_float(w.x << 16); x[1] = __uint_as_float(w.x & 0xffff0000u);
    x[2] = __uint_as_float(w.y << 16); x[3] = __uint_as_float(w.y & 0xffff0000u);
    x[4] = __uint_as_float(w.z << 16); x[5] = __uint_as_float(w.z & 0xffff0000u);
    x[6] = __uint_as_float(w.w << 16); x[7] = __uint_as_float(w.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 ksr_pack8(const float* __restrict__ x) {               // 8 floats -> 8 bf16 (RNE)
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)x[j];
    return *reinterpret_cast<const uint4*>(&v);
}

// ---- lane-private loads / stores of N contiguous channels (N = 32: one token's half) --------------
template <class T, int N>
__device__ __forceinline__ void ksr_loadn(const void* __restrict__ base, long off, float* __restrict__ x) {
    if constexpr (sizeof(T) == 4) {
        const float4* p4 = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const float4 v = p4[q];
            x[4 * q + 0] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
        }
    } else {
        const uint4* p4 = reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(base) + off);
#pragma unroll
        for (int q = 0; q < N / 8; ++q) ksr_unpack8(p4[q], x + 8 * q);
    }
}
template <class T, int N>
__device__ __forceinline__ void ksr_storen(void* __restrict__ base, long off, const float* __restrict__ x) {
    if constexpr (sizeof(T) == 4) {
        float4* p4 = reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off);
#pragma unroll
        for (int q = 0; q < N / 4; ++q) p4[q] = make_float4(x[4 * q + 0], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
    } else {
        uint4* p4 = reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(base) + off);
#pragma unroll
        for (int q = 0; q < N / 8; ++q) p4[q] = ksr_pack8(x + 8 * q);
    }
}
// fp32 shorthands (pointer already at the lane's first element)
__device__ __forceinline__ void ksr_load32(const float* __restrict__ p, float* __restrict__ x) { ksr_loadn<float, 32>(p, 0, x); }
__device__ __forceinline__ void ksr_store32(float* __restrict__ p, const float* __restrict__ x) { ksr_storen<float, 32>(p, 0, x); }
__device__ __forceinline__ void ksr_load16(const float* __restrict__ p, float* __restrict__ x) { ksr_loadn<float, 16>(p, 0, x); }
__device__ __forceinline__ void ksr_store16(float* __restrict__ p, const float* __restrict__ x) { ksr_storen<float, 16>(p, 0, x); }

// ---- coalesced HL loads / stores through a wave-private LDS transposer --------------------------
// A lane moving its own 128 contiguous bytes with eight 16-byte accesses makes every access of the wave
// touch 64 different cache lines; such streams top out near 3.5 TB/s on MI355X where lane-linear 16-byte
// accesses reach 5+ TB/s.  These helpers move a wave's tile (32 consecutive rows x 64 channels) with
// lane-linear accesses (16 lanes cover one fp32 row's 256 bytes, 8 lanes a bf16 row's 128 bytes) and
// transpose to / from the half layout in a wave-private fp32 LDS buffer of KSR_XPOSE_FLOATS floats (rows
// padded to 68 floats: conflict-free both ways).  Only the owning wave touches its buffer and LDS
// operations of one wave complete in order, so no workgroup barrier is needed; the compiler barriers keep
// the phases in program order.
constexpr int KSR_XPOSE_LD = 68;
constexpr int KSR_XPOSE_FLOATS = 32 * KSR_XPOSE_LD;

__device__ __forceinline__ void ksr_wave_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// registers of one fetched 32 x 64 tile
template <class T> struct KsrTile { float4 v[8]; };
template <> struct KsrTile<__bf16> { uint4 v[4]; };

// Phase 1: issue the global loads of rows [n0, n0+32) (nvalid real rows, the rest read as 0), channels
// [c0, c0+64) of a tensor with ld elements per row.  Several tiles can be in flight.
template <class T, bool PREDICATED = false>
__device__ __forceinline__ void ksr_co_fetch(const void* __restrict__ base, long n0, int nvalid, int ld, int c0, int lane,
                                             KsrTile<T>& t) {
    if constexpr (PREDICATED) {
        // per-row predicated loads: fewer instructions when the tile is requested one loop iteration ahead
        if constexpr (sizeof(T) == 4) {
            const int r4 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 4 * i + r4;
                t.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < nvalid) t.v[i] = ksr_ld_stream(reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + (n0 + row) * ld + c0 + c4));
            }
        } else {
            const int r8 = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + r8;
                t.v[i] = make_uint4(0u, 0u, 0u, 0u);
                if (row < nvalid) t.v[i] = ksr_ld_stream(reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(base) + (n0 + row) * ld + c0 + c8));
            }
        }
        return;
    }

    // no branches: rows past nvalid are read from row 0 of the tile (or of the tensor when the tile is empty) and
    // zeroed afterwards, so that the loads can stay in flight across loop iterations and uniform branches
    const long nb = nvalid > 0 ? n0 : 0;
    if constexpr (sizeof(T) == 4) {
        const int r4 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + r4;
            const bool ok = row < nvalid;
            const float4 v = ksr_ld_stream(reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + (nb + (ok ? row : 0)) * ld + c0 + c4));
            t.v[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    } else {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + r8;
            const bool ok = row < nvalid;
            const uint4 v = ksr_ld_stream(reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(base) + (nb + (ok ? row : 0)) * ld + c0 + c8));
            t.v[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
        }
    }
}
// The same request with neither branches nor selects on the loaded values: rows past nvalid re-read the tile's last real
// row (row 0 of the tensor when the tile is empty), so the result registers are not touched between the request and their
// first use and the loads can stay in flight across a loop iteration.  For callers that never store or sum rows past
// nvalid (token-per-lane kernels, whose columns do not mix).
template <class T>
__device__ __forceinline__ void ksr_co_fetch_clamped(const void* __restrict__ base, long n0, int nvalid, int ld, int c0, int lane,
                                                     KsrTile<T>& t) {
    const long nb = nvalid > 0 ? n0 : 0;
    const int last = nvalid > 0 ? nvalid - 1 : 0;
    if constexpr (sizeof(T) == 4) {
        const int r4 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + r4;
            t.v[i] = ksr_ld_stream(reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + (nb + (row < last ? row : last)) * ld + c0 + c4));
        }
    } else {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + r8;
            t.v[i] = ksr_ld_stream(reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(base) + (nb + (row < last ? row : last)) * ld + c0 + c8));
        }
    }
}
// fetched tile -> fp32 LDS tile [32][KSR_XPOSE_LD] (caller fences)
template <class T>
__device__ __forceinline__ void ksr_co_to_lds(float* __restrict__ xb, int lane, const KsrTile<T>& t) {
    if constexpr (sizeof(T) == 4) {
        const int r4 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(xb + (4 * i + r4) * KSR_XPOSE_LD + c4) = t.v[i];
    } else {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x[8];
            ksr_unpack8(t.v[i], x);
            float* d = xb + (8 * i + r8) * KSR_XPOSE_LD + c8;
            *reinterpret_cast<float4*>(d) = make_float4(x[0], x[1], x[2], x[3]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(x[4], x[5], x[6], x[7]);
        }
    }
}
// Phase 2: transpose one fetched tile into the half layout (this lane: row lane&31, channels 32*(lane>>5)..+31).
template <class T>
__device__ __forceinline__ void ksr_co_finish(float* __restrict__ xb, int lane, const KsrTile<T>& t, float (&x)[32]) {
    ksr_wave_fence();
    ksr_co_to_lds<T>(xb, lane, t);
    ksr_wave_fence();
    const float* rp = xb + (lane & 31) * KSR_XPOSE_LD + 32 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(rp + 4 * q);
        x[4 * q + 0] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
}
template <class T>
__device__ __forceinline__ void ksr_load32_co(float* __restrict__ xb, const void* __restrict__ base, long n0, int nvalid,
                                              int ld, int c0, int lane, float (&x)[32]) {
    KsrTile<T> t;
    ksr_co_fetch<T>(base, n0, nvalid, ld, c0, lane, t);
    ksr_co_finish<T>(xb, lane, t, x);
}
// fp32 LDS tile -> rows [n0, n0 + nvalid) of the tensor (caller has written the tile and fenced)
template <class T, bool TWO_PATH = true, bool NT = false>
__device__ __forceinline__ void ksr_co_from_lds(const float* __restrict__ xb, void* __restrict__ base, long n0, int nvalid,
                                                int ld, int c0, int lane) {
    // full tiles (all but the last one of a tensor) take a path without per-row predicates; nvalid is wave-uniform.
    // TWO_PATH = false keeps the single predicated path: a kernel that has loads in flight across this call (tiles
    // requested one iteration ahead) loses them at the branch's join otherwise
    auto body = [&](auto guard) {
        constexpr bool GUARD = decltype(guard)::value;

    if constexpr (sizeof(T) == 4) {
        const int r4 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + r4;
            const float4 v = *reinterpret_cast<const float4*>(xb + row * KSR_XPOSE_LD + c4);
            if (!GUARD || row < nvalid) ksr_st16<NT>(reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + (n0 + row) * ld + c0 + c4), v);
        }
    } else {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + r8;
            const float* sp = xb + row * KSR_XPOSE_LD + c8;
            const float4 v0 = *reinterpret_cast<const float4*>(sp), v1 = *reinterpret_cast<const float4*>(sp + 4);
            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (!GUARD || row < nvalid) ksr_st16<NT>(reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(base) + (n0 + row) * ld + c0 + c8), ksr_pack8(x));
        }
    }
    };
    if (TWO_PATH && __builtin_amdgcn_readfirstlane(nvalid) == 32) body(std::false_type{});
    else body(std::true_type{});
}
// The reverse of ksr_load32_co: x (half layout) -> rows [n0, n0 + nvalid).
template <class T, bool TWO_PATH = true, bool NT = false>
__device__ __forceinline__ void ksr_store32_co(float* __restrict__ xb, void* __restrict__ base, long n0, int nvalid,
                                               int ld, int c0, int lane, const float (&x)[32]) {
    float* wp = xb + (lane & 31) * KSR_XPOSE_LD + 32 * (lane >> 5);
    ksr_wave_fence();
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(wp + 4 * q) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
    ksr_wave_fence();
    ksr_co_from_lds<T, TWO_PATH, NT>(xb, base, n0, nvalid, ld, c0, lane);
}

// 32-channel rows (lane half h holds channels 16h .. 16h+15): same scheme with rows padded to 36 floats.
constexpr int KSR_XPOSE16_LD = 36;
// fp32 LDS tile [32][36] <-> rows of a tensor with ld elements per row, channels c0 .. c0+31
template <class T, bool TWO_PATH = true, bool NT = false, bool FULLTILE = false>     // FULLTILE: the caller knows nvalid == 32: no predicate, no branch
__device__ __forceinline__ void ksr_co16_from_lds(const float* __restrict__ xb, void* __restrict__ base, long n0, int nvalid,
                                                  int ld, int c0, int lane) {
    // full tiles (all but the last one of a tensor) take a path without per-row predicates; nvalid is wave-uniform.
    // TWO_PATH = false keeps the single predicated path: a kernel that has loads in flight across this call (tiles
    // requested one iteration ahead) loses them at the branch's join otherwise
    auto body = [&](auto guard) {
        constexpr bool GUARD = decltype(guard)::value;

    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = i * 64 + lane, row = idx >> 3, c4 = (idx & 7) * 4;
            const float4 v = *reinterpret_cast<const float4*>(xb + row * KSR_XPOSE16_LD + c4);
            if (!GUARD || row < nvalid) ksr_st16<NT>(reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + (n0 + row) * ld + c0 + c4), v);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = i * 64 + lane, row = idx >> 2, c8 = (idx & 3) * 8;
            const float* sp = xb + row * KSR_XPOSE16_LD + c8;
            const float4 v0 = *reinterpret_cast<const float4*>(sp), v1 = *reinterpret_cast<const float4*>(sp + 4);
            const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if (!GUARD || row < nvalid) ksr_st16<NT>(reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(base) + (n0 + row) * ld + c0 + c8), ksr_pack8(x));
        }
    }
    };
    if constexpr (FULLTILE) body(std::false_type{});
    else if (TWO_PATH && __builtin_amdgcn_readfirstlane(nvalid) == 32) body(std::false_type{});
    else body(std::true_type{});
}
// registers of one fetched 32 x 32 tile
template <class T> struct KsrTile16 { float4 v[4]; };
template <> struct KsrTile16<__bf16> { uint4 v[2]; };
// Phase 1 for a tensor with 32 channels per row (rows [n0, n0+32), nvalid real ones)
template <class T>
__device__ __forceinline__ void ksr_co16_fetch(const void* __restrict__ base, long n0, int nvalid, int lane, KsrTile16<T>& t) {
    const long nb = nvalid > 0 ? n0 : 0;      // branch-free, as ksr_co_fetch
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = i * 64 + lane, row = idx >> 3, c4 = (idx & 7) * 4;
            const bool ok = row < nvalid;
            const float4 v = ksr_ld_stream(reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + (nb + (ok ? row : 0)) * 32 + c4));
            t.v[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = i * 64 + lane, row = idx >> 2, c8 = (idx & 3) * 8;
            const bool ok = row < nvalid;
            const uint4 v = ksr_ld_stream(reinterpret_cast<const uint4*>(reinterpret_cast<const __bf16*>(base) + (nb + (ok ? row : 0)) * 32 + c8));
            t.v[i] = make_uint4(ok ? v.x : 0u, ok ? v.y : 0u, ok ? v.z : 0u, ok ? v.w : 0u);
        }
    }
}
// Phase 2: this lane's 16 channels (16 * (lane >> 5) ..) of row lane & 31
template <class T>
__device__ __forceinline__ void ksr_co16_finish(float* __restrict__ xb, int lane, const KsrTile16<T>& t, float (&x)[16]) {
    ksr_wave_fence();
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = i * 64 + lane, row = idx >> 3, c4 = (idx & 7) * 4;
            *reinterpret_cast<float4*>(xb + row * KSR_XPOSE16_LD + c4) = t.v[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = i * 64 + lane, row = idx >> 2, c8 = (idx & 3) * 8;
            float f[8];
            ksr_unpack8(t.v[i], f);
            float* d = xb + row * KSR_XPOSE16_LD + c8;
            *reinterpret_cast<float4*>(d) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(f[4], f[5], f[6], f[7]);
        }
    }
    ksr_wave_fence();
    const float* rp = xb + (lane & 31) * KSR_XPOSE16_LD + 16 * (lane >> 5);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(rp + 4 * q);
        x[4 * q + 0] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
    }
}
template <class T>
__device__ __forceinline__ void ksr_load16_co(float* __restrict__ xb, const void* __restrict__ base, long n0, int nvalid,
                                              int lane, float (&x)[16]) {
    KsrTile16<T> t;
    ksr_co16_fetch<T>(base, n0, nvalid, lane, t);
    ksr_co16_finish<T>(xb, lane, t, x);
}
template <class T, bool TWO_PATH = true, bool NT = false, bool FULLTILE = false>
__device__ __forceinline__ void ksr_store16_co(float* __restrict__ xb, void* __restrict__ base, long n0, int nvalid,
                                               int lane, const float (&x)[16]) {
    float* wp = xb + (lane & 31) * KSR_XPOSE16_LD + 16 * (lane >> 5);
    ksr_wave_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(wp + 4 * q) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
    ksr_wave_fence();
    ksr_co16_from_lds<T, TWO_PATH, NT, FULLTILE>(xb, base, n0, nvalid, 32, 0, lane);
}

// Half-width forms for the kernels whose LDS is taken by the three weight planes of the split-fp32 products: the same
// coalesced fp32 tile loads / stores, one 32-channel half at a time through a [32][36] buffer (4.6 KB per wave instead of
// 8.7 KB; a row's 128 bytes are covered by 8 lanes).
constexpr int KSR_XPOSE_HALF_LD = 36;
constexpr int KSR_XPOSE_HALF_FLOATS = 32 * KSR_XPOSE_HALF_LD;
template <int PREC> __host__ __device__ constexpr int ksr_xpose_floats_p() { return PREC == KSR_F32X3 ? KSR_XPOSE_HALF_FLOATS : KSR_XPOSE_FLOATS; }

// tile fetched by ksr_co_fetch<float> -> half layout
__device__ __forceinline__ void ksr_co_finish_half(float* __restrict__ xb, int lane, const KsrTile<float>& t, float (&x)[32]) {
    const int r4 = lane >> 4, c16 = lane & 15, h = lane >> 5;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        ksr_wave_fence();
        if ((c16 >> 3) == p) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(xb + (4 * i + r4) * KSR_XPOSE_HALF_LD + 4 * (c16 & 7)) = t.v[i];
        }
        ksr_wave_fence();
        if (h == p) {
            const float* rp = xb + (lane & 31) * KSR_XPOSE_HALF_LD;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(rp + 4 * q);
                x[4 * q + 0] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
            }
        }
    }
    ksr_wave_fence();
}
// x (half layout) -> rows [n0, n0 + nvalid), channels [c0, c0 + 64) of an fp32 tensor
template <bool NT = false>
__device__ __forceinline__ void ksr_store32_co_half(float* __restrict__ xb, float* __restrict__ base, long n0, int nvalid, int ld,
                                                    int c0, int lane, const float (&x)[32]) {
    const int h = lane >> 5;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        ksr_wave_fence();
        if (h == p) {
            float* wp = xb + (lane & 31) * KSR_XPOSE_HALF_LD;
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(wp + 4 * q) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
        }
        ksr_wave_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = i * 64 + lane, row = idx >> 3, c4 = (idx & 7) * 4;
            const float4 v = *reinterpret_cast<const float4*>(xb + row * KSR_XPOSE_HALF_LD + c4);
            if (row < nvalid) ksr_st16<NT>(reinterpret_cast<float4*>(base + (n0 + row) * ld + c0 + 32 * p + c4), v);
        }
    }
    ksr_wave_fence();
}
// the form a kernel instance's mode asks for (split fp32: half width; tensors are fp32 there)
template <int PREC, class T, bool TWO_PATH = true, bool NT = false>
__device__ __forceinline__ void ksr_store32_co_p(float* __restrict__ xb, void* __restrict__ base, long n0, int nvalid, int ld,
                                                 int c0, int lane, const float (&x)[32]) {
    if constexpr (PREC == KSR_F32X3) ksr_store32_co_half<NT>(xb, reinterpret_cast<float*>(base), n0, nvalid, ld, c0, lane, x);
    else ksr_store32_co<T, TWO_PATH, NT>(xb, base, n0, nvalid, ld, c0, lane, x);
}

// Deferred form of ksr_store32_co: a tile's output is transposed right away but stays in registers (the store layout:
// lane-linear 16-byte pieces) and is stored one loop iteration later, after the next tile's loads have been waited for.
// The vector-memory counter completes in issue order, so a kernel that stores at the end of an iteration and waits for
// its prefetched input at the top of the next one waits for those stores as well; stored an iteration late, both the
// loads and the stores that a wait covers were issued a whole tile's arithmetic earlier.
template <class T> struct KsrPend { ksr_f32x4_t v[8]; long n0; int nvalid; };       // native vectors: the struct must live in registers
template <> struct KsrPend<__bf16> { ksr_u32x4_t v[4]; long n0; int nvalid; };

template <class T>
__device__ __forceinline__ void ksr_pend32_co(float* __restrict__ xb, long n0, int nvalid, int lane, const float (&x)[32],
                                              KsrPend<T>& p) {
    float* wp = xb + (lane & 31) * KSR_XPOSE_LD + 32 * (lane >> 5);
    ksr_wave_fence();
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(wp + 4 * q) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
    ksr_wave_fence();
    if constexpr (sizeof(T) == 4) {
        const int r4 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) p.v[i] = *reinterpret_cast<const ksr_f32x4_t*>(xb + (4 * i + r4) * KSR_XPOSE_LD + c4);
    } else {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* sp = xb + (8 * i + r8) * KSR_XPOSE_LD + c8;
            const float4 v0 = *reinterpret_cast<const float4*>(sp), v1 = *reinterpret_cast<const float4*>(sp + 4);
            const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const uint4 u = ksr_pack8(xs);
            p.v[i] = ksr_u32x4_t{u.x, u.y, u.z, u.w};
        }
    }
    p.n0 = n0; p.nvalid = nvalid;
    ksr_wave_fence();
}
template <class T>
__device__ __forceinline__ void ksr_pend_store(void* __restrict__ base, int ld, int c0, int lane, KsrPend<T>& p) {
    if constexpr (sizeof(T) == 4) {
        const int r4 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * i + r4;
            if (row < p.nvalid) *reinterpret_cast<ksr_f32x4_t*>(reinterpret_cast<float*>(base) + (p.n0 + row) * ld + c0 + c4) = p.v[i];
        }
    } else {
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + r8;
            if (row < p.nvalid) *reinterpret_cast<ksr_u32x4_t*>(reinterpret_cast<__bf16*>(base) + (p.n0 + row) * ld + c0 + c8) = p.v[i];
        }
    }
    p.nvalid = 0;
}

// ksr_pend32_co through the half-width buffer (fp32 tensors): piece i of half p = row (i * 64 + lane) >> 3, floats
// 32 p + 4 ((i * 64 + lane) & 7) .. + 3.
struct KsrPendHalf { ksr_f32x4_t v[8]; long n0; int nvalid; };
__device__ __forceinline__ void ksr_pend32_co_half(float* __restrict__ xb, long n0, int nvalid, int lane, const float (&x)[32],
                                                   KsrPendHalf& pd) {
    const int h = lane >> 5;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        ksr_wave_fence();
        if (h == p) {
            float* wp = xb + (lane & 31) * KSR_XPOSE_HALF_LD;
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(wp + 4 * q) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
        }
        ksr_wave_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = i * 64 + lane;
            pd.v[4 * p + i] = *reinterpret_cast<const ksr_f32x4_t*>(xb + (idx >> 3) * KSR_XPOSE_HALF_LD + (idx & 7) * 4);
        }
    }
    pd.n0 = n0; pd.nvalid = nvalid;
    ksr_wave_fence();
}
__device__ __forceinline__ void ksr_pend_store_half(float* __restrict__ base, int ld, int c0, int lane, KsrPendHalf& pd) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = i * 64 + lane, row = idx >> 3, c4 = (idx & 7) * 4;
            if (row < pd.nvalid) *reinterpret_cast<ksr_f32x4_t*>(base + (pd.n0 + row) * ld + c0 + 32 * p + c4) = pd.v[4 * p + i];
        }
    pd.nvalid = 0;
}

// Sum of v over the two lanes that share a token (lanes l and l^32).
// (v_permlane32_swap exchanges the upper half of one register with the lower half of another: after swapping a register
// with a copy of itself the two results hold v[l] and v[l ^ 32] in some order - one vector instruction instead of a
// ds_bpermute round trip through the LDS crossbar.)
__device__ __forceinline__ float ksr_pair_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float ksr_pair_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// LayerNorm over the 64 channels of a token held as x[32] on each of the two half-lanes.
// Writes xhat (normalised, before gain/shift) and returns rstd.
__device__ __forceinline__ float ksr_layernorm64(const float (&x)[32], float (&xhat)[32], float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += x[i];
    const float mean = ksr_pair_sum(s) * (1.f / 64.f);
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { const float d = x[i] - mean; v += d * d; }
    const float var = ksr_pair_sum(v) * (1.f / 64.f);
    const float rstd = 1.f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 32; ++i) xhat[i] = (x[i] - mean) * rstd;
    return rstd;
}

// Column sums over a wave's 32 tokens (LayerNorm gain / shift gradients and the like): s_gx += sum_tok g * xhat,
// s_g += sum_tok g, for channel c = lane.  The products
// go through the wave's transposer tile (rows of 68 floats); lane c then adds up column c in row order, so the result
// does not depend on scheduling.  Rows past the end of the tensor hold zeros in g.
__device__ __forceinline__ void ksr_colsum32(float* __restrict__ xb, int lane, const float (&g)[32], const float (&xh)[32],
                                          float& s_gx, float& s_g) {
    float* wp = xb + (lane & 31) * KSR_XPOSE_LD + 32 * (lane >> 5);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        ksr_wave_fence();
#pragma unroll
        for (int q = 0; q < 8; ++q)
            *reinterpret_cast<float4*>(wp + 4 * q) =
                pass == 0 ? make_float4(g[4 * q] * xh[4 * q], g[4 * q + 1] * xh[4 * q + 1], g[4 * q + 2] * xh[4 * q + 2], g[4 * q + 3] * xh[4 * q + 3])
                          : make_float4(g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
        ksr_wave_fence();
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int r = 0; r < 32; r += 4) {
            t0 += xb[(r + 0) * KSR_XPOSE_LD + lane]; t1 += xb[(r + 1) * KSR_XPOSE_LD + lane];
            t2 += xb[(r + 2) * KSR_XPOSE_LD + lane]; t3 += xb[(r + 3) * KSR_XPOSE_LD + lane];
        }
        (pass == 0 ? s_gx : s_g) += (t0 + t1) + (t2 + t3);
    }
}


// The same with one multiplier per token (lane pair) instead of one per element: s_gv += sum_tok g * v, s_g += sum_tok g.
__device__ __forceinline__ void ksr_colsum32_scalar(float* __restrict__ xb, int lane, const float (&g)[32], float v,
                                                    float& s_gv, float& s_g) {
    float* wp = xb + (lane & 31) * KSR_XPOSE_LD + 32 * (lane >> 5);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const float m = pass == 0 ? v : 1.f;
        ksr_wave_fence();
#pragma unroll
        for (int q = 0; q < 8; ++q)
            *reinterpret_cast<float4*>(wp + 4 * q) = make_float4(g[4 * q] * m, g[4 * q + 1] * m, g[4 * q + 2] * m, g[4 * q + 3] * m);
        ksr_wave_fence();
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int r = 0; r < 32; r += 4) {
            t0 += xb[(r + 0) * KSR_XPOSE_LD + lane]; t1 += xb[(r + 1) * KSR_XPOSE_LD + lane];
            t2 += xb[(r + 2) * KSR_XPOSE_LD + lane]; t3 += xb[(r + 3) * KSR_XPOSE_LD + lane];
        }
        (pass == 0 ? s_gv : s_g) += (t0 + t1) + (t2 + t3);
    }
}

#endif  // __HIPCC__
