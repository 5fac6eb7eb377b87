// Parameter-gradient kernels: reductions over all N pixels.
//
//   wgrad      dW[j][k] = sum_n dZ[n][j] * A[n][k]  (+ db[j] = sum_n dZ[n][j])  on MFMA, pixels on the
//              k axis of v_mfma_f32_32x32x2_f32 (2 pixels per instruction)
//   colreduce  out[0][c] = sum_n a[n][c]*b[n][c], out[1][c] = sum_n a[n][c]       (LayerNorm gain/shift)
//   dwgrad     depthwise 3x3 weight/bias gradient (stencil)
//   stemgrad   the 12 scalars of the 1 -> 1 -> 1 front of the first FFN
// Every kernel writes one partial result per workgroup into a slab; ksr_reduce sums the slabs in a
// fixed order, so results are bitwise reproducible (no float atomics).
#include "ksr_conv_common.h"

namespace {

constexpr int WG_MAXP = 6;     // at most this many dz blocks per wave
constexpr int WG_GRID = 1024;  // at most this many workgroups / partial slabs

struct WgArgs {
    const float* dz; int dz_ld, dz_c0, J;
    const float* a1; int a1_ld, a1_c0, K;
    const float* a2; int a2_ld, a2_c0;
    float* slab;
    long N;        // pixels per batch entry (blockIdx.y selects the entry: rows [y*N, (y+1)*N) of every tensor)
    int Jv;        // valid channels of dz (multiple of 4, J - 32 < Jv <= J); rows >= Jv of dW are zero
    int sq;        // also emit sum dz^2 per row and sum a^2 per column
    long stride;   // slab floats per workgroup: J*K + J (+ J + K with sq)
};

// Each workgroup walks its pixel range in chunks of WG_T pixels.  A chunk's rows [dz | a1 | a2] are copied
// global -> LDS by 16-byte LDS-DMA loads (lane-linear image = row-major [WG_T][RL]), double buffered: the
// loads of chunk c+1 are in flight while the MFMAs of chunk c run.  Operands are then read from LDS with
// conflict-free ds_read_b32 (lane = column), four pixel pairs ahead of the MFMAs that consume them.
// Wave w owns column block kb = w % (K/32) of A and the dz blocks jb = w / (K/32) + p * (4 / (K/32)),
// p < NP; blocks past the end (and channels >= Jv of a partial block) are multiplied by 0.
// PREC = KSR_BF16: the products run on v_mfma_f32_32x32x16_bf16 (operands rounded to bf16 after the fp32
// LDS read, fp32 accumulation; 16 pixels per instruction); the bias / square sums stay exact fp32.
template <int PREC, int WG_T, int NP, bool HAS2>   // pixels per staged chunk; dz blocks per wave; a2 present
__global__ void __launch_bounds__(256, 2) wgrad_kernel(WgArgs a) {
    using T = typename KsrStore<PREC>::T;          // storage type of dz / a1 / a2 (staged as stored)
    constexpr int EPP = 16 / sizeof(T);            // elements per 16-byte piece
    extern __shared__ __attribute__((aligned(16))) float wlds_raw[];
    T* wlds = reinterpret_cast<T*>(wlds_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kk = lane >> 5;
    const int NJB = a.J / 32, NKB = a.K / 32;
    const int kb = wv % NKB, jb0 = wv / NKB, jbs = 4 / NKB;
    const int RL = a.Jv + a.K + (HAS2 ? a.K : 0);      // elements per staged row
    const int RL4 = RL / EPP;                           // 16-byte pieces per row
    const unsigned inv = (unsigned)((0x100000000ULL + RL4 - 1) / RL4);
    const int nvec = WG_T * RL4;                        // 16-byte pieces per chunk
    const int npass = (nvec + 255) / 256;

    int joff[NP];
    float jmask[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int jb = jb0 + p * jbs;
        const bool vb = jb < NJB;
        const int j = 32 * (vb ? jb : jb0) + col;       // < Jv + 32 <= Jv + K: inside the staged row
        joff[p] = j;
        jmask[p] = (vb && j < a.Jv) ? 1.f : 0.f;
    }
    const int boff = a.Jv + 32 * kb + col;

    f32x16 acc[NP];
    float bsum[NP], sqa[NP], sqb = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        bsum[p] = 0.f; sqa[p] = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[p][q] = 0.f;
    }
    const long per = (((a.N + gridDim.x - 1) / gridDim.x) + WG_T - 1) / WG_T * WG_T;
    const long base = (long)blockIdx.y * a.N;
    const long n0 = base + (long)blockIdx.x * per;
    const long n1 = ((long)(blockIdx.x + 1) * per < a.N) ? n0 + per : base + a.N;
    const int nchunks = n1 > n0 ? (int)((n1 - n0 + WG_T - 1) / WG_T) : 0;

    auto issue = [&](int c) {
        T* dstbuf = wlds + (c & 1) * (WG_T * RL);
        const long t0 = n0 + (long)c * WG_T;
        for (int i = 0; i < npass; ++i) {
            const unsigned f = (unsigned)(i * 256 + tid);
            if ((int)f >= nvec) break;                  // ragged last pass: the remaining lanes stage nothing
            const unsigned trow = __umulhi(f, inv);
            const int q = (int)(f - trow * RL4) * EPP;
            long tokn = t0 + trow;
            if (tokn >= n1) tokn = n1 - 1;
            const T* s_dz = reinterpret_cast<const T*>(a.dz) + tokn * a.dz_ld + a.dz_c0 + q;
            const T* s_a1 = reinterpret_cast<const T*>(a.a1) + tokn * a.a1_ld + a.a1_c0 + (q - a.Jv);
            const T* src = q < a.Jv ? s_dz : s_a1;
            if (HAS2) {
                const T* s_a2 = reinterpret_cast<const T*>(a.a2) + tokn * a.a2_ld + a.a2_c0 + (q - a.Jv - a.K);
                src = q < a.Jv + a.K ? src : s_a2;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dstbuf + (i * 256 + wv * 64) * EPP),
                                             16, 0, 2 /* nt: operands are read once */);
        }
    };

    constexpr bool BF = PREC != KSR_F32;
    constexpr int UG = BF ? 8 : 4;       // operand registers per group: fp32 4 pixel pairs, bf16 one 16-pixel step
    constexpr int NG = WG_T / 2 / UG;
    static_assert(NG % 2 == 0, "operand groups are consumed in pairs");
    auto load_group = [&](const T* buf, int g, float (&bv)[UG], float (&av)[UG][NP]) {
#pragma unroll
        for (int i = 0; i < UG; ++i) {
            // pixel of register i: fp32 MFMA k = 2*pair + kk; bf16 MFMA k = 8*kk + i within the 16-pixel step
            const int px = BF ? 16 * g + 8 * kk + i : 2 * (g * UG + i) + kk;
            const T* row = buf + px * RL;
            bv[i] = (float)row[boff];
            if (HAS2) bv[i] *= (float)row[boff + a.K];
#pragma unroll
            for (int p = 0; p < NP; ++p) av[i][p] = (float)row[joff[p]];
        }
    };

    if (nchunks > 0) issue(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                 // chunk c has landed (vmcnt(0) + barrier); buffer (c+1)&1 is free
        if (c + 1 < nchunks) issue(c + 1);
        const T* buf = wlds + (c & 1) * (WG_T * RL);
        const long t0 = n0 + (long)c * WG_T;
        const int nvalid = (n1 - t0 < WG_T) ? (int)(n1 - t0) : WG_T;
        if constexpr (PREC == KSR_BF16S) {
            // bf16 rows in LDS: a 16-pixel MFMA operand (8 consecutive pixels of one channel per lane) is two
            // ds_read_b64_tr_b16 (each 16-lane group reads a 4-pixel x 16-channel block transposed); no conversion.
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            const int li = lane & 15;
            const int lane_off = (8 * kk + (li >> 2)) * RL + 16 * ((lane >> 4) & 1) + 4 * (li & 3);
            const int bbase = a.Jv + 32 * kb;
            auto tr = [&](const T* ptr) -> uint2 {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
                return *reinterpret_cast<const uint2*>(&v);
            };
            auto ld_step = [&](int g, uint2 (&bq)[HAS2 ? 4 : 2], uint2 (&aq)[NP][2]) {
                const T* base = buf + 16 * g * RL + lane_off;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    bq[r] = tr(base + 4 * r * RL + bbase);
                    if (HAS2) bq[2 + r] = tr(base + 4 * r * RL + bbase + a.K);
#pragma unroll
                    for (int p = 0; p < NP; ++p) aq[p][r] = tr(base + 4 * r * RL + (joff[p] - col));
                }
            };
            auto unpack = [](const uint2 (&q)[2], float (&x)[8]) {
                x[0] = __uint_as_float(q[0].x << 16); x[1] = __uint_as_float(q[0].x & 0xffff0000u);
                x[2] = __uint_as_float(q[0].y << 16); x[3] = __uint_as_float(q[0].y & 0xffff0000u);
                x[4] = __uint_as_float(q[1].x << 16); x[5] = __uint_as_float(q[1].x & 0xffff0000u);
                x[6] = __uint_as_float(q[1].y << 16); x[7] = __uint_as_float(q[1].y & 0xffff0000u);
            };
            auto use_step = [&](int g, const uint2 (&bq)[HAS2 ? 4 : 2], const uint2 (&aq)[NP][2]) {
                const bool ragged = nvalid < WG_T;                 // wave-uniform: only a range's last chunk
                // pixels of this lane's 8 operand elements: 16g + 8kk + e; zero the ones past the range
                unsigned m[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
                if (ragged) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int px = 16 * g + 8 * kk + 2 * d;
                        m[d] = (px < nvalid ? 0x0000ffffu : 0u) | (px + 1 < nvalid ? 0xffff0000u : 0u);
                    }
                }
                uint4 bw = make_uint4(bq[0].x & m[0], bq[0].y & m[1], bq[1].x & m[2], bq[1].y & m[3]);
                if (HAS2) {
                    float x1[8], x2[8];
                    const uint2 q1[2] = {make_uint2(bw.x, bw.y), make_uint2(bw.z, bw.w)};
                    const uint2 q2[2] = {bq[2], bq[3]};
                    unpack(q1, x1); unpack(q2, x2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x1[e] *= x2[e];
                    bw = ksr_pack8(x1);
                }
                if (a.sq) {
                    float x[8];
                    const uint2 q1[2] = {make_uint2(bw.x, bw.y), make_uint2(bw.z, bw.w)};
                    unpack(q1, x);
#pragma unroll
                    for (int e = 0; e < 8; ++e) sqb += x[e] * x[e];
                }
                const bf16x8 bpk = *reinterpret_cast<const bf16x8*>(&bw);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const unsigned jm = jmask[p] != 0.f ? 0xffffffffu : 0u;
                    const uint4 aw = make_uint4(aq[p][0].x & m[0] & jm, aq[p][0].y & m[1] & jm, aq[p][1].x & m[2] & jm,
                                                aq[p][1].y & m[3] & jm);
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&aw), bpk, acc[p], 0, 0, 0);
                    if (kb == 0) {                                   // wave-uniform: bias (and square) sums of dz
                        float x[8];
                        const uint2 q1[2] = {make_uint2(aw.x, aw.y), make_uint2(aw.z, aw.w)};
                        unpack(q1, x);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { bsum[p] += x[e]; sqa[p] += x[e] * x[e]; }
                    }
                }
            };
            constexpr int NS = WG_T / 16;                            // 16-pixel steps per chunk (2 or 4)
            uint2 bqA[HAS2 ? 4 : 2], aqA[NP][2], bqB[HAS2 ? 4 : 2], aqB[NP][2];
            ld_step(0, bqA, aqA);
#pragma unroll 1
            for (int g = 0; g < NS; g += 2) {
                ld_step(g + 1, bqB, aqB);
                __builtin_amdgcn_sched_barrier(0);
                use_step(g, bqA, aqA);
                __builtin_amdgcn_sched_barrier(0);
                if (g + 2 < NS) ld_step(g + 2, bqA, aqA);
                __builtin_amdgcn_sched_barrier(0);
                use_step(g + 1, bqB, aqB);
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
        float bvA[UG], avA[UG][NP], bvB[UG], avB[UG][NP];
        auto consume = [&](int g, const float (&bv)[UG], const float (&av)[UG][NP]) {
            bf16x8 bpk, apk[NP];
#pragma unroll
            for (int i = 0; i < UG; ++i) {
                // rows past n1 were staged from pixel n1-1 (finite): zero them by multiplication
                const int px = BF ? 16 * g + 8 * kk + i : 2 * (g * UG + i) + kk;
                const float okf = (px < nvalid) ? 1.f : 0.f;
                const float b = bv[i];
                sqb += okf * b * b;
                if (BF) bpk[i] = (__bf16)b;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const float v = av[i][p] * (jmask[p] * okf);
                    if (BF) apk[p][i] = (__bf16)v;
                    else acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, b, acc[p], 0, 0, 0);
                    bsum[p] += v;
                    sqa[p] += v * v;
                }
            }
            if (BF) {
#pragma unroll
                for (int p = 0; p < NP; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(apk[p], bpk, acc[p], 0, 0, 0);
            }
        };
        load_group(buf, 0, bvA, avA);
#pragma unroll 1
        for (int g = 0; g < NG; g += 2) {
            load_group(buf, g + 1, bvB, avB);             // NG is even
            __builtin_amdgcn_sched_barrier(0);
            consume(g, bvA, avA);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < NG) load_group(buf, g + 2, bvA, avA);
            __builtin_amdgcn_sched_barrier(0);
            consume(g + 1, bvB, avB);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float* slab = a.slab + ((long)blockIdx.y * gridDim.x + blockIdx.x) * a.stride;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int jb = jb0 + p * jbs;
        if (jb < NJB) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * kk;
                slab[(long)(32 * jb + row) * a.K + 32 * kb + col] = acc[p][q];
            }
            if (kb == 0) {
                const float bs = ksr_pair_sum(bsum[p]);
                if (kk == 0) slab[(long)a.J * a.K + 32 * jb + col] = bs;
                if (a.sq) {
                    const float ss = ksr_pair_sum(sqa[p]);
                    if (kk == 0) slab[(long)a.J * a.K + a.J + 32 * jb + col] = ss;
                }
            }
        }
    }
    if (a.sq && jb0 == 0) {
        const float ss = ksr_pair_sum(sqb);
        if (kk == 0) slab[(long)a.J * a.K + 2 * a.J + 32 * kb + col] = ss;
    }
}

// The same accumulation for the shapes of the training step in bf16 storage (no product operand, no square sums, every
// channel of dz valid, staged rows of RL4 16-byte pieces with RL4 a multiple of 4, one batch entry), with the bookkeeping
// of the general kernel taken out of the chunk loop - that kernel spends half of its time issuing vector instructions
// (address arithmetic of a run-time piece -> (row, tensor) map for every transfer, range masks, unpacking for the bias
// sums): each thread's RL4 / 4 source pointers are formed once and advance by a constant per chunk; only a range's last
// chunk takes the masked path; the bias sums are one more matrix product against a vector of ones.
template <int RL4, int NP>
__global__ void __launch_bounds__(256, 2) wgrad_fast_kernel(WgArgs a) {
    typedef __bf16 T;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    constexpr int WG_T = 64, RL = RL4 * 8, NPASS = RL4 / 4, NS = WG_T / 16;
    extern __shared__ __attribute__((aligned(16))) float wlds_raw[];
    T* wlds = reinterpret_cast<T*>(wlds_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kk = lane >> 5;
    const int NJB = a.J / 32, NKB = a.K / 32;
    const int kb = wv % NKB, jb0 = wv / NKB, jbs = 4 / NKB;
    int jcol[NP];
    unsigned jm[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int jb = jb0 + p * jbs;
        jcol[p] = 32 * (jb < NJB ? jb : jb0);
        jm[p] = jb < NJB ? 0xffffffffu : 0u;
    }
    f32x16 acc[NP], bacc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[p][q] = 0.f; bacc[p][q] = 0.f; }
    const long per = (((a.N + gridDim.x - 1) / gridDim.x) + WG_T - 1) / WG_T * WG_T;
    const long n0 = (long)blockIdx.x * per;
    const long n1 = ((long)(blockIdx.x + 1) * per < a.N) ? n0 + per : a.N;
    const int nchunks = n1 > n0 ? (int)((n1 - n0 + WG_T - 1) / WG_T) : 0;

    // piece f = i * 256 + tid of a chunk: staged row f / RL4, 16-byte piece f % RL4 of [dz | a1]
    const T* src[NPASS];
    int trow[NPASS], adv[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int f = i * 256 + tid;
        trow[i] = f / RL4;
        const int q = (f - trow[i] * RL4) * 8;
        const bool is_dz = q < a.J;
        const int ld = is_dz ? a.dz_ld : a.a1_ld;
        src[i] = (is_dz ? reinterpret_cast<const T*>(a.dz) + a.dz_c0 + q : reinterpret_cast<const T*>(a.a1) + a.a1_c0 + (q - a.J))
                 + (n0 + trow[i]) * ld;
        adv[i] = WG_T * ld;
    }
    auto issue = [&](int c) {
        T* dstbuf = wlds + (c & 1) * (WG_T * RL);
        const long t0 = n0 + (long)c * WG_T;
        const bool ragged = t0 + WG_T > n1;                    // wave-uniform: rows past the range repeat its last row
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const T* sp = src[i];
            if (ragged && t0 + trow[i] >= n1) sp -= (t0 + trow[i] - (n1 - 1)) * (long)(adv[i] / WG_T);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                             (__attribute__((address_space(3))) void*)(dstbuf + (i * 256 + wv * 64) * 8),
                                             16, 0, 2 /* nt: operands are read once */);
            src[i] += adv[i];
        }
    };
    const int li = lane & 15;
    const int lane_off = (8 * kk + (li >> 2)) * RL + 16 * ((lane >> 4) & 1) + 4 * (li & 3);
    const int bbase = a.J + 32 * kb;
    auto tr = [&](const T* ptr) -> uint2 {
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
        return *reinterpret_cast<const uint2*>(&v);
    };
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.f;

    if (nchunks > 0) issue(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                 // chunk c has landed; buffer (c+1)&1 is free
        if (c + 1 < nchunks) issue(c + 1);
        const T* buf = wlds + (c & 1) * (WG_T * RL);
        const long t0 = n0 + (long)c * WG_T;
        const int nvalid = (n1 - t0 < WG_T) ? (int)(n1 - t0) : WG_T;
        const bool ragged = nvalid < WG_T;
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            const T* base = buf + 16 * g * RL + lane_off;
            uint2 bq[2], aq[NP][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                bq[r] = tr(base + 4 * r * RL + bbase);
#pragma unroll
                for (int p = 0; p < NP; ++p) aq[p][r] = tr(base + 4 * r * RL + jcol[p]);
            }
            unsigned m[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            if (ragged) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int px = 16 * g + 8 * kk + 2 * d;
                    m[d] = (px < nvalid ? 0x0000ffffu : 0u) | (px + 1 < nvalid ? 0xffff0000u : 0u);
                }
            }
            const uint4 bw = make_uint4(bq[0].x, bq[0].y, bq[1].x, bq[1].y);
            const bf16x8 bpk = *reinterpret_cast<const bf16x8*>(&bw);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const uint4 aw = make_uint4(aq[p][0].x & m[0] & jm[p], aq[p][0].y & m[1] & jm[p], aq[p][1].x & m[2] & jm[p],
                                            aq[p][1].y & m[3] & jm[p]);
                const bf16x8 apk = *reinterpret_cast<const bf16x8*>(&aw);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(apk, bpk, acc[p], 0, 0, 0);
                if (kb == 0) bacc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(apk, ones, bacc[p], 0, 0, 0);   // wave-uniform
            }
        }
    }
    float* slab = a.slab + (long)blockIdx.x * a.stride;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int jb = jb0 + p * jbs;
        if (jb < NJB) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * kk;
                slab[(long)(32 * jb + row) * a.K + 32 * kb + col] = acc[p][q];
                if (kb == 0 && col == 0) slab[(long)a.J * a.K + 32 * jb + row] = bacc[p][q];
            }
        }
    }
}

// out[j*sj + k*sk] = sum_g slab[g][j*K + k];  bias[j] = sum_g slab[g][tail_off + j] for j < tail_len.
// With kan_O > 0 the rows are in lane order (half, e) of the fused feed+KAN kernel and are scattered to
// the reference's channel numbering (rows that are padding are dropped).  blockIdx.y selects a batch
// entry: slab advances by G*stride, out by out_bstride, bias by tail_len.
struct RedArgs {
    const float* slab; int G; int J, K; long stride;
    float* out; int sj, sk;
    float* bias;
    int kan_O, kan_Kc, kan_nbh;   // nbh = per-half padded length (NB*16)
    int tail_off, tail_len;
    long out_bstride;
};

constexpr int RED_GROUPS = 32;     // slab groups per workgroup (1024 threads)
constexpr int RED_GROUPS_FEW = 8;  // ... when there are few partial rows (batched Gram products: a handful per image)

template <int GROUPS>
__device__ __forceinline__ void reduce_body(const RedArgs& a, int bx, int by, float (*part)[33]) {
    // 32 outputs x GROUPS slab groups per workgroup (lanes = consecutive outputs: 128-byte rows of the slabs); every
    // thread keeps 4 independent partial sums; all partials are combined in a fixed order
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int idx = bx * 32 + o;
    const int total = a.J * a.K + (a.bias ? a.tail_len : 0);
    const bool is_bias = idx >= a.J * a.K;
    const float* slab = a.slab + (long)by * a.G * a.stride + (is_bias ? a.tail_off + idx - a.J * a.K : idx);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (idx < total) {
        int g = grp;
        for (; g + 3 * GROUPS < a.G; g += 4 * GROUPS) {
            s0 += slab[(long)g * a.stride];
            s1 += slab[(long)(g + GROUPS) * a.stride];
            s2 += slab[(long)(g + 2 * GROUPS) * a.stride];
            s3 += slab[(long)(g + 3 * GROUPS) * a.stride];
        }
        for (; g < a.G; g += GROUPS) s0 += slab[(long)g * a.stride];
    }
    part[grp][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp != 0 || idx >= total) return;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < GROUPS; ++r) s += part[r][o];
    int j, k;
    if (is_bias) { j = idx - a.J * a.K; k = 0; }
    else { j = idx / a.K; k = idx - j * a.K; }
    if (a.kan_O > 0) {
        const int h = j / a.kan_nbh, e = j - h * a.kan_nbh;
        j = MapKan{a.kan_O, a.kan_Kc}(h, e);
        if (j < 0) return;
    }
    if (is_bias) a.bias[(long)by * a.tail_len + j] = s;
    else a.out[(long)by * a.out_bstride + (long)j * a.sj + (long)k * a.sk] = s;
}

template <int GROUPS>
__global__ void __launch_bounds__(32 * GROUPS) reduce_kernel(RedArgs a) {
    __shared__ float part[GROUPS][33];
    reduce_body<GROUPS>(a, blockIdx.x, blockIdx.y, part);
}

// Several reductions in one launch (the deferred form, ksr_reduce_defer / ksr_reduce_flush): workgroup -> (entry, block
// of 32 outputs) through the prefix table.  Same arithmetic and summation order per output as reduce_kernel<RED_GROUPS>.
constexpr int RED_BATCH_MAX = 30;
struct RedBatch {
    RedArgs e[RED_BATCH_MAX];
    int first[RED_BATCH_MAX + 1];      // first workgroup of entry i; first[n] = grid size
    int n;
};
__global__ void __launch_bounds__(32 * RED_GROUPS) reduce_batch_kernel(RedBatch b) {
    __shared__ float part[RED_GROUPS][33];
    int e = 0;
    while (e + 1 < b.n && (int)blockIdx.x >= b.first[e + 1]) ++e;
    reduce_body<RED_GROUPS>(b.e[e], (int)blockIdx.x - b.first[e], 0, part);
}

// ---------------------------------------------------------------------------------------------
struct ColArgs {
    const float* a; int a_ld, a_c0;                  // a [N, a_ld], 64 channels from a_c0
    const float* b; int b_scalar;                    // b [N,64] or [N]
    float* slab;                                     // [grid][2][64]
    long N;
};

template <class T>
__device__ __forceinline__ float4 ld4(const void* base, long off) {
    if constexpr (sizeof(T) == 4) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
    else {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(base) + off);
        return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                           __uint_as_float(w.y & 0xffff0000u));
    }
}

template <class T>   // storage type of a (and of b unless it is the per-pixel fp32 scalar)
__global__ void __launch_bounds__(256) colreduce_kernel(ColArgs a) {
    __shared__ float red[2][16][64];
    const int c4 = (threadIdx.x & 15) * 4, rg = threadIdx.x >> 4;     // 16 row groups x 16 channel quads
    const long per = (a.N + gridDim.x - 1) / gridDim.x;
    const long n0 = (long)blockIdx.x * per;
    const long n1 = (n0 + per < a.N) ? n0 + per : a.N;
    float sab[4] = {0.f, 0.f, 0.f, 0.f}, sa[4] = {0.f, 0.f, 0.f, 0.f};
    for (long n = n0 + rg; n < n1; n += 64) {
        float4 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long nn = n + 16 * u;
            const bool ok = nn < n1;
            const long nc = ok ? nn : n;
            av[u] = ld4<T>(a.a, nc * a.a_ld + a.a_c0 + c4);
            if (a.b_scalar) { const float s = a.b[nc]; bv[u] = make_float4(s, s, s, s); }
            else bv[u] = ld4<T>(a.b, nc * 64 + c4);
            if (!ok) av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sab[0] += av[u].x * bv[u].x; sab[1] += av[u].y * bv[u].y; sab[2] += av[u].z * bv[u].z; sab[3] += av[u].w * bv[u].w;
            sa[0] += av[u].x; sa[1] += av[u].y; sa[2] += av[u].z; sa[3] += av[u].w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][rg][c4 + e] = sab[e]; red[1][rg][c4 + e] = sa[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int w = threadIdx.x >> 6, c = threadIdx.x & 63;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[w][r][c];
        a.slab[(long)blockIdx.x * 128 + w * 64 + c] = s;
    }
}

// ---------------------------------------------------------------------------------------------
struct DwgArgs {
    const float* in; int in_ld, in_c0;     // conv input (NHWC)
    const float* ds; int ds_ld, ds_c0;     // gradient of the conv output, 64 channels from ds_c0
    float* slab;                           // [grid][10][64]: 9 taps + bias
    int B, H, W;
    const float* aff_v; const float* aff_ab;   // AFF: in[n][c] = relu(A[64 + c] v[n] + B[64 + c]) (v [N], A|B [256]) instead of a tensor
};

template <class T, bool AFF = false>   // storage type of in and ds; AFF: the conv input is rebuilt from one scalar per pixel
__global__ void __launch_bounds__(256) dwgrad_kernel(DwgArgs a) {
    __shared__ __attribute__((aligned(16))) float halo[AFF ? HALO_H * HALO_W : HALO_FLOATS];
    __shared__ float red[10][4][64];
    const int tid = threadIdx.x, c = tid & 63, row = tid >> 6;
    // blockIdx.y = group of 64 channels (several depthwise convs over neighbouring channel blocks of the same tensors)
    a.in_c0 += 64 * (int)blockIdx.y;
    a.ds_c0 += 64 * (int)blockIdx.y;
    a.slab += (long)blockIdx.y * gridDim.x * 640;
    const int tiles_x = (a.W + 31) / 32, tiles_y = (a.H + 3) / 4;
    const long ntiles = (long)a.B * tiles_y * tiles_x;
    const long HW = (long)a.H * a.W;
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.f;
    // Both inputs of a tile are requested one tile ahead: the 32 gradient values of this thread's (row, channel) and
    // its share of the halo tile (rows / columns outside the image are loaded from a clamped address and zeroed later)
    KsrHalo<T> hr;
    T dnext[32];
    float vnext = 0.f;
    auto fetch = [&](long t) {
        const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y);
        const int b = (int)(t / ((long)tiles_x * tiles_y));
        const int y0 = ty * 4, x0 = tx * 32;
        const int y = y0 + row;
        const int yc = y < a.H ? y : a.H - 1;
        const int xmax = (a.W - x0) < 32 ? (a.W - x0) : 32;
        const T* dsp = reinterpret_cast<const T*>(a.ds) + ((long)b * HW + (long)yc * a.W + x0) * a.ds_ld + a.ds_c0 + c;
#pragma unroll
        for (int col = 0; col < 32; ++col) {
            const int cc = col < xmax ? col : xmax - 1;
            dnext[col] = dsp[(long)cc * a.ds_ld];
        }
        if constexpr (AFF) {
            // this thread's pixel of the 6 x 34 window of v (-1 marks a pixel outside the image; v >= 0 inside)
            const int hy = tid / HALO_W, hx = tid % HALO_W, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool in = tid < HALO_H * HALO_W && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            vnext = in ? a.aff_v[(long)b * HW + (long)gy * a.W + gx] : -1.f;
        } else {
            ksr_halo_fetch<T>(a.in, a.in_ld, a.in_c0, b, y0, x0, a.H, a.W, tid, hr);
        }
    };
    const float Ac = AFF ? a.aff_ab[64 + c] : 0.f, Bc = AFF ? a.aff_ab[192 + c] : 0.f;
    const KsrTileWalk tw = ksr_xcd_tile_walk(ntiles);
    if (tw.first < tw.end) fetch(tw.first);
    for (long tile = tw.first; tile < tw.end; tile += tw.step) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
        const int y0 = ty * 4, x0 = tx * 32;
        const int y = y0 + row;
        const int xmax = (a.W - x0) < 32 ? (a.W - x0) : 32;
        float dv[32];
#pragma unroll
        for (int col = 0; col < 32; ++col) dv[col] = (float)dnext[col];
        __syncthreads();
        if constexpr (AFF) { if (tid < HALO_H * HALO_W) halo[tid] = vnext; }
        else ksr_halo_commit<T>(halo, tid, hr);
        __syncthreads();
        if (tile + tw.step < tw.end) fetch(tile + tw.step);
        {
#pragma unroll
            for (int col = 0; col < 32; ++col) {
                dv[col] = (col < xmax && y < a.H) ? dv[col] : 0.f;
                acc[9] += dv[col];
            }
            // sliding 3-wide window along the row: 34 LDS reads per tap row instead of 96
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float* hp = AFF ? halo + (row + dy) * HALO_W : halo + (row + dy) * HALO_W * HALO_LD + c;
                auto at = [&](int k) -> float {
                    if constexpr (AFF) { const float vv = hp[k]; return vv >= 0.f ? fmaxf(fmaf(Ac, vv, Bc), 0.f) : 0.f; }
                    else return hp[k * HALO_LD];
                };
                float h0 = at(0), h1 = at(1);
#pragma unroll
                for (int col = 0; col < 32; ++col) {
                    const float h2 = at(col + 2);
                    acc[dy * 3 + 0] += dv[col] * h0;
                    acc[dy * 3 + 1] += dv[col] * h1;
                    acc[dy * 3 + 2] += dv[col] * h2;
                    h0 = h1; h1 = h2;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) red[i][row][c] = acc[i];
    __syncthreads();
    for (int i = tid; i < 640; i += 256) {
        const int t = i >> 6, cc = i & 63;
        a.slab[(long)blockIdx.x * 640 + i] = red[t][0][cc] + red[t][1][cc] + red[t][2][cc] + red[t][3][cc];
    }
}

// ---------------------------------------------------------------------------------------------
// First FFN front (pointwise1 1->1, depthwise 3x3 on one channel): u = w1*Y + b1 inside the image,
// pre = bd + sum_tap wd[tap]*u(.+tap), v = relu(pre).  Given dpre [N]:
//   slab[g][0] = sum du*Y; [1] = sum du; [2..10] = sum dpre * u(.+tap); [11] = sum dpre,
//   du = sum_tap wd[tap] * dpre(.-tap).
struct StemgArgs {
    const float* x_img; long img_bstride;
    const float* dpre;
    const float* p1w; const float* p1b; const float* dww;
    float* slab;    // [grid][12]
    int B, H, W;
};

// Every one of the 12 sums is sum_m dpre[m] * f(m) with f a function of the 3x3 window of Y around m: the two sums over
// du are re-indexed (n = m + tap): sum_n du[n] Y[n] = sum_m dpre[m] sum_tap wd[tap] Y(m + tap), m + tap inside the image.
// A thread owns 4 neighbouring pixels of a row (VEC4: W % 4 == 0, 16-byte loads) or one pixel.
template <bool VEC4>
__global__ void __launch_bounds__(256) stemgrad_kernel(StemgArgs a) {
    __shared__ float red[12][4];
    constexpr int PX = VEC4 ? 4 : 1;
    const long HW = (long)a.H * a.W;
    const long N = (long)a.B * HW;
    const float w1 = a.p1w[0], b1 = a.p1b[0];
    float wd[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wd[t] = a.dww[t];
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (long n = ((long)blockIdx.x * blockDim.x + threadIdx.x) * PX; n < N; n += (long)gridDim.x * blockDim.x * PX) {
        const int b = (int)(n / HW);
        const int hw = (int)(n % HW);
        const int y = hw / a.W, x = hw % a.W;
        const float* Y = a.x_img + (long)b * a.img_bstride;
        float d[PX];
        if constexpr (VEC4) {
            const float4 t = *reinterpret_cast<const float4*>(a.dpre + n);
            d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
        } else {
            d[0] = a.dpre[n];
        }
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            const bool rowin = yy >= 0 && yy < a.H;
            const float* Yr = Y + (long)(rowin ? yy : y) * a.W;
            float win[PX + 2];      // Y(yy, x - 1 .. x + PX), 0 outside the image
            bool in[PX + 2];
            in[0] = rowin && x > 0;
            win[0] = in[0] ? Yr[x - 1] : 0.f;
            if constexpr (VEC4) {
                const float4 t = *reinterpret_cast<const float4*>(Yr + x);
                win[1] = rowin ? t.x : 0.f; win[2] = rowin ? t.y : 0.f; win[3] = rowin ? t.z : 0.f; win[4] = rowin ? t.w : 0.f;
                in[1] = in[2] = in[3] = in[4] = rowin;
            } else {
                in[1] = rowin;
                win[1] = rowin ? Yr[x] : 0.f;
            }
            in[PX + 1] = rowin && x + PX < a.W;
            win[PX + 1] = in[PX + 1] ? Yr[x + PX] : 0.f;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int tap = (dy + 1) * 3 + (dx + 1);
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    const float yt = win[p + 1 + dx];
                    const float one = in[p + 1 + dx] ? 1.f : 0.f;
                    acc[2 + tap] += d[p] * fmaf(w1, yt, b1 * one);
                    acc[0] += d[p] * (wd[tap] * yt);
                    acc[1] += d[p] * (wd[tap] * one);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[11] += d[p];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[i][wv] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) a.slab[(long)blockIdx.x * 12 + threadIdx.x] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// dW[j][k] = dW[j][k] * gain[k] + db[j] * shift[k]
__global__ void __launch_bounds__(256) affine_fold_kernel(float* dW, const float* db, const float* gain, const float* shift,
                                                          int J, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= J * K) return;
    const int j = idx / K, k = idx - j * K;
    dW[idx] = fmaf(dW[idx], gain[k], db[j] * shift[k]);
}

// out[j][k] = r[j]*u[k] + c[j]*w[k]
__global__ void __launch_bounds__(256) outer2_kernel(float* out, const float* r, const float* u, const float* c,
                                                     const float* w, int J, int K, float* c_copy) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= J * K) return;
    const int j = idx / K, k = idx - j * K;
    out[idx] = r[j] * u[k] + c[j] * w[k];
    if (c_copy && k == 0) c_copy[j] = c[j];
}

// Deferred reductions of the calling thread (ksr_reduce_defer): collected here instead of launched, until ksr_reduce_flush.
// One list per stream that recorded (at most RED_STREAMS streams per thread: the backward's chain and its second stream):
// a list that fills its window is run on its own stream without regard to the others.
constexpr int RED_STREAMS = 4;
struct RedList { hipStream_t stream; RedBatch b; };
thread_local bool g_red_defer = false;
thread_local int g_red_window = 0;           // > 0: a stream's list is run as soon as it holds that many entries
thread_local RedList g_red[RED_STREAMS];     // entries with b.n == 0 are free

int red_pending() {
    int n = 0;
    for (int i = 0; i < RED_STREAMS; ++i) n += g_red[i].b.n;
    return n;
}

int flush_one(RedBatch& b, hipStream_t s) {
    if (b.n == 0) return KSR_OK;
    const int blocks = b.first[b.n];
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)blocks), dim3(32 * RED_GROUPS), 0, s, b);
    b.n = 0;
    KSR_LAUNCH_CHECK("reduce(batch)");
    return KSR_OK;
}

// every list, in the order of the streams' first use, on stream s (the caller has ordered s after the producers)
int flush_reduce_batch(hipStream_t s) {
    for (int i = 0; i < RED_STREAMS; ++i)
        if (int rc = flush_one(g_red[i].b, s)) return rc;
    return KSR_OK;
}

RedBatch* red_list_of(hipStream_t s) {
    for (int i = 0; i < RED_STREAMS; ++i)
        if (g_red[i].b.n > 0 && g_red[i].stream == s) return &g_red[i].b;
    for (int i = 0; i < RED_STREAMS; ++i)
        if (g_red[i].b.n == 0) { g_red[i].stream = s; return &g_red[i].b; }
    return nullptr;
}

int launch_reduce(const float* slab, int G, int J, int K, long stride, float* out, int sj, int sk, float* bias,
                  int kan_O, int kan_Kc, int kan_nbh, hipStream_t s, int tail_off = -1, int tail_len = -1, int nb = 1,
                  long out_bstride = 0) {
    if (tail_off < 0) { tail_off = J * K; tail_len = J; }
    RedArgs r{slab, G, J, K, stride, out, sj, sk, bias, kan_O, kan_Kc, kan_nbh, tail_off, tail_len, out_bstride};
    const int total = J * K + (bias ? tail_len : 0);
    if (g_red_defer && nb == 1 && G > 4 * RED_GROUPS_FEW) {
        // (the few-row form sums in another grouping and batched entries keep their own launch: results stay bit-equal)
        RedBatch* bp = red_list_of(s);
        KSR_UNSUPPORTED(bp == nullptr, "reduce: more than %d streams with recorded reductions", RED_STREAMS);
        RedBatch& b = *bp;
        if (b.n == RED_BATCH_MAX) {
            if (int rc = flush_one(b, s)) return rc;
        }
        if (b.n == 0) b.first[0] = 0;
        b.e[b.n] = r;
        b.first[b.n + 1] = b.first[b.n] + (total + 31) / 32;
        ++b.n;
        if (g_red_window > 0 && b.n >= g_red_window) return flush_one(b, s);
        return KSR_OK;
    }
    // few partial rows (G <= 32: the per-image Gram products): 8 groups keep every thread of a workgroup at work
    if (G <= 4 * RED_GROUPS_FEW) hipLaunchKernelGGL(reduce_kernel<RED_GROUPS_FEW>, dim3((total + 31) / 32, nb), dim3(32 * RED_GROUPS_FEW), 0, s, r);
    else hipLaunchKernelGGL(reduce_kernel<RED_GROUPS>, dim3((total + 31) / 32, nb), dim3(32 * RED_GROUPS), 0, s, r);
    KSR_LAUNCH_CHECK("reduce");
    return KSR_OK;
}

// Launches the accumulation kernel on a (g, nb) grid; a.stride and a.slab are set by the caller.
template <int PREC, int T, int NP, bool HAS2>
int launch_wgrad_inst_p(const WgArgs& a, size_t lds_bytes, dim3 grid, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_kernel<PREC, T, NP, HAS2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { ksr_set_error("wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e)); return KSR_ERR_HIP; }
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_kernel<PREC, T, NP, HAS2>), grid, dim3(256), lds_bytes, stream, a);
    KSR_LAUNCH_CHECK("wgrad");
    return KSR_OK;
}

template <int T, int NP, bool HAS2>
int launch_wgrad_inst(const WgArgs& a, size_t lds_bytes, dim3 grid, hipStream_t stream) {
#define KSR_CALL(P) return launch_wgrad_inst_p<P, T, NP, HAS2>(a, lds_bytes, grid, stream)
    KSR_DISPATCH_PREC(KSR_CALL);
#undef KSR_CALL
    return KSR_OK;
}

int launch_wgrad_kernel(const WgArgs& a, int per_wave, int T, size_t lds_bytes, unsigned g, unsigned nb, hipStream_t stream) {
    const dim3 grid(g, nb);
    if (a.a2) {     // the one layer with a product operand (fn_2 of the gated blocks) has one block per wave
        KSR_UNSUPPORTED(per_wave != 1, "wgrad: a2 is only supported for J = K = 64");
        return T == 64 ? launch_wgrad_inst<64, 1, true>(a, lds_bytes, grid, stream)
                       : launch_wgrad_inst<32, 1, true>(a, lds_bytes, grid, stream);
    }
#define KSR_WG_CASE(NP)                                                                             \
    case NP: return T == 64 ? launch_wgrad_inst<64, (NP <= 3 ? NP : 1), false>(a, lds_bytes, grid, stream) \
                            : launch_wgrad_inst<32, NP, false>(a, lds_bytes, grid, stream);
    switch (per_wave) {     // wgrad_shape picks 64-pixel chunks only for per_wave <= 3
        KSR_WG_CASE(1) KSR_WG_CASE(2) KSR_WG_CASE(3) KSR_WG_CASE(4) KSR_WG_CASE(5) KSR_WG_CASE(6)
    }
#undef KSR_WG_CASE
    ksr_set_error("wgrad: %d row blocks per wave not supported", per_wave);
    return KSR_ERR_UNSUPPORTED;
}

// Chunk length and LDS bytes for staged rows of RL floats: 64 pixels while two buffers stay within 64 KiB
// (more bytes in flight per workgroup) and a wave owns at most 3 row blocks (register budget), else 32.
void wgrad_shape(int RL, int per_wave, int* T, size_t* lds_bytes) {
    const int esz = ksr_prec() == KSR_BF16S ? 2 : 4;       // bytes per stored element
    *T = (2 * 64 * RL * esz <= 64 * 1024 && per_wave <= 3) ? 64 : 32;
    *lds_bytes = (size_t)2 * *T * RL * esz;
}
int wgrad_align() { return ksr_prec() == KSR_BF16S ? 8 : 4; }   // elements per 16-byte piece

// Workgroups per batch entry of a batched Gram product (nb entries of `tokens` pixels, Jp = J rounded up to 32).
long gram_grid(int Jp, int K, int nb, long tokens) {
    int T; size_t lds;
    const int NKB = K / 32;
    wgrad_shape(Jp + K, (Jp / 32 + (4 / NKB) - 1) / (4 / NKB), &T, &lds);
    long cap = WG_GRID / nb;
    if (lds > 40 * 1024) cap = (WG_GRID / 2) / nb;
    if (cap < 1) cap = 1;
    long g = (tokens + 4 * T - 1) / (4 * T);
    return g < 1 ? 1 : (g > cap ? cap : g);
}

}  // namespace

extern "C" int ksr_wgrad_slab_floats(int J, int K) { return WG_GRID * (J * K + J); }

// Deferred slab reductions.  Every entry point that accumulates a parameter gradient writes per-workgroup partial rows into
// a workspace ("slab") and finishes with a small fixed-order reduction launch.  After ksr_reduce_defer(1) the calling thread's
// entry points only RECORD that reduction (up to 30; a full list is flushed on the stream of the call that overflows it), and
// ksr_reduce_flush(stream) runs all recorded ones in a single launch: same sums, same order, same bits.  The caller must
// then give every such call its OWN slab and keep slabs and outputs alive and untouched until the flush, and must order the
// flush after every producer (same stream, or an event).  ksr_reduce_defer(0) ends the mode (the list must be empty).
// ksr_reduce_defer(k), k >= 2: as 1, and the reductions recorded by calls on one stream are run ON THAT STREAM as soon as
// there are k of them (lists are kept per stream): fewer launches while the slabs are still in the caches.
bool ksr_reduce_deferring() { return g_red_defer; }

extern "C" int ksr_reduce_defer(int on) {
    if (on < 0) {                 // abandon: drop what was recorded (error paths of the caller)
        for (int i = 0; i < RED_STREAMS; ++i) g_red[i].b.n = 0;
        ksr_unet_reduce_clear();
        g_red_defer = false;
        g_red_window = 0;
        return KSR_OK;
    }
    const int pending = red_pending() + ksr_unet_reduce_pending();
    KSR_CHECK_ARG(on || pending == 0, "reduce_defer(0): %d recorded reductions were never flushed", pending);
    g_red_defer = on != 0;
    g_red_window = on >= 2 ? on : 0;
    return KSR_OK;
}
extern "C" int ksr_reduce_flush(hipStream_t stream) {
    if (int rc = flush_reduce_batch(stream)) return rc;
    return ksr_unet_reduce_flush(stream);
}
extern "C" int ksr_reduce_pending(void) { return red_pending() + ksr_unet_reduce_pending(); }

// Fixed-order sum of G partial rows [J*K | J] (written by a kernel that accumulated the products itself) to dW, db.
int ksr_slab_reduce(const float* slab, int G, int J, int K, float* dW, float* db, hipStream_t stream) {
    return launch_reduce(slab, G, J, K, (long)J * K + J, dW, K, 1, db, 0, 0, 0, stream);
}

// The same for partial rows that hold [J*K | J] at the start of `stride` floats each (several gradients per row).
int ksr_slab_reduce_strided(const float* slab, int G, int J, int K, long stride, float* dW, float* db, hipStream_t stream) {
    return launch_reduce(slab, G, J, K, stride, dW, K, 1, db, 0, 0, 0, stream);
}

// out[j * out_row_stride + k] = sum over G partial rows of `stride` floats that hold [J][K] at their start (no bias part).
int ksr_slab_reduce_rows(const float* slab, int G, int J, int K, long stride, float* out, int out_row_stride, hipStream_t stream) {
    return launch_reduce(slab, G, J, K, stride, out, out_row_stride, 1, nullptr, 0, 0, 0, stream);
}

// dW [J][K] (row-major, or scattered through the KAN channel map when kan_upscale > 0) and db [J].
// slab: workspace of ksr_wgrad_slab_floats(J, K) floats.
extern "C" int ksr_wgrad(const float* dz, int dz_ld, int dz_c0, int J, const float* a1, int a1_ld, int a1_c0, int K,
                         const float* a2, int a2_ld, int a2_c0, int64_t N, float* slab, float* dW, float* db,
                         int kan_upscale, int kan_n_coef, hipStream_t stream) {
    KSR_CHECK_ARG(dz && a1 && slab && dW, "wgrad: null pointer");
    KSR_CHECK_ARG(J > 0 && K > 0 && J % 32 == 0 && (K == 32 || K == 64), "wgrad: J must be a multiple of 32 and K 32 or 64");
    KSR_CHECK_ARG(N > 0, "wgrad: N must be positive");
    const int NKB = K / 32, per_wave = (J / 32 + (4 / NKB) - 1) / (4 / NKB);
    KSR_UNSUPPORTED(per_wave > WG_MAXP, "wgrad: J=%d too large", J);
    const int al = wgrad_align();
    KSR_CHECK_ARG(dz_ld % al == 0 && dz_c0 % al == 0 && a1_ld % al == 0 && a1_c0 % al == 0 && a2_ld % al == 0 && a2_c0 % al == 0,
                  "wgrad: leading dimensions and channel offsets must be multiples of %d", al);
    WgArgs a{dz, dz_ld, dz_c0, J, a1, a1_ld, a1_c0, K, a2, a2_ld, a2_c0, slab, (long)N, J, 0, (long)J * K + J};
    int T; size_t lds_bytes;
    wgrad_shape(J + K + (a2 ? K : 0), per_wave, &T, &lds_bytes);
    KSR_UNSUPPORTED(lds_bytes > 160 * 1024, "wgrad: J=%d needs %zu bytes of LDS", J, lds_bytes);
    long g = (N + 4 * T - 1) / (4 * T);
    if (g > WG_GRID) g = WG_GRID;
    if (lds_bytes > 40 * 1024 && g > WG_GRID / 2) g = WG_GRID / 2;     // 2 workgroups per CU fit
    bool fast = false;
    if (ksr_prec() == KSR_BF16S && !a2 && T == 64) {
        const int rl4 = (J + K) / 8;
#define KSR_WG_FAST(R4, NPV)                                                                                             \
        if (rl4 == R4 && per_wave == NPV) {                                                                              \
            static bool attr_set = false;                                                                                \
            if (!attr_set) {                                                                                             \
                hipError_t e = hipFuncSetAttribute((const void*)wgrad_fast_kernel<R4, NPV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                if (e != hipSuccess) { ksr_set_error("wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e)); return KSR_ERR_HIP; } \
                attr_set = true;                                                                                         \
            }                                                                                                            \
            hipLaunchKernelGGL((wgrad_fast_kernel<R4, NPV>), dim3((unsigned)g), dim3(256), lds_bytes, stream, a);        \
            KSR_LAUNCH_CHECK("wgrad");                                                                                   \
            fast = true;                                                                                                 \
        }
        KSR_WG_FAST(12, 1) else KSR_WG_FAST(16, 1) else KSR_WG_FAST(24, 2) else KSR_WG_FAST(28, 3)
        else KSR_WG_FAST(32, 3)      // 192 x 64: the q | k | v projections of the channel-attention head
#undef KSR_WG_FAST
    }
    if (!fast)
        if (int rc = launch_wgrad_kernel(a, per_wave, T, lds_bytes, (unsigned)g, 1, stream)) return rc;
    int kan_O = 0, kan_nbh = 0;
    if (kan_upscale > 0) {
        kan_O = kan_upscale * kan_upscale;
        kan_nbh = J / 2;
    }
    return launch_reduce(slab, (int)g, J, K, (long)J * K + J, dW, K, 1, db, kan_O, kan_n_coef, kan_nbh, stream);
}

extern "C" int64_t ksr_gram_slab_floats(int J, int K, int nb, int64_t tokens_per_entry) {
    const int Jp = (J + 31) / 32 * 32;
    if (J <= 0 || K <= 0 || nb <= 0 || tokens_per_entry <= 0) return 0;
    return (int64_t)nb * gram_grid(Jp, K, nb, (long)tokens_per_entry) * ((long)Jp * K + 2 * Jp + K);
}

// Batched Gram product over pixels: for each of nb entries (consecutive runs of tokens_per_entry pixels),
//   out[e][j][k] = sum_n dz[n][j] * a[n][k],  tail[e][0:J] = sum_n dz[n][j]
// and with want_sq (J a multiple of 32)  tail[e][J:2J] = sum_n dz[n][j]^2,  tail[e][2J:2J+K] = sum_n a[n][k]^2.
// J a multiple of 4 (the last 32-row block may be partial), K 32 or 64.  slab: ksr_gram_slab_floats floats.
extern "C" int ksr_gram_batched(const float* dz, int dz_ld, int dz_c0, int J, const float* a1, int a1_ld, int a1_c0,
                                int K, int64_t tokens_per_entry, int nb, float* slab, float* out, float* tail,
                                int want_sq, hipStream_t stream) {
    KSR_CHECK_ARG(dz && a1 && slab && out, "gram_batched: null pointer");
    KSR_CHECK_ARG(J > 0 && J % 4 == 0 && (K == 32 || K == 64), "gram_batched: J must be a multiple of 4 and K 32 or 64");
    KSR_CHECK_ARG(tokens_per_entry > 0 && nb > 0 && nb <= 65535, "gram_batched: bad batch");
    KSR_CHECK_ARG(!want_sq || (J % 32 == 0 && tail), "gram_batched: sums of squares need J % 32 == 0 and a tail buffer");
    const int al = wgrad_align();
    KSR_CHECK_ARG(J % al == 0 && dz_ld % al == 0 && dz_c0 % al == 0 && a1_ld % al == 0 && a1_c0 % al == 0,
                  "gram_batched: J, leading dimensions and channel offsets must be multiples of %d", al);
    const int Jp = (J + 31) / 32 * 32;
    const int NKB = K / 32, per_wave = (Jp / 32 + (4 / NKB) - 1) / (4 / NKB);
    KSR_UNSUPPORTED(per_wave > WG_MAXP, "gram_batched: J=%d too large", J);
    const long stride = (long)Jp * K + 2 * Jp + K;
    WgArgs a{dz, dz_ld, dz_c0, Jp, a1, a1_ld, a1_c0, K, nullptr, 0, 0, slab, (long)tokens_per_entry, J, want_sq ? 1 : 0, stride};
    int T; size_t lds_bytes;
    wgrad_shape(Jp + K, per_wave, &T, &lds_bytes);        // sized for padded rows: the masked reads stay inside the buffer
    KSR_UNSUPPORTED(lds_bytes > 160 * 1024, "gram_batched: J=%d needs %zu bytes of LDS", J, lds_bytes);
    const long g = gram_grid(Jp, K, nb, (long)tokens_per_entry);
    if (int rc = launch_wgrad_kernel(a, per_wave, T, lds_bytes, (unsigned)g, (unsigned)nb, stream)) return rc;
    const int tail_len = want_sq ? 2 * J + K : J;
    return launch_reduce(slab, (int)g, J, K, stride, out, K, 1, tail, 0, 0, 0, stream, Jp * K, tail_len, nb, (long)J * K);
}

// ksr_gram_batched for the fused feed+KAN kernel with per-image weights: dz = gparam [N, J] holds the gradient of the
// per-pixel KAN parameters in that kernel's lane order (J = 2 * ksr_feed_kan_lane_len); the rows of the products are
// scattered to the reference's channel numbering, padding rows dropped:
//   out[e][p][k] = sum_n gparam[n][lane(p)] * a[n][k],  tail[e][p] = sum_n gparam[n][lane(p)]   (p < P = r^2 (n_coef + 2))
extern "C" int ksr_gram_batched_kan(const float* gparam, int J, const float* a1, int a1_ld, int a1_c0, int K,
                                    int64_t tokens_per_entry, int nb, float* slab, float* out, float* tail, int kan_upscale,
                                    int kan_n_coef, hipStream_t stream) {
    KSR_CHECK_ARG(gparam && a1 && slab && out && tail, "gram_batched_kan: null pointer");
    KSR_CHECK_ARG(J > 0 && J % 32 == 0 && (K == 32 || K == 64), "gram_batched_kan: J must be a multiple of 32 and K 32 or 64");
    KSR_CHECK_ARG(tokens_per_entry > 0 && nb > 0 && nb <= 65535, "gram_batched_kan: bad batch");
    KSR_CHECK_ARG(kan_upscale > 0 && kan_n_coef > 0 && kan_upscale * kan_upscale / 2 * (kan_n_coef + 2) <= J / 2,
                  "gram_batched_kan: J does not hold the parameters of upscale %d, n_coef %d", kan_upscale, kan_n_coef);
    const int al = wgrad_align();
    KSR_CHECK_ARG(a1_ld % al == 0 && a1_c0 % al == 0, "gram_batched_kan: leading dimension and channel offset must be multiples of %d", al);
    const int NKB = K / 32, per_wave = (J / 32 + (4 / NKB) - 1) / (4 / NKB);
    KSR_UNSUPPORTED(per_wave > WG_MAXP, "gram_batched_kan: J=%d too large", J);
    const long stride = (long)J * K + 2 * J + K;
    WgArgs a{gparam, J, 0, J, a1, a1_ld, a1_c0, K, nullptr, 0, 0, slab, (long)tokens_per_entry, J, 0, stride};
    int T; size_t lds_bytes;
    wgrad_shape(J + K, per_wave, &T, &lds_bytes);
    KSR_UNSUPPORTED(lds_bytes > 160 * 1024, "gram_batched_kan: J=%d needs %zu bytes of LDS", J, lds_bytes);
    const long g = gram_grid(J, K, nb, (long)tokens_per_entry);
    if (int rc = launch_wgrad_kernel(a, per_wave, T, lds_bytes, (unsigned)g, (unsigned)nb, stream)) return rc;
    const int P = kan_upscale * kan_upscale * (kan_n_coef + 2);
    return launch_reduce(slab, (int)g, J, K, stride, out, K, 1, tail, kan_upscale * kan_upscale, kan_n_coef, J / 2, stream, J * K, J,
                         nb, (long)P * K);
}

// out[0][c] = sum_n a[n][c]*b, out[1][c] = sum_n a[n][c] for 64 channels c; rows out_row_stride apart.
// slab: 1024*128 floats.
extern "C" int ksr_colreduce(const float* a, int a_ld, int a_c0, const float* b, int b_is_per_pixel_scalar, int64_t N,
                             float* slab, float* out, int out_row_stride, hipStream_t stream) {
    KSR_CHECK_ARG(a && b && slab && out && N > 0 && a_ld >= 64 && out_row_stride >= 64, "colreduce: bad argument");
    KSR_CHECK_ARG(a_ld % 4 == 0 && a_c0 % 4 == 0, "colreduce: a_ld and a_c0 must be multiples of 4");
    ColArgs c{a, a_ld, a_c0, b, b_is_per_pixel_scalar, slab, (long)N};
    long g = (N + 255) / 256;
    if (g > 1024) g = 1024;
    if (ksr_prec() == KSR_BF16S) hipLaunchKernelGGL(colreduce_kernel<__bf16>, dim3((unsigned)g), dim3(256), 0, stream, c);
    else hipLaunchKernelGGL(colreduce_kernel<float>, dim3((unsigned)g), dim3(256), 0, stream, c);
    KSR_LAUNCH_CHECK("colreduce");
    return launch_reduce(slab, (int)g, 2, 64, 128, out, out_row_stride, 1, nullptr, 0, 0, 0, stream);
}

// Depthwise 3x3 (64 channels) parameter gradients: dW [64][1][3][3], db [64]. slab: 512*640 floats.
// `groups` depthwise convs over consecutive blocks of 64 channels of the same two tensors (the q | k | v branches of the
// channel-attention head) in one launch: dW [groups*64][9], db [groups*64]; slab: groups * 512 * 640 floats.
extern "C" int ksr_dwconv_wgrad_groups(const float* in, int in_ld, int in_c0, const float* ds, int ds_ld, int ds_c0, int groups,
                                       float* slab, float* dW, float* db, int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(in && ds && slab && dW && db && B > 0 && H > 0 && W > 0 && ds_ld >= 64, "dwconv_wgrad: bad argument");
    KSR_CHECK_ARG(groups >= 1 && groups <= 8 && in_c0 + 64 * groups <= in_ld && ds_c0 + 64 * groups <= ds_ld,
                  "dwconv_wgrad: the channel groups do not fit the rows");
    DwgArgs d{in, in_ld, in_c0, ds, ds_ld, ds_c0, slab, B, H, W, nullptr, nullptr};
    long ntiles = (long)B * ((H + 3) / 4) * ((W + 31) / 32);
    const long g = ntiles < 512 ? ntiles : 512;
    const dim3 grid((unsigned)g, (unsigned)groups);
    if (ksr_prec() == KSR_BF16S) hipLaunchKernelGGL(dwgrad_kernel<__bf16>, grid, dim3(256), 0, stream, d);
    else hipLaunchKernelGGL(dwgrad_kernel<float>, grid, dim3(256), 0, stream, d);
    KSR_LAUNCH_CHECK("dwconv_wgrad");
    // slab rows: tap t (J = 9) x channel (K = 64) then the bias row handled as a 10th row via two calls; one batch entry
    // per channel group (its partial rows follow the previous group's, its outputs 576 / 64 floats on)
    if (int rc = launch_reduce(slab, (int)g, 9, 64, 640, dW, 1, 9, nullptr, 0, 0, 0, stream, -1, -1, groups, 576)) return rc;
    return launch_reduce(slab + 576, (int)g, 1, 64, 640, db, 64, 1, nullptr, 0, 0, 0, stream, -1, -1, groups, 64);
}

extern "C" int ksr_dwconv_wgrad(const float* in, int in_ld, int in_c0, const float* ds, int ds_ld, int ds_c0,
                                float* slab, float* dW, float* db, int B, int H, int W, hipStream_t stream) {
    return ksr_dwconv_wgrad_groups(in, in_ld, in_c0, ds, ds_ld, ds_c0, 1, slab, dW, db, B, H, W, stream);
}

// The same for the first gated block's depthwise conv, whose input (the gate half of t) is rebuilt from (v, ab).
extern "C" int ksr_dwconv_wgrad_v(const float* v_saved, const float* ab, const float* ds, int ds_ld, int ds_c0, float* slab,
                                  float* dW, float* db, int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(v_saved && ab && ds && slab && dW && db && B > 0 && H > 0 && W > 0 && ds_ld >= 64, "dwconv_wgrad_v: bad argument");
    DwgArgs d{nullptr, 0, 0, ds, ds_ld, ds_c0, slab, B, H, W, v_saved, ab};
    long ntiles = (long)B * ((H + 3) / 4) * ((W + 31) / 32);
    const long g = ntiles < 512 ? ntiles : 512;
    if (ksr_prec() == KSR_BF16S) hipLaunchKernelGGL((dwgrad_kernel<__bf16, true>), dim3((unsigned)g), dim3(256), 0, stream, d);
    else hipLaunchKernelGGL((dwgrad_kernel<float, true>), dim3((unsigned)g), dim3(256), 0, stream, d);
    KSR_LAUNCH_CHECK("dwconv_wgrad_v");
    if (int rc = launch_reduce(slab, (int)g, 9, 64, 640, dW, 1, 9, nullptr, 0, 0, 0, stream)) return rc;
    return launch_reduce(slab + 576, (int)g, 1, 64, 640, db, 64, 1, nullptr, 0, 0, 0, stream);
}

// Gradients of the first FFN's scalar front.  out [12]: pointwise1 weight, pointwise1 bias, depthwise
// taps (9), depthwise bias (the order of the module's parameters).  slab: 256*12 floats.
extern "C" int ksr_stem_scalar_wgrad(const float* x_img, int64_t img_batch_stride, const float* dpre,
                                     const float* pw1_w, const float* pw1_b, const float* dw_w, float* slab,
                                     float* out12, int B, int H, int W, hipStream_t stream) {
    KSR_CHECK_ARG(x_img && dpre && pw1_w && pw1_b && dw_w && slab && out12 && B > 0 && H > 0 && W > 0,
                  "stem_scalar_wgrad: bad argument");
    StemgArgs s{x_img, (long)img_batch_stride, dpre, pw1_w, pw1_b, dw_w, slab, B, H, W};
    const long N = (long)B * H * W;
    const bool vec4 = W % 4 == 0 && img_batch_stride % 4 == 0 && ((uintptr_t)x_img & 15) == 0 && ((uintptr_t)dpre & 15) == 0;
    long g = (N / (vec4 ? 4 : 1) + 255) / 256;
    if (g > 256) g = 256;
    if (vec4) hipLaunchKernelGGL(stemgrad_kernel<true>, dim3((unsigned)g), dim3(256), 0, stream, s);
    else hipLaunchKernelGGL(stemgrad_kernel<false>, dim3((unsigned)g), dim3(256), 0, stream, s);
    KSR_LAUNCH_CHECK("stem_scalar_wgrad");
    return launch_reduce(slab, (int)g, 1, 12, 12, out12, 12, 1, nullptr, 0, 0, 0, stream);
}

// dW[j][k] = r[j]*u[k] + c[j]*w[k]: weight gradient of a 1x1 conv whose input is an affine function
// u[k]*s + w[k] of one scalar s per pixel (r = sum_n dz*s, c = sum_n dz).
// c_copy (optional) receives c: it is the bias gradient of the same conv.
extern "C" int ksr_outer2(float* out, const float* r, const float* u, const float* c, const float* w, int J, int K,
                          float* c_copy, hipStream_t stream) {
    KSR_CHECK_ARG(out && r && u && c && w && J > 0 && K > 0, "outer2: bad argument");
    hipLaunchKernelGGL(outer2_kernel, dim3((J * K + 255) / 256), dim3(256), 0, stream, out, r, u, c, w, J, K, c_copy);
    KSR_LAUNCH_CHECK("outer2");
    return KSR_OK;
}

// Weight gradient of a linear map whose input is a LayerNorm output s = xhat * gain + shift, from the product taken with
// xhat instead of s (so that s need not be kept): dW[j][k] <- dW[j][k] * gain[k] + db[j] * shift[k], in place.
extern "C" int ksr_wgrad_affine_input(float* dW, const float* db, const float* gain, const float* shift, int J, int K,
                                      hipStream_t stream) {
    KSR_CHECK_ARG(dW && db && gain && shift && J > 0 && K > 0, "wgrad_affine_input: bad argument");
    hipLaunchKernelGGL(affine_fold_kernel, dim3((J * K + 255) / 256), dim3(256), 0, stream, dW, db, gain, shift, J, K);
    KSR_LAUNCH_CHECK("wgrad_affine_input");
    return KSR_OK;
}
