// Backward of the attention in-projection (nn.MultiheadAttention's in_proj, gazetr.py:39) in ONE pass over dqkv, bf16
// storage mode:
//     ds[n][k]  = resid[n][k] + sum_j dqkv[n][j] W[j][k]      (times y (1 - y), y = the layer input, for the first layer)
//     dW[j][k]  = sum_n dqkv[n][j] x[n][k],   db[j] = sum_n dqkv[n][j]
// The data gradient and the weight gradient both consume dqkv [N,192], the widest tensor of the transformer backward;
// as two kernels (ksr_in_proj_bwd, ksr_wgrad) it is read twice.  Here a workgroup walks its pixel range in chunks of 32
// pixels whose rows [dqkv 192 | x 64 | resid 64] are copied global -> LDS by 16-byte LDS-DMA loads (three stages,
// non-temporal), and both products read their operands from that one staged copy:
//   * the weight gradient contracts over pixels: operands are read with the transposing ds_read_b64_tr_b16 (8 pixels
//     of one channel per lane); the 6 x 2 accumulator blocks are split over the four waves (3 each);
//   * the data gradient contracts over the 192 channels: a lane's operand is 8 consecutive channels of its pixel, one
//     16-byte LDS read (rows are padded by one 16-byte piece, so the 32 rows a wave reads fall into different banks);
//     W^T sits in registers in MFMA fragment order; waves 0 and 1 each produce 32 of the 64 output channels.
// Partial weight gradients go to one slab row per workgroup and are summed in a fixed order (deterministic).
#include "ksr_gemm.h"
#include "ksr_wg_fuse.h"

namespace {

constexpr int IP_J = 192, IP_K = 64, IP_T = 32;
constexpr int IP_RL = IP_J + 2 * IP_K;            // staged elements per row
constexpr int IP_RL4 = IP_RL / 8 + 1;             // 16-byte pieces per row incl. one of padding
constexpr int IP_RLP = IP_RL4 * 8;                // padded row length in elements
constexpr int IP_ROW = IP_J * IP_K + IP_J;        // slab floats per workgroup
constexpr int IP_GRID = 512;

struct InProjArgs {
    const __bf16* dqkv;     // [N,192]
    const __bf16* xin;      // [N,64] input of the layer (= y of the sigmoid for the first layer)
    const __bf16* resid;    // [N,64]
    const float* w;         // [192,64] in_proj_weight
    __bf16* ds;             // [N,64]
    float* slab;            // [gridDim.x][IP_ROW]
    long N;
};

constexpr int IP_NVEC = IP_T * IP_RL4;            // 16-byte pieces per staged chunk (1312)
constexpr int IP_NFULL = IP_NVEC / 256;           // transfer instructions every wave issues per chunk (5)
constexpr int IP_STAGE = (IP_NFULL + 1) * 256 * 8;   // elements per stage: the sixth, partial instruction of wave 0 writes a whole KB
constexpr int IP_STAGES = 3;

typedef short s16x4 __attribute__((ext_vector_type(4)));

// Three stages of 32 pixels, two chunks in flight per workgroup: with two (one in flight, 21 KB) the kernel ran at
// chunk size x workgroups per CU / memory latency = 4.3 TB/s.  The transfers are issued from inline assembly and waited
// for with counted vmcnt (ksr_dma_barrier_but): a wave may keep its youngest transfers and stores in flight.  Per chunk
// every wave issues IP_NFULL transfer instructions, wave 0 one more (the last 32 pieces; its upper lanes fetch a valid
// address into the stage's padding), waves 0 and 1 four 8-byte stores of the data gradient.
template <bool SIGGRAD>
__global__ void __launch_bounds__(256, 2) inproj_bwd_fused_kernel(InProjArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 stage[IP_STAGES * IP_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, kk = lane >> 5;

    // data gradient (waves 0, 1): W^T fragments of the wave's 32 output channels stay in registers (48):
    // element i of step s = W[16 s + 8 kk + i][32 wv + col]
    bf16x8 wreg[12];
    if (wv < 2) {
#pragma unroll
        for (int s12 = 0; s12 < 12; ++s12)
#pragma unroll
            for (int i = 0; i < 8; ++i) wreg[s12][i] = (__bf16)a.w[(16 * s12 + 8 * kk + i) * 64 + 32 * wv + col];
    }

    const long per = (((a.N + gridDim.x - 1) / gridDim.x) + IP_T - 1) / IP_T * IP_T;
    const long n0 = (long)blockIdx.x * per;
    const long n1 = (n0 + per < a.N) ? n0 + per : a.N;
    const int nchunks = n1 > n0 ? (int)((n1 - n0 + IP_T - 1) / IP_T) : 0;

    // piece f = i * 256 + tid of a chunk: staged row f / 41, piece f % 41 of [dqkv 24 | x 8 | resid 8 | padding 1]
    const __bf16* src[IP_NFULL + 1];
    int trow[IP_NFULL + 1], ldr[IP_NFULL + 1];
#pragma unroll
    for (int i = 0; i <= IP_NFULL; ++i) {
        int f = i * 256 + tid;
        if (f >= IP_NVEC) f = IP_NVEC - 1;                // upper lanes of wave 0's sixth instruction
        trow[i] = f / IP_RL4;
        const int piece = f - trow[i] * IP_RL4;
        const __bf16* base = a.dqkv + 8 * (piece < 24 ? piece : 0);      // also the padding piece
        ldr[i] = IP_J;
        if (piece >= 24 && piece < 32) { base = a.xin + 8 * (piece - 24); ldr[i] = IP_K; }
        if (piece >= 32 && piece < 40) { base = a.resid + 8 * (piece - 32); ldr[i] = IP_K; }
        src[i] = base + (n0 + trow[i]) * ldr[i];
    }
    auto issue = [&](int c) {
        __bf16* dstbuf = stage + (c % IP_STAGES) * IP_STAGE;
        const long t0 = n0 + (long)c * IP_T;
        const bool ragged = t0 + IP_T > n1;               // wave-uniform: rows past the range repeat its last row (masked later)
#pragma unroll
        for (int i = 0; i <= IP_NFULL; ++i) {
            if (i == IP_NFULL && wv != 0) break;          // wave-uniform
            const __bf16* sp = src[i];
            if (ragged && t0 + trow[i] >= n1) sp -= (t0 + trow[i] - (n1 - 1)) * (long)ldr[i];
            ksr_dma16<true>(sp, dstbuf + (i * 256 + wv * 64) * 8);      // nt: every row is read once
            src[i] += IP_T * ldr[i];
        }
    };

    // weight gradient: wave wv owns column block kb of x and row blocks jb0, jb0 + 2, jb0 + 4 of dqkv
    const int kb = wv & 1, jb0 = wv >> 1;
    f32x16 acc[3];
    float bsum[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[p][q] = 0.f;
    const int li = lane & 15;
    const int lane_off = (8 * kk + (li >> 2)) * IP_RLP + 16 * ((lane >> 4) & 1) + 4 * (li & 3);
    auto tr = [&](const __bf16* ptr) -> uint2 {
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)ptr);
        return *reinterpret_cast<const uint2*>(&v);
    };

    // Vector-memory instructions a wave issues after the transfers of chunk c and before it needs them: the stores of
    // chunk c-2 (S), the transfers of chunk c+1 (D), the stores of chunk c-1 (S); fewer for the first two chunks; the
    // last chunk of the range (no younger transfers, possibly skipped stores) waits for everything.
    auto wait_chunk = [&](int c) {
        if (c == nchunks - 1) { ksr_dma_barrier(); return; }
        if (wv == 0) { if (c == 0) ksr_dma_barrier_but<6>(); else if (c == 1) ksr_dma_barrier_but<10>(); else ksr_dma_barrier_but<14>(); }
        else if (wv == 1) { if (c == 0) ksr_dma_barrier_but<5>(); else if (c == 1) ksr_dma_barrier_but<9>(); else ksr_dma_barrier_but<13>(); }
        else ksr_dma_barrier_but<5>();
    };

    if (nchunks > 0) issue(0);
    if (nchunks > 1) issue(1);
    for (int c = 0; c < nchunks; ++c) {
        wait_chunk(c);                   // chunk c has landed for every wave; the stage of chunk c-1 (= of chunk c+2) is free
        if (c + 2 < nchunks) issue(c + 2);
        const __bf16* buf = stage + (c % IP_STAGES) * IP_STAGE;
        const long t0 = n0 + (long)c * IP_T;
        const int nvalid = (n1 - t0 < IP_T) ? (int)(n1 - t0) : IP_T;

        // ---- weight gradient: two 16-pixel steps
#pragma unroll
        for (int g = 0; g < IP_T / 16; ++g) {
            const __bf16* base = buf + 16 * g * IP_RLP + lane_off;
            uint2 bq[2], aq[3][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                bq[r] = tr(base + 4 * r * IP_RLP + IP_J + 32 * kb);
#pragma unroll
                for (int p = 0; p < 3; ++p) aq[p][r] = tr(base + 4 * r * IP_RLP + 32 * (jb0 + 2 * p));
            }
            // this lane's 8 operand elements are pixels 16 g + 8 kk + e: zero the ones past the range
            unsigned m[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            if (nvalid < IP_T) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int px = 16 * g + 8 * kk + 2 * d;
                    m[d] = (px < nvalid ? 0x0000ffffu : 0u) | (px + 1 < nvalid ? 0xffff0000u : 0u);
                }
            }
            const uint4 bw = make_uint4(bq[0].x, bq[0].y, bq[1].x, bq[1].y);
            const bf16x8 bpk = *reinterpret_cast<const bf16x8*>(&bw);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint4 aw = make_uint4(aq[p][0].x & m[0], aq[p][0].y & m[1], aq[p][1].x & m[2], aq[p][1].y & m[3]);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&aw), bpk, acc[p], 0, 0, 0);
                if (kb == 0) {                               // wave-uniform: the bias gradient of this wave's row blocks
                    float x[8];
                    ksr_unpack8(aw, x);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[p] += x[e];
                }
            }
        }

        // ---- data gradient: waves 0 and 1, 32 output channels each, K = 192 in 12 steps
        if (wv < 2) {
            f32x16 d;
#pragma unroll
            for (int q = 0; q < 16; ++q) d[q] = 0.f;
            const __bf16* rowp = buf + col * IP_RLP + 8 * kk;
#pragma unroll
            for (int s = 0; s < 12; ++s)
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[s], *reinterpret_cast<const bf16x8*>(rowp + 16 * s), d, 0, 0, 0);
            if (col < nvalid) {
                __bf16* dst = a.ds + (t0 + col) * IP_K;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = 32 * wv + 8 * q + 4 * kk;            // accumulator registers 4q..4q+3 = channels c0..c0+3
                    const uint2 rr = *reinterpret_cast<const uint2*>(buf + col * IP_RLP + IP_J + IP_K + c0);
                    float o[4] = {d[4 * q] + __uint_as_float(rr.x << 16), d[4 * q + 1] + __uint_as_float(rr.x & 0xffff0000u),
                                  d[4 * q + 2] + __uint_as_float(rr.y << 16), d[4 * q + 3] + __uint_as_float(rr.y & 0xffff0000u)};
                    if (SIGGRAD) {
                        const uint2 yy = *reinterpret_cast<const uint2*>(buf + col * IP_RLP + IP_J + c0);
                        const float y[4] = {__uint_as_float(yy.x << 16), __uint_as_float(yy.x & 0xffff0000u),
                                            __uint_as_float(yy.y << 16), __uint_as_float(yy.y & 0xffff0000u)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = o[e] * y[e] * (1.f - y[e]);
                    }
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (__bf16)o[e];
                    *reinterpret_cast<uint2*>(dst + c0) = *reinterpret_cast<const uint2*>(&pk);
                }
            }
        }
    }

    float* slab = a.slab + (long)blockIdx.x * IP_ROW;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int jb = jb0 + 2 * p;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * kk;
            slab[(long)(32 * jb + row) * IP_K + 32 * kb + col] = acc[p][q];
        }
        if (kb == 0) {
            const float bs = ksr_pair_sum(bsum[p]);
            if (kk == 0) slab[(long)IP_J * IP_K + 32 * jb + col] = bs;
        }
    }
}

}  // namespace

extern "C" int64_t ksr_in_proj_bwd_wg_slab_floats(void) { return (int64_t)IP_GRID * IP_ROW; }

extern "C" int ksr_in_proj_bwd_wg(const float* dqkv, const float* in_w, const float* resid, const float* x_in, int sigmoid_input,
                                  float* ds, float* slab, float* dW, float* db, int64_t N, hipStream_t stream) {
    KSR_CHECK_ARG(dqkv && in_w && resid && x_in && ds && slab && dW && db, "in_proj_bwd_wg: null pointer");
    KSR_CHECK_ARG(N > 0 && N < (1L << 31), "in_proj_bwd_wg: bad pixel count");
    KSR_UNSUPPORTED(ksr_prec() != KSR_BF16S, "in_proj_bwd_wg: bf16 storage mode only (the other modes use ksr_in_proj_bwd + ksr_wgrad)");
    InProjArgs a{reinterpret_cast<const __bf16*>(dqkv), reinterpret_cast<const __bf16*>(x_in), reinterpret_cast<const __bf16*>(resid),
                 in_w, reinterpret_cast<__bf16*>(ds), slab, (long)N};
    long g = (N + 4 * IP_T - 1) / (4 * IP_T);
    if (g > IP_GRID) g = IP_GRID;
    if (sigmoid_input) hipLaunchKernelGGL(inproj_bwd_fused_kernel<true>, dim3((unsigned)g), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(inproj_bwd_fused_kernel<false>, dim3((unsigned)g), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("in_proj_bwd_wg");
    return ksr_slab_reduce(slab, (int)g, IP_J, IP_K, dW, db, stream);
}
