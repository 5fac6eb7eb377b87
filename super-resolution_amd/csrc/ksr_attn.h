// Shared between the fp32 (ksr_attn.hip) and bf16 (ksr_attn_bf16.hip) batch-axis attention kernels.
#pragma once
#include "ksr_gemm.h"

#if defined(__HIPCC__)

struct AttnArgs {
    const float* qkv;   // [N,192]
    float* ctx;         // [N,64]
    int S, Sp, HW;
    float p_drop;
    unsigned long long seed;
    const unsigned long long* seed_dev;   // optional device word mixed into `seed` at kernel start (ksr_set_dropout_seed_source)
};

struct AttnBwdArgs {
    const float* qkv;    // [N,192]
    const float* ctx;    // [N,64]
    const float* dctx;   // [N,64]
    float* dqkv;         // [N,192]
    int S, Sp, HW;
    float p_drop;
    unsigned long long seed;
    const unsigned long long* seed_dev;   // optional device word mixed into `seed` at kernel start (ksr_set_dropout_seed_source)
};

// key (or query) index inside a 32-slot block held by accumulator register r of lane half h
__device__ __forceinline__ int key_of_reg(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Slot decoding without integer division.  Sp >= 32: a wave's blocks all belong to its own pixel, the image
// index of block kb / index i is 32*kb + i.  Sp < 32 (a power of two): slot = 32*wave + i, pixel = slot >> lg.
struct SlotMap {
    int Sp, lg, big;
    __device__ __forceinline__ void decode(int base0, int kb, int i, int my_pl, int& same_pixel, int& bb) const {
        if (big) { same_pixel = 1; bb = 32 * kb + i; }
        else { const int s = base0 + i; same_pixel = ((s >> lg) == my_pl); bb = s & (Sp - 1); }
    }
};

// bf16-operand variants (ksr_attn_bf16.hip); nw = key blocks per query (1..4)
int ksr_attn_bf16_fwd_launch(const AttnArgs& a, int nw, unsigned grid, hipStream_t stream);
int ksr_attn_bf16_bwd_launch(const AttnBwdArgs& a, int nw, unsigned grid, hipStream_t stream);
// bf16-storage variants that stage their operand rows with the LDS-DMA (ksr_attn_dma.hip)
bool ksr_attn_dma_supported(int Sp);
int ksr_attn_dma_fwd_launch(const AttnArgs& a, hipStream_t stream);
int ksr_attn_dma_bwd_launch(const AttnBwdArgs& a, hipStream_t stream);
int ksr_qkv_attn_dma_fwd_launch(const float* x, const float* w, const float* bias, float* qkv, float* ctx, int S, int Sp,
                                int HW, float p_drop, unsigned long long seed, const unsigned long long* seed_dev,
                                hipStream_t stream);

#endif
