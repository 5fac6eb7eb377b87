// UNet hyper-encoder, backward (conv_sep_kan/unet.py:30-42 differentiated; blocks conv_sep_kan/unet_blocks.py:9-178).
// Data gradients of the stride-1 convolutions, the 1x1 convolutions and the transposed convolution reuse the forward
// convolution kernel with re-laid-out weights; this file holds what has no forward twin: the weight-gradient
// reduction, BatchNorm + ReLU backward, the merge of the gradients that meet at an encoder level (2x2 max-pool
// routing included), the data gradient of the stride-2 reflect-padded projections and the per-image algebra of the
// anchor attention.
#include "ksr_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ weight gradients
#define WG_TILE_FLOATS 12288      // LDS tile: gradient row segment + activation rows (48 KB)
#define WG_ENTRIES4 1024          // groups of 4 output channels per workgroup (4 per lane)

struct WgArgs {
    const float* a0; const float* a1; int ca0, ca1;   // A side: activations (two concatenated sources), or none (mode 2)
    const float* a_scale; const float* a_shift;       // optional relu(x*scale+shift) on a0
    const float* g; int cg;                           // gradient side [B,Hg,Wg,cg]
    float* partials;                                  // [B][nwg][E]
    int ks, stride, reflect, mode;                    // mode 0: conv (rows of g), 1: transposed conv 2x2 (rows of A), 2: column sums of g
    int Ha, Wa, Hg, Wg;
    int cols, segs, nwg, E;                           // tile = one row segment of `cols` pixels; nwg workgroups per image
    int ones;                                         // 1: a virtual activation channel of ones is appended (bias gradient)
};

// grid (nwg, ceil(E/4 / WG_ENTRIES4), B).  A workgroup walks over row segments of one image; each is staged in LDS
// (activation rows with the padding materialised, BatchNorm + ReLU of the producer applied) and every lane
// accumulates groups of 4 neighbouring output channels of one (tap, ci) entry.  With few entries the lanes split the
// pixels of a segment between them and the slices are summed through LDS at the end.
__global__ void __launch_bounds__(256) unet_wgrad_kernel(WgArgs a) {
    // sized by the launch to the tile of this layer (at most WG_TILE_FLOATS): the shallow levels need a third of the
    // maximum, and the workgroups that then fit on a CU together are what hides the staging latency of each
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ float red[1024];
    const int b = blockIdx.z, t = threadIdx.x;
    const int ca = a.mode == 2 ? 1 : a.ca0 + a.ca1, cg4 = a.cg >> 2;
    const int E4 = a.E >> 2;
    const int Hi = a.mode == 1 ? a.Ha : a.Hg, Wi = a.mode == 1 ? a.Wa : a.Wg;
    const int pad = a.ks == 3 ? 1 : 0;
    // tile geometry
    const int a_rows = a.mode == 0 ? a.ks : 1, g_rows = a.mode == 1 ? 2 : 1;
    const int as = a.mode == 0 ? a.stride : 1, gs = a.mode == 1 ? 2 : 1;
    const int a_w = a.mode == 0 ? (a.cols - 1) * a.stride + a.ks : a.cols, g_w = a.cols * gs;
    const int g_floats = g_rows * g_w * a.cg;
    float* gl = tile;
    float* al = tile + g_floats;
    // lane -> entries
    const int nsl = E4 >= 256 ? 1 : 256 / E4;
    const int slice = nsl == 1 ? 0 : t / E4;
    const bool idle = nsl > 1 && slice >= nsl;
    const int cae = ca + a.ones;                      // entries per tap on the activation side
    int a_off[4], g_off[4];
    bool live[4], one[4];
    float acc[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e4 = nsl == 1 ? blockIdx.y * WG_ENTRIES4 + k * 256 + t : (k == 0 ? t % E4 : E4);
        live[k] = e4 < E4 && !idle;
        const int ec = live[k] ? e4 : 0;
        const int co4 = ec % cg4, ci = (ec / cg4) % cae, tap = ec / (cg4 * cae);
        const int ky = tap / a.ks, kx = tap % a.ks;
        one[k] = a.mode == 2 || ci == ca;
        a_off[k] = one[k] ? 0 : (a.mode == 0 ? (ky * a_w + kx) * ca + ci : ci);
        g_off[k] = (a.mode == 1 ? (ky * g_w + kx) * a.cg : 0) + co4 * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;
    }
    const int tiles = Hi * a.segs;
    for (int tl = blockIdx.x; tl < tiles; tl += a.nwg) {
        const int y = tl / a.segs, x0 = (tl - y * a.segs) * a.cols;
        const int cols = x0 + a.cols <= Wi ? a.cols : Wi - x0;
        __syncthreads();
        // ---- gradient rows: g[b][y*gs + r][x0*gs .. (x0+cols)*gs)[cg]
        for (int r = 0; r < g_rows; ++r) {
            const float4* src = reinterpret_cast<const float4*>(a.g + (((long)b * a.Hg + y * gs + r) * a.Wg + (long)x0 * gs) * a.cg);
            float4* dst = reinterpret_cast<float4*>(gl + r * g_w * a.cg);
            for (int i = t; i < cols * gs * cg4; i += 256) dst[i] = src[i];
        }
        // ---- activation rows, padding and the producer's BatchNorm + ReLU materialised
        if (a.mode != 2) {
            const int wcols = a.mode == 0 ? (cols - 1) * a.stride + a.ks : cols;
            if (((a.ca0 | a.ca1) & 3) == 0) {                 // 4 channels per load
                const int ca4 = ca >> 2;
                const bool pow2 = (ca4 & (ca4 - 1)) == 0;
                const int sh = 31 - __builtin_clz(ca4);
                const int per_row = wcols * ca4, total = a_rows * per_row;
                // all rows as one index space, four loads in flight per thread before the first is stored (a row of the
                // shallow levels is 258 pixels: row by row, 256 threads took two dependent round trips for each)
                for (int base = 0; base < total; base += 1024) {
                    float4 v[4];
                    int dsti[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = base + u * 256 + t;
                        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        dsti[u] = -1;
                        if (i < total) {
                            const int r = i / per_row, ir = i - r * per_row;
                            const int j = pow2 ? ir >> sh : ir / ca4, ch = (ir - j * ca4) << 2;
                            int iy = a.mode == 0 ? y * a.stride + r - pad : y;
                            int ix = a.mode == 0 ? x0 * a.stride + j - pad : x0 + j;
                            bool ok = true;
                            if (a.reflect) {
                                iy = iy < 0 ? -iy : (iy >= a.Ha ? 2 * a.Ha - 2 - iy : iy);
                                ix = ix < 0 ? -ix : (ix >= a.Wa ? 2 * a.Wa - 2 - ix : ix);
                            } else ok = iy >= 0 && iy < a.Ha && ix >= 0 && ix < a.Wa;
                            dsti[u] = (r * a_w + j) * ca + ch;
                            if (ok) {
                                const long pix = ((long)b * a.Ha + iy) * a.Wa + ix;
                                if (ch < a.ca0) {
                                    v[u] = *reinterpret_cast<const float4*>(a.a0 + pix * a.ca0 + ch);
                                    if (a.a_scale) {
                                        const float4 s4 = *reinterpret_cast<const float4*>(a.a_scale + ch), t4 = *reinterpret_cast<const float4*>(a.a_shift + ch);
                                        v[u] = make_float4(fmaxf(fmaf(v[u].x, s4.x, t4.x), 0.f), fmaxf(fmaf(v[u].y, s4.y, t4.y), 0.f),
                                                           fmaxf(fmaf(v[u].z, s4.z, t4.z), 0.f), fmaxf(fmaf(v[u].w, s4.w, t4.w), 0.f));
                                    }
                                } else v[u] = *reinterpret_cast<const float4*>(a.a1 + pix * a.ca1 + ch - a.ca0);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (dsti[u] >= 0) *reinterpret_cast<float4*>(al + dsti[u]) = v[u];
                }
            } else
            for (int i = t; i < a_rows * wcols * ca; i += 256) {
                const int ch = i % ca, j = (i / ca) % wcols, r = i / (ca * wcols);
                int iy = a.mode == 0 ? y * a.stride + r - pad : y;
                int ix = a.mode == 0 ? x0 * a.stride + j - pad : x0 + j;
                bool ok = true;
                if (a.reflect) {
                    iy = iy < 0 ? -iy : (iy >= a.Ha ? 2 * a.Ha - 2 - iy : iy);
                    ix = ix < 0 ? -ix : (ix >= a.Wa ? 2 * a.Wa - 2 - ix : ix);
                } else ok = iy >= 0 && iy < a.Ha && ix >= 0 && ix < a.Wa;
                float v = 0.f;
                if (ok) {
                    const long pix = ((long)b * a.Ha + iy) * a.Wa + ix;
                    if (ch < a.ca0) {
                        v = a.a0[pix * a.ca0 + ch];
                        if (a.a_scale) v = fmaxf(fmaf(v, a.a_scale[ch], a.a_shift[ch]), 0.f);
                    } else v = a.a1[pix * a.ca1 + ch - a.ca0];
                }
                al[(r * a_w + j) * ca + ch] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!live[k]) continue;
            // four pixels in flight (same order of the sums: the loop is a chain of LDS latencies otherwise)
            int x = slice;
            for (; x + 3 * nsl < cols; x += 4 * nsl) {
                float av[4];
                float4 gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    av[u] = one[k] ? 1.f : al[a_off[k] + (x + u * nsl) * as * ca];
                    gv[u] = *reinterpret_cast<const float4*>(gl + g_off[k] + (x + u * nsl) * gs * a.cg);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[k][0] = fmaf(av[u], gv[u].x, acc[k][0]); acc[k][1] = fmaf(av[u], gv[u].y, acc[k][1]);
                    acc[k][2] = fmaf(av[u], gv[u].z, acc[k][2]); acc[k][3] = fmaf(av[u], gv[u].w, acc[k][3]);
                }
            }
            for (; x < cols; x += nsl) {
                const float av = one[k] ? 1.f : al[a_off[k] + x * as * ca];
                const float4 gv = *reinterpret_cast<const float4*>(gl + g_off[k] + x * gs * a.cg);
                acc[k][0] = fmaf(av, gv.x, acc[k][0]); acc[k][1] = fmaf(av, gv.y, acc[k][1]);
                acc[k][2] = fmaf(av, gv.z, acc[k][2]); acc[k][3] = fmaf(av, gv.w, acc[k][3]);
            }
        }
    }
    float* out = a.partials + ((long)b * a.nwg + blockIdx.x) * a.E;
    if (nsl == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e4 = blockIdx.y * WG_ENTRIES4 + k * 256 + t;
            if (e4 < E4) *reinterpret_cast<float4*>(out + 4 * e4) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
        }
    } else {
        __syncthreads();
        if (!idle) {
#pragma unroll
            for (int i = 0; i < 4; ++i) red[slice * a.E + 4 * (t % E4) + i] = acc[0][i];
        }
        __syncthreads();
        for (int e = t; e < a.E; e += 256) {
            float s = 0.f;
            for (int k = 0; k < nsl; ++k) s += red[k * a.E + e];
            out[e] = s;
        }
    }
}

struct WgRedArgs { const float* partials; float* out; int B, chunks, E, per_image; };

// 16 entries x 16 slices of the partial rows per workgroup (64 contiguous bytes per row), eight rows in flight per thread;
// fixed summation order
constexpr int WGR_OUT = 16, WGR_SL = 16;
__device__ __forceinline__ void unet_wgrad_reduce_body(const WgRedArgs& a, int block, long K, long base, float* dst) {
    const int le = threadIdx.x & (WGR_OUT - 1), sl = threadIdx.x / WGR_OUT;
    const int e = block * WGR_OUT + le;
    __shared__ double red[WGR_SL][WGR_OUT];
    double s = 0.0;
    if (e < a.E) {
        const float* p = a.partials + base * a.E + e;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        long k = sl;
        for (; k + 7 * WGR_SL < K; k += 8 * WGR_SL) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = p[(k + i * WGR_SL) * a.E];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
        for (; k < K; k += WGR_SL) acc[0] += p[k * a.E];
        s = (((double)acc[0] + (double)acc[1]) + ((double)acc[2] + (double)acc[3])) +
            (((double)acc[4] + (double)acc[5]) + ((double)acc[6] + (double)acc[7]));
    }
    red[sl][le] = s;
    __syncthreads();
    if (sl == 0 && e < a.E) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < WGR_SL; ++k) tot += red[k][le];
        dst[e] = (float)tot;
    }
}

__global__ void __launch_bounds__(256) unet_wgrad_reduce_kernel(WgRedArgs a) {
    const long K = a.per_image ? a.chunks : (long)a.B * a.chunks;
    const long base = a.per_image ? (long)blockIdx.y * a.chunks : 0;
    unet_wgrad_reduce_body(a, (int)blockIdx.x, K, base, a.out + (a.per_image ? (long)blockIdx.y * a.E : 0));
}

// Several of those reductions in one launch (the deferred form: ksr_reduce_defer / ksr_reduce_flush): workgroup -> (entry,
// block of 16 outputs) through the prefix table; per output the same sums in the same order as the kernel above.
constexpr int UNET_RED_BATCH_MAX = 40;
struct WgRedBatch {
    WgRedArgs e[UNET_RED_BATCH_MAX];
    int first[UNET_RED_BATCH_MAX + 1];
    int n;
};
__global__ void __launch_bounds__(256) unet_wgrad_reduce_batch_kernel(WgRedBatch b) {
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.first[i + 1]) ++i;
    const WgRedArgs a = b.e[i];
    unet_wgrad_reduce_body(a, (int)blockIdx.x - b.first[i], (long)a.B * a.chunks, 0, a.out);
}
thread_local WgRedBatch g_unet_red;

// ------------------------------------------------------------------------------------------------ BatchNorm + ReLU
struct BnRedArgs { const float* dout; const float* raw; const float* scale; const float* shift; float* partials; long npix; int C; };

#define BN_PIX_PER_WG 512

// partial sums over pixels of dy and dy*raw with dy = dout * (raw*scale+shift > 0): [wg][2][C]
__global__ void __launch_bounds__(256) unet_bn_bwd_reduce_kernel(BnRedArgs a) {
    const int c4n = a.C >> 2;                              // 1, 2, 4, 8 or 16 channel quads
    const int q = threadIdx.x % c4n, slot = threadIdx.x / c4n, nslot = 256 / c4n;
    const long p0 = (long)blockIdx.x * BN_PIX_PER_WG;
    const long p1 = p0 + BN_PIX_PER_WG < a.npix ? p0 + BN_PIX_PER_WG : a.npix;
    const float4 s = *reinterpret_cast<const float4*>(a.scale + 4 * q), t = *reinterpret_cast<const float4*>(a.shift + 4 * q);
    float sd[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f};
    for (long p = p0 + slot; p < p1; p += nslot) {
        const float4 d4 = *reinterpret_cast<const float4*>(a.dout + p * a.C + 4 * q);
        const float4 r4 = *reinterpret_cast<const float4*>(a.raw + p * a.C + 4 * q);
        const float d[4] = {d4.x, d4.y, d4.z, d4.w}, r[4] = {r4.x, r4.y, r4.z, r4.w};
        const float ss[4] = {s.x, s.y, s.z, s.w}, tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dy = fmaf(r[i], ss[i], tt[i]) > 0.f ? d[i] : 0.f;
            sd[i] += dy;
            sx[i] = fmaf(dy, r[i], sx[i]);
        }
    }
    __shared__ float red[4][2][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        for (int m = 32; m >= c4n; m >>= 1) { sd[i] += __shfl_xor(sd[i], m, 64); sx[i] += __shfl_xor(sx[i], m, 64); }
        if (lane < c4n) { red[wave][0][4 * lane + i] = sd[i]; red[wave][1][4 * lane + i] = sx[i]; }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * a.C) {
        const int k = threadIdx.x / a.C, c = threadIdx.x % a.C;
        a.partials[((long)blockIdx.x * 2 + k) * a.C + c] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}

struct BnBwdFinArgs {
    const float* partials; int nparts; long count; const float* gamma; const float* saved;   // saved: mean | invstd
    int train; float* coef; float* dgamma; float* dbeta; int C;
};

// per channel: dgamma, dbeta and the coefficients of  dx = A * dy + Bc * raw + Cc
__global__ void __launch_bounds__(256) unet_bn_bwd_finalize_kernel(BnBwdFinArgs a) {
    const int c = blockIdx.x;
    __shared__ double red[2][256];
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < a.nparts; i += 256) {
        s += (double)a.partials[((long)i * 2) * a.C + c];
        q += (double)a.partials[((long)i * 2 + 1) * a.C + c];
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) { red[0][threadIdx.x] += red[0][threadIdx.x + m]; red[1][threadIdx.x] += red[1][threadIdx.x + m]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = a.saved[c], invstd = a.saved[a.C + c], n = (double)a.count;
        const double dbeta = red[0][0], dgamma = invstd * (red[1][0] - mean * red[0][0]);
        const double A = (double)a.gamma[c] * invstd;
        a.dgamma[c] = (float)dgamma;
        a.dbeta[c] = (float)dbeta;
        a.coef[c] = (float)A;
        a.coef[a.C + c] = a.train ? (float)(-A * invstd * dgamma / n) : 0.f;
        a.coef[2 * a.C + c] = a.train ? (float)(-A * dbeta / n + A * mean * invstd * dgamma / n) : 0.f;
    }
}

struct BnApplyArgs { const float* dout; const float* raw; const float* scale; const float* shift; const float* coef; float* dx; long n4; int C; };

__global__ void __launch_bounds__(256) unet_bn_relu_bwd_kernel(BnApplyArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.n4) return;
    const int c = (int)(idx % (a.C >> 2)) * 4;
    const float4 d4 = reinterpret_cast<const float4*>(a.dout)[idx], r4 = reinterpret_cast<const float4*>(a.raw)[idx];
    const float d[4] = {d4.x, d4.y, d4.z, d4.w}, r[4] = {r4.x, r4.y, r4.z, r4.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float dy = fmaf(r[i], a.scale[c + i], a.shift[c + i]) > 0.f ? d[i] : 0.f;
        o[i] = fmaf(a.coef[c + i], dy, fmaf(a.coef[a.C + c + i], r[i], a.coef[2 * a.C + c + i]));
    }
    reinterpret_cast<float4*>(a.dx)[idx] = make_float4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------------------------------------ gradient merge
struct MergeArgs { const float* g0; const float* g1; const float* gp; const float* act; float* out; int B, H, W, C; };

// out = g0 + g1 + maxpool2x2 backward of gp (to the first maximum of each window, as torch routes it); one 2x2 block
// of pixels and 4 channels per lane.  g1, gp optional.
__global__ void __launch_bounds__(256) unet_grad_merge_kernel(MergeArgs a) {
    const int c4n = a.C >> 2, Hb = (a.H + 1) >> 1, Wb = (a.W + 1) >> 1, Hp = a.H >> 1, Wp = a.W >> 1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)a.B * Hb * Wb * c4n) return;
    const int c = (int)(idx % c4n) * 4;
    long r = idx / c4n;
    const int px = (int)(r % Wb); r /= Wb;
    const int py = (int)(r % Hb);
    const int b = (int)(r / Hb);
    const bool window = py < Hp && px < Wp;          // odd sizes: the last row / column is in no pooling window
    long off[4];
    bool in[4];
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        in[k] = 2 * py + (k >> 1) < a.H && 2 * px + (k & 1) < a.W;
        off[k] = (((long)b * a.H + 2 * py + (k >> 1)) * a.W + 2 * px + (k & 1)) * a.C + c;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in[k]) t = *reinterpret_cast<const float4*>(a.act + off[k]);
        v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
    }
    float gpv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.gp && window) {
        const float4 t = *reinterpret_cast<const float4*>(a.gp + (((long)b * Hp + py) * Wp + px) * a.C + c);
        gpv[0] = t.x; gpv[1] = t.y; gpv[2] = t.z; gpv[3] = t.w;
    }
    int arg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int best = 0;
        float m = v[0][i];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (v[k][i] > m) { m = v[k][i]; best = k; }
        arg[i] = best;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!in[k]) continue;
        const float4 t0 = *reinterpret_cast<const float4*>(a.g0 + off[k]);
        float o[4] = {t0.x, t0.y, t0.z, t0.w};
        if (a.g1) {
            const float4 t1 = *reinterpret_cast<const float4*>(a.g1 + off[k]);
            o[0] += t1.x; o[1] += t1.y; o[2] += t1.z; o[3] += t1.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += arg[i] == k ? gpv[i] : 0.f;
        *reinterpret_cast<float4*>(a.out + off[k]) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------------------------------------ stride-2 reflect dgrad
struct S2Args { const float* dz; const float* weight; float* dx; int cz, cin, H, W, Ho, Wo; };

// rows (before reflection) of the padded input that land on input row i: i itself, -1 when i == 1, H when i == H-2
__device__ __forceinline__ int s2_candidate(int k, int i, int n) {
    return k == 0 ? i : (k == 1 ? (i == 1 ? -1 : -99) : (i == n - 2 ? n : -99));
}

// data gradient of conv3x3(stride 2, reflect pad 1): one input pixel, CT input channels per lane.  weight [9][cz][cin].
// LPP > 1 (small images: a few thousand pixels would leave most of the chip idle behind serial chains of cz x taps
// loads): LPP neighbouring lanes share a pixel and split the cz gradient channels, a butterfly adds their sums.
template <int CT, int LPP>
__global__ void __launch_bounds__(256) unet_s2_dgrad_kernel(S2Args a) {
    const int b = blockIdx.y, c0 = blockIdx.z * CT;
    const int sub = threadIdx.x % LPP;
    const int pr = blockIdx.x * (256 / LPP) + threadIdx.x / LPP;
    const bool live = pr < a.H * a.W;
    const int p = live ? pr : 0;                             // idle lanes redo pixel 0 (they take part in the butterfly)
    const int iy = p / a.W, ix = p - iy * a.W;
    // this workgroup's slice of the weights, [9][cz][CT], in LDS: the innermost loop reads CT of them per gradient channel,
    // which from global memory was a chain of cache latencies (49 us for 8 x 64 x 64 pixels, 80 -> 32 channels)
    extern __shared__ __attribute__((aligned(16))) float s2w[];
    for (int i = threadIdx.x; i < 9 * a.cz * (CT / 4); i += 256) {
        const int row = i / (CT / 4), q = i - row * (CT / 4);
        *reinterpret_cast<float4*>(s2w + row * CT + 4 * q) = *reinterpret_cast<const float4*>(a.weight + (long)row * a.cin + c0 + 4 * q);
    }
    __syncthreads();
    float acc[CT];
#pragma unroll
    for (int i = 0; i < CT; ++i) acc[i] = 0.f;
    for (int cy = 0; cy < 3; ++cy) {
        const int ry = s2_candidate(cy, iy, a.H);
        if (ry == -99) continue;
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = ry + 1 - ky;
            if (ty < 0 || (ty & 1) || (ty >> 1) >= a.Ho) continue;
            for (int cx = 0; cx < 3; ++cx) {
                const int rx = s2_candidate(cx, ix, a.W);
                if (rx == -99) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int tx = rx + 1 - kx;
                    if (tx < 0 || (tx & 1) || (tx >> 1) >= a.Wo) continue;
                    const float* g = a.dz + (((long)b * a.Ho + (ty >> 1)) * a.Wo + (tx >> 1)) * a.cz;
                    const float* w = s2w + (ky * 3 + kx) * a.cz * CT;
                    for (int co = 4 * sub; co < a.cz; co += 4 * LPP) {
                        const float4 g4 = *reinterpret_cast<const float4*>(g + co);
                        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float wr[CT];
#pragma unroll
                            for (int i = 0; i < CT; i += 4) {
                                const float4 w4 = *reinterpret_cast<const float4*>(w + (co + e) * CT + i);
                                wr[i] = w4.x; wr[i + 1] = w4.y; wr[i + 2] = w4.z; wr[i + 3] = w4.w;
                            }
#pragma unroll
                            for (int i = 0; i < CT; ++i) acc[i] = fmaf(gv[e], wr[i], acc[i]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int m = LPP >> 1; m >= 1; m >>= 1) acc[i] += __shfl_xor(acc[i], m, 64);
    if (!live || sub != 0) return;
    float* o = a.dx + ((long)b * a.H * a.W + p) * a.cin + c0;
#pragma unroll
    for (int i = 0; i < CT; i += 4) *reinterpret_cast<float4*>(o + i) = make_float4(acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
}

// ------------------------------------------------------------------------------------------------ anchor attention algebra
// entry e of the packed products -> the pair of channels of z = (q | k | a): q_i a_j, then a_j k_i, then the squared norms
__device__ __forceinline__ void gram_entry(int e, int c, int& A, int& Bc) {
    const int ca = c >> 1, nqa = c * ca;
    if (e < nqa) { A = e / ca; Bc = 2 * c + e % ca; }
    else if (e < 2 * nqa) { const int f = e - nqa; A = 2 * c + f / c; Bc = c + f % c; }
    else { A = Bc = e - 2 * nqa; }
}

struct MixBwdArgs {
    const float* g; const float* dT; int c, heads;      // g: Gram matrix z^T z [B][zc][zc]
    const float* temp_a; const float* temp_v; const float* wv; const float* wp; float vscale;
    float* D;        // [B][zc][zc]: dz[n] = z[n] D (symmetric), zero outside the 2.5c live channels
    int zc;
    float* dparams;  // [B][2c^2 + 2 heads]: d v_proj | d project_out | d temperature_a | d temperature_v, per image
};

__global__ void __launch_bounds__(256) unet_att_mix_bwd_kernel(MixBwdArgs a) {
    const int b = blockIdx.x, c = a.c, ca = c >> 1, nqa = c * ca, cc = c * c;
    const int E = cc + (5 * c) / 2;
    const int ch = c / a.heads, cah = ca / a.heads;
    const int t = threadIdx.x;
    __shared__ float g[32 * 32 + 80], at_a[32 * 16], at_k[16 * 32], M[32 * 32], T1[32 * 32], dTs[32 * 32], dT1[32 * 32], dM[32 * 32];
    __shared__ float s_a[32 * 16], s_k[16 * 32], da[32 * 16], dk[16 * 32], dN[80], dtemp[2][32 + 16];
    // the two c x c weights are read inside the products' inner loops: from global memory each of those loops was a chain
    // of memory latencies (67 us for c = 32)
    __shared__ float wvs[32 * 32], wps[32 * 32];
    for (int i = t; i < cc; i += 256) { wvs[i] = a.wv[i]; wps[i] = a.wp[i]; }
    for (int e = t; e < E; e += 256) {
        int A, Bc;
        gram_entry(e, c, A, Bc);
        g[e] = a.g[((long)b * a.zc + A) * a.zc + Bc];
    }
    for (int i = t; i < cc; i += 256) dTs[i] = a.dT[(long)b * cc + i];       // [ci][co]
    for (int i = t; i < 80; i += 256) dN[i] = 0.f;
    __syncthreads();
    const float* nrm2 = g + 2 * nqa;
    // ---- forward recomputation: normalised products, softmaxes
    if (t < c) {
        const int i = t, h = i / ch;
        const float nq = fmaxf(sqrtf(nrm2[i]), 1e-12f);
        float mx = -INFINITY;
        for (int jj = 0; jj < cah; ++jj) {
            const int j = h * cah + jj;
            const float sv = g[i * ca + j] / (nq * fmaxf(sqrtf(nrm2[2 * c + j]), 1e-12f));
            s_a[i * ca + j] = sv;
            mx = fmaxf(mx, sv * a.temp_a[h]);
        }
        float sum = 0.f;
        for (int jj = 0; jj < cah; ++jj) { const int j = h * cah + jj; const float ex = expf(s_a[i * ca + j] * a.temp_a[h] - mx); at_a[i * ca + j] = ex; sum += ex; }
        for (int jj = 0; jj < cah; ++jj) at_a[i * ca + h * cah + jj] /= sum;
    } else if (t >= 64 && t < 64 + ca) {
        const int j = t - 64, h = j / cah;
        const float na = fmaxf(sqrtf(nrm2[2 * c + j]), 1e-12f);
        float mx = -INFINITY;
        for (int ii = 0; ii < ch; ++ii) {
            const int i = h * ch + ii;
            const float sv = g[nqa + j * c + i] / (na * fmaxf(sqrtf(nrm2[c + i]), 1e-12f));
            s_k[j * c + i] = sv;
            mx = fmaxf(mx, sv * a.temp_v[h]);
        }
        float sum = 0.f;
        for (int ii = 0; ii < ch; ++ii) { const int i = h * ch + ii; const float ex = expf(s_k[j * c + i] * a.temp_v[h] - mx); at_k[j * c + i] = ex; sum += ex; }
        for (int ii = 0; ii < ch; ++ii) at_k[j * c + h * ch + ii] /= sum;
    }
    __syncthreads();
    for (int idx = t; idx < cc; idx += 256) {
        const int i = idx / c, i2 = idx % c, h = i / ch;
        float s = 0.f;
        if (i2 / ch == h)
            for (int jj = 0; jj < cah; ++jj) s = fmaf(at_a[i * ca + h * cah + jj], at_k[(h * cah + jj) * c + i2], s);
        M[idx] = s;
    }
    __syncthreads();
    for (int idx = t; idx < cc; idx += 256) {      // T1[i][ci] = vscale * sum_i' M[i][i'] Wv[i'][ci]
        const int i = idx / c, ci = idx % c, h = i / ch;
        float s = 0.f;
        for (int ii = 0; ii < ch; ++ii) s = fmaf(M[i * c + h * ch + ii], wvs[(h * ch + ii) * c + ci], s);
        T1[idx] = s * a.vscale;
    }
    __syncthreads();
    // ---- T[co][ci] = sum_i Wp[co][i] T1[i][ci]
    float* dpar = a.dparams + (long)b * (2 * cc + 2 * a.heads);
    for (int idx = t; idx < cc; idx += 256) {
        const int r = idx / c, q = idx % c;
        float s1 = 0.f, s2 = 0.f;
        for (int k = 0; k < c; ++k) {
            s1 = fmaf(dTs[k * c + r], T1[q * c + k], s1);          // dWp[co=r][i=q] = sum_ci dT[ci][co] T1[i][ci]
            s2 = fmaf(wps[k * c + r], dTs[q * c + k], s2);         // dT1[i=r][ci=q] = sum_co Wp[co][i] dT[ci][co]
        }
        dpar[cc + idx] = s1;
        dT1[idx] = s2;
    }
    __syncthreads();
    for (int idx = t; idx < cc; idx += 256) {
        const int r = idx / c, q = idx % c;
        float s1 = 0.f, s2 = 0.f;
        const bool same = r / ch == q / ch;
        for (int k = 0; k < c; ++k) {
            if (same) s1 = fmaf(dT1[r * c + k], wvs[q * c + k], s1);   // dM[i=r][i'=q] = vscale sum_ci dT1[i][ci] Wv[i'][ci]
            s2 = fmaf(M[k * c + r], dT1[k * c + q], s2);                 // dWv[i'=r][ci=q] = vscale sum_i M[i][i'] dT1[i][ci]
        }
        dM[idx] = same ? s1 * a.vscale : 0.f;
        dpar[idx] = s2 * a.vscale;
    }
    __syncthreads();
    // ---- softmax backward, row-wise; logits = s * temperature
    if (t < c) {
        const int i = t, h = i / ch;
        float dot = 0.f;
        for (int jj = 0; jj < cah; ++jj) {
            const int j = h * cah + jj;
            float d = 0.f;
            for (int ii = 0; ii < ch; ++ii) d = fmaf(dM[i * c + h * ch + ii], at_k[j * c + h * ch + ii], d);   // d attn_a[i][j]
            da[i * ca + j] = d;
            dot = fmaf(d, at_a[i * ca + j], dot);
        }
        float dtemp_i = 0.f;
        for (int jj = 0; jj < cah; ++jj) {
            const int j = h * cah + jj;
            const float dl = at_a[i * ca + j] * (da[i * ca + j] - dot);
            dtemp_i = fmaf(dl, s_a[i * ca + j], dtemp_i);
            da[i * ca + j] = dl * a.temp_a[h];                       // d s_a
        }
        dtemp[0][i] = dtemp_i;
    } else if (t >= 64 && t < 64 + ca) {
        const int j = t - 64, h = j / cah;
        float dot = 0.f;
        for (int ii = 0; ii < ch; ++ii) {
            const int i2 = h * ch + ii;
            float d = 0.f;
            for (int i = h * ch; i < (h + 1) * ch; ++i) d = fmaf(at_a[i * ca + j], dM[i * c + i2], d);         // d attn_k[j][i2]
            dk[j * c + i2] = d;
            dot = fmaf(d, at_k[j * c + i2], dot);
        }
        float dtemp_j = 0.f;
        for (int ii = 0; ii < ch; ++ii) {
            const int i2 = h * ch + ii;
            const float dl = at_k[j * c + i2] * (dk[j * c + i2] - dot);
            dtemp_j = fmaf(dl, s_k[j * c + i2], dtemp_j);
            dk[j * c + i2] = dl * a.temp_v[h];
        }
        dtemp[1][j] = dtemp_j;
    }
    __syncthreads();
    if (t < a.heads) {
        float s = 0.f;
        for (int i = t * ch; i < (t + 1) * ch; ++i) s += dtemp[0][i];
        dpar[2 * cc + t] = s;
    } else if (t >= 64 && t < 64 + a.heads) {
        const int h = t - 64;
        float s = 0.f;
        for (int j = h * cah; j < (h + 1) * cah; ++j) s += dtemp[1][j];
        dpar[2 * cc + a.heads + h] = s;
    }
    // ---- s = G / (n1 n2): gradients of the squared norms (one lane per channel, no races)
    if (t >= 128 && t < 128 + 2 * c + ca) {
        const int chn = t - 128;
        float s = 0.f;
        if (chn < c) {                                  // q_i
            const int i = chn, h = i / ch;
            for (int jj = 0; jj < cah; ++jj) { const int j = h * cah + jj; s = fmaf(da[i * ca + j], s_a[i * ca + j], s); }
        } else if (chn < 2 * c) {                       // k_i
            const int i = chn - c, h = i / ch;
            for (int jj = 0; jj < cah; ++jj) { const int j = h * cah + jj; s = fmaf(dk[j * c + i], s_k[j * c + i], s); }
        } else {                                        // a_j: both products
            const int j = chn - 2 * c, h = j / cah;
            for (int ii = 0; ii < ch; ++ii) {
                const int i = h * ch + ii;
                s = fmaf(da[i * ca + j], s_a[i * ca + j], s);
                s = fmaf(dk[j * c + i], s_k[j * c + i], s);
            }
        }
        dN[chn] = -s / (2.f * fmaxf(nrm2[chn], 1e-24f));
    }
    __syncthreads();
    // ---- D: z -> dz, symmetric [zc][zc]
    float* D = a.D + (long)b * a.zc * a.zc;
    for (int idx = t; idx < a.zc * a.zc; idx += 256) {
        const int r = idx / a.zc, q = idx % a.zc;
        const int lo = r < q ? r : q, hi = r < q ? q : r;
        float v = 0.f;
        if (hi < 2 * c + ca) {
            if (lo == hi) v = 2.f * dN[lo];
            else if (hi >= 2 * c) {
                const int j = hi - 2 * c;
                if (lo < c) {                           // q_i with a_j
                    const int i = lo;
                    if (i / ch == j / cah) v = da[i * ca + j] / (fmaxf(sqrtf(nrm2[i]), 1e-12f) * fmaxf(sqrtf(nrm2[2 * c + j]), 1e-12f));
                } else if (lo < 2 * c) {                // k_i with a_j
                    const int i = lo - c;
                    if (i / ch == j / cah) v = dk[j * c + i] / (fmaxf(sqrtf(nrm2[c + i]), 1e-12f) * fmaxf(sqrtf(nrm2[2 * c + j]), 1e-12f));
                }
            }
        }
        D[idx] = v;
    }
}

}  // namespace

int ksr_unet_reduce_flush(hipStream_t stream) {
    WgRedBatch& b = g_unet_red;
    if (b.n == 0) return KSR_OK;
    const int blocks = b.first[b.n];
    hipLaunchKernelGGL(unet_wgrad_reduce_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, b);
    b.n = 0;
    KSR_LAUNCH_CHECK("unet_wgrad_reduce(batch)");
    return KSR_OK;
}
int ksr_unet_reduce_pending() { return g_unet_red.n; }
void ksr_unet_reduce_clear() { g_unet_red.n = 0; }


struct WgPlan { int cols, segs, nwg, E, tile_floats; };

static bool wgrad_plan(int mode, int ks, int stride, int ca, int cg, int B, int Ha, int Wa, int Hg, int Wg, WgPlan& p) {
    const int Hi = mode == 1 ? Ha : Hg, Wi = mode == 1 ? Wa : Wg;
    const int a_rows = mode == 0 ? ks : 1, g_rows = mode == 1 ? 2 : 1, gs = mode == 1 ? 2 : 1;
    const int a_ch = mode == 2 ? 0 : ca;
    const int fixed = mode == 0 ? a_rows * (ks - stride) * a_ch : 0;
    const int per_col = a_rows * (mode == 0 ? stride : 1) * a_ch + g_rows * gs * cg;
    int cols_max = (WG_TILE_FLOATS - 4 - fixed) / per_col;
    if (cols_max < 1) return false;
    if (cols_max > Wi) cols_max = Wi;
    p.segs = (Wi + cols_max - 1) / cols_max;
    p.cols = (Wi + p.segs - 1) / p.segs;
    const int tiles = Hi * p.segs;
    // as many workgroups as the GPU can hold at once (8 per CU when the tile is small), each with the same number of tiles
    int per_image = 2048 / B;
    if (per_image < 1) per_image = 1;
    const int rounds = (tiles + per_image - 1) / per_image;
    p.nwg = (tiles + rounds - 1) / rounds;
    p.E = (mode == 2 ? 1 : ks * ks * ca) * cg;
    const int a_w = mode == 0 ? (p.cols - 1) * stride + ks : p.cols;
    p.tile_floats = g_rows * p.cols * gs * cg + (mode == 2 ? 0 : a_rows * a_w * ca) + 4;
    return true;
}

extern "C" int64_t ksr_unet_wgrad_slab_floats(int mode, int ks, int ca, int cg, int B, int Ha, int Wa, int Hg, int Wg) {
    const int taps = mode == 0 ? ks * ks : (mode == 1 ? 4 : 1);
    const int64_t E = (int64_t)taps * (mode == 2 ? 1 : ca) * cg;
    const int per_image = 2048 / B > 0 ? 2048 / B : 1;      // upper bound of the workgroups per image
    return (int64_t)B * per_image * E;
}

extern "C" int ksr_unet_wgrad(const float* a0, int ca0, const float* a1, int ca1, const float* a_scale, const float* a_shift,
                              const float* g, int cg, float* slab, float* out, int per_image, int add_ones, int mode, int ks, int stride,
                              int reflect, int B, int Ha, int Wa, int Hg, int Wg, hipStream_t stream) {
    KSR_CHECK_ARG(g && slab && out && B > 0 && cg > 0 && Hg > 0 && Wg > 0, "unet_wgrad: bad argument");
    KSR_CHECK_ARG(mode == 2 || (a0 && ca0 > 0 && Ha > 0 && Wa > 0), "unet_wgrad: activations missing");
    KSR_CHECK_ARG((a1 != nullptr) == (ca1 > 0), "unet_wgrad: a1 and ca1 disagree");
    KSR_UNSUPPORTED(mode < 0 || mode > 2, "unet_wgrad: mode %d", mode);
    KSR_UNSUPPORTED(mode == 0 && (ks < 1 || ks > 3 || (stride != 1 && stride != 2)), "unet_wgrad: kernel %d stride %d", ks, stride);
    KSR_UNSUPPORTED(cg % 4 != 0, "unet_wgrad: cg = %d must be a multiple of 4", cg);
    KSR_UNSUPPORTED(B > 65535, "unet_wgrad: batch %d too large for one launch", B);
    if (mode == 1) { ks = 2; stride = 2; reflect = 0; KSR_CHECK_ARG(Hg == 2 * Ha && Wg == 2 * Wa, "unet_wgrad: transposed conv sizes"); }
    if (mode == 2) { ks = 1; stride = 1; reflect = 0; Ha = Hg; Wa = Wg; }
    if (mode == 0) {
        const int pad = ks == 3 ? 1 : 0;
        KSR_CHECK_ARG(Hg == (Ha + 2 * pad - ks) / stride + 1 && Wg == (Wa + 2 * pad - ks) / stride + 1, "unet_wgrad: conv sizes disagree");
        KSR_UNSUPPORTED(reflect && (Ha < 2 || Wa < 2), "unet_wgrad: reflect padding needs at least 2 pixels per side");
    }
    const int ca = mode == 2 ? 1 : ca0 + ca1;
    add_ones = (add_ones && mode != 2) ? 1 : 0;
    WgPlan p;
    KSR_UNSUPPORTED(!wgrad_plan(mode, ks, stride, ca, cg, B, Ha, Wa, Hg, Wg, p), "unet_wgrad: %d + %d channels do not fit the LDS tile", ca, cg);
    p.E = (mode == 2 ? 1 : ks * ks * (ca + add_ones)) * cg;
    WgArgs a{a0, a1, ca0, ca1, a_scale, a_shift, g, cg, slab, ks, stride, reflect, mode, Ha, Wa, Hg, Wg, p.cols, p.segs, p.nwg, p.E, add_ones};
    const int E4 = p.E / 4;
    hipLaunchKernelGGL(unet_wgrad_kernel, dim3((unsigned)p.nwg, (unsigned)((E4 + WG_ENTRIES4 - 1) / WG_ENTRIES4), (unsigned)B), dim3(256),
                       (size_t)p.tile_floats * sizeof(float), stream, a);
    KSR_LAUNCH_CHECK("unet_wgrad");
    WgRedArgs r{slab, out, B, p.nwg, p.E, per_image};
    if (!per_image && ksr_reduce_deferring()) {
        // recorded, not run: the caller keeps slab and out until ksr_reduce_flush (a full list is flushed here)
        WgRedBatch& b = g_unet_red;
        if (b.n == UNET_RED_BATCH_MAX) {
            if (int rc = ksr_unet_reduce_flush(stream)) return rc;
        }
        if (b.n == 0) b.first[0] = 0;
        b.e[b.n] = r;
        b.first[b.n + 1] = b.first[b.n] + (p.E + WGR_OUT - 1) / WGR_OUT;
        ++b.n;
        return KSR_OK;
    }
    hipLaunchKernelGGL(unet_wgrad_reduce_kernel, dim3((unsigned)((p.E + WGR_OUT - 1) / WGR_OUT), (unsigned)(per_image ? B : 1)), dim3(256), 0, stream, r);
    KSR_LAUNCH_CHECK("unet_wgrad_reduce");
    return KSR_OK;
}

extern "C" int64_t ksr_unet_bn_bwd_partials(int64_t npix) { return (npix + BN_PIX_PER_WG - 1) / BN_PIX_PER_WG; }

extern "C" int ksr_unet_bn_relu_bwd(const float* dout, const float* raw, const float* scale, const float* shift, const float* gamma,
                                    const float* saved_mean_invstd, int train, float* partials, float* coef, float* dgamma,
                                    float* dbeta, float* dx, int64_t npix, int C, hipStream_t stream) {
    KSR_CHECK_ARG(dout && raw && scale && shift && gamma && saved_mean_invstd && partials && coef && dgamma && dbeta && dx && npix > 0,
                  "unet_bn_relu_bwd: bad argument");
    KSR_UNSUPPORTED(C != 4 && C != 8 && C != 16 && C != 32 && C != 64, "unet_bn_relu_bwd: C = %d (4, 8, 16, 32 or 64)", C);
    const int nparts = (int)((npix + BN_PIX_PER_WG - 1) / BN_PIX_PER_WG);
    BnRedArgs r{dout, raw, scale, shift, partials, (long)npix, C};
    hipLaunchKernelGGL(unet_bn_bwd_reduce_kernel, dim3((unsigned)nparts), dim3(256), 0, stream, r);
    KSR_LAUNCH_CHECK("unet_bn_bwd_reduce");
    BnBwdFinArgs f{partials, nparts, (long)npix, gamma, saved_mean_invstd, train, coef, dgamma, dbeta, C};
    hipLaunchKernelGGL(unet_bn_bwd_finalize_kernel, dim3((unsigned)C), dim3(256), 0, stream, f);
    KSR_LAUNCH_CHECK("unet_bn_bwd_finalize");
    const long n4 = (long)npix * (C / 4);
    BnApplyArgs ap{dout, raw, scale, shift, coef, dx, n4, C};
    hipLaunchKernelGGL(unet_bn_relu_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, ap);
    KSR_LAUNCH_CHECK("unet_bn_relu_bwd");
    return KSR_OK;
}

extern "C" int ksr_unet_grad_merge(const float* g0, const float* g1, const float* g_pooled, const float* act, float* out, int B, int H,
                                   int W, int C, hipStream_t stream) {
    KSR_CHECK_ARG(g0 && act && out && B > 0 && H > 0 && W > 0, "unet_grad_merge: bad argument");
    KSR_UNSUPPORTED(C % 4 != 0, "unet_grad_merge: C = %d must be a multiple of 4", C);
    MergeArgs a{g0, g1, g_pooled, act, out, B, H, W, C};
    const long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(unet_grad_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("unet_grad_merge");
    return KSR_OK;
}

extern "C" int ksr_unet_s2_dgrad(const float* dz, int cz, const float* weight, float* dx, int cin, int B, int H, int W,
                                 hipStream_t stream) {
    KSR_CHECK_ARG(dz && weight && dx && B > 0 && H >= 2 && W >= 2, "unet_s2_dgrad: bad argument");
    KSR_UNSUPPORTED(cz % 4 != 0 || cin % 4 != 0, "unet_s2_dgrad: channel counts (%d, %d) must be multiples of 4", cz, cin);
    KSR_UNSUPPORTED(B > 65535, "unet_s2_dgrad: batch %d too large for one launch", B);
    KSR_UNSUPPORTED((size_t)9 * cz * 8 * sizeof(float) > 64 * 1024, "unet_s2_dgrad: %d gradient channels do not fit the weight tile", cz);
    S2Args a{dz, weight, dx, cz, cin, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
    const bool split = (long)B * H * W <= 32768 && cz >= 16;             // 4 lanes per pixel
    const unsigned gx = (unsigned)((H * W + (split ? 63 : 255)) / (split ? 64 : 256));
    if (cin % 8 == 0) {
        if (split) hipLaunchKernelGGL((unet_s2_dgrad_kernel<8, 4>), dim3(gx, (unsigned)B, (unsigned)(cin / 8)), dim3(256), (size_t)9 * cz * 8 * sizeof(float), stream, a);
        else hipLaunchKernelGGL((unet_s2_dgrad_kernel<8, 1>), dim3(gx, (unsigned)B, (unsigned)(cin / 8)), dim3(256), (size_t)9 * cz * 8 * sizeof(float), stream, a);
    } else {
        if (split) hipLaunchKernelGGL((unet_s2_dgrad_kernel<4, 4>), dim3(gx, (unsigned)B, (unsigned)(cin / 4)), dim3(256), (size_t)9 * cz * 4 * sizeof(float), stream, a);
        else hipLaunchKernelGGL((unet_s2_dgrad_kernel<4, 1>), dim3(gx, (unsigned)B, (unsigned)(cin / 4)), dim3(256), (size_t)9 * cz * 4 * sizeof(float), stream, a);
    }
    KSR_LAUNCH_CHECK("unet_s2_dgrad");
    return KSR_OK;
}

extern "C" int ksr_unet_att_mix_bwd(const float* gram, const float* dT, int c, int heads, int zc, int B, const float* temp_a,
                                    const float* temp_v, const float* wv, const float* wp, float vscale, float* D, float* dparams,
                                    hipStream_t stream) {
    KSR_CHECK_ARG(gram && dT && temp_a && temp_v && wv && wp && D && dparams && B > 0, "unet_att_mix_bwd: bad argument");
    KSR_UNSUPPORTED(c < 4 || c > 32 || (c & (c - 1)), "unet_att_mix_bwd: dim = %d (4, 8, 16 or 32)", c);
    KSR_UNSUPPORTED(heads < 1 || (c / 2) % heads != 0, "unet_att_mix_bwd: %d heads do not divide dim/2 = %d", heads, c / 2);
    KSR_CHECK_ARG(zc >= 2 * c + c / 2, "unet_att_mix_bwd: zc too small");
    MixBwdArgs a{gram, dT, c, heads, temp_a, temp_v, wv, wp, vscale, D, zc, dparams};
    hipLaunchKernelGGL(unet_att_mix_bwd_kernel, dim3((unsigned)B), dim3(256), 0, stream, a);
    KSR_LAUNCH_CHECK("unet_att_mix_bwd");
    return KSR_OK;
}
