// SSIM term of the reference's training loss (src/ml/pipelines/default.py:79-81: loss = L1(Y) + 0.15 (1 - SSIM)),
// SSIM = torchmetrics' StructuralSimilarityIndexMeasure(data_range=(0, 1)) with its defaults, restated from the
// published definition (torchmetrics is not part of the reference tree):
//   inputs clamped to [0, 1]; Gaussian window 11 x 11, sigma 1.5 (separable, normalised); c1 = 0.01^2, c2 = 0.03^2;
//   mu_p, mu_t, E[p^2], E[t^2], E[pt] by windowed averaging; variances clamped at 0;
//   S = (2 mu_p mu_t + c1)(2 cov + c2) / ((mu_p^2 + mu_t^2 + c1)(var_p + var_t + c2));
//   only windows that lie entirely inside the image count (the (H-10) x (W-10) interior); mean over pixels,
//   channels and images.
// Forward: column strips walked top to bottom (vertical filter in registers, horizontal filter through LDS, see
// ssim_fwd_cols_kernel), per-workgroup partial sums reduced in a fixed order.  Backward (luma plane only: the chroma planes do not depend on the parameters): the
// forward also writes dS/dmu_p, dS/dE[p^2], dS/dE[pt] per window; the gradient kernel filters those three maps
// with the same window and combines them with p and t.
#include "ksr_common.h"

#include <algorithm>
#include <type_traits>

namespace {

constexpr int SS_K = 11, SS_R = 5;
constexpr int SS_T = 32;                 // the workspace sizes (ksr_ssim_slab_floats, ksr_ssim_l1_slab_floats) count 32 x 32 tiles

struct SsimWin { float g[SS_K]; };
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct SsimArgs {
    const float* pred; long pred_bs;     // [B, C, H, W] with images pred_bs elements apart
    const float* tgt; long tgt_bs;
    float* maps;                         // [3][B][Hi][Wi] or null: derivative maps of channel 0
    float* slab;
    int C, B, H, W;
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// ---------------------------------------------------------------------------------------------------------------
// Forward, column-first.  A workgroup of 128 threads owns a strip of <= 128 input columns (<= 118 window columns) and a
// band of window rows (the whole height whenever strips x planes alone fill the chip: no row is then filtered twice),
// and walks down the band: thread = input column.  VERTICAL pass in registers: an input row
// adds g[k] * (p, t, pp, tt, pt) to the running sums of the 11 window rows it belongs to (scatter form; the sums rotate
// through 11 register sets, the loop over an 11-row batch is unrolled so that every index is static; the tap that
// starts a window row overwrites its set) and completes one window row, whose five column sums go to LDS.  HORIZONTAL
// pass after each batch: one thread per (row of the batch, run of 11 windows) slides over 21 columns (1.9 LDS reads per
// window and quantity group) and evaluates S and, for the luma channel, the three derivative maps.  The next batch's
// pixels are requested before the horizontal pass.  Against the tile version (32 x 32 windows from 42 x 42 pixels,
// filter passes that keep 3 and 2 of 4 waves busy): no rows are filtered twice, 6 % of the columns are, every thread
// works in both passes.
// The kernel is bound by the vector ALUs' throughput, not by latency: with taps removed from one part at a time (64 x 3 x
// 512 x 512, 266 us) the vertical filter accounts for 76 us, the horizontal one for 90, the evaluation of S for 47 and
// loads / LDS / barriers for the rest, the parts adding up; a variant with three waves per SIMD instead of two (runs of 5
// windows, 6-row sub-batches: 168 registers, 17 KB of LDS) ran in the same time.  About 5 cycles per vector instruction and
// SIMD (two thirds of them packed) is what this chip sustains on such a stream.
constexpr int SV_T = 128;                       // two waves: four workgroups give every SIMD of a CU its two waves (three-wave
                                                // workgroups left two of the four SIMDs with one)
constexpr int SV_WO = SV_T - (SS_K - 1);        // window columns per strip at most
constexpr int SV_R = SS_K;                      // input rows per batch
constexpr int SV_RUN = 11, SV_SPAN = SV_RUN + SS_K - 1;
constexpr int SV_V1_LD = SV_T + 25;             // row pitch 153: the horizontal pass's reads (lane = batch row x run of 11 columns: 11 columns
                                                // apart inside a row, six rows per wave) are free of bank conflicts for both the 16-byte and
                                                // the 4-byte rows (144 made them three- and two-way; no measurable difference)
static_assert(((SV_WO + SV_RUN - 1) / SV_RUN - 1) * SV_RUN + SV_SPAN <= SV_V1_LD, "the last run's columns stay inside a padded row");
constexpr int SV_GRID = 256 * 4;                // persistent workgroups: four fit a CU (registers)

__global__ void __launch_bounds__(SV_T, 2) ssim_fwd_cols_kernel(SsimArgs a, SsimWin w, int ns, int wo, int nb, int bh) {
    __shared__ float4 v4[SV_R][SV_V1_LD];        // (mu_p, mu_t, E[pp], E[tt]) column sums of a batch's window rows; the 16
                                                 // columns of padding are only read for windows past the strip
    __shared__ float v1[SV_R][SV_V1_LD];         // E[pt]
    __shared__ float red[SV_T / 64];
    const int Hi = a.H - 2 * SS_R, Wi = a.W - 2 * SS_R;
    const int total = ns * nb * a.C * a.B;
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = (int)ksr_xcd_contiguous(blockIdx.x, gridDim.x) * chunk, t_end = min(total, t_begin + chunk);
    const long plane = (long)a.H * a.W;
    const int tid = threadIdx.x;
    const int nruns = (wo + SV_RUN - 1) / SV_RUN;
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    float sum = 0.f;
    // One (strip, band, image, plane) task.  LUMA: the plane whose derivative maps the gradient kernel reads: five filtered
    // quantities (p, t, pp, tt, pt), the two variances clamped at zero one by one, as the gradient needs the sign of the
    // prediction's.  The chroma planes only contribute their value, in which the variances appear as their sum: four
    // quantities (p, t, pp + tt, pt), a third fewer multiply-adds in both filter passes and no second LDS row; the sum is
    // clamped as a whole, which differs from the separate clamps only where a window is flat to rounding (|d| < 1e-8
    // against c2 = 9e-4).
    auto task = [&](auto luma_tag, int strip, int band, int b, int c) {
        constexpr bool LUMA = decltype(luma_tag)::value;
        const int x0 = strip * wo, y0 = band * bh;
        const int xo_end = min(Wi, x0 + wo), yo_end = min(Hi, y0 + bh);      // windows [y0, yo_end) x [x0, xo_end)
        const int y_in_end = yo_end + SS_K - 1;                                  // input rows [y0, y_in_end)
        const int nbatch = (y_in_end - y0 + SV_R - 1) / SV_R;
        // uniform row pointers + the thread's column as an unsigned 32-bit offset: the address form with a scalar base.
        // Rows past the band and columns past the strip are clamped into the plane and loaded like the others: they
        // only reach window rows / windows that are never evaluated.
        const float* P = a.pred + (long)b * a.pred_bs + (long)c * plane + x0;
        const float* T = a.tgt + (long)b * a.tgt_bs + (long)c * plane + x0;
        const unsigned tcol = (unsigned)(min(x0 + tid, a.W - 1) - x0);
        float pv[SV_R], tv[SV_R];
        auto fetch = [&](int m) {
#pragma unroll
            for (int rr = 0; rr < SV_R; ++rr) {
                const long row = (long)min(y0 + SV_R * m + rr, a.H - 1) * a.W;
                pv[rr] = (P + row)[tcol];
                tv[rr] = (T + row)[tcol];
            }
        };
        fetch(0);
        f32x2 sa[SS_K], sb[SS_K];
        float sc[LUMA ? SS_K : 1];
#pragma unroll
        for (int j = 0; j < SS_K; ++j) { sa[j] = f32x2{0.f, 0.f}; sb[j] = f32x2{0.f, 0.f}; if constexpr (LUMA) sc[j] = 0.f; }
        for (int m = 0; m < nbatch; ++m) {
            // ---- vertical: input row y0 + 11 m + rr is tap k of window row y - k, kept in register set (rr - k) mod 11
#pragma unroll
            for (int rr = 0; rr < SV_R; ++rr) {
                const f32x2 x = {clamp01(pv[rr]), clamp01(tv[rr])};
                const f32x2 xx = x * x;
                const float pt = x[0] * x[1];
                const f32x2 y2 = LUMA ? xx : f32x2{xx[0] + xx[1], pt};      // chroma: (pp + tt, pt)
#pragma unroll
                for (int k = 0; k < SS_K; ++k) {
                    const int j = (rr - k + SS_K) % SS_K;
                    const float g = w.g[k < 6 ? k : 10 - k];      // the window is symmetric: six weights live in scalar registers
                    if (k == 0) { sa[j] = g * x; sb[j] = g * y2; if constexpr (LUMA) sc[j] = g * pt; }
                    else { sa[j] += g * x; sb[j] += g * y2; if constexpr (LUMA) sc[j] += g * pt; }
                }
                const int jc = (rr + 1) % SS_K;             // window row y - 10 is complete
                v4[rr][tid] = make_float4(sa[jc][0], sa[jc][1], sb[jc][0], sb[jc][1]);
                if constexpr (LUMA) v1[rr][tid] = sc[jc];
            }
            if (m + 1 < nbatch) fetch(m + 1);
            __syncthreads();
            // ---- horizontal: thread = (batch row rr, run j of 11 windows), j fastest
            const bool want_maps = LUMA && a.maps;         // uniform over the workgroup
            const int hrr = tid / nruns, hj = tid - hrr * nruns, q0 = hj * SV_RUN;
            const int uy = y0 + SV_R * m + hrr - (SS_K - 1);
            const bool hrow = tid < SV_R * nruns && uy >= y0 && uy < yo_end;
            float dmap[SV_RUN][3];
            if (hrow) {
                const int rr = hrr;
                f32x2 ha[SV_RUN], hb[SV_RUN];
                float hc[LUMA ? SV_RUN : 1];
#pragma unroll
                for (int o = 0; o < SV_RUN; ++o) { ha[o] = f32x2{0.f, 0.f}; hb[o] = f32x2{0.f, 0.f}; if constexpr (LUMA) hc[o] = 0.f; }
#pragma unroll
                for (int i = 0; i < SV_SPAN; ++i) {
                    // at most four columns' reads ahead of their use: hoisting all 21 costs 100 registers
                    if (i % 4 == 0) __builtin_amdgcn_sched_barrier(0);
                    const int q = q0 + i;        // < SV_V1_LD; columns past the strip only feed windows past it
                    const float4 v = v4[rr][q];
                    float vc = 0.f;
                    if constexpr (LUMA) vc = v1[rr][q];
                    const f32x2 va = {v.x, v.y}, vb = {v.z, v.w};
#pragma unroll
                    for (int o = 0; o < SV_RUN; ++o) {
                        if (i - o < 0 || i - o >= SS_K) continue;
                        const float g = w.g[i - o < 6 ? i - o : 10 - (i - o)];
                        ha[o] += g * va; hb[o] += g * vb;
                        if constexpr (LUMA) hc[o] += g * vc;
                    }
                }
#pragma unroll
                for (int o = 0; o < SV_RUN; ++o) {
                    const int ux = x0 + q0 + o;
                    dmap[o][0] = dmap[o][1] = dmap[o][2] = 0.f;
                    if (ux >= xo_end) continue;
                    const float m1 = ha[o][0], m2 = ha[o][1];
                    float vp_raw = 0.f, cov, D2;
                    if constexpr (LUMA) {
                        const float e11 = hb[o][0], e22 = hb[o][1], e12 = hc[o];
                        vp_raw = e11 - m1 * m1;
                        const float vt_raw = e22 - m2 * m2;
                        cov = e12 - m1 * m2;
                        D2 = fmaxf(vp_raw, 0.f) + fmaxf(vt_raw, 0.f) + c2;
                    } else {
                        // the two variances enter S only as their sum: one filtered quantity E[pp + tt] for both
                        cov = hb[o][1] - m1 * m2;
                        D2 = fmaxf(hb[o][0] - m1 * m1 - m2 * m2, 0.f) + c2;
                    }
                    const float N1 = 2.f * m1 * m2 + c1, D1 = m1 * m1 + m2 * m2 + c1;
                    const float N2 = 2.f * cov + c2;
                    const float i1 = __builtin_amdgcn_rcpf(D1), i2 = __builtin_amdgcn_rcpf(D2);
                    const float inv = i1 * i2;
                    const float S = N1 * N2 * inv;
                    sum += S;
                    if (want_maps) {
                        const float vp_on = vp_raw > 0.f ? 1.f : 0.f;
                        dmap[o][2] = 2.f * N1 * inv;
                        dmap[o][1] = -S * i2 * vp_on;
                        dmap[o][0] = 2.f * m2 * (N2 - N1) * inv - 2.f * m1 * S * (i1 - i2 * vp_on);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // one window's temporaries at a time
                }
            }
            if (want_maps) {
                // The derivative maps leave through the (now free) column-sum rows: a thread's 11 windows are 11 columns
                // of one row, so storing them directly puts 64 lanes on 64 separate 44-byte strides per instruction (that
                // cost a third of the kernel's time); from LDS thread = column stores whole rows.
                __syncthreads();           // every thread has read its columns of v4
                if (hrow) {
#pragma unroll
                    for (int o = 0; o < SV_RUN; ++o) v4[hrr][q0 + o] = make_float4(dmap[o][0], dmap[o][1], dmap[o][2], 0.f);
                }
                __syncthreads();
                const long ms = (long)a.B * Hi * Wi;
                if (x0 + tid < xo_end) {
#pragma unroll
                    for (int rr = 0; rr < SV_R; ++rr) {
                        const int vy = y0 + SV_R * m + rr - (SS_K - 1);
                        if (vy < y0 || vy >= yo_end) continue;
                        const float4 d = v4[rr][tid];
                        float* mp = a.maps + ((long)b * Hi + vy) * Wi + x0;
                        mp[tid] = d.x; (mp + ms)[tid] = d.y; (mp + 2 * ms)[tid] = d.z;
                    }
                }
            }
            __syncthreads();       // the next batch's vertical pass overwrites v4 / v1
        }
    };
    for (int t = t_begin; t < t_end; ++t) {
        // channel fastest: a workgroup's chunk of tasks holds the planes of one (strip, band, image) - the luma plane, which
        // also writes the derivative maps, costs 1.4 x a chroma plane, and every workgroup is resident from the start
        const int c = t % a.C, strip = (t / a.C) % ns, band = (t / (a.C * ns)) % nb, b = t / (a.C * ns * nb);
        if (c == 0) task(std::true_type{}, strip, band, b, c); else task(std::false_type{}, strip, band, b, c);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = red[0];
#pragma unroll
        for (int i = 1; i < SV_T / 64; ++i) s += red[i];
        a.slab[blockIdx.x] = s;
    }
}

// One workgroup: strided partial sums in double, a butterfly inside each wave, then the 16 wave sums in order
// (deterministic).
// With loss_out: loss_out[0] = out[0] + ssim_weight * (1 - ssim[0]), every operation rounded on its own (the sum a training
// loop would otherwise form from the two one-element tensors).
__global__ void __launch_bounds__(1024) ssim_finish_kernel(const float* slab, long G, double scale, float* out,
                                                            const float* ssim = nullptr, float ssim_weight = 0.f,
                                                            float* loss_out = nullptr) {
    __shared__ double part[16];
    double s = 0.0;
    for (long i = threadIdx.x; i < G; i += 1024) s += (double)slab[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w];
        const float v = (float)(t * scale);
        out[0] = v;
        if (loss_out) loss_out[0] = __fadd_rn(v, __fmul_rn(__fsub_rn(1.0f, ssim[0]), ssim_weight));
    }
}

struct SsimGradArgs {
    const float* maps;                   // [3][B][Hi][Wi]
    const float* pred; long pred_bs;     // luma planes
    const float* tgt; long tgt_bs;
    float* grad; long grad_bs;           // accumulated into
    int B, H, W;
    float scale;
    float l1_inv;                        // L1: gradient magnitude of the L1 term, gscale / (B H W)
    float* l1_slab;                      // L1: [gridDim.x] partial sums of |pred - tgt|
};

// grad[x] += scale * clampmask(p[x]) * sum over windows u covering x of g(x - u) (A_u + 2 p_x B_u + t_x C_u)
// L1: the luma L1 term rides along (the kernel reads both planes anyway): grad[x] = l1_inv * sign(p - t) + the above
// (written, not accumulated) and the workgroup's sum of |p - t| goes to l1_slab.
// Column-first like the forward kernel (which it mirrors line by line): with the three maps zero-padded by 10 on every
// side, grad = the "valid" 11 x 11 filter of (A, B, C) combined with the pixel, so a workgroup of 128 threads
// (thread = padded map column) walks down a strip of <= 118 output columns over a band of output rows: vertical filter in
// registers in scatter form (11 register sets), horizontal filter after each 11-row batch through LDS (thread = row x run
// of 11 outputs), the filtered triples back through LDS so that thread = column combines them with its pixels of both
// planes (requested before the horizontal pass) and stores whole rows.  Against the 32 x 32 tile kernel it replaces (0.167
// ms for 64 planes of 512 x 512, now 0.114): no row or column of the maps is filtered twice inside a strip, every thread
// works in both passes (the tile kernel kept 126 and 96 of 256 busy), three barriers per 11 rows of ~100 pixels instead of
// two per 32 x 32 tile.
template <bool L1>
__global__ void __launch_bounds__(SV_T, 2) ssim_grad_cols_kernel(SsimGradArgs a, SsimWin w, int ns, int wo, int nb, int bh) {
    __shared__ float4 v4[SV_R][SV_V1_LD];        // column sums of a batch's rows, then the filtered (A, B, C) of its pixels
    __shared__ float l1red[SV_T / 64];
    const int Hi = a.H - 2 * SS_R, Wi = a.W - 2 * SS_R;
    const int total = ns * nb * a.B;
    const int chunk = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = (int)ksr_xcd_contiguous(blockIdx.x, gridDim.x) * chunk, t_end = min(total, t_begin + chunk);
    const long msz = (long)a.B * Hi * Wi;
    const int tid = threadIdx.x;
    const int nruns = (wo + SV_RUN - 1) / SV_RUN;
    float l1acc = 0.f;
    for (int t = t_begin; t < t_end; ++t) {
        const int strip = t % ns, band = (t / ns) % nb, b = t / (ns * nb);
        const int x0 = strip * wo, y0 = band * bh;
        const int xo_end = min(a.W, x0 + wo), yo_end = min(a.H, y0 + bh);        // pixels [y0, yo_end) x [x0, xo_end)
        const int r_end = yo_end + SS_K - 1;                                       // padded map rows [y0, r_end)
        const int nbatch = (r_end - y0 + SV_R - 1) / SV_R;
        // every load is issued from an address clamped into the maps / planes (scalar base + 32-bit offset); map entries
        // outside the image's windows contribute zero
        const float* mb = a.maps + (long)b * Hi * Wi;
        const int cm = x0 + tid - (SS_K - 1);                                      // this thread's map column
        const bool col_in = cm >= 0 && cm < Wi;
        const unsigned cmc = (unsigned)min(max(cm, 0), Wi - 1);
        const float* P = a.pred + (long)b * a.pred_bs;
        const float* T = a.tgt + (long)b * a.tgt_bs;
        float* Gd = a.grad + (long)b * a.grad_bs;
        const unsigned pcol = (unsigned)min(x0 + tid, a.W - 1);
        float mv[SV_R][3];
        auto fetch = [&](int m) {
#pragma unroll
            for (int rr = 0; rr < SV_R; ++rr) {
                const int rm = y0 + SV_R * m + rr - (SS_K - 1);                    // map row of padded row y0 + 11 m + rr
                const bool in = col_in && rm >= 0 && rm < Hi;
                const unsigned o = (unsigned)(min(max(rm, 0), Hi - 1) * Wi) + cmc;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float v = (mb + k * msz)[o];
                    mv[rr][k] = in ? v : 0.f;
                }
            }
        };
        fetch(0);
        f32x2 sa[SS_K];
        float sc[SS_K];
#pragma unroll
        for (int j = 0; j < SS_K; ++j) { sa[j] = f32x2{0.f, 0.f}; sc[j] = 0.f; }
        for (int m = 0; m < nbatch; ++m) {
            // ---- vertical: padded row y0 + 11 m + rr is tap k of output row (that row) - k, kept in register set (rr - k) mod 11
#pragma unroll
            for (int rr = 0; rr < SV_R; ++rr) {
                const f32x2 x = {mv[rr][0], mv[rr][1]};
                const float cv = mv[rr][2];
#pragma unroll
                for (int k = 0; k < SS_K; ++k) {
                    const int j = (rr - k + SS_K) % SS_K;
                    const float g = w.g[k < 6 ? k : 10 - k];
                    if (k == 0) { sa[j] = g * x; sc[j] = g * cv; }
                    else { sa[j] += g * x; sc[j] += g * cv; }
                }
                const int jc = (rr + 1) % SS_K;             // output row (that row) - 10 is complete
                v4[rr][tid] = make_float4(sa[jc][0], sa[jc][1], sc[jc], 0.f);
            }
            // this batch's pixels of the thread's column (output rows y0 + 11 m + rr - 10) and the next batch's map rows:
            // in flight during the horizontal pass
            float pin[SV_R], tin[SV_R];
#pragma unroll
            for (int rr = 0; rr < SV_R; ++rr) {
                const int vy = y0 + SV_R * m + rr - (SS_K - 1);
                const unsigned off = (unsigned)(min(max(vy, 0), a.H - 1) * a.W) + pcol;
                pin[rr] = P[off];
                tin[rr] = T[off];
            }
            if (m + 1 < nbatch) fetch(m + 1);
            __syncthreads();
            // ---- horizontal: thread = (batch row, run of 11 pixels), run fastest
            const int hrr = tid / nruns, hj = tid - hrr * nruns, q0 = hj * SV_RUN;
            const int uy = y0 + SV_R * m + hrr - (SS_K - 1);
            const bool hrow = tid < SV_R * nruns && uy >= y0 && uy < yo_end;
            f32x2 ha[SV_RUN];
            float hc[SV_RUN];
            if (hrow) {
#pragma unroll
                for (int o = 0; o < SV_RUN; ++o) { ha[o] = f32x2{0.f, 0.f}; hc[o] = 0.f; }
#pragma unroll
                for (int i = 0; i < SV_SPAN; ++i) {
                    if (i % 4 == 0) __builtin_amdgcn_sched_barrier(0);       // at most four columns' reads ahead of their use
                    const float4 v = v4[hrr][q0 + i];
                    const f32x2 va = {v.x, v.y};
#pragma unroll
                    for (int o = 0; o < SV_RUN; ++o) {
                        if (i - o < 0 || i - o >= SS_K) continue;
                        const float g = w.g[i - o < 6 ? i - o : 10 - (i - o)];
                        ha[o] += g * va; hc[o] += g * v.z;
                    }
                }
            }
            __syncthreads();           // every thread has read its columns of v4
            if (hrow) {
#pragma unroll
                for (int o = 0; o < SV_RUN; ++o) v4[hrr][q0 + o] = make_float4(ha[o][0], ha[o][1], hc[o], 0.f);
            }
            __syncthreads();
            if (x0 + tid < xo_end) {
#pragma unroll
                for (int rr = 0; rr < SV_R; ++rr) {
                    const int vy = y0 + SV_R * m + rr - (SS_K - 1);
                    if (vy < y0 || vy >= yo_end) continue;
                    const float4 d = v4[rr][tid];
                    const float p = pin[rr], traw = tin[rr], tg = clamp01(traw);
                    const float inside = (p >= 0.f && p <= 1.f) ? 1.f : 0.f;
                    const float gs = a.scale * inside * (d.x + 2.f * clamp01(p) * d.y + tg * d.z);
                    float* gp = Gd + (long)vy * a.W + x0;
                    if constexpr (L1) {
                        const float df = p - traw;
                        l1acc += fabsf(df);
                        gp[tid] = (df > 0.f ? a.l1_inv : (df < 0.f ? -a.l1_inv : 0.f)) + gs;
                    } else {
                        gp[tid] += gs;
                    }
                }
            }
            __syncthreads();       // the next batch's vertical pass overwrites v4
        }
    }
    if constexpr (L1) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) l1acc += __shfl_xor(l1acc, off, 64);
        if ((threadIdx.x & 63) == 0) l1red[threadIdx.x >> 6] = l1acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = l1red[0];
#pragma unroll
            for (int i = 1; i < SV_T / 64; ++i) s += l1red[i];
            a.l1_slab[blockIdx.x] = s;
        }
    }
}

int fill_window(SsimWin& w) {
    // normalised fp32 Gaussian, sigma 1.5, taps -5..5
    float s = 0.f;
    for (int k = 0; k < SS_K; ++k) {
        const float d = (float)(k - SS_R) / 1.5f;
        w.g[k] = expf(-d * d / 2.f);
        s += w.g[k];
    }
    for (int k = 0; k < SS_K; ++k) w.g[k] /= s;
    return KSR_OK;
}

}  // namespace

extern "C" int64_t ksr_ssim_slab_floats(int C, int B, int H, int W) {
    if (C <= 0 || B <= 0 || H <= 2 * SS_R || W <= 2 * SS_R) return 0;
    const long tiles = (long)((H - 2 * SS_R + SS_T - 1) / SS_T) * ((W - 2 * SS_R + SS_T - 1) / SS_T);
    return tiles * C * B;
}

// ssim_out[0] = SSIM(pred, tgt) over C channels of B images (see the file header).  maps (optional):
// 3 * B * (H-10) * (W-10) floats receiving the per-window derivative maps of channel 0 for ksr_ssim_y_grad.
extern "C" int ksr_ssim_fwd(const float* pred, int64_t pred_batch_stride, const float* tgt, int64_t tgt_batch_stride, int C,
                            int B, int H, int W, float* maps, float* slab, float* ssim_out, hipStream_t stream) {
    KSR_CHECK_ARG(pred && tgt && slab && ssim_out && C > 0 && B > 0, "ssim_fwd: bad argument");
    KSR_CHECK_ARG(H > 2 * SS_R && W > 2 * SS_R, "ssim_fwd: the image must be larger than the 11 x 11 window");
    SsimWin w;
    fill_window(w);
    SsimArgs a{pred, (long)pred_batch_stride, tgt, (long)tgt_batch_stride, maps, slab, C, B, H, W};
    const int Hi = H - 2 * SS_R, Wi = W - 2 * SS_R;
    // strips of equal width (<= 118 window columns); bands of equal height, as few as give every workgroup slot a task
    // (a band re-filters the 10 rows above it), at least 23 window rows (three batches of input rows) each
    const int ns = (Wi + SV_WO - 1) / SV_WO, wo = (Wi + ns - 1) / ns;
    int nb = 1;
    while ((long)ns * nb * B < SV_GRID * 9 / 10 && Hi / (nb + 1) >= 2 * SS_K + 1) ++nb;
    const int bh = (Hi + nb - 1) / nb;
    KSR_CHECK_ARG((long)ns * nb * C * B < (1L << 31), "ssim_fwd: too many strips for one launch");
    const unsigned grid = (unsigned)std::min<long>((long)ns * nb * B, SV_GRID);      // a workgroup takes all planes of its (strip, band, image)s
    hipLaunchKernelGGL(ssim_fwd_cols_kernel, dim3(grid), dim3(SV_T), 0, stream, a, w, ns, wo, nb, bh);
    KSR_LAUNCH_CHECK("ssim_fwd");
    hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(1024), 0, stream, slab, (long)grid,
                       1.0 / ((double)B * C * Hi * Wi), ssim_out);
    KSR_LAUNCH_CHECK("ssim_fwd(finish)");
    return KSR_OK;
}

// Strips, bands and workgroups of the column-first gradient kernel, chosen as for the forward kernel; never more workgroups
// than the L1 slab (ksr_ssim_l1_slab_floats) has entries.
struct SsimGradPlan { int ns, wo, nb, bh; unsigned grid; };
static SsimGradPlan ssim_grad_plan(int B, int H, int W) {
    SsimGradPlan p;
    p.ns = (W + SV_WO - 1) / SV_WO;
    p.wo = (W + p.ns - 1) / p.ns;
    p.nb = 1;
    while ((long)p.ns * p.nb * B < SV_GRID * 9 / 10 && H / (p.nb + 1) >= 2 * SS_K + 1) ++p.nb;
    p.bh = (H + p.nb - 1) / p.nb;
    const long slab = (long)((H + SS_T - 1) / SS_T) * ((W + SS_T - 1) / SS_T) * B;
    long g = std::min<long>((long)p.ns * p.nb * B, SV_GRID);
    p.grid = (unsigned)std::min<long>(g, slab);
    return p;
}

// grad[b, y, x] += scale * dSSIM_sum/dpred[b, 0, y, x], where SSIM_sum is the un-normalised sum of the window values
// of channel 0 (scale carries the loss weight and 1 / (B * C * (H-10) * (W-10))).  pred / tgt point at the luma planes.
extern "C" int ksr_ssim_y_grad(const float* maps, const float* pred, int64_t pred_batch_stride, const float* tgt,
                               int64_t tgt_batch_stride, float* grad, int64_t grad_batch_stride, int B, int H, int W,
                               float scale, hipStream_t stream) {
    KSR_CHECK_ARG(maps && pred && tgt && grad && B > 0, "ssim_y_grad: bad argument");
    KSR_CHECK_ARG(H > 2 * SS_R && W > 2 * SS_R, "ssim_y_grad: the image must be larger than the 11 x 11 window");
    SsimWin w;
    fill_window(w);
    SsimGradArgs a{maps, pred, (long)pred_batch_stride, tgt, (long)tgt_batch_stride, grad, (long)grad_batch_stride, B, H, W, scale,
                   0.f, nullptr};
    const unsigned tiles = (unsigned)(((H + SS_T - 1) / SS_T) * ((W + SS_T - 1) / SS_T));
    KSR_CHECK_ARG((long)tiles * B < (1L << 31), "ssim_y_grad: too many tiles for one launch");
    const SsimGradPlan pl = ssim_grad_plan(B, H, W);
    hipLaunchKernelGGL(ssim_grad_cols_kernel<false>, dim3(pl.grid), dim3(SV_T), 0, stream, a, w, pl.ns, pl.wo, pl.nb, pl.bh);
    KSR_LAUNCH_CHECK("ssim_y_grad");
    return KSR_OK;
}

extern "C" int64_t ksr_ssim_l1_slab_floats(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (int64_t)((H + SS_T - 1) / SS_T) * ((W + SS_T - 1) / SS_T) * B;
}

// ksr_l1_y_loss and ksr_ssim_y_grad in one pass over the luma planes: grad[b, y, x] = l1_grad_scale * sign(pred - tgt) /
// (B H W) + scale * dSSIM_sum/dpred (written, not accumulated), l1_out[0] = mean |pred - tgt|.
// l1_slab: ksr_ssim_l1_slab_floats(B, H, W) floats of workspace.  loss_out (optional, with ssim_value = the device scalar
// ksr_ssim_fwd wrote): loss_out[0] = l1_out[0] + ssim_weight * (1 - ssim_value[0]).
extern "C" int ksr_ssim_y_grad_l1(const float* maps, const float* pred, int64_t pred_batch_stride, const float* tgt,
                                  int64_t tgt_batch_stride, float* grad, int64_t grad_batch_stride, int B, int H, int W,
                                  float scale, float l1_grad_scale, float* l1_slab, float* l1_out, const float* ssim_value,
                                  float ssim_weight, float* loss_out, hipStream_t stream) {
    KSR_CHECK_ARG(maps && pred && tgt && grad && l1_slab && l1_out && B > 0, "ssim_y_grad_l1: bad argument");
    KSR_CHECK_ARG(!loss_out || ssim_value, "ssim_y_grad_l1: loss_out needs ssim_value");
    KSR_CHECK_ARG(H > 2 * SS_R && W > 2 * SS_R, "ssim_y_grad_l1: the image must be larger than the 11 x 11 window");
    SsimWin w;
    fill_window(w);
    const double numel = (double)B * H * W;
    SsimGradArgs a{maps, pred, (long)pred_batch_stride, tgt, (long)tgt_batch_stride, grad, (long)grad_batch_stride, B, H, W, scale,
                   (float)(l1_grad_scale / numel), l1_slab};
    const unsigned tiles = (unsigned)(((H + SS_T - 1) / SS_T) * ((W + SS_T - 1) / SS_T));
    KSR_CHECK_ARG((long)tiles * B < (1L << 31), "ssim_y_grad_l1: too many tiles for one launch");
    const SsimGradPlan pl = ssim_grad_plan(B, H, W);
    const unsigned grid = pl.grid;
    hipLaunchKernelGGL(ssim_grad_cols_kernel<true>, dim3(grid), dim3(SV_T), 0, stream, a, w, pl.ns, pl.wo, pl.nb, pl.bh);
    KSR_LAUNCH_CHECK("ssim_y_grad_l1");
    hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(1024), 0, stream, l1_slab, (long)grid, 1.0 / numel, l1_out,
                       ssim_value, ssim_weight, loss_out);
    KSR_LAUNCH_CHECK("ssim_y_grad_l1(finish)");
    return KSR_OK;
}
