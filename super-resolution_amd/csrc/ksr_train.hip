// Training-step support: the Y-plane L1 loss with its gradient, a fused Adam step over the flat
// parameter buffer, and the gradient all-reduce (RCCL over xGMI).
#include "ksr_common.h"
#include "ksr.h"
#include <math.h>
#include <atomic>
#include <rccl/rccl.h>

// Partial sums of ksr_l1_y_loss_static (one float per workgroup, at most 256 workgroups): 1 KB that every GPU the
// library runs on holds until the process ends.  In host code this name is only a handle for hipGetSymbolAddress;
// the device address is looked up once per device and kept in g_scratch.
#define KSR_L1_STATIC_FLOATS 256
static __device__ __align__(256) float scratch[KSR_L1_STATIC_FLOATS];

namespace {

// loss = mean |pred[:,0] - tgt[:,0]| ; g[:,0] = sign(pred - tgt) * gscale / numel.  One partial sum per
// workgroup goes to slab; l1_finish_kernel sums them in order (deterministic).
__global__ void __launch_bounds__(256) l1_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                 float* __restrict__ g, long plane, long pred_bs, long tgt_bs,
                                                 long g_bs, int B, float gscale, float* __restrict__ slab) {
    __shared__ float red[4];
    const long total4 = (long)B * plane / 4;
    const float inv = gscale / (float)((double)B * (double)plane);
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long e = i * 4;
        const long b = e / plane, off = e - b * plane;
        const float4 p = *reinterpret_cast<const float4*>(pred + b * pred_bs + off);
        const float4 t = *reinterpret_cast<const float4*>(tgt + b * tgt_bs + off);
        const float d0 = p.x - t.x, d1 = p.y - t.y, d2 = p.z - t.z, d3 = p.w - t.w;
        s += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
        float4 o;
        o.x = d0 > 0.f ? inv : (d0 < 0.f ? -inv : 0.f);
        o.y = d1 > 0.f ? inv : (d1 < 0.f ? -inv : 0.f);
        o.z = d2 > 0.f ? inv : (d2 < 0.f ? -inv : 0.f);
        o.w = d3 > 0.f ? inv : (d3 < 0.f ? -inv : 0.f);
        *reinterpret_cast<float4*>(g + b * g_bs + off) = o;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) slab[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// one wave: strided partial sums, then a butterfly (fixed order)
__global__ void __launch_bounds__(64) l1_finish_kernel(const float* slab, int G, float scale, float* loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < G; i += 64) s += slab[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) loss[0] = s * scale;
}

// Adam (torch.optim.Adam semantics, amsgrad off, maximize off): one pass over the flat buffers.
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                   float beta1, float beta2, float eps, float wd, float bc1,
                                                   float bc2_sqrt, float gscale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// The same step with the step count and the learning rate read from device memory (a recorded HIP graph replays its
// by-value arguments; these two change every step / every epoch).
__global__ void __launch_bounds__(256) adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long n,
                                                       const float* __restrict__ lr_dev, float beta1, float beta2, float eps,
                                                       float wd, const unsigned long long* __restrict__ step_dev,
                                                       float gscale) {
    const double st = (double)*step_dev;
    const float bc1 = (float)(1.0 - pow((double)beta1, st));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, st));
    const float lr = *lr_dev;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * gscale;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

ncclComm_t g_comm = nullptr;
int g_nranks = 1;

constexpr int KSR_MAX_DEVICES = 64;
std::atomic<float*> g_scratch[KSR_MAX_DEVICES];      // device address of `scratch` on each GPU (zero-initialised)

// Address of the current device's copy of `scratch`.  Two threads that race here resolve the same address.
int scratch_of_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= KSR_MAX_DEVICES) {
        ksr_set_error("l1_y_loss_static: no usable current device (%s)", hipGetErrorString(e));
        return KSR_ERR_HIP;
    }
    float* p = g_scratch[dev].load(std::memory_order_acquire);
    if (p == nullptr) {
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr) {
            ksr_set_error("l1_y_loss_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return KSR_ERR_HIP;
        }
        g_scratch[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return KSR_OK;
}

int l1_launch(const float* pred, int64_t pred_batch_stride, const float* tgt, int64_t tgt_batch_stride, float* grad,
              int64_t grad_batch_stride, int B, int64_t plane_elems, float grad_scale, float* slab256, float* loss_out,
              hipStream_t stream) {
    const long total4 = (long)B * plane_elems / 4;
    long g = (total4 + 255) / 256;
    if (g > KSR_L1_STATIC_FLOATS) g = KSR_L1_STATIC_FLOATS;
    hipLaunchKernelGGL(l1_kernel, dim3((unsigned)g), dim3(256), 0, stream, pred, tgt, grad, (long)plane_elems,
                       (long)pred_batch_stride, (long)tgt_batch_stride, (long)grad_batch_stride, B, grad_scale, slab256);
    KSR_LAUNCH_CHECK("l1_y_loss");
    hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(64), 0, stream, slab256, (int)g,
                       1.0f / (float)((double)B * (double)plane_elems), loss_out);
    KSR_LAUNCH_CHECK("l1_y_loss(finish)");
    return KSR_OK;
}

}  // namespace

extern "C" int ksr_l1_y_loss(const float* pred, int64_t pred_batch_stride, const float* tgt, int64_t tgt_batch_stride,
                             float* grad, int64_t grad_batch_stride, int B, int64_t plane_elems, float grad_scale,
                             float* slab256, float* loss_out, hipStream_t stream) {
    KSR_CHECK_ARG(pred && tgt && grad && slab256 && loss_out, "l1_y_loss: null pointer");
    KSR_CHECK_ARG(B > 0 && plane_elems > 0 && plane_elems % 4 == 0 && pred_batch_stride % 4 == 0 &&
                  tgt_batch_stride % 4 == 0 && grad_batch_stride % 4 == 0, "l1_y_loss: sizes must be multiples of 4");
    return l1_launch(pred, pred_batch_stride, tgt, tgt_batch_stride, grad, grad_batch_stride, B, plane_elems, grad_scale,
                     slab256, loss_out, stream);
}

// The same loss for callers that cannot allocate: the partial sums live in this library's own `scratch` array instead of
// a caller-provided workspace.  Every partial sum the finish kernel reads is written by the first kernel of the same
// call, so nothing has to be cleared between calls.  Calls that overlap on one GPU share the array and race.
extern "C" int ksr_l1_y_loss_static(const float* pred, int64_t pred_batch_stride, const float* tgt, int64_t tgt_batch_stride,
                                    float* grad, int64_t grad_batch_stride, int B, int64_t plane_elems, float grad_scale,
                                    float* loss_out, hipStream_t stream) {
    KSR_CHECK_ARG(pred && tgt && grad && loss_out, "l1_y_loss_static: null pointer");
    KSR_CHECK_ARG(B > 0 && plane_elems > 0 && plane_elems % 4 == 0 && pred_batch_stride % 4 == 0 &&
                  tgt_batch_stride % 4 == 0 && grad_batch_stride % 4 == 0, "l1_y_loss_static: sizes must be multiples of 4");
    KSR_CHECK_ARG((int64_t)B * plane_elems <= KSR_L1_STATIC_MAX_ELEMS,
                  "l1_y_loss_static: %lld elements exceed the maximum of %lld (2^32) this entry point accepts: each of its %d "
                  "partial sums would add more than 2^24 fp32 terms; use ksr_l1_y_loss",
                  (long long)B * (long long)plane_elems, (long long)KSR_L1_STATIC_MAX_ELEMS, KSR_L1_STATIC_FLOATS);
    float* slab = nullptr;
    const int rc = scratch_of_current_device(&slab);
    if (rc != KSR_OK) return rc;
    return l1_launch(pred, pred_batch_stride, tgt, tgt_batch_stride, grad, grad_batch_stride, B, plane_elems, grad_scale, slab,
                     loss_out, stream);
}

extern "C" int ksr_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             hipStream_t stream) {
    KSR_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad argument");
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)g), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, (long)n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
    KSR_LAUNCH_CHECK("adam_step");
    return KSR_OK;
}

extern "C" int ksr_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                                 const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                                 const uint64_t* step_dev, float grad_scale, hipStream_t stream) {
    KSR_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && lr_dev && step_dev && n > 0, "adam_step_dev: bad argument");
    long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)g), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, (long)n,
                       lr_dev, beta1, beta2, eps, weight_decay, reinterpret_cast<const unsigned long long*>(step_dev),
                       grad_scale);
    KSR_LAUNCH_CHECK("adam_step_dev");
    return KSR_OK;
}

// ------------------------------------------------------------------------------------------ RCCL
extern "C" int ksr_comm_unique_id(void* out128) {
    KSR_CHECK_ARG(out128 != nullptr, "comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == 128, "unexpected ncclUniqueId size");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { ksr_set_error("ncclGetUniqueId: %s", ncclGetErrorString(r)); return KSR_ERR_COMM; }
    memcpy(out128, &id, 128);
    return KSR_OK;
}

extern "C" int ksr_comm_init(int rank, int nranks, const void* unique_id128) {
    KSR_CHECK_ARG(unique_id128 && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad argument");
    KSR_CHECK_ARG(g_comm == nullptr, "comm_init: communicator already initialised");
    ncclUniqueId id;
    memcpy(&id, unique_id128, 128);
    ncclResult_t r = ncclCommInitRank(&g_comm, nranks, id, rank);
    if (r != ncclSuccess) { g_comm = nullptr; ksr_set_error("ncclCommInitRank: %s", ncclGetErrorString(r)); return KSR_ERR_COMM; }
    g_nranks = nranks;
    return KSR_OK;
}

// In-place sum all-reduce of n fp32 values; the 1/nranks of the mean is applied by the consumer
// (ksr_adam_step's grad_scale) so no extra pass over the buffer is needed.
extern "C" int ksr_allreduce_sum(float* buf, int64_t n, hipStream_t stream) {
    KSR_CHECK_ARG(buf && n > 0, "allreduce_sum: bad argument");
    KSR_CHECK_ARG(g_comm != nullptr, "allreduce_sum: communicator not initialised");
    ncclResult_t r = ncclAllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, g_comm, stream);
    if (r != ncclSuccess) { ksr_set_error("ncclAllReduce: %s", ncclGetErrorString(r)); return KSR_ERR_COMM; }
    return KSR_OK;
}

extern "C" int ksr_comm_nranks(void) { return g_comm ? g_nranks : 0; }

extern "C" int ksr_comm_destroy(void) {
    if (g_comm) {
        ncclCommDestroy(g_comm);
        g_comm = nullptr;
        g_nranks = 1;
    }
    return KSR_OK;
}
