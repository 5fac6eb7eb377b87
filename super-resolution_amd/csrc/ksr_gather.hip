// libksr: batched strided gather.  One launch executes a table of small re-layouts
//     dst[sum_d i_d * dstride_d] = sum_{r < nred} src[sum_d i_d * sstride_d + r * rstride] * (src2 ? src2[...] : 1)
// over 4-D index spaces: the permutes / flips / diagonal embeddings / slices / small reductions that turn the UNet
// hyper-encoder's parameters into the layouts its kernels read (torch [cout][cin][kh][kw] -> [tap][cin][cout], the
// flipped + transposed copies for the data gradients, the dense embedding and composition of the anchor attention's
// depthwise projections) and that turn the kernels' gradient layouts back into the parameters' shapes, including the
// chain rule through those compositions.  Each of them is a few hundred to a few ten thousand elements; launched one
// by one (as torch ops) they were about 260 of the 470 launches of a training step.
#include "ksr_common.h"
#include "ksr.h"

namespace {

// KsrGatherBlock: 256 consecutive elements of one table entry per workgroup
__global__ void __launch_bounds__(256) gather_kernel(const KsrGatherDesc* __restrict__ descs, const KsrGatherBlock* __restrict__ blocks) {
    const KsrGatherBlock blk = blocks[blockIdx.x];
    const KsrGatherDesc& d = descs[blk.desc];
    const int i = blk.first + threadIdx.x;
    if (i >= d.count) return;
    int rem = i;
    const int i3 = rem % d.shape[3]; rem /= d.shape[3];
    const int i2 = rem % d.shape[2]; rem /= d.shape[2];
    const int i1 = rem % d.shape[1];
    const int i0 = rem / d.shape[1];
    const long so = (long)i0 * d.sstride[0] + (long)i1 * d.sstride[1] + (long)i2 * d.sstride[2] + (long)i3 * d.sstride[3];
    const long dof = (long)i0 * d.dstride[0] + (long)i1 * d.dstride[1] + (long)i2 * d.dstride[2] + (long)i3 * d.dstride[3];
    float acc;
    if (d.src2) {
        const long s2 = (long)i0 * d.s2stride[0] + (long)i1 * d.s2stride[1] + (long)i2 * d.s2stride[2] + (long)i3 * d.s2stride[3];
        acc = d.src[so] * d.src2[s2];
        for (int r = 1; r < d.nred; ++r) acc += d.src[so + (long)r * d.rstride] * d.src2[s2 + (long)r * d.r2stride];   // fixed order
    } else {
        acc = d.src[so];
        for (int r = 1; r < d.nred; ++r) acc += d.src[so + (long)r * d.rstride];
    }
    d.dst[dof] = acc;
}

}  // namespace

extern "C" int ksr_gather_batch(const KsrGatherDesc* descs, const KsrGatherBlock* blocks, int nblocks, hipStream_t stream) {
    KSR_CHECK_ARG(descs && blocks && nblocks > 0, "gather_batch: bad argument");
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, descs, blocks);
    KSR_LAUNCH_CHECK("gather_batch");
    return KSR_OK;
}
