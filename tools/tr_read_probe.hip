// Probe of the gfx950 transposed LDS read (ds_read_b64_tr_b16) used by the bf16-storage weight-gradient kernel:
// prints which 16-bit elements each lane receives from a [64][72] tile holding value 100*row + col.
// Expected: lane i of every 16-lane group g receives column i of rows 4g..4g+3 (element q = row 4g+q).
//   hipcc -O2 --offload-arch=gfx950 tools/tr_read_probe.hip -o /tmp/tr_read_probe && /tmp/tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(const short* in, short* out) {
    __shared__ short lds[64 * 72];
    for (int i = threadIdx.x; i < 64 * 72; i += 64) lds[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x, grp = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
    // lane 4q+p of a group supplies the address of row q, columns 4p..4p+3 of the group's 4 x 16 block
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + (4 * grp + q) * 72 + 4 * p));
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    static short h[64 * 72], r[256];
    short *d, *o;
    for (int i = 0; i < 64 * 72; ++i) h[i] = (short)((i / 72) * 100 + (i % 72));
    if (hipMalloc(&d, sizeof(h)) != hipSuccess || hipMalloc(&o, sizeof(r)) != hipSuccess) return 1;
    if (hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    if (hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) bad += r[4 * l + e] != (short)((4 * (l >> 4) + e) * 100 + (l & 15));
    printf("ds_read_b64_tr_b16 layout %s\n", bad ? "DIFFERS from the expected one" : "as expected");
    return bad != 0;
}
