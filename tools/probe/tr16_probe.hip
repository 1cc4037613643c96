// Probe of ds_read_b64_tr_b16 (gfx950): which LDS halfwords does lane l receive when every lane passes its own address?
// Prints, for two address patterns, the 4 halfword indices each lane got.  (development tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s;

__global__ void probe(int* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int addr_hw;                         // halfword index this lane passes
    if (mode == 0) addr_hw = l * 4;      // consecutive 8-byte pieces
    else addr_hw = (l & 15) * 64 + (l >> 4) * 4;   // lane i of a 16-group -> row i (64 halfwords apart), group g -> col block g
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(lds + addr_hw));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

int main() {
    int* d;
    hipMalloc(&d, 256 * sizeof(int));
    int h[256];
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
