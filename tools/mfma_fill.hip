// How many VALU / SALU / LDS fillers hide behind one v_mfma_f32_16x16x32_bf16 (development microbenchmark).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F, int KIND>
__global__ void k(float* out, int iters) {
    __shared__ float lds[4096];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f); b[e] = (__bf16)(e * 0.01f); }
    float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    int sacc = iters;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(m * F + f) & 7]));
                if (KIND == 1) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc) : : "scc");
                if (KIND == 5) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(m * F + f) & 7])); asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc) : : "scc"); }
                if (KIND == 2) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"((threadIdx.x & 63) * 4)); asm volatile("s_waitcnt lgkmcnt(8)"); }
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(double*)&v[((m * F + f) & 3) * 2]));
                if (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(v[(m * F + f) & 7]));
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int e = 0; e < 8; ++e) s += v[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + sacc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (iters * 4);
}

template <int F, int KIND>
void run(const char* name, int threads) {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    k<F, KIND><<<256, threads>>>(d, 2000);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<F, KIND><<<256, threads>>>(d, 20000);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
    printf("%-8s fillers/MFMA %d  waves/SIMD %d : %.1f clock64 ticks per MFMA per wave, %.2f ns per MFMA per wave\n", name, F, threads / 256, cyc, ms * 1e6 / (20000 * 4));
    hipFree(d);
}

int main() {
    for (int th : {256, 512}) {
        run<0, 0>("none", th); run<1, 0>("valu", th); run<2, 0>("valu", th); run<3, 0>("valu", th); run<4, 0>("valu", th); run<6, 0>("valu", th); run<8, 0>("valu", th);
        run<2, 1>("salu", th); run<4, 1>("salu", th); run<8, 1>("salu", th);
        run<2, 5>("v+s", th); run<3, 5>("v+s", th);
        run<1, 2>("lds", th); run<2, 2>("lds", th);
        run<2, 3>("pkadd", th); run<2, 4>("cvtpk", th); run<4, 4>("cvtpk", th);
    }
    return 0;
}
