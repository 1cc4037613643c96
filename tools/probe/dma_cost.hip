// What one 1 KiB staging piece costs a wave that is otherwise issuing MFMAs (development microbenchmark, gfx950):
// LDS-DMA (global_load_lds_dwordx4) against a register-staged copy (global_load_dwordx4 + ds_write_b128 one trip later),
// for contiguous and row-gathered sources, with and without LDS fragment reads beside them.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/dma_cost.hip -o tools/probe/dma_cost && tools/probe/dma_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m) : "memory", "m0");
}

// KIND 0: no copy; 1: LDS-DMA; 2: register staged.  PAT 0: 1 KiB contiguous; 1: 8 rows x 128 B (pitch 3440 B, 16-byte aligned only);
// 2: 2 rows x 512 B.  NM MFMAs and NR ds_read_b128 per trip and wave; PIECES copies per trip and wave.
template <int KIND, int PAT, int NM, int NR, int PIECES>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t src_bytes, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    u32x4 a = {1u, 2u, 3u, 4u}, b = {5u, 6u, 7u, 8u};
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    long off_lane = PAT == 0 ? lane * 16 : PAT == 1 ? (lane >> 3) * 3440 + (lane & 7) * 16 : (lane >> 5) * 3440 + (lane & 31) * 16;
    char* stage = lds + 65536 + wid * 8192;
    u32x4 pend[PIECES];
    for (int p = 0; p < PIECES; ++p) pend[p] = (u32x4){0u, 0u, 0u, 0u};
    size_t pos = ((size_t)blockIdx.x * 8 + wid) * 65536;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            pos += 8 * 3440 + 48;                             // a fresh, 16-byte aligned place every time
            if (pos + 65536 > src_bytes) pos -= src_bytes - 65536;
            const char* s = src + (pos & ~(size_t)15) + off_lane;
            if (KIND == 1) glds16(s, stage + (p & 3) * 1024);
            if (KIND == 2) {
                *reinterpret_cast<u32x4*>(stage + (p & 3) * 1024 + lane * 16) = pend[p];      // last trip's piece
                pend[p] = *reinterpret_cast<const u32x4*>(s);
            }
        }
        u32x4 r[NR > 0 ? NR : 1];
#pragma unroll
        for (int q = 0; q < NR; ++q) r[q] = *reinterpret_cast<const u32x4*>(lds + ((q * 64 + lane + it) & 4095) * 16);
#pragma unroll
        for (int m = 0; m < NM; ++m)
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NR; ++q) a ^= r[q];
        if (KIND == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    for (int p = 0; p < PIECES; ++p) s += pend[p][0];
    s += reinterpret_cast<float*>(stage)[lane];
    out[blockIdx.x * 512 + threadIdx.x] = s + a[0];
}

template <int KIND, int PAT, int NM, int NR, int PIECES>
void run(const char* src, size_t bytes, float* out) {
    auto kern = k<KIND, PAT, NM, NR, PIECES>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    const int iters = 4000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536 + 65536, 0, src, bytes, out, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 65536 + 65536, 0, src, bytes, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters;
    static double base[8][8];
    if (KIND == 0) base[NM / 8][NR / 4] = ns;
    const char* kn[3] = {"none", "lds-dma", "reg+ds_write"};
    const char* pn[3] = {"1 KiB contiguous", "8 x 128 B rows", "2 x 512 B rows"};
    printf("%-13s %-17s %2d MFMA %2d ds_read_b128 %d pieces per trip and wave: %7.1f ns per trip", kn[KIND], KIND ? pn[PAT] : "-", NM, NR,
           KIND ? PIECES : 0, ns);
    if (KIND) printf("  (+%.1f ns = %.0f cycles @2.1 GHz per piece and CU)", ns - base[NM / 8][NR / 4], (ns - base[NM / 8][NR / 4]) * 2.1 / (8 * PIECES));
    printf("\n");
    fflush(stdout);
}

int main() {
    const size_t bytes = 256u << 20;
    char* src;
    float* out;
    hipMalloc(&src, bytes);
    hipMemset(src, 1, bytes);
    hipMalloc(&out, 256 * 512 * 4);
#define ROW(NM, NR)                                                                                                   \
    run<0, 0, NM, NR, 1>(src, bytes, out);                                                                           \
    run<1, 0, NM, NR, 1>(src, bytes, out); run<1, 1, NM, NR, 1>(src, bytes, out); run<1, 2, NM, NR, 1>(src, bytes, out); \
    run<2, 0, NM, NR, 1>(src, bytes, out); run<2, 1, NM, NR, 1>(src, bytes, out); run<2, 2, NM, NR, 1>(src, bytes, out); \
    run<1, 1, NM, NR, 2>(src, bytes, out); run<2, 1, NM, NR, 2>(src, bytes, out);
    ROW(32, 0)
    ROW(32, 8)
    ROW(16, 8)
    return 0;
}
