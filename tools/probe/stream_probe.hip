// Bare MFMA + LDS-fragment-read streams of the L16 forward kernel's step, for the two fp16 MFMA shapes and several wave tiles
// (development microbenchmark: no copies, no barriers, results meaningless).  One workgroup of 8 waves per CU, like the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe tools/probe/stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// SHAPE 16: wave tile = TM x TN tiles of 16 x 16, K = 32 per step: A frags TM x 2 limbs, B frags TN x 2 limbs, TM * TN * 3 MFMAs
// SHAPE 32: wave tile = TM x TN tiles of 32 x 32, K = 32 per step = 2 k-halves: A TM x 2 x 2, B TN x 2 x 2, TM * TN * 2 * 3 MFMAs
template <int SHAPE, int TM, int TN, bool LDS = true>
__global__ __launch_bounds__(512) void stream(float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const unsigned long long ck0 = __builtin_readcyclecounter(), cr0 = __builtin_amdgcn_s_memrealtime();
    for (int i = threadIdx.x; i < 32 * 1024 / 4 * 4; i += 512) smem[i] = (float)(i & 1023) * 1e-6f;
    __syncthreads();
    const u32x4* wl = reinterpret_cast<const u32x4*>(smem) + lane;           // A: 64 KB ring of 1 KB fragments
    const u32x4* bl = reinterpret_cast<const u32x4*>(smem) + 4096 + lane;    // B: behind it
    constexpr int KH = SHAPE == 32 ? 2 : 1;
    constexpr int NA = TM * KH * 2, NB = TN * KH * 2;
    u32x4 a[2][NA], b[2][NB];
    f32x4 acc16[SHAPE == 16 ? TM * TN : 1];
    f32x16 acc32[SHAPE == 32 ? TM * TN : 1];
    for (auto& v : acc16) v = (f32x4){0, 0, 0, 0};
    for (auto& v : acc32) for (int e = 0; e < 16; ++e) v[e] = 0;
    auto rd = [&](int s, u32x4* ad, u32x4* bd) {
        const int slot = (s & 3) * 16 * 64;                                   // 16 KB per slot
#pragma unroll
        for (int i = 0; i < NA; ++i) ad[i] = wl[slot + (i & 15) * 64];
#pragma unroll
        for (int i = 0; i < NB; ++i) bd[i] = bl[(s & 1) * 512 + (i & 7) * 64];
    };
    rd(0, a[0], b[0]);
    if (!LDS) rd(1, a[1], b[1]);
    auto body = [&](int s, u32x4* ac, u32x4* bc, u32x4* an, u32x4* bn) {
        if (LDS) rd(s + 1, an, bn);
        constexpr int kLa[3] = {1, 0, 0}, kLb[3] = {0, 1, 0};
        if constexpr (SHAPE == 16) {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc16[i * TN + j] = mfma16(ac[i * 2 + kLa[g]], bc[j * 2 + kLb[g]], acc16[i * TN + j]);
        } else {
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc32[i * TN + j] = mfma32(ac[(i * 2 + h) * 2 + kLa[g]], bc[(j * 2 + h) * 2 + kLb[g]], acc32[i * TN + j]);
        }
        constexpr int NM = TM * TN * 3 * KH, NR = NA + NB;
#pragma unroll
        for (int k = 0; k < NM; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (LDS && k < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int s = 0; s < steps; s += 2) {
        body(s, a[0], b[0], a[1], b[1]);
        body(s + 1, a[1], b[1], a[0], b[0]);
    }
    float r = 0;
    for (auto& v : acc16) r += v[0] + v[1] + v[2] + v[3];
    for (auto& v : acc32) for (int e = 0; e < 16; ++e) r += v[e];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[256 * 512] = (float)(__builtin_readcyclecounter() - ck0);
        out[256 * 512 + 1] = (float)(__builtin_amdgcn_s_memrealtime() - cr0);
    }
}

template <int SHAPE, int TM, int TN, bool LDS = true>
void run(const char* name) {
    float* d;
    hipMalloc(&d, 256 * 512 * 4 + 64);
    auto kern = stream<SHAPE, TM, TN, LDS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const int steps = 40000;      // ~30 ms per launch: long enough for the clock to settle
    kern<<<256, 512, 96 * 1024>>>(d, 200);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kern<<<256, 512, 96 * 1024>>>(d, steps);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double tile = SHAPE * SHAPE;
    const double flop = 2.0 * tile * 32 * TM * TN * 3 * (double)steps * 8 * 256;   // per step a wave does TM*TN tiles x K 32 x 3 products
    const int nm = TM * TN * 3 * (SHAPE == 32 ? 2 : 1), nr = (TM + TN) * 2 * (SHAPE == 32 ? 2 : 1);
    float ck[2];
    hipMemcpy(ck, d + 256 * 512, 8, hipMemcpyDeviceToHost);
    const double mhz = 100.0 * ck[0] / ck[1];
    printf("%-10s tile %3d x %3d  MFMA/step %3d  LDS reads/step %2d : %7.3f ms  %7.1f TF executed (%.2f of 2500)  clock %4.0f MHz  %.1f cycles per MFMA and SIMD\n",
           name, SHAPE * TM, SHAPE * TN, nm, nr, best, flop / best / 1e9, flop / best / 1e9 / 2500.0, mhz, ck[0] / ((double)steps * nm * 2));
    hipFree(d);
}

int main() {
    run<16, 8, 2>("16: 8x2");
    run<16, 7, 2>("16: 7x2");
    run<32, 4, 1>("32: 4x1");
    run<16, 8, 2, false>("16: 8x2 reg");
    run<16, 4, 4, false>("16: 4x4 reg");
    run<32, 4, 1, false>("32: 4x1 reg");
    run<32, 2, 2, false>("32: 2x2 reg");
    run<16, 8, 2>("16: 8x2");
    run<32, 2, 2>("32: 2x2");
    run<16, 4, 4>("16: 4x4");
    return 0;
}
