// Shared helpers for the gfx950 kernels of libfsc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fsc_hip.h"

namespace fsc {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(fsc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define FSC_CHECK_ARG(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            fsc::set_error(__VA_ARGS__);    \
            return 22; /* EINVAL */         \
        }                                   \
    } while (0)

#define FSC_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            fsc::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return (int)e__;                                                          \
        }                                                                             \
    } while (0)

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Sum over a block of NW waves; result valid in every thread.  `scratch` holds NW values.
template <typename T, int NW>
__device__ __forceinline__ T block_sum(T v, T* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    T r = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r += scratch[i];
    return r;
}

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

}  // namespace fsc
