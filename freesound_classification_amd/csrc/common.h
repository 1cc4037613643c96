// Shared helpers for the gfx950 kernels of libfsc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fsc_hip.h"

namespace fsc {

void set_error(const char* fmt, ...);
int l16_fwd_clock(double* shader_mhz);      // conv_l16.hip: shader clock of the last forward / dgrad launch (fsc_conv_l16_last_clock)

inline hipStream_t as_stream(fsc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Development switches (FSC_* environment variables), read ONCE when the first entry point needs them -- never on the per-call path.
struct EnvFlags {
    bool no_l16, no_l16_pool, no_l16_wgrad, l16_no_xcd, l16_vec1, dbg_noksplit, frontend_generic, fe_block_sync;
    int l16_cot, l16_pt, l16w_tw;        // 0 = not forced
};
const EnvFlags& env();                   // misc.hip


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

// Largest |value| a launch writes, for the operand scaling of the split-fp16 conv kernels.  The buffer has
// kAmaxFloats non-negative floats (all 0 before the launch) and the result is the maximum over ALL of them:
// workgroups publish into kAmaxSlots slots on separate 128-byte lines (a single address serialises ~2 ns per
// access in L2: 100 k waves cost more than the streaming pass they belong to), one access per workgroup, and the
// atomic is skipped when the slot already holds a larger value.  Every thread of the block must call.
constexpr int kAmaxSlots = 16, kAmaxStride = 32, kAmaxFloats = kAmaxSlots * kAmaxStride;
__device__ __forceinline__ void publish_amax(float* out, float m) {
    __shared__ float amax_s[16];
    m = wave_max(m);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) amax_s[wid] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) m = fmaxf(m, amax_s[i]);
        float* slot = out + ((blockIdx.x + blockIdx.y * 7) % kAmaxSlots) * kAmaxStride;
        if (m > *reinterpret_cast<volatile float*>(slot)) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(m));
    }
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

// fsc_conv_desc.arith == FSC_ARITH_DEFAULT means the PROCESS default (fsc_conv_default_arith(): f16x6 unless FSC_CONV_ARITH says
// otherwise) at EVERY entry point, the pre-split (L16) ones included: the descriptor is resolved once, at the boundary.
#define FSC_RESOLVE_DESC(d)                                                                      \
    fsc_conv_desc d##_resolved_;                                                                 \
    if ((d) != nullptr && (d)->arith == FSC_ARITH_DEFAULT) {                                     \
        d##_resolved_ = *(d);                                                                    \
        d##_resolved_.arith = fsc_conv_default_arith();                                          \
        (d) = &d##_resolved_;                                                                    \
    }
