// Device helpers of the L16 activation format (include/fsc_hip.h "pre-split activations"): scaled two-limb fp16
// split of fp32 values, the power-of-two scale derived from a tensor's declared maximum.
#pragma once
#include "common.h"

namespace l16 {

// fp16 two-limb split with a power-of-two scale: h = rne16(x * s), l = rne16(x * s - h), one v_fma_mix each
// (the mixed-precision FMA reads the fp32 source and the fp16 half directly and rounds once).  Returns the pair
// (x0 in the low half, x1 in the high half) of each limb.  |x * s - h - l| <= 2^-24 |x * s|.
__device__ __forceinline__ void split2_pair(float x0, float x1, float s, unsigned& h, unsigned& l) {
    unsigned hp, lp;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hp) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hp) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lp) : "v"(x0), "v"(s), "v"(hp));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lp) : "v"(x1), "v"(s), "v"(hp));
    h = hp;
    l = lp;
}
// Power-of-two scale that brings a tensor of magnitude `amax` to [2^14, 2^15) (fp16 overflows at 65504), as the
// exponent field of the scale; the inverse has field 254 - f.  Tensors below 2^-111 or above 2^125 are clamped.
__device__ __forceinline__ int scale_field(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    int f = 268 - e;
    f = f < 2 ? 2 : f > 252 ? 252 : f;
    return f;
}
__device__ __forceinline__ float field_to_float(int f) { return __uint_as_float((unsigned)f << 23); }
// Inverse of the scale.  A declared maximum of +Inf leaves no usable scale for the finite elements: the inverse is
// NaN then, so every output of the consumer is non-finite instead of silently losing the finite part.
__device__ __forceinline__ float inv_scale(int f, float amax) {
    return ((__float_as_uint(amax) >> 23) & 0xffu) == 0xffu ? __uint_as_float(0x7fc00000u) : field_to_float(254 - f);
}

// eight channels of one position -> the two 16-byte limb vectors of the L16 layout
__device__ __forceinline__ void split8(const float (&v)[8], float s, uint4& hi, uint4& lo) {
    split2_pair(v[0], v[1], s, hi.x, lo.x);
    split2_pair(v[2], v[3], s, hi.y, lo.y);
    split2_pair(v[4], v[5], s, hi.z, lo.z);
    split2_pair(v[6], v[7], s, hi.w, lo.w);
}

// An L16 tensor's declared maximum travels as an FSC_AMAX_FLOATS buffer whose maximum is the value.  Producers that
// know the value up front store it with this (every thread of a block of >= 64 threads calls; one block does it).
__device__ __forceinline__ void store_amax(float* amax, float value) {
    for (int i = threadIdx.x; i < fsc::kAmaxFloats; i += blockDim.x) amax[i] = i == 0 ? value : 0.f;
}

}  // namespace l16
