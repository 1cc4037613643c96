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

// ---- three exact bf16 limbs (arith 9 / 8 / 6): x = h + m + l with h = rne_bf16(x), m = rne_bf16(x - h), l = x - h - m (8 + 8 + 8
// significand bits; bf16 has the fp32 exponent range, so there is no scale and no declared maximum).
__host__ __device__ inline bool is_bf3(int arith) { return arith == 9 || arith == 8 || arith == 6; }
typedef float f32x2_l16 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_l16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {      // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_l16){lo, hi}, bf16x2_l16));
}
__device__ __forceinline__ float fsub1(float a, float b) {                 // one v_sub_f32 (hipcc pairs them into v_pk_add_f32 otherwise)
    float r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned hp = cvt_pk_bf16(x0, x1);
    const float r0 = fsub1(x0, __uint_as_float(hp << 16)), r1 = fsub1(x1, __uint_as_float(hp & 0xffff0000u));
    const unsigned mp = cvt_pk_bf16(r0, r1);
    const float s0 = fsub1(r0, __uint_as_float(mp << 16)), s1 = fsub1(r1, __uint_as_float(mp & 0xffff0000u));
    h = hp;
    m = mp;
    l = cvt_pk_bf16(s0, s1);
}
// eight channels of one position -> the three 16-byte limb vectors
__device__ __forceinline__ void split8_bf3(const float (&v)[8], uint4& h, uint4& m, uint4& l) {
    split3_pair(v[0], v[1], h.x, m.x, l.x);
    split3_pair(v[2], v[3], h.y, m.y, l.y);
    split3_pair(v[4], v[5], h.z, m.z, l.z);
    split3_pair(v[6], v[7], h.w, m.w, l.w);
}

}  // namespace l16

// conv_l3.hip: the bf16-limb (three limbs, arith 9 / 8 / 6) kernels behind the fsc_conv_l16_* / fsc_l16_* entry points
namespace fsc {
namespace l3 {
size_t tensor_bytes(int n, int c, long hw);
int pack(const float* x, int n, int c, long hw, void* out, hipStream_t st);
int unpack(const void* in, int n, int c, long hw, float* x, hipStream_t st);
int supported(const fsc_conv_desc* d, int dgrad);
int pool_supported(const fsc_conv_desc* d);
size_t packed_floats(const fsc_conv_desc* d, int dgrad);
int pack_weights_pair(const fsc_conv_desc* d, const float* weight, float* packed_fwd, float* packed_dgrad, hipStream_t st);
int pack_weights_multi(int count, const fsc_conv_desc* descs, const float* const* weights, float* const* packed_fwd,
                       float* const* packed_dgrad, hipStream_t st);
int fwd(const fsc_conv_desc* d, const void* in_l16, const float* packed, const float* bias, int dgrad, int accumulate, float* out,
        const float* stat_pivot, void* stat_rec, hipStream_t st);
int pool_fwd(const fsc_conv_desc* d, const void* in_l16, const float* packed, const float* bias, float* pooled, uint8_t* idx,
             const float* stat_pivot, void* stat_rec, hipStream_t st);
int stats_layout(const fsc_conv_desc* d, int pool, int* out4);
int plan_describe(const fsc_conv_desc* d, int dgrad, char* buf, size_t buf_len);
int last_clock(double* shader_mhz);
}  // namespace l3
}  // namespace fsc
