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


// ---- three SCALED fp16 limbs (arith 10, "f16x6"): with the power-of-two scale s of the two-limb format, x * s = h + m + l with
// h = rne_f16(x s), m = rne_f16(x s - h), l = rne_f16(x s - h - m): 11 + 11 + 11 significand bits -- the 24 bits of x exactly
// while the last bit of x s is >= 2^-24 (the fp16 subnormal step), i.e. |x| >= 2^-16 max|tensor|; smaller elements keep an
// ABSOLUTE error <= 2^-25 / s <= 2^-39 max|tensor| (an fp32 accumulation of such a tensor rounds at 2^-24 of its partial sums).
// Six of the nine limb products (all but m*l, l*m, l*l <= 2^-32 |a b|) form the product of two fp32 operands to 2^-32: as exact,
// for an fp32 accumulator, as all nine of the unscaled bf16 limbs, at 6 / 9 of the MFMA work.
__host__ __device__ inline bool is_f3(int arith) { return arith == 10; }
__host__ __device__ inline bool is_l3(int arith) { return is_bf3(arith) || is_f3(arith); }
// L16 format codes of the `limbs` arguments of include/fsc_hip.h
constexpr int kFmtF16x2 = 2, kFmtBf16x3 = 3, kFmtF16x3 = 4;
__host__ __device__ constexpr int fmt_limbs(int fmt) { return fmt == kFmtF16x2 ? 2 : 3; }
__host__ __device__ constexpr bool fmt_scaled(int fmt) { return fmt != kFmtBf16x3; }
// (eight mixed-precision FMAs per pair of values: h = rne16(x s), m = rne16(x s - h) in one rounding each, r = x s - h exactly in
// fp32, l = rne16(r - m))
__device__ __forceinline__ void split3s_pair(float x0, float x1, float s, unsigned& h, unsigned& m, unsigned& l) {
    unsigned hp, mp, lp;
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hp) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hp) : "v"(x1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(mp) : "v"(x0), "v"(s), "v"(hp));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(mp) : "v"(x1), "v"(s), "v"(hp));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(s), "v"(hp));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(s), "v"(hp));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lp) : "v"(r0), "v"(mp));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lp) : "v"(r1), "v"(mp));
    h = hp;
    m = mp;
    l = lp;
}
__device__ __forceinline__ void split8_f3(const float (&v)[8], float s, uint4& h, uint4& m, uint4& l) {
    split3s_pair(v[0], v[1], s, h.x, m.x, l.x);
    split3s_pair(v[2], v[3], s, h.y, m.y, l.y);
    split3s_pair(v[4], v[5], s, h.z, m.z, l.z);
    split3s_pair(v[6], v[7], s, h.w, m.w, l.w);
}

}  // namespace l16

// conv_l3.hip: the three-limb (bf16: arith 9 / 8 / 6; scaled fp16: arith 10) kernels behind the fsc_conv_l16_* / fsc_l16_* entry points
namespace fsc {
namespace l3 {
size_t tensor_bytes(int n, int c, long hw);
// (amax / in_amax: the FSC_AMAX_FLOATS buffer of the scaled fp16 format, NULL for bf16 limbs)
int pack(const float* x, int n, int c, long hw, const float* amax, void* out, hipStream_t st);
int unpack(const void* in, int n, int c, long hw, const float* amax, float* x, hipStream_t st);
int supported(const fsc_conv_desc* d, int dgrad);
int pool_supported(const fsc_conv_desc* d);
size_t packed_floats(const fsc_conv_desc* d, int dgrad);
int pack_weights_pair(const fsc_conv_desc* d, const float* weight, float* packed_fwd, float* packed_dgrad, hipStream_t st);
int pack_weights_multi(int count, const fsc_conv_desc* descs, const float* const* weights, float* const* packed_fwd,
                       float* const* packed_dgrad, hipStream_t st);
int fwd(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias, int dgrad,
        int accumulate, float* out, const float* stat_pivot, void* stat_rec, hipStream_t st);
int pool_fwd(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias, float* pooled,
             uint8_t* idx, const float* stat_pivot, void* stat_rec, hipStream_t st);
int fwd_act(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias,
            const float* scale, const float* shift, const float* alpha, void* out_l16, const float* out_amax, float* seen_max,
            hipStream_t st);
int pool_fwd_act(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias,
                 const float* scale, const float* shift, const float* alpha, float* out, void* out_l16, const float* out_amax,
                 float* seen_max, hipStream_t st);
int stats_layout(const fsc_conv_desc* d, int pool, int* out4);
int plan_describe(const fsc_conv_desc* d, int dgrad, char* buf, size_t buf_len);
int last_clock(double* shader_mhz);
}  // namespace l3
}  // namespace fsc
