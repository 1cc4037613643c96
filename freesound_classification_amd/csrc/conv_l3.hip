// Convolutions at full fp32 precision on PRE-SPLIT bf16 limbs ("L16 tensors with three limbs", arith 9 / 8 / 6) for gfx950.
//
// An fp32 value splits EXACTLY into three bf16 limbs, x = h + m + l (8 + 8 + 8 significand bits, bf16 has the fp32 exponent
// range: no scaling, no operand maximum), and every limb product is exact in the fp32 accumulator of
// v_mfma_f32_16x16x32_bf16.  With all nine limb products (arith 9) a convolution product is the exact product of the two fp32
// operands -- what an IEEE fp32 multiply-add chain forms before its rounding -- accumulated in fp32: arithmetic no narrower than
// the reference's nn.Conv2d on fp32 tensors (networks/classifiers.py:526-531, 77-81), at 2500 / 9 = 278 TF against the 157 TF
// of the fp32 matrix / vector pipes.  (arith 8 drops l*l <= 2^-32 |a*b|; arith 6 also m*l + l*m <= 2^-23 |a*b|.)
//
//     3-limb L16 tensor of a logical (N, C, H, W) fp32 tensor, OCT = ceil(C / 8):
//         bf16[N][OCT][3 limbs][H * W][8 channels]          (16 bytes per (limb, position); pad channels are 0)
//
// conv.hip's conv_fwd_x3_kernel<.., 9> reads fp32 activations and splits them beside the MFMAs, once per tap; the producers of
// norm_act.hip write the limbs once.  With nine MFMAs per pair of fragments the kernel is bound by the matrix pipe, not by its
// operand streams, and is organised around that:
//   * 8 waves = 2 channel groups x 4 pixel groups; a wave owns CT channel tiles x PTW pixel tiles (16 x 16 each) and keeps
//     CT * PTW accumulators: per MFMA step (one tap x 32 channels) it issues CT * PTW * 9 MFMAs for 3 (CT + PTW) fragment loads;
//   * the WEIGHT fragments never touch LDS: a wave loads the 3 limb fragments of its own channel tiles straight from global
//     memory (L2-resident: one channel block's fragments are re-read by every worker) into the registers of the MFMA A operand,
//     the fragments of tile i for step S + 1 right behind tile i's MFMAs of step S -- no ring, no per-step barrier;
//   * the halo'd input box of a 32-channel chunk (12 (octet, limb) planes) is staged by 16-byte LDS-DMA, two stages; the workgroup
//     synchronises ONCE PER CHUNK (nine steps of a 3x3 layer);
//   * B fragments are single ds_read_b128s; those of step S + 1 are read behind the MFMAs of the wave's last channel tile of
//     step S, low limb first (the products are ordered by B limb so that a limb's registers are free early).
// Epilogue (16-byte stores through a per-wave LDS transpose), the fused 2x2 max-pool and the BatchNorm statistics of the output
// are those of conv_l16.hip.
#include "common.h"
#include "l16.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int kWaves = 8;
constexpr int kChunk = 32;           // channels per K chunk = 4 octets
constexpr int kNptMax = 6;           // input DMA instructions per (octet, limb) plane: plane <= 384 positions
constexpr int kScr = 20;             // epilogue scratch row stride (floats)
constexpr int kUnits = 12;           // (octet, limb) planes of a staged chunk
constexpr int kNstg = 2;

struct L3Geom {
    int n, cin, cout, h, w;
    long hw;
    int oct_in;
    int nb, th, tw;
    int tiles_n, tiles_h, tiles_w;
    int rows, cols;
    int plane;                // positions per staged (octet, limb) plane, multiple of 8
    int npos, npix, npt;
    int nfull, tail_oct, tail_steps, steps;
    int coblk;                // channel blocks
    int cot;                  // channel tiles per block (as packed)
    int tiles_total;          // channel tiles of the layer
    long img_stride;          // 16-byte units between images of the input: oct_in * 3 * hw
    int xcd;
    int spat;                 // item index v -> tile of XCD (v mod 8)'s contiguous eighth (see tile_of)
};

__device__ __attribute__((aligned(16))) float g_zero16_3[4] = {0.f, 0.f, 0.f, 0.f};
__device__ unsigned long long g_l3_clock[2];

// Development (-DFSC_L16_PROFILE): shader-clock stamps, summed per wave of workgroup 0 into g_l3_prof[wave][phase]; read and cleared
// by fsc_debug_l3_prof.  Phases: 0 chunk hand-over (wait, barrier, copy issue), 1 the MFMA steps, 2 epilogue, 3 first fragments of an
// item, 4 steps counted, 5 whole kernel, 6 the same in 100 MHz reference ticks, 7 the wait + barrier part of the hand-over (0 = its copy issue).
#ifdef FSC_L16_PROFILE
__device__ unsigned long long g_l3_prof[8][8];
#define P3_DECL unsigned long long pf_t = __builtin_readcyclecounter(), pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long pf_k0 = pf_t, pf_r0 = __builtin_amdgcn_s_memrealtime();
#define P3_ADD(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pf_acc[i] += n_ - pf_t; pf_t = n_; } while (0)
#else
#define P3_DECL
#define P3_ADD(i)
#endif

__device__ __forceinline__ int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {      // see conv_l16.hip
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m) : "memory", "m0");
}
__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma_l(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return mfma_bf(a, b, c);
}
constexpr int kWmaxBlocks3 = 64;     // partial maxima of a weight tensor (scaled fp16 limbs)
// maximum over an FSC_AMAX_FLOATS buffer, by one wave (uniform result)
__device__ __forceinline__ float wave_amax512(const float* __restrict__ amax) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < fsc::kAmaxFloats / 64; ++i) m = fmaxf(m, amax[(threadIdx.x & 63) + 64 * i]);
    return fsc::wave_max(m);
}
__device__ __forceinline__ float dpp_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor8(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));
}
__device__ __forceinline__ void raw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// limb products (A limb, B limb), 0 = h, 1 = m, 2 = l, ordered by B limb (low first) so that a B limb's registers are free for
// the next step's fragments as early as possible; within a group the small terms come first
template <int NPROD> struct Prods;
template <> struct Prods<9> { static constexpr int la[9] = {2, 1, 0, 2, 1, 0, 2, 1, 0}; static constexpr int lb[9] = {2, 2, 2, 1, 1, 1, 0, 0, 0}; };
template <> struct Prods<8> { static constexpr int la[8] = {1, 0, 2, 1, 0, 2, 1, 0}; static constexpr int lb[8] = {2, 2, 1, 1, 1, 0, 0, 0}; };
template <> struct Prods<6> { static constexpr int la[6] = {0, 1, 0, 2, 1, 0}; static constexpr int lb[6] = {2, 1, 1, 0, 0, 0}; };

// -------------------------------------------------------------------------------------------
// fp32 NCHW <-> three bf16 limbs (stand-alone producer / inverse: tests, and tensors no fused producer writes)
// (amax != nullptr: three SCALED fp16 limbs, l16.h)
__global__ __launch_bounds__(256) void l3_pack_kernel(const float* __restrict__ x, int n, int c, long hw, const float* __restrict__ amax,
                                                      uint4* __restrict__ out) {
    float sc = 0.f;
    if (amax != nullptr) sc = l16::field_to_float(l16::scale_field(wave_amax512(amax)));
    const int oct = (c + 7) / 8;
    const long total = (long)n * oct * hw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long p = idx % hw;
        const long no = idx / hw;
        const int o = (int)(no % oct);
        const long img = no / oct;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = o * 8 + e;
            v[e] = ch < c ? x[(img * c + ch) * hw + p] : 0.f;
        }
        uint4 h, m, l;
        if (amax != nullptr) l16::split8_f3(v, sc, h, m, l);
        else l16::split8_bf3(v, h, m, l);
        out[(no * 3) * hw + p] = h;
        out[(no * 3 + 1) * hw + p] = m;
        out[(no * 3 + 2) * hw + p] = l;
    }
}

__global__ __launch_bounds__(256) void l3_unpack_kernel(const uint4* __restrict__ in, int n, int c, long hw,
                                                        const float* __restrict__ amax, float* __restrict__ x) {
    float inv = 1.f;
    if (amax != nullptr) {
        const float mx = wave_amax512(amax);
        inv = l16::inv_scale(l16::scale_field(mx), mx);
    }
    const int oct = (c + 7) / 8;
    const long total = (long)n * oct * hw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long p = idx % hw;
        const long no = idx / hw;
        const int o = (int)(no % oct);
        const long img = no / oct;
        const uint4 h = in[(no * 3) * hw + p], m = in[(no * 3 + 1) * hw + p], l = in[(no * 3 + 2) * hw + p];
        const unsigned hh[4] = {h.x, h.y, h.z, h.w}, mm[4] = {m.x, m.y, m.z, m.w}, ll[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = o * 8 + e;
            if (ch >= c) continue;
            const int sh = (e & 1) * 16;
            if (amax != nullptr) {
                const float hv = (float)__builtin_bit_cast(_Float16, (unsigned short)(hh[e >> 1] >> sh)),
                            mv = (float)__builtin_bit_cast(_Float16, (unsigned short)(mm[e >> 1] >> sh)),
                            lv = (float)__builtin_bit_cast(_Float16, (unsigned short)(ll[e >> 1] >> sh));
                x[(img * c + ch) * hw + p] = (hv + (mv + lv)) * inv;  // (m + l is exact: 22 significant bits)
                continue;
            }
            const float hv = __uint_as_float((hh[e >> 1] >> sh) << 16), mv = __uint_as_float((mm[e >> 1] >> sh) << 16),
                        lv = __uint_as_float((ll[e >> 1] >> sh) << 16);
            x[(img * c + ch) * hw + p] = hv + (mv + lv);              // (m + l is exact: 16 significant bits)
        }
    }
}

// weight (c_out, c_in, kh, kw) -> A fragments: packed[co block][step][channel tile][limb][lane][8 bf16]; lane = (kq, m); its 8
// values are the channels of octet `oct` at tap `tap`, (tap, oct) = divmod(4 * step_in_chunk + kq, octets of the chunk) -- the
// step structure of conv_l16.hip, three exact bf16 limbs instead of two scaled fp16 ones (no weight maximum, one launch).
struct PackDir3 {
    uint4* packed;
    float* w_amax;           // scaled fp16 limbs: receives max |w| (behind the fragments; read by the conv kernel), else nullptr
    int part_home;           // the partial maxima of the pack live behind THIS direction's w_amax (else: those words are zeroed)
    int cot, co_blocks, nfull, tail_oct, steps, dgrad;
    long frags;              // (fragment, lane) pairs: co_blocks * steps * cot * 64
    int blocks;
};
__device__ __forceinline__ void pack3_block(const float* __restrict__ w, int c_out, int c_in, int taps, const PackDir3& dir, int blk,
                                            const float* __restrict__ wmax_part) {
    float sw = 0.f;
    if (dir.w_amax != nullptr) {                           // every block folds the partial maxima of l3_wmax_kernel
        float wm = 0.f;
        for (int i = 0; i < kWmaxBlocks3; ++i) wm = fmaxf(wm, wmax_part[i]);
        if (blk == 0) {                                    // (every word of the tail defined: packed buffers compare bit for bit)
            if (threadIdx.x == 0) dir.w_amax[0] = wm;
            if (threadIdx.x >= 1 && threadIdx.x < 4) dir.w_amax[threadIdx.x] = 0.f;
            if (!dir.part_home && threadIdx.x < kWmaxBlocks3) dir.w_amax[4 + threadIdx.x] = 0.f;
        }
        sw = l16::field_to_float(l16::scale_field(wm));
    }
    const int cot = dir.cot, steps = dir.steps, nfull = dir.nfull;
    for (long idx = (long)blk * blockDim.x + threadIdx.x; idx < dir.frags; idx += (long)dir.blocks * blockDim.x) {
        const int lane = (int)(idx & 63);
        long rest = idx >> 6;
        const int i = (int)(rest % cot); rest /= cot;
        const int S = (int)(rest % steps);
        const int cb = (int)(rest / steps);
        const int c = S < nfull * taps ? S / taps : nfull;
        const int s = S - c * taps;
        const int noct = c < nfull ? 4 : dir.tail_oct;
        const int gi = 4 * s + (lane >> 4);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (gi < taps * noct) {
            const int tap = gi / noct, oct = gi - tap * noct;
            const int k0 = c * kChunk + oct * 8, m = (cb * cot + i) * 16 + (lane & 15);
            const int tp = dir.dgrad ? taps - 1 - tap : tap;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int co = dir.dgrad ? k0 + e : m, ci = dir.dgrad ? m : k0 + e;
                if (co < c_out && ci < c_in) v[e] = w[((long)co * c_in + ci) * taps + tp];
            }
        }
        uint4 h, m, l;
        if (dir.w_amax != nullptr) l16::split8_f3(v, sw, h, m, l);
        else l16::split8_bf3(v, h, m, l);
        uint4* o = dir.packed + ((((long)cb * steps + S) * cot + i) * 3) * 64 + lane;
        o[0] = h;
        o[64] = m;
        o[128] = l;
    }
}
constexpr int kMultiPack3 = 12;
struct PackJob3 {
    const float* w;
    float* part;             // scaled fp16 limbs: kWmaxBlocks3 partial maxima (behind the first direction's fragments)
    long count;              // elements of w
    int c_out, c_in, taps;
    PackDir3 a, b;
    int first_block;
};
struct PackJobs3 { PackJob3 j[kMultiPack3]; int n; };
__global__ void l3_pack_w_multi_kernel(PackJobs3 jobs) {
    int job = 0;
#pragma unroll 1
    for (int k = 1; k < jobs.n; ++k)
        if ((int)blockIdx.x >= jobs.j[k].first_block) job = k;
    const PackJob3& pj = jobs.j[job];
    const int rel = (int)blockIdx.x - pj.first_block;
    const bool second = rel >= pj.a.blocks;
    pack3_block(pj.w, pj.c_out, pj.c_in, pj.taps, second ? pj.b : pj.a, second ? rel - pj.a.blocks : rel, pj.part);
}
// max |w| of up to kMultiPack3 weight tensors: kWmaxBlocks3 partial maxima each (no clear needed)
__global__ __launch_bounds__(256) void l3_wmax_multi_kernel(PackJobs3 jobs) {
    __shared__ float sm[4];
    const int job = blockIdx.x / kWmaxBlocks3, blk = blockIdx.x - job * kWmaxBlocks3;
    const PackJob3& pj = jobs.j[job];
    float m = 0.f;
    for (long i = (long)blk * 256 + threadIdx.x; i < pj.count; i += (long)kWmaxBlocks3 * 256) m = fmaxf(m, fabsf(pj.w[i]));
    m = fsc::wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) pj.part[blk] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// -------------------------------------------------------------------------------------------
// Forward / dgrad.  See the head of the file.  POOL / STATS as in conv_l16_fwd_kernel.
// F16: the limbs are scaled fp16 (arith 10, l16.h) -- v_mfma_f32_16x16x32_f16, and the accumulators leave through
// 1 / (input scale * weight scale) (powers of two: exact), from the declared maxima `in_amax` (FSC_AMAX_FLOATS) and `w_amax`.
// ACT16 (inference, fsc_conv_l16_fwd_act): the epilogue applies a per-channel affine (an eval-mode BatchNorm: y = z scale + shift)
// and PReLU to the convolution output z and writes y as the three-limb L16 tensor the NEXT convolution reads -- straight from the
// accumulators, 6 bytes per element instead of 4 written + 4 read + 6 written by the separate pass.  Same expressions as the
// two-pass route (fwd_l16_kernel of norm_act.hip): bit-identical limbs.  Scaled fp16 limbs take the DECLARED maximum `out_amax`
// (a calibrated bound); the largest |y| actually written is max-ed into `seen` for the caller's saturation check.
struct ActArgs3 {
    const float* scale;       // per output channel, or nullptr (identity)
    const float* shift;
    const float* alpha;       // PReLU slopes, or nullptr (no activation)
    uint2* out16;             // the L16 output in 8-byte units; nullptr = not an ACT16 launch
    const float* out_amax;    // scaled fp16 limbs: FSC_AMAX_FLOATS declared maximum of y
    unsigned* seen;           // receives max |y| as its fp32 bit pattern (atomicMax; zeroed by the caller), or nullptr
    int pooled;               // 1: the POOL variant -- y = act(maxpool2x2(conv)), written as fp32 to `out` (if non-null) AND as limbs
};
template <int KH, int KW, int CT, int PTW, int NPROD, bool F16 = false, bool POOL = false, bool STATS = false, bool ACT16 = false>
__global__ __launch_bounds__(kWaves * 64) void conv_l3_fwd_kernel(L3Geom g, const uint4* __restrict__ in,
                                                                   const uint4* __restrict__ packed,
                                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                                   int accumulate, uint8_t* __restrict__ pool_idx,
                                                                   const float* __restrict__ stat_pivot,
                                                                   float4* __restrict__ stat_rec,
                                                                   const float* __restrict__ in_amax,
                                                                   const float* __restrict__ w_amax, ActArgs3 act) {
    static_assert(!ACT16 || !STATS, "ACT16 (inference) has no statistics epilogue");
    constexpr int TAPS = KH * KW;
    constexpr int PADH = KH / 2, PADW = KW / 2;
    using P = Prods<NPROD>;
    float oscale = 1.f;
    if constexpr (F16) {
        const float ax = wave_amax512(in_amax), aw = *w_amax;
        oscale = l16::inv_scale(l16::scale_field(ax), ax) * l16::inv_scale(l16::scale_field(aw), aw);
        oscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, oscale)));
    }
    float s_out = 1.f, seen_mx = 0.f;
    if constexpr (F16 && ACT16) {
        s_out = l16::field_to_float(l16::scale_field(wave_amax512(act.out_amax)));
        s_out = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_out)));
    }

    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    uint4* const ibase = smem4;
    const int istage = kUnits * g.plane;                  // uint4 per stage
    float* const scratch = reinterpret_cast<float*>(ibase + kNstg * istage) + (threadIdx.x >> 6) * (16 * kScr);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kq = lane >> 4;
    const int cw = wid >> 2, pw = wid & 3;               // waves w and w + 4 share a SIMD: one of each channel group
    const unsigned long long ck0 = __builtin_readcyclecounter(), cr0 = __builtin_amdgcn_s_memrealtime();

    const float inv_per = 1.0f / (float)(g.rows * g.cols), inv_cols = 1.0f / (float)g.cols;
    const float inv_tw = 1.0f / (float)g.tw, inv_thw = 1.0f / (float)(g.th * g.tw);
    int pix_b[PTW];               // byte offset of this lane's pixel inside a staged plane
#pragma unroll
    for (int pt = 0; pt < PTW; ++pt) {
        const int t = pw * PTW + pt;
        const int p = t * 16 + lm;
        int pl = 0;
        if (p < g.npix) {
            if (POOL) {                                  // tile = a 2 x 8 block: (image, row pair, column octet)
                const int tpr = g.tw >> 3, tpi = (g.th >> 1) * tpr;
                const int b = t / tpi, rem = t - b * tpi;
                const int tr = rem / tpr, tc = rem - tr * tpr;
                pl = (b * g.rows + 2 * tr + (lm >> 3)) * g.cols + 8 * tc + (lm & 7);
            } else {
                const int per = g.th * g.tw;
                const int b = fdiv(p, inv_thw), rem = p - b * per;
                const int r = fdiv(rem, inv_tw), c = rem - r * g.tw;
                pl = (b * g.rows + r) * g.cols + c;
            }
        }
        pix_b[pt] = pl * 16;
    }
    const int ntiles = g.tiles_n * g.tiles_h * g.tiles_w;
    // items = (pixel tile, channel block); a worker keeps ONE channel block (conv_l16.hip: item order, XCD-aware form)
    int cb_w, t0;
    const int ts = (int)gridDim.x / g.coblk;
    if (g.xcd) {
        const int l = (int)blockIdx.x >> 3;
        cb_w = l % g.coblk;
        t0 = ((int)blockIdx.x & 7) + 8 * (l / g.coblk);
    } else {
        cb_w = (int)blockIdx.x % g.coblk;
        t0 = (int)blockIdx.x / g.coblk;
    }
#ifndef FSC_L3_XCDSPAT
#define FSC_L3_XCDSPAT 1
#endif
    // XCD-aware form, second half: the item index v (v mod 8 = the XCD of the worker that takes it) names tile base(v mod 8) + v / 8,
    // base(x) = the start of XCD x's CONTIGUOUS eighth of the tiles -- the 32 workers of an XCD work on 32 adjacent boxes at any
    // time, so the halo rows and columns two boxes share are fetched into that XCD's L2 once (with tile = v, an XCD's boxes were 8
    // apart and every halo came from HBM twice)
    const int tq = ntiles >> 3, tr = ntiles & 7;
    auto tile_of = [&](int v) -> int {
        if (!FSC_L3_XCDSPAT || !g.spat) return v;
        const int x = v & 7, j = v >> 3;
        return x * tq + (x < tr ? x : tr) + j;
    };
    const int nchunks = g.nfull + (g.tail_oct ? 1 : 0);
    // this wave's channel tiles inside the block: the live tiles of the block are halved between the two channel groups
    int blk_live = g.tiles_total - cb_w * g.cot;
    blk_live = blk_live > g.cot ? g.cot : blk_live;
    const int half0 = (blk_live + 1) >> 1;
    const int tile0 = cw ? half0 : 0;
    const int live = cw ? blk_live - half0 : half0;

    // source offset (16-byte units, relative to unit 0 of image 0) of staged positions; -1 = zeros.  Recomputed for every chunk
    // (once per nine steps) instead of kept in registers across the steps.
    auto plan_input = [&](int tile, int (&pos_off)[kNptMax]) {
        int t = tile;
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h; t /= g.tiles_h;
        const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
#pragma unroll
        for (int q = 0; q < kNptMax; ++q) {
            const int pos = q * 64 + lane_o;
            pos_off[q] = -1;
            if (q < g.npt && pos < g.npos) {
                const int per = g.rows * g.cols;
                const int b = fdiv(pos, inv_per), rem = pos - b * per;
                const int rr = fdiv(rem, inv_cols), cc = rem - rr * g.cols;
                const int gh = h0 + rr - PADH, gw = w0 + cc - PADW;
                if (n0 + b < g.n && gh >= 0 && gh < g.h && gw >= 0 && gw < g.w)
                    pos_off[q] = (int)((long)(n0 + b) * g.img_stride + (long)gh * g.w + gw);
            }
        }
    };

    f32x4 acc[CT][PTW];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int NST = STATS ? CT : 1;
    float st_s1[NST], st_s2[NST], st_mn[NST], st_mx[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) { st_s1[i] = 0.f; st_s2[i] = 0.f; st_mn[i] = INFINITY; st_mx[i] = -INFINITY; }
    auto stat_add = [&](int i, float y, float pv) {
        const float a = y - pv;
        st_s1[i] += a;
        st_s2[i] = fmaf(a, a, st_s2[i]);
        st_mn[i] = fminf(st_mn[i], y);
        st_mx[i] = fmaxf(st_mx[i], y);
    };

    // ---- input DMA: chunk c into `stage`; wave wid copies units wid and wid + 8 (unit = octet * 3 + limb)
    const uint4* const zero = reinterpret_cast<const uint4*>(g_zero16_3);
#ifndef FSC_L3_COPYGRP
#define FSC_L3_COPYGRP 1
#endif
    // FSC_L3_COPYGRP: the copies of a chunk are issued by the waves of channel group 1 alone (three planes each; the group with the
    // smaller half of an odd block).  Its partner on the SIMD (same pixel group, channel group 0) goes straight from the barrier into
    // its MFMAs, so the matrix pipe works through the ~0.6 k cycles of address arithmetic and copy issue instead of standing still
    // while all eight waves issue (cycle stamps: 630 - 1900 cycles per step of hand-over, independent of the number of products).
    auto issue_i = [&](int stage, int c, int tile) {
        if (FSC_L3_COPYGRP && cw == 0) return;
        int pos_off[kNptMax];
        plan_input(tile, pos_off);
#pragma unroll
        for (int k = 0; k < (FSC_L3_COPYGRP ? 3 : 2); ++k) {
            const int u = FSC_L3_COPYGRP ? pw + 4 * k : wid + 8 * k;
            if (u < kUnits) {
                const int ol = u / 3;
                const int oct = c * 4 + ol;
                const bool live_u = oct < g.oct_in;
                const uint4* src = in + (long)(oct * 3 + (u - ol * 3)) * g.hw;
                uint4* dst = ibase + stage * istage + u * g.plane;
#pragma unroll
                for (int q = 0; q < kNptMax; ++q) {
                    if (q < g.npt && q * 64 + lane < g.plane) {
                        const bool lv = live_u && pos_off[q] >= 0;
                        glds16(lv ? src + pos_off[q] : zero, dst + q * 64);
                    }
                }
            }
        }
    };
    int ip_item = t0, ip_c = 0, ip_stg = 0;
    auto produce_i = [&]() -> bool {
        if (ip_item >= ntiles) return false;
        issue_i(ip_stg, ip_c, tile_of(ip_item));
        ip_stg ^= 1;
        if (++ip_c == nchunks) {
            ip_c = 0;
            ip_item += ts;
        }
        return true;
    };

    // ---- B operand address of (stage, chunk, step): byte offset from ibase of this lane group's octet, high limb, at the step's
    //      tap (without the pixel)
    const int limb_b = g.plane * 16;
    const int stage_b = kUnits * limb_b;
    const int kq_b = kq * 3 * limb_b;
    auto b_off_tail = [&](int stage, int s) -> int {        // remainder chunk: (tap, octet) flattened over lane groups
        const int noct = g.tail_oct;
        int gi = 4 * s + kq;
        if (gi >= TAPS * noct) gi = 0;                     // its weights are zero
        const int tap = TAPS == 1 ? 0 : fdiv(gi, 1.0f / (float)noct);
        const int oct = gi - tap * noct;
        const int ty = TAPS == 1 ? 0 : fdiv(tap, 1.0f / (float)KW), tx = tap - ty * KW;
        return stage * stage_b + oct * 3 * limb_b + (ty * g.cols + tx) * 16;
    };
    auto b_off = [&](int stage, int c, int s) -> int {
        if (c >= g.nfull) return b_off_tail(stage, s);
        const int ty = TAPS == 1 ? 0 : (s * 11) >> 5, tx = s - ty * KW;          // s / 3 for s < 10
        return stage * stage_b + (ty * g.cols + tx) * 16 + kq_b;
    };

    u32x4 A[CT][3], B[PTW][3];
    const char* const ib = reinterpret_cast<const char*>(ibase);
    // this lane's pointer to limb 0 of channel tile `tile0` of step 0 of the block's fragments
    const long a_step = (long)g.cot * 3 * 64;             // uint4 per step
    const u32x4* const a_base = reinterpret_cast<const u32x4*>(packed) + (long)cb_w * g.steps * a_step + lane;
    int a_tile[CT];                                        // (uniform) fragment offsets of this wave's tiles, dead ones clamped
#pragma unroll
    for (int i = 0; i < CT; ++i) {
        int t = tile0 + i;
        t = t < g.cot ? t : g.cot - 1;
        a_tile[i] = t * 3 * 64;
    }

    P3_DECL
#if defined(FSC_L3_PRIO) && (FSC_L3_PRIO == 3 || FSC_L3_PRIO == 4)
    // development: a fixed priority for one channel group (3: the copying group, 4: the other)
    if ((FSC_L3_PRIO == 3) == (cw == 1)) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#endif
    int item = t0;
    // consumer position (all uniform): step S of the item, chunk c, step sc of nst inside it, input stage stg
    int S = 0, c = 0, sc = 0, nst = g.nfull > 0 ? TAPS : g.tail_steps, stg = 0;
    bool drain = false;
    if (item < ntiles) {
        produce_i();
        produce_i();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_barrier();
    }
    // fragments of an item's first step: read at the top of the item, NOT carried across the previous item's epilogue (96 registers)
    auto load_first = [&]() {
        const int off0 = b_off(stg, 0, 0);
#pragma unroll
        for (int j = 0; j < PTW; ++j)
#pragma unroll
            for (int l = 0; l < 3; ++l) B[j][l] = *reinterpret_cast<const u32x4*>(ib + off0 + pix_b[j] + l * limb_b);
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int l = 0; l < 3; ++l) A[i][l] = a_base[a_tile[i] + l * 64];
    };

    // One MFMA step = one tap x 32 channels for LIVE channel tiles x PTW pixel tiles.
    auto step = [&](auto live_c) {
        constexpr int LIVE = decltype(live_c)::value;
        // ---- chunk hand-over, once per chunk at the start of its LAST step: every wave holds the B fragments of this step in
        //      registers, so the stage is free for the chunk after next; the next chunk's stage is published by the barrier (each
        //      wave waits for its own copies: loads retire in order and at most the A fragments of this step are younger)
        if (sc == nst - 1) {
            P3_ADD(1);
            if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LIVE) : "memory");
            drain = false;
            raw_barrier();
            P3_ADD(7);
            produce_i();           // (issued piecewise between the channel tiles of the step instead: +2.6 % cycles -- DESIGN 4.3)
            P3_ADD(0);
        }
        // ---- the next step: its B offset and A fragments
        int noff;
        {
            int sn = sc + 1;
            if (sn < nst) {
                noff = b_off(stg, c, sn);
            } else {
                sn = 0;
                stg ^= 1;
                c = c + 1 < nchunks ? c + 1 : 0;
                nst = c < g.nfull ? TAPS : g.tail_steps;
                noff = b_off(stg, c, 0);
            }
            sc = sn;
        }
        S = S + 1 < g.steps ? S + 1 : 0;
        const u32x4* const an = a_base + (long)S * a_step;
        const char* const bn = ib + noff;
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, LIVE>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            constexpr bool last = i == LIVE - 1;
#ifndef FSC_L3_PRIO
#define FSC_L3_PRIO 1
#endif
            // The two waves of a SIMD (w and w + 4: one of each channel group) share its matrix pipe, and issue arbitration is by
            // priority, then AGE: left alone, the older wave runs its MFMAs at the full rate, the younger one gets the leftovers, the
            // older waits ~18 k cycles at every chunk hand-over while the younger finishes ALONE with all its own stalls exposed
            // (cycle stamps: 2980 against 4500 cycles per step).  Alternating the priority per channel tile (1: per step) in
            // opposite phase makes them take turns, each covering the other's waits.
            if (FSC_L3_PRIO == 2 || (FSC_L3_PRIO == 1 && i == 0)) {
                const int ph = FSC_L3_PRIO == 2 ? (i & 1) : (S & 1);
                if ((ph ^ cw) & 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
#ifndef FSC_L3_TMPACC
#define FSC_L3_TMPACC 1
#endif
            // The nine limb products of a step are summed in a SEPARATE accumulator that starts from zero, smallest terms first, and
            // reach the running sum with one fp32 addition: accumulated straight into `acc` each of the nine MFMAs rounds at the
            // magnitude of the whole running sum (nine roundings per 32-channel tap instead of the one an fp32 FMA chain spends on
            // it: forward / input-gradient errors 2.8 ... 4.6x those of PyTorch's fp32 convolution).
            f32x4 tmp[PTW];
            static_for<0, NPROD>([&](auto p_c) {
                constexpr int p = decltype(p_c)::value;
#pragma unroll
                for (int j = 0; j < PTW; ++j) {
                    if (FSC_L3_TMPACC) tmp[j] = mfma_l<F16>(A[i][P::la[p]], B[j][P::lb[p]], p == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : tmp[j]);
                    else acc[i][j] = mfma_l<F16>(A[i][P::la[p]], B[j][P::lb[p]], acc[i][j]);
                }
                if constexpr (last && (p + 1 == NPROD || P::lb[p + 1 < NPROD ? p + 1 : p] != P::lb[p])) {
                    // this B limb is done for the step: the next step's fragments of it go behind these MFMAs
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < PTW; ++j)
                        B[j][P::lb[p]] = *reinterpret_cast<const u32x4*>(bn + pix_b[j] + P::lb[p] * limb_b);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // tile i's fragments of the next step, behind its MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 0; l < 3; ++l) A[i][l] = an[a_tile[i] + l * 64];
            __builtin_amdgcn_sched_barrier(0);
            if (FSC_L3_TMPACC) {
                // (plain v_add_f32: hipcc pairs them into v_pk_add_f32, which costs ~15 cycles beside MFMAs on gfx950)
#pragma unroll
                for (int j = 0; j < PTW; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v;
                        asm("v_add_f32_e32 %0, %1, %2" : "=v"(v) : "v"(acc[i][j][r]), "v"(tmp[j][r]));
                        acc[i][j][r] = v;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    const bool add_bias = bias != nullptr;
    const int CO_BLK = g.cot * 16;
    auto run = [&](auto live_c) {
        constexpr int LIVE = decltype(live_c)::value;
        for (; item < ntiles; item += ts) {
            const int tile = tile_of(item);
            const int co0 = cb_w * CO_BLK + tile0 * 16;
            load_first();
            P3_ADD(3);
#pragma unroll 1
            for (int k = 0; k < g.steps; ++k) step(live_c);
            P3_ADD(1);
#ifdef FSC_L16_PROFILE
            pf_acc[4] += g.steps;
#endif

            // (the lane-derived parts of the epilogue's address decode are item-invariant: behind an opaque copy of the lane index they
            // are recomputed per item instead of hoisted out of the item loop and kept in ~80 registers across the MFMA steps)
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int lm_e = lane_e & 15, kq_e = lane_e >> 4;
            if constexpr (ACT16 && !POOL) {
                // ---- affine + PReLU + limb split: lane (kq, lm) holds channels kq * 4 + r of pixel lm -- four channels = one 8-byte
                //      half of the (octet, limb) vector of that pixel; lanes kq, kq ^ 1 complete the 16 bytes, 16 pixels make 256
                long gpix[PTW];                           // 16-byte unit of the pixel in limb 0 of octet 0 of its image, or -1
                {
                    int t = tile;
                    const int twi = t % g.tiles_w; t /= g.tiles_w;
                    const int thi = t % g.tiles_h; t /= g.tiles_h;
                    const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
                    const long img_u = (long)((g.cout + 7) >> 3) * 3 * g.hw;
#pragma unroll
                    for (int pt = 0; pt < PTW; ++pt) {
                        const int p = (pw * PTW + pt) * 16 + lm_e;
                        gpix[pt] = -1;
                        if (p < g.npix) {
                            const int per = g.th * g.tw;
                            const int b = fdiv(p, inv_thw), rem = p - b * per;
                            const int r = fdiv(rem, inv_tw), cq = rem - r * g.tw;
                            if (n0 + b < g.n && h0 + r < g.h && w0 + cq < g.w)
                                gpix[pt] = (long)(n0 + b) * img_u + (long)(h0 + r) * g.w + (w0 + cq);
                        }
                    }
                }
                long hw_t = g.hw;
                const float* bias_t = bias;
                asm volatile("" : "+s"(hw_t), "+s"(bias_t));
                const int oct_out = (g.cout + 7) >> 3;
                const bool has_aff = act.scale != nullptr, has_alpha = act.alpha != nullptr;
#pragma unroll
                for (int i = 0; i < LIVE; ++i) {
                    if (i >= live) continue;
                    float bv[4], sc[4], sh[4], al[4];
                    bool okc[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cob = co0 + i * 16 + kq_e * 4 + r;
                        okc[r] = cob < g.cout;
                        bv[r] = (add_bias && okc[r]) ? bias_t[cob] : 0.f;
                        sc[r] = (has_aff && okc[r]) ? act.scale[cob] : 1.f;
                        sh[r] = (has_aff && okc[r]) ? act.shift[cob] : 0.f;
                        al[r] = (has_alpha && okc[r]) ? act.alpha[cob] : 0.f;
                    }
                    const int oct = ((co0 + i * 16) >> 3) + (kq_e >> 1);
                    if (oct < oct_out) {
#pragma unroll
                        for (int j = 0; j < PTW; ++j) {
                            if (gpix[j] < 0) continue;
                            float y[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float z = F16 ? fmaf(acc[i][j][r], oscale, bv[r]) : acc[i][j][r] + bv[r];
                                const float t = has_aff ? fmaf(z, sc[r], sh[r]) : z;
                                const float v = (has_alpha && !(t > 0.f)) ? al[r] * t : t;
                                y[r] = okc[r] ? v : 0.f;
                                seen_mx = fmaxf(seen_mx, fabsf(y[r]));
                            }
                            unsigned h0, m0, l0, h1, m1, l1;
                            if constexpr (F16) {
                                l16::split3s_pair(y[0], y[1], s_out, h0, m0, l0);
                                l16::split3s_pair(y[2], y[3], s_out, h1, m1, l1);
                            } else {
                                l16::split3_pair(y[0], y[1], h0, m0, l0);
                                l16::split3_pair(y[2], y[3], h1, m1, l1);
                            }
                            uint2* o = act.out16 + (gpix[j] + (long)oct * 3 * hw_t) * 2 + (kq_e & 1);
                            o[0] = make_uint2(h0, h1);
                            o[2 * hw_t] = make_uint2(m0, m1);
                            o[4 * hw_t] = make_uint2(l0, l1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (POOL) {
                // ---- pooled epilogue (conv_l16.hip): the four pixels of a pooling window sit in lanes lm, lm + 1, lm + 8, lm + 9
                int t = tile;
                const int twi = t % g.tiles_w; t /= g.tiles_w;
                const int thi = t % g.tiles_h; t /= g.tiles_h;
                const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
                const int oh = g.h >> 1, ow = g.w >> 1;
                const int tpr = g.tw >> 3, tpi = (g.th >> 1) * tpr;
                long pool_g[PTW];
                long pool_u[PTW];                            // (ACT16) 16-byte unit of the pooled pixel in limb 0 of octet 0 of its image
#pragma unroll
                for (int pt = 0; pt < PTW; ++pt) {
                    const int tt = pw * PTW + pt;
                    pool_g[pt] = -1;
                    pool_u[pt] = -1;
                    if (tt * 16 < g.npix) {
                        const int b = tt / tpi, rem = tt - b * tpi;
                        const int tr = rem / tpr, tc = rem - tr * tpr;
                        const int pr = (h0 >> 1) + tr, pc = (w0 >> 1) + 4 * tc + (lane_e & 3);
                        if (n0 + b < g.n && pr < oh && pc < ow) {
                            pool_g[pt] = ((long)(n0 + b) * g.cout * oh + pr) * ow + pc;
                            pool_u[pt] = (long)(n0 + b) * ((g.cout + 7) >> 3) * 3 * ((long)oh * ow) + (long)pr * ow + pc;
                        }
                    }
                }
                const long ohw = (long)oh * ow;
                const float* bias_p = bias;
                asm volatile("" : "+s"(bias_p));
                const int chp = lane_e >> 2;
#pragma unroll
                for (int i = 0; i < LIVE; ++i) {
                    if (i >= live) continue;                 // (a wave of a short last block: its clamped tiles belong to nobody)
                    float bv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cob = co0 + i * 16 + kq_e * 4 + r;
                        bv[r] = (bias_p != nullptr && cob < g.cout) ? bias_p[cob] : 0.f;
                    }
                    const int co = co0 + i * 16 + chp;
                    float pv = 0.f;
                    if (STATS && stat_pivot != nullptr && co < g.cout) pv = stat_pivot[co];
#pragma unroll
                    for (int j = 0; j < PTW; ++j) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v0 = F16 ? fmaf(acc[i][j][r], oscale, bv[r]) : acc[i][j][r] + bv[r];
                            const float v1 = dpp_xor1(v0);
                            const float v2 = dpp_xor8(v0);
                            const float v3 = dpp_xor8(v1);
                            float best = v0;                     // first maximum in window order, NaN wins (fsc_maxpool_fwd)
                            int bi = 0;
                            if (v1 > best || v1 != v1) { best = v1; bi = 1; }
                            if ((v2 > best || v2 != v2) && best == best) { best = v2; bi = 2; }
                            if ((v3 > best || v3 != v3) && best == best) { best = v3; bi = 3; }
                            if ((lm_e & 9) == 0) {
                                scratch[(kq_e * 4 + r) * kScr + (lm_e >> 1)] = best;
                                scratch[(kq_e * 4 + r) * kScr + 8 + (lm_e >> 1)] = __int_as_float(bi);
                            }
                        }
                        const float val = scratch[chp * kScr + (lane_e & 3)];
                        if constexpr (ACT16) {
                            // ---- inference (fsc_conv_l16_pool_fwd_act): the POOLED value goes through the eval-mode BatchNorm and PReLU
                            //      behind the pool (classifiers.py:532-534) and leaves as fp32 (the residual reads it) AND as the limbs the
                            //      next convolution reads; the pooled pre-activation itself is never written
                            float yv = 0.f;
                            if (co < g.cout) {
                                const float tt = act.scale != nullptr ? fmaf(val, act.scale[co], act.shift[co]) : val;
                                yv = (act.alpha != nullptr && !(tt > 0.f)) ? act.alpha[co] * tt : tt;
                            }
                            if (co < g.cout && pool_g[j] >= 0) {
                                if (out != nullptr) out[pool_g[j] + (long)co * ohw] = yv;
                                seen_mx = fmaxf(seen_mx, fabsf(yv));
                            }
                            scratch[chp * kScr + (lane_e & 3)] = yv;           // (its own slot) -> lanes 0..15 = (channel quad, pooled pixel)
                            if (lane_e < 16) {
                                const int kq2 = lane_e >> 2;
                                float y[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) y[r] = scratch[(kq2 * 4 + r) * kScr + (lane_e & 3)];
                                const int oct = ((co0 + i * 16) >> 3) + (kq2 >> 1);
                                if (oct < ((g.cout + 7) >> 3) && pool_u[j] >= 0) {
                                    unsigned h0, m0, l0, h1, m1, l1;
                                    if constexpr (F16) {
                                        l16::split3s_pair(y[0], y[1], s_out, h0, m0, l0);
                                        l16::split3s_pair(y[2], y[3], s_out, h1, m1, l1);
                                    } else {
                                        l16::split3_pair(y[0], y[1], h0, m0, l0);
                                        l16::split3_pair(y[2], y[3], h1, m1, l1);
                                    }
                                    uint2* o = act.out16 + (pool_u[j] + (long)oct * 3 * ohw) * 2 + (kq2 & 1);
                                    o[0] = make_uint2(h0, h1);
                                    o[2 * ohw] = make_uint2(m0, m1);
                                    o[4 * ohw] = make_uint2(l0, l1);
                                }
                            }
                        } else {
                            const int bidx = __float_as_int(scratch[chp * kScr + 8 + (lane_e & 3)]);
                            if (co < g.cout && pool_g[j] >= 0) {
                                out[pool_g[j] + (long)co * ohw] = val;
                                pool_idx[pool_g[j] + (long)co * ohw] = (uint8_t)bidx;
                                if (STATS) stat_add(i, val, pv);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // ---- epilogue: D row = channel (kq_e * 4 + r), column = pixel (lm) -> scratch[ch][px] -> lane = (channel, quad)
                long quad_g[PTW];
                int quad_ok[PTW];
                {
                    int t = tile;
                    const int twi = t % g.tiles_w; t /= g.tiles_w;
                    const int thi = t % g.tiles_h; t /= g.tiles_h;
                    const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
#pragma unroll
                    for (int pt = 0; pt < PTW; ++pt) {
                        const int p = (pw * PTW + pt) * 16 + (lane_e & 3) * 4;
                        quad_g[pt] = 0;
                        quad_ok[pt] = 0;
                        if (p < g.npix) {
                            const int per = g.th * g.tw;
                            const int b = fdiv(p, inv_thw), rem = p - b * per;
                            const int r = fdiv(rem, inv_tw), cq = rem - r * g.tw;
                            if (n0 + b < g.n && h0 + r < g.h) {
                                quad_g[pt] = (long)(n0 + b) * g.cout * g.hw + (long)(h0 + r) * g.w + (w0 + cq);
                                const int left = g.w - (w0 + cq), inbox = g.npix - p;
                                const int nv = left < inbox ? left : inbox;
                                quad_ok[pt] = nv >= 4 ? 15 : nv <= 0 ? 0 : (1 << nv) - 1;
                            }
                        }
                    }
                }
                long hw_t = g.hw;
                const float* bias_t = bias;
                asm volatile("" : "+s"(hw_t), "+s"(bias_t));
                const int ch = lane_e >> 2;
#pragma unroll
                for (int i = 0; i < LIVE; ++i) {
                    if (i >= live) continue;
                    float bv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cob = co0 + i * 16 + kq_e * 4 + r;
                        bv[r] = (add_bias && cob < g.cout) ? bias_t[cob] : 0.f;
                    }
                    const int co = co0 + i * 16 + ch;
                    float pv = 0.f;
                    if (STATS && stat_pivot != nullptr && co < g.cout) pv = stat_pivot[co];
#pragma unroll
                    for (int j = 0; j < PTW; ++j) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            scratch[(kq_e * 4 + r) * kScr + lm_e] = F16 ? fmaf(acc[i][j][r], oscale, bv[r]) : acc[i][j][r] + bv[r];
                        const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + ch * kScr + (lane_e & 3) * 4);
                        if (STATS && co < g.cout && quad_ok[j]) {
                            if (quad_ok[j] == 15) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) stat_add(i, v[k], pv);
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (quad_ok[j] & (1 << k)) stat_add(i, v[k], pv);
                            }
                        }
                        if (co < g.cout && quad_ok[j]) {
                            float* o = out + quad_g[j] + (long)co * hw_t;
                            if (!accumulate && quad_ok[j] == 15) {
                                *reinterpret_cast<f32x4*>(o) = v;
                            } else if (quad_ok[j] == 15) {
                                const f32x4 old = *reinterpret_cast<const f32x4*>(o);
                                *reinterpret_cast<f32x4*>(o) = old + v;
                            } else {
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                    if (quad_ok[j] & (1 << k)) o[k] = accumulate ? o[k] + v[k] : v[k];
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < PTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            drain = true;                                       // the stores above share the counter of the copies and retire out of order
            P3_ADD(2);
        }
    };
    if (CT > 1 && live == CT - 1) run(std::integral_constant<int, (CT > 1 ? CT - 1 : 1)>{});
    else run(std::integral_constant<int, CT>{});

#ifdef FSC_L16_PROFILE
    if (blockIdx.x == 0 && lane == 0) {
        pf_acc[5] = __builtin_readcyclecounter() - pf_k0;
        pf_acc[6] = __builtin_amdgcn_s_memrealtime() - pf_r0;
        for (int i = 0; i < 8; ++i) atomicAdd(&g_l3_prof[wid][i], pf_acc[i]);
    }
#endif
    if (blockIdx.x == 0 && tid == 0) {
        g_l3_clock[0] = __builtin_readcyclecounter() - ck0;
        g_l3_clock[1] = __builtin_amdgcn_s_memrealtime() - cr0;
    }
    if constexpr (ACT16) {
        seen_mx = fsc::wave_max(seen_mx);
        if (act.seen != nullptr && lane == 0) atomicMax(act.seen, __float_as_uint(seen_mx));
    }
    if constexpr (STATS) {
        // records [(worker * 8 + wave) * CO_BLK + channel in block]: this wave's tiles carry its sums, the block's other
        // channels neutral records (fsc_bn_records_fold_conv folds all eight waves of a worker)
        const int nlive = live < CT ? live : CT;
        float4* const rec = stat_rec + ((long)blockIdx.x * kWaves + wid) * CO_BLK;
        for (int k = lane; k < CO_BLK; k += 64) {
            const int t = k >> 4;
            if (t < tile0 || t >= tile0 + nlive) rec[k] = make_float4(0.f, 0.f, INFINITY, -INFINITY);
        }
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            float a = st_s1[i], b = st_s2[i], mn = st_mn[i], mx = st_mx[i];
            a += dpp_xor1(a); a += dpp_xor2(a);
            b += dpp_xor1(b); b += dpp_xor2(b);
            mn = fminf(mn, dpp_xor1(mn)); mn = fminf(mn, dpp_xor2(mn));
            mx = fmaxf(mx, dpp_xor1(mx)); mx = fmaxf(mx, dpp_xor2(mx));
            if ((lane & 3) == 0 && i < nlive) rec[(tile0 + i) * 16 + (lane >> 2)] = make_float4(a, b, mn, mx);
        }
    }
}

// -------------------------------------------------------------------------------------------
// host-side planning (box choice as conv_l16.hip plan_l16_pt)
struct L3Plan {
    L3Geom g;
    int ct, ptw, co_blocks, nprod;
    bool f16;                 // scaled fp16 limbs (arith 10)
    size_t lds_bytes;
    long tiles, workers;
};

bool plan_l3_pt(const fsc_conv_desc& d_in, int dgrad, int ptw, L3Plan* out, bool pool = false) {
    L3Plan p{};
    L3Geom& g = p.g;
    fsc_conv_desc d = d_in;
    const int taps = d.kh * d.kw;
    if (!((d.kh == 3 && d.kw == 3) || (d.kh == 1 && d.kw == 1))) return false;
    if (taps == 1) {
        d.w = d.h * d.w;
        d.h = 1;
    }
    g.n = d.n; g.h = d.h; g.w = d.w; g.hw = (long)d.h * d.w;
    g.cin = dgrad ? d.c_out : d.c_in;
    g.cout = dgrad ? d.c_in : d.c_out;
    if (g.cin < 32 || g.cout < 48) return false;
    g.oct_in = (g.cin + 7) / 8;
    g.img_stride = (long)g.oct_in * 3 * g.hw;
    if ((long)g.n * g.img_stride >= (1L << 31)) return false;
    if (g.w >= 1024 && taps > 1) return false;
    // channel tiles per block: at most 10 (two wave groups of five).  One block when the layer has <= 10 tiles; otherwise the
    // block size with the fewest executed tile slots (a block of n tiles runs 2 * ceil(n / 2)) plus a fixed part per block
    const int tiles = fsc::ceil_div(g.cout, 16);
    int cot = tiles < 2 ? 2 : tiles;
    if (tiles > 10) {
        long best = -1;
        for (int cc = 6; cc <= 10; ++cc) {
            const int bb = fsc::ceil_div(tiles, cc);
            const int last = tiles - (bb - 1) * cc;
            const long cost = (long)(bb - 1) * (2 * ((cc + 1) / 2) * 9 + 2) + (2 * ((last + 1) / 2) * 9 + 2);
            if (best < 0 || cost < best) { best = cost; cot = cc; }
        }
    }
    const int force_cot = fsc::env().l16_cot;               // development (FSC_L16_COT): force the channel tiles per block
    if (force_cot >= 2 && force_cot <= 10) cot = force_cot;
    p.ct = (cot + 1) / 2;
    g.cot = cot;
    g.tiles_total = tiles;
    p.co_blocks = fsc::ceil_div(tiles, cot);
    p.ptw = ptw;
    const int pix_cap = 4 * ptw * 16;
    const size_t lds_total = 160 * 1024;
    const size_t scratch = (size_t)kWaves * 16 * kScr * sizeof(float);
    int cap_pos = (int)((lds_total - scratch) / ((size_t)kNstg * kUnits * 16));
    cap_pos &= ~15;
    if (cap_pos > 64 * kNptMax) cap_pos = 64 * kNptMax;
    long bcost = -1;
    int bnb = 1, bth = 1, btw = 1;
    for (int tw = 4; tw <= ((d.w + 3) & ~3) && tw <= pix_cap; tw += 4) {
        if (pool && (tw & 7)) continue;
        int th = pix_cap / tw;
        if (th > d.h) th = pool ? ((d.h + 1) & ~1) : d.h;
        if (pool) th &= ~1;
        if (th < (pool ? 2 : 1)) continue;
        int nb = 1;
        if (th >= d.h && tw >= d.w) {
            nb = pix_cap / (th * tw);
            if (nb > d.n) nb = d.n;
            if (nb < 1) nb = 1;
        }
        while (nb > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) --nb;
        while (th > (pool ? 2 : 1) && nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) th -= pool ? 2 : 1;
        if (nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) continue;
        const long nt = (long)fsc::ceil_div(d.n, nb) * fsc::ceil_div(d.h, th) * fsc::ceil_div(d.w, tw);
        const long pen = (tw >= 32 || tw >= d.w) ? 100 : tw >= 16 ? 102 : tw >= 8 ? 108 : 125;
        const long cost = nt * pen;
        if (bcost < 0 || cost < bcost || (cost == bcost && tw > btw)) {
            bcost = cost; bnb = nb; bth = th; btw = tw;
        }
    }
    if (bcost < 0) return false;
    g.nb = bnb; g.th = bth; g.tw = btw;
    g.rows = bth + d.kh - 1; g.cols = btw + d.kw - 1;
    g.npix = bnb * bth * btw; g.npos = bnb * g.rows * g.cols;
    g.tiles_n = fsc::ceil_div(d.n, bnb); g.tiles_h = fsc::ceil_div(d.h, bth); g.tiles_w = fsc::ceil_div(d.w, btw);
    p.tiles = (long)g.tiles_n * g.tiles_h * g.tiles_w;
    if ((double)d.n * g.hw < (ptw == 4 ? 0.7 : 0.55) * (double)p.tiles * pix_cap) return false;
    // (a multiple of 16 positions: the four lane groups of a ds_read_b128 address octets 3 limb planes apart, which must be a
    // multiple of 256 bytes for their 16-byte pieces to fall into disjoint banks -- with a multiple of 8, 47 % of the LDS cycles of
    // the 100 -> 100 layer were bank conflicts)
    g.plane = (g.npos + 15) & ~15;
    g.npt = fsc::ceil_div(g.plane, 64);
    if (g.npt > kNptMax) return false;
    const int rem = g.cin % kChunk;
    g.nfull = g.cin / kChunk + (rem > kChunk - 8 ? 1 : 0);
    g.tail_oct = (rem > 0 && rem <= kChunk - 8) ? fsc::ceil_div(rem, 8) : 0;
    g.tail_steps = fsc::ceil_div(taps * g.tail_oct, 4);
    g.steps = g.nfull * taps + g.tail_steps;
    if (g.steps < 3) return false;
    g.coblk = p.co_blocks;
    p.lds_bytes = (size_t)kNstg * kUnits * 16 * g.plane + scratch;
    if (p.lds_bytes > lds_total) return false;
    const long items = p.tiles * p.co_blocks;
    if (items < 128) return false;
    p.workers = items < 256 ? items : 256;
    if (items > 256 && items <= 512) p.workers = (items + 1) / 2;
    p.workers -= p.workers % p.co_blocks;
    g.xcd = (p.co_blocks > 1 && p.workers % (8 * p.co_blocks) == 0 && !fsc::env().l16_no_xcd) ? 1 : 0;
    // (one channel block: item v = worker + workers * k, and worker mod 8 is the worker's XCD when the grid is a multiple of 8)
    g.spat = (g.xcd || (p.co_blocks == 1 && p.workers % 8 == 0 && !fsc::env().l16_no_xcd)) ? 1 : 0;
    *out = p;
    return true;
}

int nprod_of(int arith) { return arith == 6 || l16::is_f3(arith) ? 6 : arith == 8 ? 8 : 9; }

// the three-limb arithmetics THIS BUILD has kernels for: 9 (bf16, nine products) and 10 (scaled fp16, six products); the six- and
// eight-product bf16 variants exist under -DFSC_L3_ALL_PRODS only, so without it their descriptors have NO pre-split tiling
// (supported / packed_floats / pack_weights say so up front instead of failing with rc 22 at launch time: ADVICE r5) and the
// caller's fp32-input route (fsc_conv_fwd, arith 6) serves them
bool l3_built(int arith) {
#ifdef FSC_L3_ALL_PRODS
    return l16::is_l3(arith);
#else
    return arith == 9 || l16::is_f3(arith);
#endif
}

bool plan_l3(const fsc_conv_desc& d, int dgrad, L3Plan* out) {
    if (!l3_built(d.arith)) return false;
    if (fsc::env().no_l16) return false;
    const int force_pt = fsc::env().l16_pt;                 // development (FSC_L16_PT = 2 | 1): 256- / 128-pixel tiles
    bool ok = false;
    if ((!force_pt || force_pt == 2) && plan_l3_pt(d, dgrad, 4, out)) ok = true;
    else if (force_pt != 2 && plan_l3_pt(d, dgrad, 2, out)) ok = true;
    if (ok) {
        out->nprod = nprod_of(d.arith);
        out->f16 = l16::is_f3(d.arith);
    }
    return ok;
}

bool plan_l3_pool(const fsc_conv_desc& d, L3Plan* out) {
    if (!l3_built(d.arith)) return false;
    if (fsc::env().no_l16 || fsc::env().no_l16_pool) return false;
    if (d.kh != 3 || d.kw != 3 || d.h < 2 || d.w < 8) return false;
    L3Plan plain;
    if (!plan_l3(d, 0, &plain) || plain.ptw != 4) return false;
    if (!plan_l3_pt(d, 0, 4, out, true)) return false;
    out->nprod = nprod_of(d.arith);
    out->f16 = l16::is_f3(d.arith);
    return out->g.cot == plain.g.cot && out->co_blocks == plain.co_blocks && out->g.steps == plain.g.steps;
}

size_t l3_packed_u4(const L3Plan& p) { return (size_t)p.co_blocks * p.g.steps * p.g.cot * 3 * 64; }
// scaled fp16 limbs: behind the fragments, max |w| (4 floats: 16-byte alignment) and the partial maxima of the pack
constexpr size_t kWTailFloats = 4 + kWmaxBlocks3;

bool stats_ok3(const L3Plan& p) { return p.workers >= p.co_blocks; }

struct StatArgs3 { const float* pivot; float4* rec; };
struct ScaleArgs3 { const float* in_amax; const float* w_amax; };

template <int KH, int KW, int CT, int PTW, int NPROD, bool F16, bool POOL, bool STATS, bool ACT16 = false>
int launch3(const L3Plan& p, const uint4* in, const uint4* packed, const float* bias, float* out, int accumulate, uint8_t* idx,
            StatArgs3 sa, ScaleArgs3 sc, ActArgs3 act, hipStream_t st) {
    auto kern = conv_l3_fwd_kernel<KH, KW, CT, PTW, NPROD, F16, POOL, STATS, ACT16>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)p.workers), dim3(kWaves * 64), p.lds_bytes, st, p.g, in, packed, bias, out, accumulate, idx,
                       sa.pivot, sa.rec, sc.in_amax, sc.w_amax, act);
    FSC_LAUNCH_CHECK("fsc_conv_l16_fwd(three limbs)");
    return 0;
}

template <int KH, int KW, int CT, int PTW, int NPROD, bool F16>
int launch3_var(const L3Plan& p, const uint4* in, const uint4* packed, const float* bias, float* out, int accumulate, uint8_t* idx,
                StatArgs3 sa, ScaleArgs3 sc, ActArgs3 act, hipStream_t st) {
    if (act.out16 && act.pooled) {
        if constexpr (KH == 3 && PTW == 4)
            return launch3<KH, KW, CT, PTW, NPROD, F16, true, false, true>(p, in, packed, bias, out, 0, nullptr, sa, sc, act, st);
        fsc::set_error("fsc_conv_l16_pool_fwd_act: internal: no pooled instantiation");
        return 22;
    }
    if (act.out16) return launch3<KH, KW, CT, PTW, NPROD, F16, false, false, true>(p, in, packed, bias, nullptr, 0, nullptr, sa, sc, act, st);
    if (idx) {
        if constexpr (KH == 3 && PTW == 4) {
            if (sa.rec) return launch3<KH, KW, CT, PTW, NPROD, F16, true, true>(p, in, packed, bias, out, 0, idx, sa, sc, act, st);
            return launch3<KH, KW, CT, PTW, NPROD, F16, true, false>(p, in, packed, bias, out, 0, idx, sa, sc, act, st);
        }
        fsc::set_error("fsc_conv_l16_pool_fwd: internal: no pooled instantiation");
        return 22;
    }
    if (sa.rec) return launch3<KH, KW, CT, PTW, NPROD, F16, false, true>(p, in, packed, bias, out, 0, nullptr, sa, sc, act, st);
    return launch3<KH, KW, CT, PTW, NPROD, F16, false, false>(p, in, packed, bias, out, accumulate, nullptr, sa, sc, act, st);
}

template <int KH, int KW, int NPROD, bool F16>
int launch3_ct(const L3Plan& p, const uint4* in, const uint4* packed, const float* bias, float* out, int accumulate, uint8_t* idx,
               StatArgs3 sa, ScaleArgs3 sc, ActArgs3 act, hipStream_t st) {
#define FSC_L3_CASE(CT_)                                                                                                             \
    case CT_:                                                                                                                        \
        if (p.ptw == 4) return launch3_var<KH, KW, CT_, 4, NPROD, F16>(p, in, packed, bias, out, accumulate, idx, sa, sc, act, st); \
        return launch3_var<KH, KW, CT_, 2, NPROD, F16>(p, in, packed, bias, out, accumulate, idx, sa, sc, act, st);
    switch (p.ct) {
#ifndef FSC_L16_DEV
        FSC_L3_CASE(2)
        FSC_L3_CASE(3)
#endif
        FSC_L3_CASE(4)
        FSC_L3_CASE(5)
        default: break;
    }
#undef FSC_L3_CASE
    fsc::set_error("fsc_conv_l16_fwd: internal: no three-limb instantiation for %d channel tiles per wave", p.ct);
    return 22;
}

int launch3_any(const L3Plan& p, int taps, const uint4* in, const uint4* packed, const float* bias, float* out, int accumulate,
                uint8_t* idx, StatArgs3 sa, const float* in_amax, hipStream_t st, ActArgs3 act = ActArgs3{}) {
    ScaleArgs3 sc{nullptr, nullptr};
    if (p.f16) {
        sc.in_amax = in_amax;
        sc.w_amax = reinterpret_cast<const float*>(packed + l3_packed_u4(p));
#ifndef FSC_L3_NO_F16
        if (taps == 9) return launch3_ct<3, 3, 6, true>(p, in, packed, bias, out, accumulate, idx, sa, sc, act, st);
        return launch3_ct<1, 1, 6, true>(p, in, packed, bias, out, accumulate, idx, sa, sc, act, st);
#endif
    }
#ifndef FSC_L3_NO_BF16
    if (p.nprod == 9) {
        if (taps == 9) return launch3_ct<3, 3, 9, false>(p, in, packed, bias, out, accumulate, idx, sa, sc, act, st);
        return launch3_ct<1, 1, 9, false>(p, in, packed, bias, out, accumulate, idx, sa, sc, act, st);
    }
#endif
#ifdef FSC_L3_ALL_PRODS
    if (p.nprod == 8) {
        if (taps == 9) return launch3_ct<3, 3, 8, false>(p, in, packed, bias, out, accumulate, idx, sa, sc, act, st);
        return launch3_ct<1, 1, 8, false>(p, in, packed, bias, out, accumulate, idx, sa, sc, act, st);
    }
    if (p.nprod == 6) {
        if (taps == 9) return launch3_ct<3, 3, 6, false>(p, in, packed, bias, out, accumulate, idx, sa, sc, act, st);
        return launch3_ct<1, 1, 6, false>(p, in, packed, bias, out, accumulate, idx, sa, sc, act, st);
    }
#endif
    fsc::set_error("fsc_conv_l16_fwd: this build has no three-limb kernels with %d products%s", p.nprod, p.f16 ? " (fp16)" : "");
    return 22;
}

bool valid3(const fsc_conv_desc* d) {
    if (!d || d->n <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->h <= 0 || d->w <= 0) return false;
    const long big = 1L << 31;
    return (long)d->n * d->c_in * d->h * d->w < big && (long)d->n * d->c_out * d->h * d->w < big;
}

PackDir3 make_dir3(const L3Plan& p, float* packed, int dgrad) {
    PackDir3 r{};
    r.packed = reinterpret_cast<uint4*>(packed);
    r.w_amax = p.f16 ? packed + l3_packed_u4(p) * 4 : nullptr;
    r.cot = p.g.cot; r.co_blocks = p.co_blocks; r.nfull = p.g.nfull; r.tail_oct = p.g.tail_oct; r.steps = p.g.steps;
    r.dgrad = dgrad;
    r.frags = (long)p.co_blocks * p.g.steps * p.g.cot * 64;
    long xb = (r.frags + 255) / 256;
    r.blocks = (int)(xb > 4096 ? 4096 : xb);
    return r;
}

}  // namespace

// ------------------------------------------------------------------------------------------- internal interface (l16.h)
namespace fsc {
namespace l3 {

size_t tensor_bytes(int n, int c, long hw) { return (size_t)n * ((c + 7) / 8) * 3 * (size_t)hw * 16; }

int pack(const float* x, int n, int c, long hw, const float* amax, void* out, hipStream_t st) {
    const long total = (long)n * ((c + 7) / 8) * hw;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(l3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, c, hw, amax, reinterpret_cast<uint4*>(out));
    FSC_LAUNCH_CHECK("fsc_l16_pack(three limbs)");
    return 0;
}

int unpack(const void* in, int n, int c, long hw, const float* amax, float* x, hipStream_t st) {
    const long total = (long)n * ((c + 7) / 8) * hw;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(l3_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const uint4*>(in), n, c, hw, amax, x);
    FSC_LAUNCH_CHECK("fsc_l16_unpack(three limbs)");
    return 0;
}

int supported(const fsc_conv_desc* d, int dgrad) {
    L3Plan p;
    return valid3(d) && plan_l3(*d, dgrad, &p) ? 1 : 0;
}

int pool_supported(const fsc_conv_desc* d) {
    L3Plan p;
    return valid3(d) && plan_l3_pool(*d, &p) ? 1 : 0;
}

size_t packed_floats(const fsc_conv_desc* d, int dgrad) {
    L3Plan p;
    if (!valid3(d) || !plan_l3(*d, dgrad, &p)) return 0;
    return l3_packed_u4(p) * 4 + (p.f16 ? kWTailFloats : 0);
}

int pack_weights_multi(int count, const fsc_conv_desc* descs, const float* const* weights, float* const* packed_fwd,
                       float* const* packed_dgrad, hipStream_t st);
int pack_weights_pair(const fsc_conv_desc* d, const float* weight, float* packed_fwd, float* packed_dgrad, hipStream_t st) {
    FSC_CHECK_ARG(valid3(d) && weight && (packed_fwd || packed_dgrad), "fsc_conv_l16_pack_weights_pair: bad arguments");
    return pack_weights_multi(1, d, &weight, &packed_fwd, &packed_dgrad, st);
}

int pack_weights_multi(int count, const fsc_conv_desc* descs, const float* const* weights, float* const* packed_fwd,
                       float* const* packed_dgrad, hipStream_t st) {
    for (int base = 0; base < count; base += kMultiPack3) {
        PackJobs3 jobs{};
        jobs.n = count - base < kMultiPack3 ? count - base : kMultiPack3;
        int blocks = 0;
        bool f16 = false;
        for (int k = 0; k < jobs.n; ++k) {
            const fsc_conv_desc* d = descs + base + k;
            float* pf_out = packed_fwd[base + k];
            float* pd_out = packed_dgrad[base + k];
            FSC_CHECK_ARG(valid3(d) && weights[base + k] && (pf_out || pd_out), "fsc_conv_l16_pack_weights_multi: bad entry %d", base + k);
            L3Plan pf{}, pd{};
            FSC_CHECK_ARG(!pf_out || plan_l3(*d, 0, &pf), "fsc_conv_l16_pack_weights_multi: no forward tiling for entry %d", base + k);
            FSC_CHECK_ARG(!pd_out || plan_l3(*d, 1, &pd), "fsc_conv_l16_pack_weights_multi: no dgrad tiling for entry %d", base + k);
            PackJob3& pj = jobs.j[k];
            pj.w = weights[base + k];
            pj.c_out = d->c_out; pj.c_in = d->c_in; pj.taps = d->kh * d->kw;
            pj.count = (long)d->c_out * d->c_in * pj.taps;
            f16 = f16 || l16::is_f3(d->arith);
            PackDir3 a{}, b{};
            if (pf_out) a = make_dir3(pf, pf_out, 0);
            if (pd_out) b = make_dir3(pd, pd_out, 1);
            if (!pf_out) { a = b; b = PackDir3{}; }
            a.part_home = 1;
            pj.a = a; pj.b = b;
            pj.part = a.w_amax ? a.w_amax + 4 : nullptr;
            pj.first_block = blocks;
            blocks += a.blocks + b.blocks;
        }
        if (f16) hipLaunchKernelGGL(l3_wmax_multi_kernel, dim3((unsigned)(jobs.n * kWmaxBlocks3)), dim3(256), 0, st, jobs);
        hipLaunchKernelGGL(l3_pack_w_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, jobs);
    }
    FSC_LAUNCH_CHECK("fsc_conv_l16_pack_weights_multi(three limbs)");
    return 0;
}

int fwd(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias, int dgrad,
        int accumulate, float* out, const float* stat_pivot, void* stat_rec, hipStream_t st) {
    L3Plan p;
    FSC_CHECK_ARG(valid3(d) && in_l16 && packed && out, "fsc_conv_l16_fwd: bad descriptor or null pointer");
    FSC_CHECK_ARG(!l16::is_f3(d->arith) || in_amax, "fsc_conv_l16_fwd: scaled fp16 limbs need the input's declared maximum");
    FSC_CHECK_ARG(!(dgrad && bias), "fsc_conv_l16_fwd: dgrad takes no bias");
    FSC_CHECK_ARG(plan_l3(*d, dgrad, &p), "fsc_conv_l16_fwd: unsupported shape (see fsc_conv_l16_supported)");
    FSC_CHECK_ARG(!stat_rec || stats_ok3(p), "fsc_conv_l16_fwd_stats: unsupported shape (see fsc_conv_l16_stats_layout)");
    return launch3_any(p, d->kh * d->kw, reinterpret_cast<const uint4*>(in_l16), reinterpret_cast<const uint4*>(packed), bias, out,
                       accumulate, nullptr, StatArgs3{stat_pivot, reinterpret_cast<float4*>(stat_rec)}, in_amax, st);
}

int fwd_act(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias,
            const float* scale, const float* shift, const float* alpha, void* out_l16, const float* out_amax, float* seen_max,
            hipStream_t st) {
    L3Plan p;
    FSC_CHECK_ARG(valid3(d) && in_l16 && packed && out_l16, "fsc_conv_l16_fwd_act: bad descriptor or null pointer");
    FSC_CHECK_ARG((scale == nullptr) == (shift == nullptr), "fsc_conv_l16_fwd_act: scale and shift come together");
    FSC_CHECK_ARG(!l16::is_f3(d->arith) || (in_amax && out_amax), "fsc_conv_l16_fwd_act: scaled fp16 limbs need both declared maxima");
    FSC_CHECK_ARG(plan_l3(*d, 0, &p), "fsc_conv_l16_fwd_act: unsupported shape (see fsc_conv_l16_supported)");
    ActArgs3 act{scale, shift, alpha, reinterpret_cast<uint2*>(out_l16), out_amax, reinterpret_cast<unsigned*>(seen_max), 0};
    return launch3_any(p, d->kh * d->kw, reinterpret_cast<const uint4*>(in_l16), reinterpret_cast<const uint4*>(packed), bias, nullptr,
                       0, nullptr, StatArgs3{nullptr, nullptr}, in_amax, st, act);
}

// conv 3x3 -> MaxPool2d(2) -> eval-mode BatchNorm -> PReLU in one launch (reference networks/classifiers.py:526-534): `out` (may be
// NULL) receives the result as fp32 (N, c_out, H / 2, W / 2), `out_l16` as the three-limb L16 tensor of that shape
int pool_fwd_act(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias,
                 const float* scale, const float* shift, const float* alpha, float* out, void* out_l16, const float* out_amax,
                 float* seen_max, hipStream_t st) {
    L3Plan p;
    FSC_CHECK_ARG(valid3(d) && in_l16 && packed && out_l16, "fsc_conv_l16_pool_fwd_act: bad descriptor or null pointer");
    FSC_CHECK_ARG((scale == nullptr) == (shift == nullptr), "fsc_conv_l16_pool_fwd_act: scale and shift come together");
    FSC_CHECK_ARG(!l16::is_f3(d->arith) || (in_amax && out_amax), "fsc_conv_l16_pool_fwd_act: scaled fp16 limbs need both declared maxima");
    FSC_CHECK_ARG(plan_l3_pool(*d, &p), "fsc_conv_l16_pool_fwd_act: unsupported shape (see fsc_conv_l16_pool_supported)");
    ActArgs3 act{scale, shift, alpha, reinterpret_cast<uint2*>(out_l16), out_amax, reinterpret_cast<unsigned*>(seen_max), 1};
    return launch3_any(p, 9, reinterpret_cast<const uint4*>(in_l16), reinterpret_cast<const uint4*>(packed), bias, out, 0, nullptr,
                       StatArgs3{nullptr, nullptr}, in_amax, st, act);
}

int pool_fwd(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias, float* pooled,
             uint8_t* idx, const float* stat_pivot, void* stat_rec, hipStream_t st) {
    L3Plan p;
    FSC_CHECK_ARG(valid3(d) && in_l16 && packed && pooled && idx, "fsc_conv_l16_pool_fwd: bad descriptor or null pointer");
    FSC_CHECK_ARG(!l16::is_f3(d->arith) || in_amax, "fsc_conv_l16_pool_fwd: scaled fp16 limbs need the input's declared maximum");
    FSC_CHECK_ARG(plan_l3_pool(*d, &p), "fsc_conv_l16_pool_fwd: unsupported shape (see fsc_conv_l16_pool_supported)");
    FSC_CHECK_ARG(!stat_rec || stats_ok3(p), "fsc_conv_l16_pool_fwd_stats: unsupported shape (see fsc_conv_l16_stats_layout)");
    return launch3_any(p, 9, reinterpret_cast<const uint4*>(in_l16), reinterpret_cast<const uint4*>(packed), bias, pooled, 0, idx,
                       StatArgs3{stat_pivot, reinterpret_cast<float4*>(stat_rec)}, in_amax, st);
}

int stats_layout(const fsc_conv_desc* d, int pool, int* out4) {
    L3Plan p;
    if (!valid3(d) || !out4) return 0;
    if (!(pool ? plan_l3_pool(*d, &p) : plan_l3(*d, 0, &p)) || !stats_ok3(p)) return 0;
    out4[0] = (int)p.workers; out4[1] = p.co_blocks; out4[2] = p.g.cot * 16; out4[3] = p.g.xcd;
    return 1;
}

int plan_describe(const fsc_conv_desc* d, int dgrad, char* buf, size_t buf_len) {
    L3Plan p;
    FSC_CHECK_ARG(valid3(d) && buf && buf_len > 0 && plan_l3(*d, dgrad, &p), "fsc_conv_l16_plan_describe: unsupported shape");
    // (the name is the kernel instantiation: <KH, KW, channel tiles per WAVE, pixel tiles per wave, limb products>; `cot` = channel
    // tiles per block, halved between the two wave groups)
    snprintf(buf, buf_len, "conv_l3_fwd_kernel<%d,%d,%d,%d,%d%s> cot=%d box=%dx%dx%d items=%ldx%d workers=%ld steps=%d lds=%zu", d->kh,
             d->kw, p.ct, p.ptw, p.nprod, p.f16 ? ",f16" : "", p.g.cot, p.g.nb, p.g.th, p.g.tw, p.tiles, p.co_blocks, p.workers, p.g.steps, p.lds_bytes);
    return 0;
}

int last_clock(double* shader_mhz) {
    unsigned long long v[2] = {0, 0};
    hipError_t e = hipMemcpyFromSymbol(v, HIP_SYMBOL(g_l3_clock), sizeof(v));
    if (e != hipSuccess) { fsc::set_error("fsc_conv_l16_last_clock: %s", hipGetErrorString(e)); return (int)e; }
    *shader_mhz = v[1] ? 100.0 * (double)v[0] / (double)v[1] : 0.0;
    return 0;
}

}  // namespace l3
}  // namespace fsc

#ifdef FSC_L16_PROFILE
extern "C" int fsc_debug_l3_prof(unsigned long long* out64) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_l3_prof), sizeof(unsigned long long) * 64);
    unsigned long long z[64] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_l3_prof), z, sizeof(z));
    return 0;
}
#endif
