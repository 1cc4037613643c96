// Convolutions on PRE-SPLIT activations ("L16" tensors) for gfx950.
//
// conv.hip's split-fp16 kernels read fp32 NCHW activations and split every value into two scaled fp16 limbs
// beside the MFMAs -- once per TAP, nine times for a 3x3 layer -- which leaves them latency-bound at ~30 % MFMA
// busy (profiles/r01g_pmc_conv_f16x3.txt).  Here the producer of an activation (the BN / PReLU apply kernels,
// norm_act.hip, or fsc_l16_pack) writes the limbs ONCE, in the layout the MFMA B operand wants:
//
//     L16 tensor of a logical (N, C, H, W) fp32 tensor, OCT = ceil(C / 8):
//         half[N][OCT][2 limbs][H * W][8 channels]         (16 bytes per (limb, position); pad channels are 0)
//     with one power-of-two scale per tensor taken from its declared maximum (the FSC_AMAX_FLOATS slot buffer):
//         h = rne16(x * s), l = rne16(x * s - h)            (same arithmetic as conv.hip split2_pair)
//
// 4 bytes per element like fp32.  The conv kernel then
//   * copies the halo'd box of a 32-channel chunk HBM -> LDS with 16-byte LDS-DMA (a quarter of the copy
//     instructions of the fp32 kernel), as [octet * 2 + limb][position][8];
//   * reads a B fragment (8 channels of one pixel, one limb) with ONE ds_read_b128 -- no split, no VALU;
// everything else (pre-split weight fragments streamed through a ring of step slots, persistent workgroups,
// counted waits across raw barriers, 16-byte epilogue stores) follows conv_fwd_x3_kernel.  The products and their
// order are those of the f16x3 arithmetic of conv.hip (hl, lh, hh per k-step), so results are bit-identical to it.
//
// Replaces nn.Conv2d 3x3 / 1x1 forward and input gradient (reference networks/classifiers.py:526-531, 77-81).
#include "common.h"
#include "l16.h"
#include <vector>
#include <stdlib.h>
#include <string.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(1))) const void* glb_ptr;

constexpr int kWaves = 8;            // two waves per SIMD
constexpr int kChunk = 32;           // channels per K chunk = 4 octets
constexpr int kNptMax = 6;           // input DMA instructions per (octet, limb) plane: plane <= 384 positions
constexpr int kScr = 20;             // epilogue scratch row stride (floats)
constexpr int kWmaxBlocks = 64;      // partial maxima of the weights (l16_wmax_kernel)

struct LGeom {
    int n, cin, cout, h, w;
    long hw;
    int oct_in;               // octets of the input tensor
    int nb, th, tw;           // pixel box: images x rows x cols
    int tiles_n, tiles_h, tiles_w;
    int rows, cols;           // staged box incl. halo
    int plane;                // positions per staged (octet, limb) plane, multiple of 8
    int npos, npix, npt;
    int nfull, tail_oct, tail_steps, steps;
    int coblk;
    long img_stride;          // 16-byte units between images of the input: oct_in * 2 * hw
    int xcd;                  // item order: the channel blocks of a tile on workers of one XCD (see the kernel)
    int spat;                 // item index v -> tile of XCD (v mod 8)'s contiguous eighth of the tiles (tile_of in the kernel)
};

__device__ __attribute__((aligned(16))) float g_zero16_l[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// Shader clock of the last launch: workgroup 0 stamps the shader-cycle counter (s_memtime) and the constant 100 MHz reference
// counter (s_memrealtime) at both ends of the kernel; their ratio is the clock the chip actually ran the kernel at (the MFMA-bound
// launches of cfg 2 run near 1.6 GHz under the power limit, not at the 2.4 GHz the peak is quoted for).  fsc_conv_l16_last_clock.
__device__ unsigned long long g_l16_clock[2];

// Development (-DFSC_L16_PROFILE): shader-clock stamps around the phases of a step, summed per wave of workgroup 0 into
// g_l16_prof[wave][phase]; read back and cleared by fsc_debug_l16_prof.  Phases: 0 wait + barrier, 1 barrier -> first MFMA group
// (copy issue of the un-spread variant, control, fresh A fragments), 2 MFMA groups (with the spread copies between them),
// 3 epilogue, 4 steps counted, 5 whole kernel.
#ifdef FSC_L16_PROFILE
__device__ unsigned long long g_l16_prof[8][8];
__device__ __forceinline__ unsigned long long prof_now() { return __builtin_readcyclecounter(); }
#define PROF_DECL unsigned long long pf_t = 0, pf_acc[7] = {0, 0, 0, 0, 0, 0, 0}; const unsigned long long pf_k0 = prof_now(), pf_r0 = __builtin_amdgcn_s_memrealtime();
#define PROF_MARK() (pf_t = prof_now())
#define PROF_ADD(i) do { const unsigned long long n_ = prof_now(); pf_acc[i] += n_ - pf_t; pf_t = n_; } while (0)
#else
#define PROF_DECL
#define PROF_MARK()
#define PROF_ADD(i)
#endif

// 16-byte LDS-DMA (lane i writes lds_wave_base + 16 i).  Issued as inline assembly on purpose: hipcc knows that the builtin
// writes LDS and, unable to tell the stages of the dynamic LDS block apart, puts `s_waitcnt vmcnt(0)` in front of the next LDS
// read -- i.e. waits for the copy it has just issued for a LATER step before computing the current one (measured: the kernels
// ran 20-37 % faster with the copies removed, and not at all faster with the explicit waits removed).  The waits of these
// kernels are explicit (`s_waitcnt vmcnt(N)` before the barrier that publishes a stage).
__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m) : "memory", "m0");
}

using l16::split2_pair;
using l16::scale_field;
using l16::field_to_float;
using l16::inv_scale;
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// lane ^ 1 and lane ^ 8 inside a row of 16 lanes as DPP moves (quad_perm [1,0,3,2], row_ror:8): __shfl_xor goes through
// ds_bpermute and the LDS queue
__device__ __forceinline__ float dpp_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {      // quad_perm [2,3,0,1]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor8(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));
}
__device__ __forceinline__ void raw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// maximum over an FSC_AMAX_FLOATS slot buffer (one float per thread of a 512-thread block), uniform result
__device__ __forceinline__ float block_amax512(const float* amax, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float mw = fsc::wave_max(amax[tid]);
    if (lane == 0) red[wid] = mw;
    __syncthreads();
    float ax = red[0];
#pragma unroll
    for (int i = 1; i < kWaves; ++i) ax = fmaxf(ax, red[i]);
    ax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ax)));
    __syncthreads();
    return ax;
}

// -------------------------------------------------------------------------------------------
// fp32 NCHW -> L16 (stand-alone producer: tests, and tensors no fused producer writes).  One thread = one
// (image, octet, position): reads 8 channel planes (coalesced along positions), writes 16 + 16 bytes.
__global__ __launch_bounds__(256) void l16_pack_kernel(const float* __restrict__ x, int n, int c, long hw,
                                                       const float* __restrict__ amax, uint4* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < fsc::kAmaxFloats; i += 256) m = fmaxf(m, amax[i]);
    m = fsc::wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = field_to_float(scale_field(m));
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
        uint4 hi, lo;
        split2_pair(v[0], v[1], s, hi.x, lo.x);
        split2_pair(v[2], v[3], s, hi.y, lo.y);
        split2_pair(v[4], v[5], s, hi.z, lo.z);
        split2_pair(v[6], v[7], s, hi.w, lo.w);
        out[(no * 2) * hw + p] = hi;
        out[(no * 2 + 1) * hw + p] = lo;
    }
}

// L16 -> fp32 NCHW: (h + l) / s (exact in fp32 up to the final rounding).  Tests / debugging.
__global__ __launch_bounds__(256) void l16_unpack_kernel(const uint4* __restrict__ in, int n, int c, long hw,
                                                         const float* __restrict__ amax, float* __restrict__ x) {
    __shared__ float red[4];
    float m = 0.f;
    for (int i = threadIdx.x; i < fsc::kAmaxFloats; i += 256) m = fmaxf(m, amax[i]);
    m = fsc::wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int f = scale_field(m);
    const float inv = inv_scale(f, m);
    const int oct = (c + 7) / 8;
    const long total = (long)n * oct * hw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long p = idx % hw;
        const long no = idx / hw;
        const int o = (int)(no % oct);
        const long img = no / oct;
        const uint4 hi = in[(no * 2) * hw + p], lo = in[(no * 2 + 1) * hw + p];
        const unsigned hh[4] = {hi.x, hi.y, hi.z, hi.w}, ll[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = o * 8 + e;
            if (ch >= c) continue;
            const unsigned short hb = (unsigned short)(hh[e >> 1] >> ((e & 1) * 16));
            const unsigned short lb = (unsigned short)(ll[e >> 1] >> ((e & 1) * 16));
            const float hv = (float)__builtin_bit_cast(_Float16, hb), lv = (float)__builtin_bit_cast(_Float16, lb);
            x[(img * c + ch) * hw + p] = (hv + lv) * inv;
        }
    }
}

// weight (c_out, c_in, kh, kw) -> A fragments: packed[co block][step][channel tile][limb][lane][8 fp16];
// lane = (kq, m); its 8 values are the channels of octet `oct` at tap `tap`, (tap, oct) = divmod(4 * step_in_chunk
// + kq, octets of the chunk).  Scaled by scale_field(max |w|).  Same format as conv.hip pack_x3_kernel (fp16 limbs).
// One launch packs up to two directions (forward and input gradient) of the same weight: blocks [0, a.blocks) write
// direction a, the rest direction b; both fold the partial maxima of l16_wmax_kernel at `wmax_part`.
struct PackDir {
    unsigned short* packed;
    float* w_amax;           // receives max |w| (read by the conv kernel)
    int cot, co_blocks, nfull, tail_oct, steps, dgrad;
    long total;
    int blocks;
};
__device__ __forceinline__ void pack_dir_block(const float* __restrict__ w, int c_out, int c_in, int taps, const PackDir& dir, int blk,
                                               const float* __restrict__ wmax_part);
__global__ void l16_pack_w_kernel(const float* __restrict__ w, int c_out, int c_in, int taps, PackDir a, PackDir b,
                                  const float* __restrict__ wmax_part) {
    const bool second = (int)blockIdx.x >= a.blocks;
    pack_dir_block(w, c_out, c_in, taps, second ? b : a, second ? blockIdx.x - a.blocks : blockIdx.x, wmax_part);
}

// max |w| in two stages without a clear: kWmaxBlocks partial maxima behind the fragments, folded by every block of the pack
// kernel (and published as w_amax[0] for the conv kernel by its first thread)
__global__ __launch_bounds__(256) void l16_wmax_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
    __shared__ float sm[4];
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    m = fsc::wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// The same two kernels for several weights at once (fsc_conv_l16_pack_weights_multi: one pair of launches per kMultiPack layers
// instead of one per layer -- 40 tiny launches per training step at cfg 2).
constexpr int kMultiPack = 8;
struct PackJob {
    const float* w;
    int c_out, c_in, taps;
    long count;              // elements of w
    float* part;             // kWmaxBlocks partial maxima (behind the first direction's fragments)
    PackDir a, b;            // b.blocks == 0: one direction only
    int first_block;         // of this job in the pack launch
};
struct PackJobs { PackJob j[kMultiPack]; int n; };

__global__ __launch_bounds__(256) void l16_wmax_multi_kernel(PackJobs jobs) {
    __shared__ float sm[4];
    const int job = blockIdx.x / kWmaxBlocks, blk = blockIdx.x - job * kWmaxBlocks;
    const PackJob& pj = jobs.j[job];
    float m = 0.f;
    for (long i = (long)blk * 256 + threadIdx.x; i < pj.count; i += (long)kWmaxBlocks * 256) m = fmaxf(m, fabsf(pj.w[i]));
    m = fsc::wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) pj.part[blk] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

__device__ __forceinline__ void pack_dir_block(const float* __restrict__ w, int c_out, int c_in, int taps, const PackDir& dir, int blk,
                                               const float* __restrict__ wmax_part) {
    float wm = 0.f;
    for (int i = 0; i < kWmaxBlocks; ++i) wm = fmaxf(wm, wmax_part[i]);
    if (blk == 0 && threadIdx.x == 0) dir.w_amax[0] = wm;
    const float sw = field_to_float(scale_field(wm));
    const int cot = dir.cot, steps = dir.steps, nfull = dir.nfull;
    // a thread = the 8 K values of one (fragment, lane): the index arithmetic once, two 16-byte stores (one element per thread
    // cost six integer divisions and a 2-byte store each: 313 us per step for 86 MB of weights)
    for (long idx = (long)blk * blockDim.x + threadIdx.x; idx < (dir.total >> 3); idx += (long)dir.blocks * blockDim.x) {
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
        uint4 hi, lo;
        split2_pair(v[0], v[1], sw, hi.x, lo.x);
        split2_pair(v[2], v[3], sw, hi.y, lo.y);
        split2_pair(v[4], v[5], sw, hi.z, lo.z);
        split2_pair(v[6], v[7], sw, hi.w, lo.w);
        const long base2 = ((((long)cb * steps + S) * cot + i) * 2) * 512 + lane * 8;
        *reinterpret_cast<uint4*>(dir.packed + base2) = hi;
        *reinterpret_cast<uint4*>(dir.packed + base2 + 512) = lo;
    }
}

__global__ void l16_pack_w_multi_kernel(PackJobs jobs) {
    int job = 0;
#pragma unroll 1
    for (int k = 1; k < jobs.n; ++k)
        if ((int)blockIdx.x >= jobs.j[k].first_block) job = k;
    const PackJob& pj = jobs.j[job];
    const int rel = (int)blockIdx.x - pj.first_block;
    const bool second = rel >= pj.a.blocks;
    pack_dir_block(pj.w, pj.c_out, pj.c_in, pj.taps, second ? pj.b : pj.a, second ? rel - pj.a.blocks : rel, pj.part);
}

// -------------------------------------------------------------------------------------------
// Forward / dgrad.  Workgroup = 8 waves, tile = COT*16 output channels x (8 * PT * 16) pixels, persistent over
// (pixel tile, channel block) items.  One MFMA step = one tap x 32 channels: COT * PT * 3 MFMAs per wave.
// POOL: forward of a convolution that is followed by MaxPool2d(2) (classifiers.py:526-532 for the blocks after the stem): a
// wave's 16-pixel tiles are 2 x 8 blocks of the box instead of 16 consecutive pixels, so the four pixels of every pooling window
// sit in lanes lm, lm + 1, lm + 8, lm + 9 of one MFMA column group -- the epilogue pools with two shuffles per value (same
// first-maximum / NaN rule as fsc_maxpool_fwd) and writes the pooled tensor and the window indices; the full-resolution output
// (1 GB at the first such layer of cfg 2) is never written and the separate max-pool pass disappears.  `out` = pooled tensor.
// STATS: the forward of a convolution whose output goes into a BatchNorm (every convolution of a block: classifiers.py:78-101,
// 524-533): the epilogue also accumulates, per lane and output channel, sum (y - pivot), sum (y - pivot)^2, min y, max y of what
// it stores (y = the pooled value with POOL) -- the statistics pass over the output disappears.  A worker keeps ONE channel
// block for all its items (the host makes the worker count a multiple of the channel blocks), so the sums stay in registers
// until the end of the kernel: one float4 record per (worker, wave, channel) in `stat_rec`, folded by
// fsc_bn_records_fold_conv.  pivot = stat_pivot[channel] (the BatchNorm's running mean: close to the batch mean) or 0.
// ACT16 (inference, fsc_conv_l16_fwd_act; conv_l3.hip has the three-limb twin): the epilogue applies an eval-mode BatchNorm's scale /
// shift and PReLU and writes the result as the two-limb L16 operand of the next convolution, scaled by the DECLARED maximum the caller
// brings; same expressions as the two-pass route: bit-identical limbs.
struct ActArgs2 {
    const float* scale;
    const float* shift;
    const float* alpha;
    uint2* out16;
    const float* out_amax;
    unsigned* seen;
};
template <int KH, int KW, int COT, int PT, bool POOL = false, bool STATS = false, bool ACT16 = false>
__global__ __launch_bounds__(kWaves * 64) void conv_l16_fwd_kernel(LGeom g, const uint4* __restrict__ in,
                                                                    const float* __restrict__ packed,
                                                                    const float* __restrict__ bias,
                                                                    float* __restrict__ out, int accumulate,
                                                                    const float* __restrict__ in_amax,
                                                                    const float* __restrict__ w_amax,
                                                                    uint8_t* __restrict__ pool_idx = nullptr,
                                                                    const float* __restrict__ stat_pivot = nullptr,
                                                                    float4* __restrict__ stat_rec = nullptr,
                                                                    ActArgs2 act = ActArgs2{}) {
    static_assert(!ACT16 || (!POOL && !STATS), "ACT16 replaces the plain epilogue");
    constexpr int TAPS = KH * KW;
    constexpr int CO_BLK = COT * 16;
    constexpr int PADH = KH / 2, PADW = KW / 2;
    constexpr int DAHEAD = TAPS == 1 ? 2 : 1;       // the input DMA runs this many chunks ahead
    constexpr int NSTG = DAHEAD + 1;
    constexpr int WUNITS = COT * 2;                 // 1 KB fragment images per step
    constexpr int WSLOT_F = WUNITS * 256;           // floats per ring slot
    constexpr int NWQ = (WUNITS + kWaves - 1) / kWaves;
    constexpr int RING = 4;                         // weight slots: W runs three steps ahead
    constexpr int NWLO = WUNITS / kWaves;           // weight DMAs every wave issues per step (at least)
    constexpr int NPAIR = (COT + 1) / 2;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wring = smem;
    uint4* const ibase = reinterpret_cast<uint4*>(smem + RING * WSLOT_F);
    const int istage = 8 * g.plane;                 // uint4 per stage
    float* const scratch = reinterpret_cast<float*>(ibase + NSTG * istage) + (threadIdx.x >> 6) * (16 * kScr);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kq = lane >> 4;
    const unsigned long long ck0 = __builtin_readcyclecounter(), cr0 = __builtin_amdgcn_s_memrealtime();

#ifndef FSC_L16_PRIO
#define FSC_L16_PRIO 0
#endif
    // the second-dispatched half of the workgroup loses every issue arbitration against its older SIMD partner (priority, then
    // age): a static priority evens the two halves out (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (FSC_L16_PRIO && wid >= 4) __builtin_amdgcn_s_setprio(1);
    // ---- operand scales (before any DMA lands in the ring)
    const float ax = block_amax512(in_amax, smem);
    const float aw = *w_amax;
    const int fx = scale_field(ax), fw = scale_field(aw);
    const float inv_x = inv_scale(fx, ax), inv_w = inv_scale(fw, aw);
    float act_s = 1.f, act_seen = 0.f;
    if constexpr (ACT16) act_s = field_to_float(scale_field(block_amax512(act.out_amax, smem)));

    const float inv_per = 1.0f / (float)(g.rows * g.cols), inv_cols = 1.0f / (float)g.cols;
    const float inv_tw = 1.0f / (float)g.tw, inv_thw = 1.0f / (float)(g.th * g.tw);
    int pix_b[PT];               // byte offset of this lane's pixel inside a staged plane
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wid * PT + pt) * 16 + lm;
        int pl = 0;
        if (p < g.npix) {
            if (POOL) {                                  // tile = a 2 x 8 block: (image, row pair, column octet)
                const int tpr = g.tw >> 3, tpi = (g.th >> 1) * tpr;
                const int t = wid * PT + pt;
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
    // Items = (pixel tile, channel block).  A worker keeps ONE channel block and walks tiles t0, t0 + ts, ... (the worker count is
    // a multiple of the channel blocks).  With g.xcd the `coblk` workers that share a tile are neighbours on one XCD (workgroups
    // go to XCDs round-robin: worker b runs on XCD b % 8), so the second read of a tile's input boxes hits that XCD's L2
    // instead of going out again: tile t belongs to XCD t % 8, local slot (t / 8) % (workers per XCD / coblk).
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
    // g.spat: the item index v (v mod 8 = the XCD of its worker) names tile base(v mod 8) + v / 8 -- an XCD's 32 workers take 32
    // ADJACENT boxes at any time, whose shared halo rows and columns reach that XCD's L2 once (conv_l3.hip: -1 % kernel time)
    const int tq = ntiles >> 3, tr = ntiles & 7;
    auto tile_of = [&](int v) -> int {
        if (!g.spat) return v;
        const int x = v & 7, j = v >> 3;
        return x * tq + (x < tr ? x : tr) + j;
    };
    const int nchunks = g.nfull + (g.tail_oct ? 1 : 0);

    int pos_off[kNptMax];        // source offset (16-byte units, relative to unit 0 of image 0) of staged positions; -1 = zeros
    auto plan_input = [&](int tile) {                       // (once per item: the decode is redone instead of kept in registers)
        int t = tile;
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h; t /= g.tiles_h;
        const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
#pragma unroll
        for (int q = 0; q < kNptMax; ++q) {
            const int pos = q * 64 + lane;
            pos_off[q] = -1;
            if (pos < g.npos) {
                const int per = g.rows * g.cols;
                const int b = fdiv(pos, inv_per), rem = pos - b * per;
                const int rr = fdiv(rem, inv_cols), cc = rem - rr * g.cols;
                const int gh = h0 + rr - PADH, gw = w0 + cc - PADW;
                if (n0 + b < g.n && gh >= 0 && gh < g.h && gw >= 0 && gw < g.w)
                    pos_off[q] = (int)((long)(n0 + b) * g.img_stride + (long)gh * g.w + gw);
            }
        }
    };

    f32x4 acc[COT][PT];
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    PROF_DECL
    constexpr int NST = STATS ? COT : 1;
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

    // ---- DMA issue
    const uint4* const zero = reinterpret_cast<const uint4*>(g_zero16_l);
    auto issue_w = [&](const float* src, int slot) {        // src: this lane's pointer into the step's fragments
        float* dst = wring + slot * WSLOT_F;
#pragma unroll
        for (int q = 0; q < NWQ; ++q) {
            const int u = q * kWaves + wid;
            if ((q + 1) * kWaves <= WUNITS || u < WUNITS) glds16(src + u * 256, dst + u * 256);
        }
    };
    // input of chunk c into `stage`: wave wid copies unit wid = (octet wid / 2, limb wid % 2); always g.npt instructions
    auto issue_i = [&](int stage, int c) {
        const int oct = c * 4 + (wid >> 1);
        const bool live_u = oct < g.oct_in;
        const uint4* src = in + (long)(oct * 2 + (wid & 1)) * g.hw;
        uint4* dst = ibase + stage * istage + wid * g.plane;
#pragma unroll
        for (int q = 0; q < kNptMax; ++q) {
            if (q < g.npt && q * 64 + lane < g.plane) {         // (lanes past the plane would land in the next unit)
                const bool live = live_u && pos_off[q] >= 0;
                glds16(live ? src + pos_off[q] : zero, dst + q * 64);
            }
        }
    };

    // ---- producers: W runs three steps ahead of the MFMA steps, the input DAHEAD chunks, both across items
    int wp_item = t0, wp_left = g.steps, wp_slot = 0;      // wp_left: steps of tile wp_item not issued yet
    const float* const wp_base = packed + (long)cb_w * g.steps * WSLOT_F + lane * 4;
    const float* wp_src = wp_base;
    auto wp_set_item = [&]() { wp_src = wp_base; };
    auto produce_w = [&]() -> bool {
        if (wp_item >= ntiles) return false;
        issue_w(wp_src, wp_slot);
        wp_src += WSLOT_F;
        wp_slot = wp_slot == RING - 1 ? 0 : wp_slot + 1;
        if (--wp_left == 0) {
            wp_left = g.steps;
            wp_item += ts;
            wp_set_item();
        }
        return true;
    };
    int ip_item = t0, ip_c = 0, ip_stg = 0;
    auto produce_i = [&]() -> bool {
        if (ip_item >= ntiles) return false;
        issue_i(ip_stg, ip_c);
        ip_stg = ip_stg == NSTG - 1 ? 0 : ip_stg + 1;
        if (++ip_c == nchunks) {
            ip_c = 0;
            ip_item += ts;
            if (ip_item < ntiles) plan_input(tile_of(ip_item));
        }
        return true;
    };

    // ---- B operand address of (stage, chunk, step) for this lane: byte offset from ibase of its octet's high limb
    //      at the step's tap (without the pixel).  Full chunks: lane group kq = octet kq, uniform tap.
    const int limb_b = g.plane * 16;                        // bytes between the limb planes of an octet
    const int stage_b = 8 * limb_b;
    const int kq_b = kq * 2 * limb_b;
    auto b_off_tail = [&](int stage, int s) -> int {        // remainder chunk: (tap, octet) flattened over lane groups
        const int noct = g.tail_oct;
        int gi = 4 * s + kq;
        if (gi >= TAPS * noct) gi = 0;                     // its weights are zero
        const int tap = TAPS == 1 ? 0 : fdiv(gi, 1.0f / (float)noct);
        const int oct = gi - tap * noct;
        const int ty = TAPS == 1 ? 0 : fdiv(tap, 1.0f / (float)KW), tx = tap - ty * KW;
        return stage * stage_b + oct * 2 * limb_b + (ty * g.cols + tx) * 16;
    };
    auto b_off = [&](int stage, int c, int s) -> int {      // (stage, c, s uniform)
        if (c >= g.nfull) return b_off_tail(stage, s);
        const int ty = TAPS == 1 ? 0 : (s * 11) >> 5, tx = s - ty * KW;          // s / 3 for s < 10
        return stage * stage_b + (ty * g.cols + tx) * 16 + kq_b;
    };

    struct Frag { u32x4 v[2][PT]; };                        // B fragments of a step: [limb][pixel tile]
    struct AFrag { u32x4 v[2][2]; };                        // A fragments of a pair of channel tiles: [tile][limb]
    Frag fb0, fb1;
    AFrag fa0, fa1;
    const char* const ib = reinterpret_cast<const char*>(ibase);
    auto read_b = [&](int off, Frag& dst) {
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            dst.v[0][j] = *reinterpret_cast<const u32x4*>(ib + off + pix_b[j]);
            dst.v[1][j] = *reinterpret_cast<const u32x4*>(ib + off + pix_b[j] + limb_b);
        }
    };
    auto read_a = [&](const u32x4* wl, int pr, AFrag& dst) {      // wl: this lane's fragments of a ring slot
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int l = 0; l < 2; ++l)
                if (pr * 2 + t < COT) dst.v[t][l] = wl[((pr * 2 + t) * 2 + l) * 64];
    };

    int item = t0;                 // (the tile of the current item)
    int slot = 0;
    if (item < ntiles) {
        plan_input(tile_of(item));
        wp_set_item();
#pragma unroll
        for (int d = 0; d < DAHEAD; ++d) produce_i();
        produce_w();
        produce_w();
        if (TAPS == 1) produce_w();                          // (3x3: the first pair's two slots; each lead tap issues the next pair)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_barrier();
        read_b(b_off(0, 0, 0), fb0);
        read_a(reinterpret_cast<const u32x4*>(wring) + lane, 0, fa0);
    }

    // consumer position (all uniform): chunk c, step sc of nst inside it, input stage stg; wait state
    int c = 0, sc = 0, nst = g.nfull > 0 ? TAPS : g.tail_steps, stg = 0;
    bool first_step = true, drain = false;

    // One MFMA step = one tap x 32 channels.  3x3 kernels synchronise every TWO taps (TWO): the workgroup barrier, the DMA
    // issue and the wait for the weights cost ~1150 cycles per barrier whatever the number of MFMAs behind it (measured: step
    // time 2260 cycles at 24 MFMAs per wave, 3090 at 42), so a `lead` tap (barrier, DMA issue for the next pair, its first A
    // fragments read fresh) is followed by a `follow` tap that runs straight on from registers: its B fragments and first A
    // fragments were read behind the lead's MFMAs (both weight slots of a pair land before its barrier).  1x1 kernels (every
    // tap opens a chunk) keep one barrier per tap with counted waits and read the next tap's first A fragments across it.
    // SPREAD (3x3): the copies of a pair are not issued in one burst behind the barrier -- eight waves x ~5 KB at once queue up in
    // front of the CU's one-line-per-clock vector memory path (~40 KB = ~650 cycles during which every wave that has a copy to issue
    // stands still, and both waves of a SIMD do so together) -- but piecewise between the MFMA groups of the lead tap: the next
    // pair's first weight slot after channel-tile pair 0, its second after pair 1, the input box (when a chunk opens) after pair 2.
    constexpr bool TWO = TAPS > 1;
#ifndef FSC_L16_SPREAD
#define FSC_L16_SPREAD 1
#endif
    constexpr bool SPREAD = TWO && FSC_L16_SPREAD != 0;
    auto step = [&](bool lead, bool has_follow, const Frag& bcur, Frag& bnxt, const AFrag& acur, AFrag& anxt) {
        bool opens = false;
        AFrag t0, t1, af;
        PROF_MARK();
        if (!TWO) {
            if (!first_step) {
                // the barrier covers W(S+1) (issued two steps ago); W(S+2) may stay in flight.  Stores share the counter and
                // retire out of order, and a dry weight producer leaves nothing younger: drain then.
                if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWLO) : "memory");
                raw_barrier();
            }
            first_step = false;
            produce_i();
            drain = !produce_w();
        } else if (lead) {
            if (!first_step) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // everything issued a pair ago has landed (weights of this
                raw_barrier();                                       // pair, a young input box, the previous item's stores)
            }
            PROF_ADD(0);
#ifndef FSC_L16_EARLY_A
#define FSC_L16_EARLY_A 1
#endif
            if (FSC_L16_EARLY_A) {        // this tap's first A fragments are on their way while the scalar bookkeeping below runs
                read_a(reinterpret_cast<const u32x4*>(wring + slot * WSLOT_F) + lane, 0, af);
                __builtin_amdgcn_sched_barrier(0);
            }
            first_step = false;
            // weights of the NEXT pair into the two slots the previous pair used; the input box DAHEAD chunks ahead when a
            // chunk opens in either tap of this pair
            opens = sc == 0 || (has_follow && sc + 1 == nst);
            if (!SPREAD) {
                produce_w();
                if (has_follow) produce_w();
                if (opens) produce_i();
            }
        }
        // the next step and the byte offset of its B operand
        int noff;
        {
            int sn = sc + 1;
            if (sn < nst) {
                noff = b_off(stg, c, sn);
            } else {                                        // first step of the next chunk (of the next item at the end)
                sn = 0;
                stg = stg == NSTG - 1 ? 0 : stg + 1;
                c = c + 1 < nchunks ? c + 1 : 0;
                nst = c < g.nfull ? TAPS : g.tail_steps;
                noff = b_off(stg, c, 0);
            }
            sc = sn;
        }
        const u32x4* wl = reinterpret_cast<const u32x4*>(wring + slot * WSLOT_F) + lane;
        slot = slot == RING - 1 ? 0 : slot + 1;
        const u32x4* wln = reinterpret_cast<const u32x4*>(wring + slot * WSLOT_F) + lane;
        const bool fresh_a = TWO && lead;                    // (uniform) this tap's first pair is read now, behind the barrier
        const bool next_a = TWO ? (lead && has_follow) : true;   // the next tap's first pair can be read at the end of this one
        if (fresh_a && !FSC_L16_EARLY_A) read_a(wl, 0, af);
        PROF_ADD(1);
        constexpr int kLa[3] = {1, 0, 0}, kLb[3] = {0, 1, 0};     // (A limb, B limb): l*h, h*l, h*h -- smallest first
#pragma unroll
        for (int pr = 0; pr < NPAIR; ++pr) {
            const AFrag& a = pr == 0 ? (fresh_a ? af : acur) : (pr & 1) ? t1 : t0;
            AFrag& an = pr + 1 == NPAIR ? anxt : (pr & 1) ? t0 : t1;
            // reads behind this pair's MFMAs: the next pair's A fragments (the next tap's first pair from the next slot at
            // the end) and, with the first pair, the next tap's B fragments
            if (pr + 1 < NPAIR) read_a(wl, pr + 1, an);
            else if (next_a) read_a(wln, 0, an);
            if (pr == 0) read_b(noff, bnxt);
            const int ntile = pr * 2 + 1 < COT ? 2 : 1;
#pragma unroll
            for (int gq = 0; gq < 3; ++gq)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int i = pr * 2 + t;
                    if (i < COT) {
#pragma unroll
                        for (int j = 0; j < PT; ++j) acc[i][j] = mfma16(a.v[t][kLa[gq]], bcur.v[kLb[gq]][j], acc[i][j]);
                    }
                }
            // one LDS read behind each of the first MFMAs of the pair
            const int nrd = (pr + 1 < NPAIR ? (((pr + 1) * 2 + 1 < COT) ? 4 : 2) : (COT > 1 ? 4 : 2)) + (pr == 0 ? 2 * PT : 0);
#pragma unroll
            for (int k = 0; k < 3 * ntile * PT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // one MFMA
                if (k < nrd) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read
            }
            __builtin_amdgcn_sched_barrier(0);
            if (SPREAD && lead) {
                if (pr == 0) produce_w();
                if (pr == (NPAIR > 1 ? 1 : 0) && has_follow) produce_w();
                if (pr == (NPAIR > 2 ? 2 : NPAIR - 1) && opens) produce_i();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        PROF_ADD(2);
#ifdef FSC_L16_PROFILE
        pf_acc[4] += 1;
#endif
    };

    const bool add_bias = bias != nullptr;
    for (; item < ntiles; item += ts) {
        const int tile = tile_of(item);
        const int co0 = cb_w * CO_BLK;
#pragma unroll 1
        for (int S = 0; S + 1 < g.steps; S += 2) {
            step(true, true, fb0, fb1, fa0, fa1);
            step(false, false, fb1, fb0, fa1, fa0);
        }
        if (g.steps & 1) {                                  // odd number of steps: the next item starts from fb0 / fa0 again
            step(true, false, fb0, fb1, fa0, fa1);
            fb0 = fb1;
            fa0 = fa1;
        }

        PROF_MARK();
        if constexpr (POOL) {
            // ---- pooled epilogue.  After the shuffles the four even lanes lm = 0, 2, 4, 6 of a column group hold the pooled
            //      value and window index of 4 channels x one window; through the scratch tile lane L = (channel L >> 2,
            //      window L & 3) stores four bytes of four consecutive pooled pixels of a channel.
            int t = tile;
            const int twi = t % g.tiles_w; t /= g.tiles_w;
            const int thi = t % g.tiles_h; t /= g.tiles_h;
            const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
            const int oh = g.h >> 1, ow = g.w >> 1;
            const int tpr = g.tw >> 3, tpi = (g.th >> 1) * tpr;
            long pool_g[PT];                                // pooled offset (channel 0) of this lane's window; -1 = outside
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int tt = wid * PT + pt;
                pool_g[pt] = -1;
                if (tt * 16 < g.npix) {
                    const int b = tt / tpi, rem = tt - b * tpi;
                    const int tr = rem / tpr, tc = rem - tr * tpr;
                    const int pr = (h0 >> 1) + tr, pc = (w0 >> 1) + 4 * tc + (lane & 3);
                    if (n0 + b < g.n && pr < oh && pc < ow) pool_g[pt] = ((long)(n0 + b) * g.cout * oh + pr) * ow + pc;
                }
            }
            const long ohw = (long)oh * ow;
            const float* bias_p = bias;
            asm volatile("" : "+s"(bias_p));
            const int chp = lane >> 2;
#pragma unroll
            for (int i = 0; i < COT; ++i) {
                float bv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cob = co0 + i * 16 + kq * 4 + r;
                    bv[r] = (bias_p != nullptr && cob < g.cout) ? bias_p[cob] : 0.f;
                }
                const int co = co0 + i * 16 + chp;
                float pv = 0.f;
                if (STATS && stat_pivot != nullptr && co < g.cout) pv = stat_pivot[co];
#pragma unroll
                for (int j = 0; j < PT; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v0 = fmaf(acc[i][j][r] * inv_x, inv_w, bv[r]);
                        const float v1 = dpp_xor1(v0);
                        const float v2 = dpp_xor8(v0);
                        const float v3 = dpp_xor8(v1);
                        float best = v0;                     // first maximum in window order, NaN wins (fsc_maxpool_fwd)
                        int bi = 0;
                        if (v1 > best || v1 != v1) { best = v1; bi = 1; }
                        if ((v2 > best || v2 != v2) && best == best) { best = v2; bi = 2; }
                        if ((v3 > best || v3 != v3) && best == best) { best = v3; bi = 3; }
                        if ((lm & 9) == 0) {                 // lanes lm = 0, 2, 4, 6: the window's first pixel
                            scratch[(kq * 4 + r) * kScr + (lm >> 1)] = best;
                            scratch[(kq * 4 + r) * kScr + 8 + (lm >> 1)] = __int_as_float(bi);
                        }
                    }
                    const float val = scratch[chp * kScr + (lane & 3)];
                    const int bidx = __float_as_int(scratch[chp * kScr + 8 + (lane & 3)]);
                    if (co < g.cout && pool_g[j] >= 0) {
                        out[pool_g[j] + (long)co * ohw] = val;
                        pool_idx[pool_g[j] + (long)co * ohw] = (uint8_t)bidx;
                        if (STATS) stat_add(i, val, pv);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < COT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            drain = true;
            PROF_ADD(3);
            continue;
        }
        if constexpr (ACT16) {
            // ---- affine + PReLU + limb split (conv_l3.hip): lane (kq, lm) holds channels kq * 4 + r of pixel lm = one 8-byte half of
            //      the (octet, limb) vector of that pixel
            long gpix[PT];
            {
                int t = tile;
                const int twi = t % g.tiles_w; t /= g.tiles_w;
                const int thi = t % g.tiles_h; t /= g.tiles_h;
                const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
                const long img_u = (long)((g.cout + 7) >> 3) * 2 * g.hw;
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const int p = (wid * PT + pt) * 16 + lm;
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
            long hw_a = g.hw;
            const float* bias_a = bias;
            asm volatile("" : "+s"(hw_a), "+s"(bias_a));
            const int oct_out = (g.cout + 7) >> 3;
            const bool has_aff = act.scale != nullptr, has_alpha = act.alpha != nullptr;
#pragma unroll
            for (int i = 0; i < COT; ++i) {
                float bv[4], sc[4], sh[4], al[4];
                bool okc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cob = co0 + i * 16 + kq * 4 + r;
                    okc[r] = cob < g.cout;
                    bv[r] = (add_bias && okc[r]) ? bias_a[cob] : 0.f;
                    sc[r] = (has_aff && okc[r]) ? act.scale[cob] : 1.f;
                    sh[r] = (has_aff && okc[r]) ? act.shift[cob] : 0.f;
                    al[r] = (has_alpha && okc[r]) ? act.alpha[cob] : 0.f;
                }
                const int oct = ((co0 + i * 16) >> 3) + (kq >> 1);
                if (oct < oct_out) {
#pragma unroll
                    for (int j = 0; j < PT; ++j) {
                        if (gpix[j] < 0) continue;
                        float y[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float z = fmaf(acc[i][j][r] * inv_x, inv_w, bv[r]);
                            const float t = has_aff ? fmaf(z, sc[r], sh[r]) : z;
                            const float v = (has_alpha && !(t > 0.f)) ? al[r] * t : t;
                            y[r] = okc[r] ? v : 0.f;
                            act_seen = fmaxf(act_seen, fabsf(y[r]));
                        }
                        unsigned h0, l0, h1, l1;
                        split2_pair(y[0], y[1], act_s, h0, l0);
                        split2_pair(y[2], y[3], act_s, h1, l1);
                        uint2* o = act.out16 + (gpix[j] + (long)oct * 2 * hw_a) * 2 + (kq & 1);
                        o[0] = make_uint2(h0, h1);
                        o[2 * hw_a] = make_uint2(l0, l1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < COT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            drain = true;
            PROF_ADD(3);
            continue;
        }
        // ---- epilogue: D row = channel (kq*4 + r), column = pixel (lm) -> scratch[ch][px] -> lane = (channel, quad).
        //      This lane's quads (four consecutive pixels of a box row) are decoded here, once per item.
        long quad_g[PT];
        int quad_ok[PT];
        {
            int t = tile;
            const int twi = t % g.tiles_w; t /= g.tiles_w;
            const int thi = t % g.tiles_h; t /= g.tiles_h;
            const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int p = (wid * PT + pt) * 16 + (lane & 3) * 4;
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
        const int ch = lane >> 2;
#pragma unroll
        for (int i = 0; i < COT; ++i) {
            float bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cob = co0 + i * 16 + kq * 4 + r;
                bv[r] = (add_bias && cob < g.cout) ? bias_t[cob] : 0.f;
            }
            const int co = co0 + i * 16 + ch;
            float pv = 0.f;
            if (STATS && stat_pivot != nullptr && co < g.cout) pv = stat_pivot[co];
#pragma unroll
            for (int j = 0; j < PT; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) scratch[(kq * 4 + r) * kScr + lm] = fmaf(acc[i][j][r] * inv_x, inv_w, bv[r]);
                const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + ch * kScr + (lane & 3) * 4);
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
#pragma unroll
        for (int i = 0; i < COT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        drain = true;                                       // (1x1) the stores above share the DMA counter
        PROF_ADD(3);
    }
#ifdef FSC_L16_PROFILE
    if (blockIdx.x == 0 && lane == 0) {
        pf_acc[5] = prof_now() - pf_k0;
        pf_acc[6] = __builtin_amdgcn_s_memrealtime() - pf_r0;          // constant 100 MHz reference clock
        for (int i = 0; i < 7; ++i) atomicAdd(&g_l16_prof[wid][i], pf_acc[i]);
    }
#endif
    if (blockIdx.x == 0 && tid == 0) {
        g_l16_clock[0] = __builtin_readcyclecounter() - ck0;
        g_l16_clock[1] = __builtin_amdgcn_s_memrealtime() - cr0;
    }
    if constexpr (ACT16) {
        act_seen = fsc::wave_max(act_seen);
        if (act.seen != nullptr && lane == 0) atomicMax(act.seen, __float_as_uint(act_seen));
    }
    if constexpr (STATS) {
        // the four lanes of a quad hold the same channel: fold them, lane (lane & 3) == 0 writes the record
#pragma unroll
        for (int i = 0; i < COT; ++i) {
            float a = st_s1[i], b = st_s2[i], mn = st_mn[i], mx = st_mx[i];
            a += dpp_xor1(a); a += dpp_xor2(a);
            b += dpp_xor1(b); b += dpp_xor2(b);
            mn = fminf(mn, dpp_xor1(mn)); mn = fminf(mn, dpp_xor2(mn));
            mx = fmaxf(mx, dpp_xor1(mx)); mx = fmaxf(mx, dpp_xor2(mx));
            if ((lane & 3) == 0)
                stat_rec[((long)blockIdx.x * kWaves + wid) * CO_BLK + i * 16 + (lane >> 2)] = make_float4(a, b, mn, mx);
        }
    }
}

// -------------------------------------------------------------------------------------------
// host-side planning
struct LPlan {
    LGeom g;
    int cot, pt, co_blocks;
    size_t lds_bytes;
    long tiles, workers;
};

bool plan_l16_pt(const fsc_conv_desc& d_in, int dgrad, int pt, int max_cot, LPlan* out, bool pool = false) {
    LPlan p{};
    LGeom& g = p.g;
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
    g.img_stride = (long)g.oct_in * 2 * g.hw;
    if ((long)g.n * g.img_stride >= (1L << 31)) return false;
    if (g.w >= 1024 && taps > 1) return false;
    const int nstg = taps == 1 ? 3 : 2;
    // channel tiles per workgroup: a step costs its MFMAs plus a fixed part (barrier, DMA issue, B reads) worth about
    // one tile; odd tile counts leave half a pair of the MFMA schedule idle only in the A prefetch, not in MFMAs
    const int tiles = fsc::ceil_div(g.cout, 16);
    int best_cot = 1, best_blocks = tiles;
    long best_cost = -1;
    const int force_cot = fsc::env().l16_cot;               // development (FSC_L16_COT): force the channel tiles per workgroup
    for (int cot = 1; cot <= max_cot; ++cot) {
        if (force_cot && cot != force_cot) continue;
        const int blocks = fsc::ceil_div(tiles, cot);
        const long cost = (long)blocks * (cot * 3 + 4);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && blocks < best_blocks)) {
            best_cot = cot; best_blocks = blocks; best_cost = cost;
        }
    }
    p.cot = best_cot;
    p.co_blocks = best_blocks;
    p.pt = pt;
    const int pix_cap = kWaves * pt * 16;
    const size_t lds_total = 160 * 1024;
    const size_t ring = (size_t)4 * p.cot * 2 * 1024;
    const size_t scratch = (size_t)kWaves * 16 * kScr * sizeof(float);
    if (ring + scratch + (size_t)nstg * 128 * 64 > lds_total) return false;
    int cap_pos = (int)((lds_total - ring - scratch) / ((size_t)nstg * 128));
    cap_pos &= ~7;
    if (cap_pos > 64 * kNptMax) cap_pos = 64 * kNptMax;
    long bcost = -1;
    int bnb = 1, bth = 1, btw = 1;
    for (int tw = 4; tw <= ((d.w + 3) & ~3) && tw <= pix_cap; tw += 4) {
        if (pool && (tw & 7)) continue;                  // pooled epilogue: tiles are 2 x 8 blocks of the box
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
    if ((double)d.n * g.hw < (pt == 2 ? 0.7 : 0.55) * (double)p.tiles * pix_cap) return false;
    g.plane = (g.npos + 7) & ~7;
    g.npt = fsc::ceil_div(g.plane, 64);
    if (g.npt > kNptMax) return false;
    const int rem = g.cin % kChunk;
    g.nfull = g.cin / kChunk + (rem > kChunk - 8 ? 1 : 0);
    g.tail_oct = (rem > 0 && rem <= kChunk - 8) ? fsc::ceil_div(rem, 8) : 0;
    g.tail_steps = fsc::ceil_div(taps * g.tail_oct, 4);
    g.steps = g.nfull * taps + g.tail_steps;
    if (g.steps < 3) return false;
    g.coblk = p.co_blocks;
    p.lds_bytes = ring + (size_t)nstg * 128 * g.plane + scratch;
    if (p.lds_bytes > lds_total) return false;
    const long items = p.tiles * p.co_blocks;
    if (items < 128) return false;                  // small late layers: the split-K kernels of conv.hip fill the chip better
    p.workers = items < 256 ? items : 256;
    if (items > 256 && items <= 512) p.workers = (items + 1) / 2;     // two items each instead of 256 + a short second wave
    p.workers -= p.workers % p.co_blocks;           // a worker keeps one channel block
    g.xcd = (p.co_blocks > 1 && p.workers % (8 * p.co_blocks) == 0 && !fsc::env().l16_no_xcd) ? 1 : 0;
    g.spat = (g.xcd || (p.co_blocks == 1 && p.workers % 8 == 0 && !fsc::env().l16_no_xcd)) ? 1 : 0;
    *out = p;
    return true;
}

bool plan_l16(const fsc_conv_desc& d, int dgrad, LPlan* out) {
    if (d.arith != 3) return false;                          // (the entry points resolve FSC_ARITH_DEFAULT: FSC_RESOLVE_DESC)
    if (fsc::env().no_l16) return false;
    const int taps = d.kh * d.kw;
    const int force_pt = fsc::env().l16_pt;                 // development (FSC_L16_PT): force the pixel tiles per wave
    // (9 / 10 tiles per block leave too little LDS for a 256-pixel box: 150-channel layers run 7 % faster as 2 x 5 tiles)
    if ((!force_pt || force_pt == 2) && plan_l16_pt(d, dgrad, 2, 8, out)) return true;
    if (force_pt == 2) return false;
    return plan_l16_pt(d, dgrad, 1, taps == 1 ? 8 : 10, out);
}

// forward convolution fused with the 2 x 2 max-pool behind it: the same channel tiling as plan_l16 (the packed weights are
// shared), a box of 2 x 8 blocks
bool plan_l16_pool(const fsc_conv_desc& d, LPlan* out) {
    if (d.arith != 3) return false;
    if (fsc::env().no_l16 || fsc::env().no_l16_pool) return false;
    if (d.kh != 3 || d.kw != 3 || d.h < 2 || d.w < 8) return false;
    LPlan plain;
    if (!plan_l16(d, 0, &plain) || plain.pt != 2) return false;
    if (!plan_l16_pt(d, 0, 2, 8, out, true)) return false;
    return out->cot == plain.cot && out->co_blocks == plain.co_blocks && out->g.steps == plain.g.steps &&
           out->cot >= 4 && out->cot <= 8;
}

size_t l16_limb_floats(const LPlan& p) { return (size_t)p.co_blocks * p.g.steps * p.cot * 2 * 256; }

// statistics records (STATS kernels): the worker count is cut to a multiple of the channel blocks so that a worker keeps its block
struct StatArgs { const float* pivot; float4* rec; };
unsigned stat_workers(const LPlan& p) { return (unsigned)p.workers; }
bool stats_ok(const LPlan& p) { return p.cot <= 8 && p.workers >= p.co_blocks; }

template <int KH, int KW, int COT, int PT>
int launch_l16(const LPlan& p, const uint4* in, const float* packed, const float* bias, float* out, int accumulate,
               const float* in_amax, hipStream_t st, StatArgs sa = StatArgs{nullptr, nullptr}, ActArgs2 act = ActArgs2{}) {
    const float* w_amax = packed + l16_limb_floats(p);
    if (act.out16) {
        auto kern = conv_l16_fwd_kernel<KH, KW, COT, PT, false, false, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        hipLaunchKernelGGL(kern, dim3((unsigned)p.workers), dim3(kWaves * 64), p.lds_bytes, st, p.g, in, packed, bias, (float*)nullptr, 0,
                           in_amax, w_amax, (uint8_t*)nullptr, (const float*)nullptr, (float4*)nullptr, act);
        FSC_LAUNCH_CHECK("fsc_conv_l16_fwd_act");
        return 0;
    }
    if constexpr (COT <= 8) {
        if (sa.rec) {
            auto kern = conv_l16_fwd_kernel<KH, KW, COT, PT, false, true>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
            hipLaunchKernelGGL(kern, dim3(stat_workers(p)), dim3(kWaves * 64), p.lds_bytes, st, p.g, in, packed, bias, out, 0, in_amax,
                               w_amax, (uint8_t*)nullptr, sa.pivot, sa.rec, ActArgs2{});
            FSC_LAUNCH_CHECK("fsc_conv_l16_fwd_stats");
            return 0;
        }
    }
    auto kern = conv_l16_fwd_kernel<KH, KW, COT, PT, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)p.workers), dim3(kWaves * 64), p.lds_bytes, st, p.g, in, packed, bias, out,
                       accumulate, in_amax, w_amax, (uint8_t*)nullptr, (const float*)nullptr, (float4*)nullptr, ActArgs2{});
    FSC_LAUNCH_CHECK("fsc_conv_l16_fwd");
    return 0;
}

template <int COT>
int launch_l16_pool(const LPlan& p, const uint4* in, const float* packed, const float* bias, float* pooled, uint8_t* idx,
                    const float* in_amax, hipStream_t st, StatArgs sa = StatArgs{nullptr, nullptr}) {
    const float* w_amax = packed + l16_limb_floats(p);
    if (sa.rec) {
        auto kern = conv_l16_fwd_kernel<3, 3, COT, 2, true, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        hipLaunchKernelGGL(kern, dim3(stat_workers(p)), dim3(kWaves * 64), p.lds_bytes, st, p.g, in, packed, bias, pooled, 0, in_amax,
                           w_amax, idx, sa.pivot, sa.rec, ActArgs2{});
        FSC_LAUNCH_CHECK("fsc_conv_l16_pool_fwd_stats");
        return 0;
    }
    auto kern = conv_l16_fwd_kernel<3, 3, COT, 2, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)p.workers), dim3(kWaves * 64), p.lds_bytes, st, p.g, in, packed, bias, pooled, 0,
                       in_amax, w_amax, idx, (const float*)nullptr, (float4*)nullptr, ActArgs2{});
    FSC_LAUNCH_CHECK("fsc_conv_l16_pool_fwd");
    return 0;
}

template <int KH, int KW, int COT>
int launch_l16_pt(const LPlan& p, const uint4* in, const float* packed, const float* bias, float* out, int accumulate,
                  const float* in_amax, hipStream_t st, StatArgs sa = StatArgs{nullptr, nullptr}, ActArgs2 act = ActArgs2{}) {
    if (p.pt == 2) return launch_l16<KH, KW, COT, 2>(p, in, packed, bias, out, accumulate, in_amax, st, sa, act);
    return launch_l16<KH, KW, COT, 1>(p, in, packed, bias, out, accumulate, in_amax, st, sa, act);
}

template <int KH, int KW>
int launch_l16_cot(const LPlan& p, const uint4* in, const float* packed, const float* bias, float* out, int accumulate,
                   const float* in_amax, hipStream_t st, StatArgs sa = StatArgs{nullptr, nullptr}, ActArgs2 act = ActArgs2{}) {
    switch (p.cot) {
#ifndef FSC_L16_DEV          // (development builds: only the instantiations of the cfg-2 layers, a third of the compile time)
        case 3: return launch_l16_pt<KH, KW, 3>(p, in, packed, bias, out, accumulate, in_amax, st, sa, act);
        case 4: return launch_l16_pt<KH, KW, 4>(p, in, packed, bias, out, accumulate, in_amax, st, sa, act);
        case 6: return launch_l16_pt<KH, KW, 6>(p, in, packed, bias, out, accumulate, in_amax, st, sa, act);
        case 9: if constexpr (KH * KW > 1) return launch_l16_pt<KH, KW, 9>(p, in, packed, bias, out, accumulate, in_amax, st, sa, act); break;
        case 10: if constexpr (KH * KW > 1) return launch_l16_pt<KH, KW, 10>(p, in, packed, bias, out, accumulate, in_amax, st, sa, act); break;
#endif
        case 5: return launch_l16_pt<KH, KW, 5>(p, in, packed, bias, out, accumulate, in_amax, st, sa, act);
        case 7: return launch_l16_pt<KH, KW, 7>(p, in, packed, bias, out, accumulate, in_amax, st, sa, act);
        case 8: return launch_l16_pt<KH, KW, 8>(p, in, packed, bias, out, accumulate, in_amax, st, sa, act);
        default: break;
    }
    fsc::set_error("fsc_conv_l16_fwd: internal: no instantiation for %d channel tiles", p.cot);
    return 22;
}

bool valid_l16_desc(const fsc_conv_desc* d) {
    if (!d || d->n <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->h <= 0 || d->w <= 0) return false;
    const long big = 1L << 31;
    return (long)d->n * d->c_in * d->h * d->w < big && (long)d->n * d->c_out * d->h * d->w < big;
}

}  // namespace

extern "C" {

size_t fsc_l16_bytes(int n, int c, long hw) { return (size_t)n * ((c + 7) / 8) * 2 * (size_t)hw * 16; }

int fsc_l16_pack(const float* x, int n, int c, long hw, const float* amax, void* out, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && amax && out && n > 0 && c > 0 && hw > 0, "fsc_l16_pack: bad arguments");
    const long total = (long)n * ((c + 7) / 8) * hw;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(l16_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, fsc::as_stream(stream), x, n, c, hw, amax,
                       reinterpret_cast<uint4*>(out));
    FSC_LAUNCH_CHECK("fsc_l16_pack");
    return 0;
}

int fsc_l16_unpack(const void* in, int n, int c, long hw, const float* amax, float* x, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && amax && in && n > 0 && c > 0 && hw > 0, "fsc_l16_unpack: bad arguments");
    const long total = (long)n * ((c + 7) / 8) * hw;
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(l16_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, fsc::as_stream(stream),
                       reinterpret_cast<const uint4*>(in), n, c, hw, amax, x);
    FSC_LAUNCH_CHECK("fsc_l16_unpack");
    return 0;
}

/* the same three with the limb format spelled out: limbs = 2 (scaled fp16 pairs, arith 3), 3 (exact bf16 triples, arith 9 / 8 / 6:
 * no scale, `amax` is not read and may be NULL) or 4 (FSC_L16_F16X3: scaled fp16 TRIPLES, arith 10) */
size_t fsc_l16_bytes_limbs(int n, int c, long hw, int limbs) {
    return limbs == 3 || limbs == 4 ? fsc::l3::tensor_bytes(n, c, hw) : fsc_l16_bytes(n, c, hw);
}

int fsc_l16_pack_limbs(const float* x, int n, int c, long hw, const float* amax, int limbs, void* out, fsc_stream_t stream) {
    FSC_CHECK_ARG(limbs >= 2 && limbs <= 4, "fsc_l16_pack_limbs: limbs must be 2, 3 or 4 (three scaled fp16 limbs)");
    if (limbs == 2) return fsc_l16_pack(x, n, c, hw, amax, out, stream);
    FSC_CHECK_ARG(x && out && n > 0 && c > 0 && hw > 0 && (limbs == 3 || amax), "fsc_l16_pack_limbs: bad arguments");
    return fsc::l3::pack(x, n, c, hw, limbs == 4 ? amax : nullptr, out, fsc::as_stream(stream));
}

int fsc_l16_unpack_limbs(const void* in, int n, int c, long hw, const float* amax, int limbs, float* x, fsc_stream_t stream) {
    FSC_CHECK_ARG(limbs >= 2 && limbs <= 4, "fsc_l16_unpack_limbs: limbs must be 2, 3 or 4 (three scaled fp16 limbs)");
    if (limbs == 2) return fsc_l16_unpack(in, n, c, hw, amax, x, stream);
    FSC_CHECK_ARG(x && in && n > 0 && c > 0 && hw > 0 && (limbs == 3 || amax), "fsc_l16_unpack_limbs: bad arguments");
    return fsc::l3::unpack(in, n, c, hw, limbs == 4 ? amax : nullptr, x, fsc::as_stream(stream));
}

int fsc_conv_l16_supported(const fsc_conv_desc* d, int dgrad) {
    FSC_RESOLVE_DESC(d)
    if (d && l16::is_l3(d->arith)) return fsc::l3::supported(d, dgrad);
    LPlan p;
    return valid_l16_desc(d) && plan_l16(*d, dgrad, &p) ? 1 : 0;
}

size_t fsc_conv_l16_packed_floats(const fsc_conv_desc* d, int dgrad) {
    FSC_RESOLVE_DESC(d)
    if (d && l16::is_l3(d->arith)) return fsc::l3::packed_floats(d, dgrad);
    LPlan p;
    if (!valid_l16_desc(d) || !plan_l16(*d, dgrad, &p)) return 0;
    return l16_limb_floats(p) + 4 + kWmaxBlocks;
}

static PackDir make_pack_dir(const LPlan& p, float* packed, int dgrad) {
    PackDir r{};
    r.packed = reinterpret_cast<unsigned short*>(packed);
    r.w_amax = packed + l16_limb_floats(p);
    r.cot = p.cot; r.co_blocks = p.co_blocks; r.nfull = p.g.nfull; r.tail_oct = p.g.tail_oct; r.steps = p.g.steps;
    r.dgrad = dgrad;
    r.total = (long)p.co_blocks * p.g.steps * p.cot * 512;
    long xb = ((r.total >> 3) + 255) / 256;                  // a thread packs 8 values
    r.blocks = (int)(xb > 4096 ? 4096 : xb);
    return r;
}

/* packs the forward (packed_fwd) and / or input-gradient (packed_dgrad) fragments of one weight in two launches total */
int fsc_conv_l16_pack_weights_pair(const fsc_conv_desc* d, const float* weight, float* packed_fwd, float* packed_dgrad,
                                   fsc_stream_t stream) {
    FSC_RESOLVE_DESC(d)
    if (d && l16::is_l3(d->arith)) return fsc::l3::pack_weights_pair(d, weight, packed_fwd, packed_dgrad, fsc::as_stream(stream));
    FSC_CHECK_ARG(valid_l16_desc(d) && weight && (packed_fwd || packed_dgrad), "fsc_conv_l16_pack_weights_pair: bad arguments");
    LPlan pf{}, pd{};
    FSC_CHECK_ARG(!packed_fwd || plan_l16(*d, 0, &pf), "fsc_conv_l16_pack_weights_pair: no forward tiling for this shape");
    FSC_CHECK_ARG(!packed_dgrad || plan_l16(*d, 1, &pd), "fsc_conv_l16_pack_weights_pair: no dgrad tiling for this shape");
    hipStream_t st = fsc::as_stream(stream);
    PackDir a{}, b{};
    if (packed_fwd) a = make_pack_dir(pf, packed_fwd, 0);
    if (packed_dgrad) b = make_pack_dir(pd, packed_dgrad, 1);
    if (!packed_fwd) { a = b; b = PackDir{}; }
    float* part = a.w_amax + 4;                            // partial maxima live behind the first direction's fragments
    hipLaunchKernelGGL(l16_wmax_kernel, dim3(kWmaxBlocks), dim3(256), 0, st, weight, (long)d->c_out * d->c_in * d->kh * d->kw, part);
    hipLaunchKernelGGL(l16_pack_w_kernel, dim3((unsigned)(a.blocks + b.blocks)), dim3(256), 0, st, weight, d->c_out, d->c_in,
                       d->kh * d->kw, a, b, part);
    FSC_LAUNCH_CHECK("fsc_conv_l16_pack_weights");
    return 0;
}

int fsc_conv_l16_pack_weights(const fsc_conv_desc* d, const float* weight, int dgrad, float* packed, fsc_stream_t stream) {
    return fsc_conv_l16_pack_weights_pair(d, weight, dgrad ? nullptr : packed, dgrad ? packed : nullptr, stream);
}

/* fsc_conv_l16_pack_weights_pair for `count` weights in ceil(count / 8) pairs of launches.  packed_fwd[i] / packed_dgrad[i] may be
 * NULL for a direction that is not wanted (never both). */
int fsc_conv_l16_pack_weights_multi(int count, const fsc_conv_desc* descs, const float* const* weights, float* const* packed_fwd,
                                    float* const* packed_dgrad, fsc_stream_t stream) {
    FSC_CHECK_ARG(count > 0 && descs && weights && packed_fwd && packed_dgrad, "fsc_conv_l16_pack_weights_multi: bad arguments");
    hipStream_t st = fsc::as_stream(stream);
    std::vector<fsc_conv_desc> resolved(descs, descs + count);       // (FSC_ARITH_DEFAULT -> the process default, entry by entry)
    for (auto& r : resolved)
        if (r.arith == FSC_ARITH_DEFAULT) r.arith = fsc_conv_default_arith();
    descs = resolved.data();
    if (l16::is_l3(descs[0].arith)) {                        // (one arithmetic per call)
        for (int k = 1; k < count; ++k)
            FSC_CHECK_ARG(descs[k].arith == descs[0].arith, "fsc_conv_l16_pack_weights_multi: entry %d mixes limb formats", k);
        return fsc::l3::pack_weights_multi(count, descs, weights, packed_fwd, packed_dgrad, st);
    }
    for (int base = 0; base < count; base += kMultiPack) {
        PackJobs jobs{};
        jobs.n = count - base < kMultiPack ? count - base : kMultiPack;
        int blocks = 0;
        for (int k = 0; k < jobs.n; ++k) {
            const fsc_conv_desc* d = descs + base + k;
            float* pf_out = packed_fwd[base + k];
            float* pd_out = packed_dgrad[base + k];
            FSC_CHECK_ARG(valid_l16_desc(d) && weights[base + k] && (pf_out || pd_out), "fsc_conv_l16_pack_weights_multi: bad entry %d", base + k);
            LPlan pf{}, pd{};
            FSC_CHECK_ARG(!pf_out || plan_l16(*d, 0, &pf), "fsc_conv_l16_pack_weights_multi: no forward tiling for entry %d", base + k);
            FSC_CHECK_ARG(!pd_out || plan_l16(*d, 1, &pd), "fsc_conv_l16_pack_weights_multi: no dgrad tiling for entry %d", base + k);
            PackJob& pj = jobs.j[k];
            pj.w = weights[base + k];
            pj.c_out = d->c_out; pj.c_in = d->c_in; pj.taps = d->kh * d->kw;
            pj.count = (long)d->c_out * d->c_in * pj.taps;
            PackDir a{}, b{};
            if (pf_out) a = make_pack_dir(pf, pf_out, 0);
            if (pd_out) b = make_pack_dir(pd, pd_out, 1);
            if (!pf_out) { a = b; b = PackDir{}; }
            pj.a = a; pj.b = b;
            pj.part = a.w_amax + 4;
            pj.first_block = blocks;
            blocks += a.blocks + b.blocks;
        }
        hipLaunchKernelGGL(l16_wmax_multi_kernel, dim3((unsigned)(jobs.n * kWmaxBlocks)), dim3(256), 0, st, jobs);
        hipLaunchKernelGGL(l16_pack_w_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, jobs);
    }
    FSC_LAUNCH_CHECK("fsc_conv_l16_pack_weights_multi");
    return 0;
}

int fsc_conv_l16_fwd(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed,
                     const float* bias, int dgrad, int accumulate, float* out, fsc_stream_t stream) {
    FSC_RESOLVE_DESC(d)
    if (d && l16::is_l3(d->arith))
        return fsc::l3::fwd(d, in_l16, in_amax, packed, bias, dgrad, accumulate, out, nullptr, nullptr, fsc::as_stream(stream));
    LPlan p;
    FSC_CHECK_ARG(valid_l16_desc(d) && in_l16 && in_amax && packed && out, "fsc_conv_l16_fwd: bad descriptor or null pointer");
    FSC_CHECK_ARG(!(dgrad && bias), "fsc_conv_l16_fwd: dgrad takes no bias");
    FSC_CHECK_ARG(plan_l16(*d, dgrad, &p), "fsc_conv_l16_fwd: unsupported shape (see fsc_conv_l16_supported)");
    hipStream_t st = fsc::as_stream(stream);
    const uint4* in = reinterpret_cast<const uint4*>(in_l16);
    if (d->kh == 3) return launch_l16_cot<3, 3>(p, in, packed, bias, out, accumulate, in_amax, st);
    return launch_l16_cot<1, 1>(p, in, packed, bias, out, accumulate, in_amax, st);
}

/* statistics records of the STATS forward: out3[4] = {workers, channel blocks, channels per block, order}; the records are
 * workers * 8 * channels-per-block float4 {sum (y - pivot), sum (y - pivot)^2, min, max}; worker w holds channel block
 * w % blocks (order 0) or (w / 8) % blocks (order 1: XCD-aware item order) */
int fsc_conv_l16_stats_layout(const fsc_conv_desc* d, int pool, int* out3) {
    FSC_RESOLVE_DESC(d)
    if (d && l16::is_l3(d->arith)) return fsc::l3::stats_layout(d, pool, out3);
    LPlan p;
    if (!valid_l16_desc(d) || !out3) return 0;
    if (!(pool ? plan_l16_pool(*d, &p) : plan_l16(*d, 0, &p)) || !stats_ok(p)) return 0;
    out3[0] = (int)stat_workers(p); out3[1] = p.co_blocks; out3[2] = p.cot * 16; out3[3] = p.g.xcd;
    return 1;
}

int fsc_conv_l16_fwd_stats(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed,
                           const float* bias, float* out, const float* stat_pivot, void* stat_rec, fsc_stream_t stream) {
    FSC_RESOLVE_DESC(d)
    if (d && l16::is_l3(d->arith)) {
        FSC_CHECK_ARG(stat_rec, "fsc_conv_l16_fwd_stats: null records");
        return fsc::l3::fwd(d, in_l16, in_amax, packed, bias, 0, 0, out, stat_pivot, stat_rec, fsc::as_stream(stream));
    }
    LPlan p;
    FSC_CHECK_ARG(valid_l16_desc(d) && in_l16 && in_amax && packed && out && stat_rec, "fsc_conv_l16_fwd_stats: bad descriptor or null pointer");
    FSC_CHECK_ARG(plan_l16(*d, 0, &p) && stats_ok(p), "fsc_conv_l16_fwd_stats: unsupported shape (see fsc_conv_l16_stats_layout)");
    hipStream_t st = fsc::as_stream(stream);
    const uint4* in = reinterpret_cast<const uint4*>(in_l16);
    const StatArgs sa{stat_pivot, reinterpret_cast<float4*>(stat_rec)};
    if (d->kh == 3) return launch_l16_cot<3, 3>(p, in, packed, bias, out, 0, in_amax, st, sa);
    return launch_l16_cot<1, 1>(p, in, packed, bias, out, 0, in_amax, st, sa);
}

int fsc_conv_l16_pool_supported(const fsc_conv_desc* d) {
    FSC_RESOLVE_DESC(d)
    if (d && l16::is_l3(d->arith)) return fsc::l3::pool_supported(d);
    LPlan p;
    return valid_l16_desc(d) && plan_l16_pool(*d, &p) ? 1 : 0;
}

static int pool_fwd_impl(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias,
                         float* pooled, uint8_t* idx, StatArgs sa, fsc_stream_t stream) {
    if (d && l16::is_l3(d->arith))
        return fsc::l3::pool_fwd(d, in_l16, in_amax, packed, bias, pooled, idx, sa.pivot, sa.rec, fsc::as_stream(stream));
    LPlan p;
    FSC_CHECK_ARG(valid_l16_desc(d) && in_l16 && in_amax && packed && pooled && idx, "fsc_conv_l16_pool_fwd: bad descriptor or null pointer");
    FSC_CHECK_ARG(plan_l16_pool(*d, &p), "fsc_conv_l16_pool_fwd: unsupported shape (see fsc_conv_l16_pool_supported)");
    FSC_CHECK_ARG(!sa.rec || stats_ok(p), "fsc_conv_l16_pool_fwd_stats: unsupported shape (see fsc_conv_l16_stats_layout)");
    hipStream_t st = fsc::as_stream(stream);
    const uint4* in = reinterpret_cast<const uint4*>(in_l16);
    switch (p.cot) {
#ifndef FSC_L16_DEV
        case 4: return launch_l16_pool<4>(p, in, packed, bias, pooled, idx, in_amax, st, sa);
        case 6: return launch_l16_pool<6>(p, in, packed, bias, pooled, idx, in_amax, st, sa);
#endif
        case 5: return launch_l16_pool<5>(p, in, packed, bias, pooled, idx, in_amax, st, sa);
        case 7: return launch_l16_pool<7>(p, in, packed, bias, pooled, idx, in_amax, st, sa);
        default: return launch_l16_pool<8>(p, in, packed, bias, pooled, idx, in_amax, st, sa);
    }
}

int fsc_conv_l16_pool_fwd(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed,
                          const float* bias, float* pooled, uint8_t* idx, fsc_stream_t stream) {
    return pool_fwd_impl(d, in_l16, in_amax, packed, bias, pooled, idx, StatArgs{nullptr, nullptr}, stream);
}

int fsc_conv_l16_pool_fwd_stats(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed,
                                const float* bias, float* pooled, uint8_t* idx, const float* stat_pivot, void* stat_rec,
                                fsc_stream_t stream) {
    FSC_RESOLVE_DESC(d)
    FSC_CHECK_ARG(stat_rec, "fsc_conv_l16_pool_fwd_stats: null records");
    return pool_fwd_impl(d, in_l16, in_amax, packed, bias, pooled, idx, StatArgs{stat_pivot, reinterpret_cast<float4*>(stat_rec)}, stream);
}

#ifdef FSC_L16_PROFILE
/* development: copies the 8 x 8 phase counters to `out64` (host) and clears them */
int fsc_debug_l16_prof(unsigned long long* out64) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_l16_prof), sizeof(unsigned long long) * 64);
    unsigned long long z[64] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_l16_prof), z, sizeof(z));
    return 0;
}
#endif

/* include/fsc_hip.h: inference -- 3x3 convolution + MaxPool2d(2) + eval-mode BatchNorm + PReLU in one launch (three-limb arithmetics) */
int fsc_conv_l16_pool_fwd_act_supported(const fsc_conv_desc* d) {
    FSC_RESOLVE_DESC(d)
    return d && l16::is_l3(d->arith) ? fsc::l3::pool_supported(d) : 0;
}

int fsc_conv_l16_pool_fwd_act(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias,
                              const float* scale, const float* shift, const float* alpha, float* out, void* out_l16,
                              const float* out_amax, float* seen_max, fsc_stream_t stream) {
    FSC_RESOLVE_DESC(d)
    FSC_CHECK_ARG(d && l16::is_l3(d->arith), "fsc_conv_l16_pool_fwd_act: the three-limb arithmetics only (arith 9, 10)");
    return fsc::l3::pool_fwd_act(d, in_l16, in_amax, packed, bias, scale, shift, alpha, out, out_l16, out_amax, seen_max,
                                 fsc::as_stream(stream));
}

/* include/fsc_hip.h: inference -- convolution + per-channel affine + PReLU, written as the L16 operand of the next convolution */
int fsc_conv_l16_fwd_act_supported(const fsc_conv_desc* d) { return fsc_conv_l16_supported(d, 0); }

int fsc_conv_l16_fwd_act(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const float* packed, const float* bias,
                         const float* scale, const float* shift, const float* alpha, void* out_l16, const float* out_amax,
                         float* seen_max, fsc_stream_t stream) {
    FSC_RESOLVE_DESC(d)
    if (d && l16::is_l3(d->arith))
        return fsc::l3::fwd_act(d, in_l16, in_amax, packed, bias, scale, shift, alpha, out_l16, out_amax, seen_max, fsc::as_stream(stream));
    LPlan p;
    FSC_CHECK_ARG(valid_l16_desc(d) && in_l16 && in_amax && packed && out_l16 && out_amax, "fsc_conv_l16_fwd_act: bad descriptor or null pointer");
    FSC_CHECK_ARG((scale == nullptr) == (shift == nullptr), "fsc_conv_l16_fwd_act: scale and shift come together");
    FSC_CHECK_ARG(plan_l16(*d, 0, &p), "fsc_conv_l16_fwd_act: unsupported shape (see fsc_conv_l16_supported)");
    hipStream_t st = fsc::as_stream(stream);
    const uint4* in = reinterpret_cast<const uint4*>(in_l16);
    ActArgs2 act{scale, shift, alpha, reinterpret_cast<uint2*>(out_l16), out_amax, reinterpret_cast<unsigned*>(seen_max)};
    if (d->kh == 3) return launch_l16_cot<3, 3>(p, in, packed, bias, nullptr, 0, in_amax, st, StatArgs{nullptr, nullptr}, act);
    return launch_l16_cot<1, 1>(p, in, packed, bias, nullptr, 0, in_amax, st, StatArgs{nullptr, nullptr}, act);
}

int fsc_conv_l16_plan_describe(const fsc_conv_desc* d, int dgrad, char* buf, size_t buf_len) {
    FSC_RESOLVE_DESC(d)
    if (d && l16::is_l3(d->arith)) return fsc::l3::plan_describe(d, dgrad, buf, buf_len);
    LPlan p;
    FSC_CHECK_ARG(valid_l16_desc(d) && buf && buf_len > 0 && plan_l16(*d, dgrad, &p), "fsc_conv_l16_plan_describe: unsupported shape");
    snprintf(buf, buf_len, "conv_l16_fwd_kernel<%d,%d,%d,%d> box=%dx%dx%d items=%ldx%d workers=%ld steps=%d lds=%zu", d->kh, d->kw,
             p.cot, p.pt, p.g.nb, p.g.th, p.g.tw, p.tiles, p.co_blocks, p.workers, p.g.steps, p.lds_bytes);
    return 0;
}

}  // extern "C"

namespace fsc {
int l16_fwd_clock(double* shader_mhz) {
    FSC_CHECK_ARG(shader_mhz, "fsc_conv_l16_last_clock: null pointer");
    unsigned long long v[2] = {0, 0};
    hipError_t e = hipMemcpyFromSymbol(v, HIP_SYMBOL(g_l16_clock), sizeof(v));
    if (e != hipSuccess) { fsc::set_error("fsc_conv_l16_last_clock: %s", hipGetErrorString(e)); return (int)e; }
    *shader_mhz = v[1] ? 100.0 * (double)v[0] / (double)v[1] : 0.0;
    return 0;
}
}  // namespace fsc
