// Weight gradient of nn.Conv2d 3x3 / 1x1 (reference networks/classifiers.py:526-531, 77-81) from PRE-SPLIT operands
// (L16 tensors, see conv_l16.hip / include/fsc_hip.h):
//
//     dW[co][ci][tap] = sum over pixels  dOut[co][p] * In[ci][p + tap]          K = pixels
//
// Both operands are activations.  conv.hip's conv_wgrad_x3_kernel reads them as fp32 planes and splits every value into
// fp16 limbs beside the MFMAs; here they arrive as limbs, 8 channels of a position per 16 bytes.  The MFMA wants the
// OTHER orientation -- a lane holds 8 consecutive pixels (K) of one channel -- which is exactly what the LDS transpose
// read of gfx950 delivers: ds_read_b64_tr_b16 hands lane l of a 16-lane group element j = halfword (l & 3) of the 8 bytes
// lane 4 j + (l >> 2) addressed (measured with tools/probe/tr16_probe.hip).  With lane i = 4 j + c4 addressing
// [position p0 + j][channels 4 c4 .. 4 c4 + 3] the group receives [channel l][positions p0 .. p0 + 3]: two reads = one
// 16x16x32 operand, for ANY position offset -- the nine taps of a 3x3 layer are nine address offsets into the one staged
// box, with no shifting, re-pairing or splitting in registers (the fp32 kernel spends 28 VALU per three taps on that).
//
// Workgroup = 8 waves = ng co groups x nt ci groups; a wave owns <= MT co tiles x (3x3: one ci tile x 9 taps | 1x1: CT ci
// tiles).  Unit of work = a box of 64 pixels (th x tw, tw in {8, 16, 32, 64}) staged by 16-byte LDS-DMA, two stages,
// split-K over boxes; partial sums [split][tap][ci][co] and the reduce kernel as in conv.hip.  Products and order per
// k-step are those of the f16x3 arithmetic (lh, hl, hh).
#include "common.h"
#include "l16.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(3))) v4s* lds_v4s;
typedef __attribute__((address_space(1))) const void* glb_ptr;

constexpr int kWaves = 8;
constexpr int kMaxXI = 4;          // input DMA instructions per (octet, limb) plane: plane <= 256 positions
#ifndef FSC_L16W_COPYW
#define FSC_L16W_COPYW 4
#endif
// 3x3 kernels: waves 0 .. 3 (one per SIMD) issue all copies of the next unit; the other wave of each SIMD goes straight to its
// MFMAs and has the matrix pipe to itself meanwhile.  With all eight waves copying, the burst (LDS-DMA accepts ~30 B per cycle
// and CU) kept the pipe idle for 2500-3300 cycles of a 9400-cycle unit: +6-9 % on every 3x3 layer of cfg 2.  (Spreading the
// copies over the MFMA slots instead lost 4 %; the short units of the 1x1 kernels are 3 % faster with all eight waves copying;
// giving the copying waves the smaller co-tile groups changed nothing.)
constexpr int kCopyWaves3x3 = FSC_L16W_COPYW;
#ifndef FSC_L16W_AHEAD
#define FSC_L16W_AHEAD 1
#endif
constexpr int kAhead = FSC_L16W_AHEAD;     // B (input) fragments are read this many MFMA slots ahead of their use (2, 3: no faster)
// Unit pitches (positions) are padded to == 2 (mod 8) where the LDS allows: an octet (two units) is then 64 bytes (mod 256) from
// the next, and the transpose read's lanes of octet A and octet B hit disjoint bank halves (unpadded: a 2-way conflict on every read).

struct WGeom {
    int n, cin, cout, h, w;
    long hw;
    int oct_in, oct_out;            // octets of the two L16 tensors
    int th, tw, tiles_h, tiles_w;   // 64-pixel box and boxes per image
    int rows, cols, plane, npos;    // staged input window incl. halo: npos = rows * cols positions, plane = unit pitch (npos padded)
    int du;                         // dOut unit pitch: 66 / 68 (padded: two / three limbs) or 64
    int xi;                         // ceil(plane / 64)
    int units, nsplit;
    int ng, nt;                     // co groups x ci groups of a workgroup (ng * nt == 8)
    int tpb, tpg;                   // co tiles per block / per group (tpg <= MT)
    int co_blocks, ci_blocks, co_pad, ci_pad;
};

__device__ __attribute__((aligned(16))) float g_zero16_w[4] = {0.f, 0.f, 0.f, 0.f};
__device__ unsigned long long g_l16w_clock[2];      // shader cycles / 100 MHz reference ticks of the last launch (see conv_l16.hip)
// Development (-DFSC_L16_PROFILE): cycle stamps around the phases of a unit, summed per wave of workgroup (0, 0):
// 0 wait for the copies + barrier, 1 copy issue, 2 MFMA k-steps, 3 units, 4 whole kernel (fsc_debug_l16w_prof)
#ifdef FSC_L16_PROFILE
__device__ unsigned long long g_l16w_prof[8][8];
#define WPROF_MARK() (pf_t = __builtin_readcyclecounter())
#define WPROF_ADD(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pf_acc[i] += n_ - pf_t; pf_t = n_; } while (0)
#else
#define WPROF_MARK()
#define WPROF_ADD(i)
#endif

// 16-byte LDS-DMA as inline assembly (see conv_l16.hip glds16: behind the builtin hipcc waits for the copy of the NEXT unit
// before the first LDS read of the current one).  The wait is explicit: vmcnt(0) before the barrier that opens a unit.
__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_ptr)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(m) : "memory", "m0");
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// limb products (A limb, B limb), smallest first.  Two scaled fp16 limbs: l*h, h*l, h*h (l*l dropped: arith 3).  Three exact bf16
// limbs (0 = h, 1 = m, 2 = l): all nine (arith 9), without l*l (8), without l*l, m*l, l*m (6).
template <int NL, int NPROD> struct WProds;
template <> struct WProds<2, 3> { static constexpr int la[3] = {1, 0, 0}; static constexpr int lb[3] = {0, 1, 0}; };
template <> struct WProds<3, 9> { static constexpr int la[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0}; static constexpr int lb[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0}; };
template <> struct WProds<3, 8> { static constexpr int la[8] = {1, 2, 0, 1, 2, 0, 1, 0}; static constexpr int lb[8] = {2, 1, 2, 1, 0, 1, 0, 0}; };
template <> struct WProds<3, 6> { static constexpr int la[6] = {0, 1, 2, 0, 1, 0}; static constexpr int lb[6] = {2, 1, 0, 1, 0, 0}; };
// two transpose reads = the 8 K values (positions p .. p + 7) of this lane's channel; `p` = byte address of the lane's piece
__device__ __forceinline__ u32x4 tr_read8(const char* p) {
    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p)));
    const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(p + 64)));
    return (u32x4){lo[0], lo[1], hi[0], hi[1]};
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// same co-tile spreading as conv.hip wgx_group
__host__ __device__ inline void w_group(int live, int ng, int q, int* start, int* count) {
    const int base = live / ng, extra = live - base * ng;
    int st = 0, cnt = 0;
    for (int k = 0; k <= q; ++k) {
        const int rank = k < (ng + 1) / 2 ? 2 * k : 2 * (ng - 1 - k) + 1;
        cnt = base + (rank < extra ? 1 : 0);
        if (k < q) st += cnt;
    }
    *start = st;
    *count = cnt;
}

__device__ __forceinline__ void xcd_block_order(int* block, int* split) {
    const unsigned bx = gridDim.x, total = gridDim.x * gridDim.y;
    const unsigned id = blockIdx.x + bx * blockIdx.y;
    const unsigned xcd = id & 7u, slot = id >> 3;
    const unsigned chunk = total >> 3, rem = total & 7u;
    const unsigned v = xcd < rem ? xcd * (chunk + 1) + slot : rem * (chunk + 1) + (xcd - rem) * chunk + slot;
    *split = (int)(v / bx);
    *block = (int)(v - (unsigned)*split * bx);
}

// F16: scaled fp16 limbs (two: arith 3; three: arith 10, l16.h) -- else three exact bf16 limbs
template <int KH, int KW, int MT, int CT, int NL = 2, int NPROD = 3, bool F16 = (NL == 2)>
__global__ __launch_bounds__(kWaves * 64) void conv_l16_wgrad_kernel(WGeom g, const uint4* __restrict__ in,
                                                                      const uint4* __restrict__ dout,
                                                                      float* __restrict__ part,
                                                                      const float* __restrict__ in_amax,
                                                                      const float* __restrict__ dout_amax) {
    constexpr int TAPS = KH * KW;
    constexpr int PADH = KH / 2, PADW = KW / 2;
    constexpr int NB = TAPS * CT;                  // B slots of a wave: (ci tile, tap)
    static_assert(TAPS == 1 || CT == 1, "3x3: one ci tile per wave");
    constexpr int kCopyWaves = TAPS == 1 ? kWaves : kCopyWaves3x3;

    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    const int co_oct = g.ng * g.tpg * 2, ci_oct = g.nt * CT * 2;             // octets staged per operand
    const int stage_u4 = co_oct * NL * g.du + ci_oct * NL * g.plane;           // uint4 per stage: dOut units, then input units

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long ck0 = __builtin_readcyclecounter(), cr0 = __builtin_amdgcn_s_memrealtime();
    const int kq = lane >> 4, li = lane & 15, c4 = li & 3, lj = li >> 2;
    const int lm = li;
    const int cog = wid / g.nt, cig = wid - cog * g.nt;
    int blk, split;
    xcd_block_order(&blk, &split);
    const int cb = blk / g.ci_blocks, ib = blk - cb * g.ci_blocks;
    const int tile0 = cb * g.tpb;                                            // first co tile of the block
    const int cit0 = ib * g.nt * CT;                                         // first ci tile of the block
    const int total_co_tiles = (g.cout + 15) >> 4, total_ci_tiles = (g.cin + 15) >> 4;
    int mt_live, g_start;
    {
        const int blk_live = tile0 + g.tpb < total_co_tiles ? g.tpb : total_co_tiles - tile0;
        w_group(blk_live > 0 ? blk_live : 0, g.ng, cog, &g_start, &mt_live);
        if (cit0 + cig * CT >= total_ci_tiles) mt_live = 0;                  // ci group wholly beyond c_in
    }

    // ---- operand scales
    float inv_ab = 1.f;
    if constexpr (F16) {
        float* red = reinterpret_cast<float*>(smem4);
        const float ma = fsc::wave_max(dout_amax[tid]), mb = fsc::wave_max(in_amax[tid]);
        if (lane == 0) { red[wid] = ma; red[kWaves + wid] = mb; }
        __syncthreads();
        float xa = red[0], xb = red[kWaves];
#pragma unroll
        for (int i = 1; i < kWaves; ++i) { xa = fmaxf(xa, red[i]); xb = fmaxf(xb, red[kWaves + i]); }
        __syncthreads();
        const int fa = l16::scale_field(xa), fb = l16::scale_field(xb);
        inv_ab = l16::inv_scale(fa, xa) * l16::inv_scale(fb, xb);
    }

    // ---- octets of the block beyond the tensors are never copied: zero them once in both stages
    const int co_oct_live = min(co_oct, g.oct_out - tile0 * 2), ci_oct_live = min(ci_oct, g.oct_in - cit0 * 2);
    for (int st = 0; st < 2; ++st) {
        uint4* dl = smem4 + st * stage_u4;
        uint4* il = dl + co_oct * NL * g.du;
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int i = (co_oct_live > 0 ? co_oct_live : 0) * NL * g.du + tid; i < co_oct * NL * g.du; i += kWaves * 64) dl[i] = z;
        for (int i = (ci_oct_live > 0 ? ci_oct_live : 0) * NL * g.plane + tid; i < ci_oct * NL * g.plane; i += kWaves * 64) il[i] = z;
    }

    // ---- unit-invariant DMA plans.  dOut: lane = pixel of the box in row-major (= run) order; input: lane + 64 j =
    //      position of the halo'd window
    const int pr = lane / g.tw, pc = lane - pr * g.tw;
    int qr[kMaxXI], qc[kMaxXI];
#pragma unroll
    for (int j = 0; j < kMaxXI; ++j) {
        const int o = lane + 64 * j;
        qr[j] = o / g.cols;
        qc[j] = o - qr[j] * g.cols;
        if (o >= g.npos) qr[j] = -1;
    }
    const uint4* const zero = reinterpret_cast<const uint4*>(g_zero16_w);
    const long do_img = (long)g.oct_out * NL * g.hw, in_img = (long)g.oct_in * NL * g.hw;
    auto issue_unit = [&](int u, int stage) {
        if (wid >= kCopyWaves) return;
        uint4* dl = smem4 + stage * stage_u4;
        uint4* il = dl + co_oct * NL * g.du;
        int t = u;
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h; t /= g.tiles_h;
        const int n0 = t, h0 = thi * g.th, w0 = twi * g.tw;
        {
            const bool live = h0 + pr < g.h && w0 + pc < g.w;
            const uint4* src = dout + (long)n0 * do_img + (long)(tile0 * 2) * NL * g.hw + (long)(h0 + pr) * g.w + (w0 + pc);
#pragma unroll 1
            for (int un = wid; un < co_oct_live * NL; un += kCopyWaves)       // unit = (octet, limb)
                glds16(live ? src + (long)un * g.hw : zero, dl + un * g.du);
        }
#pragma unroll
        for (int j = 0; j < kMaxXI; ++j) {
            if (j < g.xi && qr[j] >= 0) {
                const int gh = h0 + qr[j] - PADH, gw = w0 + qc[j] - PADW;
                const bool live = gh >= 0 && gh < g.h && gw >= 0 && gw < g.w;
                const uint4* src = in + (long)n0 * in_img + (long)(cit0 * 2) * NL * g.hw + (long)gh * g.w + gw;
#pragma unroll 1
                for (int un = wid; un < ci_oct_live * NL; un += kCopyWaves)
                    glds16(live ? src + (long)un * g.hw : zero, il + un * g.plane + j * 64);
            }
        }
    };

    f32x4 acc[NB][MT];
#pragma unroll
    for (int s = 0; s < NB; ++s)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[s][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- this lane's transpose-read pieces.  A (dOut) of tile i, limb l, k-step st:
    //      ((2 (g_start + i) + (c4 >> 1)) * 2 + l) * 64 positions + run * 8 + lj, + (c4 & 1) * 8 bytes;  run = 4 st + kq
    const int runs_shift = g.tw == 8 ? 0 : g.tw == 16 ? 1 : g.tw == 32 ? 2 : 3;      // log2(runs per box row)
    const int a_lane = (((2 * g_start + (c4 >> 1)) * NL) * g.du + kq * 8 + lj) * 16 + (c4 & 1) * 8;
    //      B (input) of slot (ci tile ct, tap), limb l: ((2 (cig * CT + ct) + (c4 >> 1)) * 2 + l) * plane + (r + ty) * cols + c0 + tx + lj
    int b_lane[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int run = 4 * st + kq;
        const int r = run >> runs_shift, c0 = (run - (r << runs_shift)) * 8;
        b_lane[st] = (((2 * cig * CT + (c4 >> 1)) * NL) * g.plane + r * g.cols + c0 + lj) * 16 + (c4 & 1) * 8;
    }
    const int limb_b = g.plane * 16;
    using WP = WProds<NL, NPROD>;

    // The whole unit loop is specialised on the wave's live co tiles (dead tiles of a block's last group are skipped without
    // a branch per MFMA; a switch INSIDE the loop makes hipcc keep two copies of the accumulators).
#ifdef FSC_L16_PROFILE
    unsigned long long pf_t = 0, pf_acc[5] = {0, 0, 0, 0, 0};
#endif
    auto run_units = [&](auto live_c) {
        constexpr int LIVE = decltype(live_c)::value;
        int stage = 0;
        if (split < g.units) issue_unit(split, 0);
#pragma unroll 1
        for (int u = split; u < g.units; u += g.nsplit) {
            WPROF_MARK();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // unit u has landed ...
            __syncthreads();                  // ... for every wave; everyone is done with the other stage
            WPROF_ADD(0);
            if (u + g.nsplit < g.units) issue_unit(u + g.nsplit, stage ^ 1);
            WPROF_ADD(1);
            const char* dl = reinterpret_cast<const char*>(smem4 + stage * stage_u4);
            const char* il = dl + (size_t)co_oct * NL * g.du * 16;
            if constexpr (LIVE > 0) {
#pragma unroll 1
                for (int st = 0; st < 2; ++st) {
                    const char* ap = dl + a_lane + st * (4 * 8 * 16);
                    const char* bp = il + b_lane[st];
                    u32x4 af[LIVE][NL];
#pragma unroll
                    for (int i = 0; i < LIVE; ++i)
#pragma unroll
                        for (int l = 0; l < NL; ++l) af[i][l] = tr_read8(ap + (i * 2 * NL + l) * g.du * 16);
                    u32x4 bf[kAhead + 1][NL];                                  // [buffer][limb]
                    auto read_slot = [&](int s, u32x4 (&dst)[NL]) {
                        const int ct = s / TAPS, tap = s - ct * TAPS;
                        const int ty = tap / KW, tx = tap - ty * KW;
                        const char* p = bp + ct * 2 * NL * limb_b + (ty * g.cols + tx) * 16;
#pragma unroll
                        for (int l = 0; l < NL; ++l) dst[l] = tr_read8(p + l * limb_b);
                    };
#pragma unroll
                    for (int s = 0; s < kAhead && s < NB; ++s) read_slot(s, bf[s]);
                    static_for<0, NB>([&](auto s_c) {
                        constexpr int s = decltype(s_c)::value;
                        if (s + kAhead < NB) read_slot(s + kAhead, bf[(s + kAhead) % (kAhead + 1)]);
#pragma unroll
                        for (int gq = 0; gq < NPROD; ++gq)
#pragma unroll
                            for (int i = 0; i < LIVE; ++i)
                                acc[s][i] = mfma16<F16>(af[i][WP::la[gq]], bf[s % (kAhead + 1)][WP::lb[gq]], acc[s][i]);
                        // the 2 * NL reads of the slot kAhead ahead go behind the first MFMAs of this one; nothing else moves across
                        // slots (unpinned, the scheduler hoists every slot's reads to the top of the k-step: +56 registers)
#pragma unroll
                        for (int k = 0; k < 2 * NL; ++k) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (s + kAhead < NB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
            }
            stage ^= 1;
            WPROF_ADD(2);
#ifdef FSC_L16_PROFILE
            pf_acc[3] += 1;
#endif
        }
    };
    switch (mt_live) {
        case 0: run_units(std::integral_constant<int, 0>{}); break;
        case 1: run_units(std::integral_constant<int, 1>{}); break;
        case 2: if constexpr (MT >= 2) run_units(std::integral_constant<int, 2>{}); break;
        case 3: if constexpr (MT >= 3) run_units(std::integral_constant<int, 3>{}); break;
        default: if constexpr (MT >= 4) run_units(std::integral_constant<int, 4>{}); break;
    }

#ifdef FSC_L16_PROFILE
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
        pf_acc[4] = __builtin_readcyclecounter() - ck0;
        for (int i = 0; i < 5; ++i) atomicAdd(&g_l16w_prof[wid][i], pf_acc[i]);
    }
#endif
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        g_l16w_clock[0] = __builtin_readcyclecounter() - ck0;
        g_l16w_clock[1] = __builtin_amdgcn_s_memrealtime() - cr0;
    }
    // partial[split][tap][ci][co]; D row = co (kq*4 + r), column = ci (lm)
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        const int ct = s / TAPS, tap = s - ct * TAPS;
        const long row = ((long)split * TAPS + tap) * g.ci_pad + (cit0 + cig * CT + ct) * 16 + lm;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (i < mt_live && cit0 + cig * CT + ct < total_ci_tiles) {
                const int tile = tile0 + g_start + i;
                const f32x4 a = acc[s][i] * inv_ab;
                *reinterpret_cast<float4*>(part + row * g.co_pad + tile * 16 + kq * 4) = make_float4(a[0], a[1], a[2], a[3]);
            }
        }
    }
}

// blockIdx.y = (tap, ci) row of the partial slices; threadIdx.x runs along co (coalesced reads), threadIdx.y over four interleaved
// groups of split-K slices (one thread walking all <= 256 slices of its element was a chain of dependent loads: 22 us per layer)
constexpr int kRedX = 64, kRedY = 4;
__global__ __launch_bounds__(kRedX * kRedY) void l16_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int c_out,
                                                                          int c_in, int taps, int ci_pad, int co_pad, int nsplit) {
    __shared__ float red[kRedY][kRedX];
    const int co = blockIdx.x * kRedX + threadIdx.x;
    const int tap = blockIdx.y / c_in, ci = blockIdx.y - tap * c_in;
    const long slice = (long)taps * ci_pad * co_pad;
    float s0 = 0.f, s1 = 0.f;
    if (co < c_out) {
        const float* p = part + ((long)tap * ci_pad + ci) * co_pad + co;
        int sp = threadIdx.y;
        for (; sp + kRedY < nsplit; sp += 2 * kRedY) {
            s0 += p[(long)sp * slice];
            s1 += p[(long)(sp + kRedY) * slice];
        }
        if (sp < nsplit) s0 += p[(long)sp * slice];
    }
    red[threadIdx.y][threadIdx.x] = s0 + s1;
    __syncthreads();
    if (threadIdx.y == 0 && co < c_out)
        dw[((long)co * c_in + ci) * taps + tap] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// -------------------------------------------------------------------------------------------
struct WPlan {
    WGeom g;
    int mt, ct, nl, nprod;
    bool f16;
    size_t lds_bytes;
};

bool plan_l16_wgrad(const fsc_conv_desc& d, WPlan* out) {
    const bool bf3 = l16::is_l3(d.arith);                   // (three limbs: bf16, or scaled fp16)
    if (d.arith != 3 && !bf3) return false;                  // (the entry points resolve FSC_ARITH_DEFAULT: FSC_RESOLVE_DESC)
    const int nl = bf3 ? 3 : 2;
    const int max_tpg = bf3 ? 3 : 4;       // co tiles per wave: 9 taps x 4 tiles x 4 registers + three-limb fragments do not fit 256
    if (fsc::env().no_l16 || fsc::env().no_l16_wgrad) return false;
    const int taps = d.kh * d.kw;
    if (!((d.kh == 3 && d.kw == 3) || (d.kh == 1 && d.kw == 1))) return false;
    if (d.c_in < 32 || d.c_out < 32) return false;
    WPlan p{};
    WGeom& g = p.g;
    int h = d.h, w = d.w;
    if (taps == 1) {                 // no halo: every (n, octet) plane is one row of h*w positions
        w = d.h * d.w;
        h = 1;
    }
    g.n = d.n; g.cin = d.c_in; g.cout = d.c_out; g.h = h; g.w = w; g.hw = (long)h * w;
    g.oct_in = (d.c_in + 7) / 8; g.oct_out = (d.c_out + 7) / 8;
    if ((long)d.n * g.oct_in * nl * g.hw >= (1L << 31) || (long)d.n * g.oct_out * nl * g.hw >= (1L << 31)) return false;
    // box = th x tw = 64 pixels: the fewest padded pixels first; among boxes within 1 % of that, the smallest halo'd window
    // (8 x 8: 100 positions = two DMA instructions per plane; 2 x 32: 136 = three.  Measured on cfg 2 with the box pinned:
    // 64 x 215 planes +5 % with 8 x 8 over 2 x 32, 32 x 107 planes +6 % / -2.5 % over 4 x 16 at equal padding)
    long best_px = -1;
    int best_npos = 0;
    const int force_tw = fsc::env().l16w_tw;                   // development (FSC_L16W_TW): pin the box width
    for (int pass = 0; pass < 2; ++pass)                       // 0: the fewest padded pixels; 1: the box
        for (int tw = 64; tw >= 8; tw >>= 1) {
            const int th = 64 / tw;
            if (force_tw && force_tw != tw && taps > 1) continue;
            if (th > 1 && h == 1) continue;
            const int npos = (th + d.kh - 1) * (tw + d.kw - 1);
            if (npos > 64 * kMaxXI - 8) continue;
            const long px = (long)fsc::ceil_div(h, th) * fsc::ceil_div(w, tw) * 64;
            if (pass == 0) {
                if (best_px < 0 || px < best_px) best_px = px;
            } else if (px * 100 <= best_px * 101 && (best_npos == 0 || npos < best_npos)) {
                best_npos = npos; g.th = th; g.tw = tw;
            }
        }
    if (best_npos > 0) best_px = (long)fsc::ceil_div(h, g.th) * fsc::ceil_div(w, g.tw) * 64;
    if (best_px < 0 || (double)h * w < 0.6 * (double)best_px) return false;
    g.tiles_h = fsc::ceil_div(h, g.th); g.tiles_w = fsc::ceil_div(w, g.tw);
    g.units = d.n * g.tiles_h * g.tiles_w;
    g.rows = g.th + d.kh - 1; g.cols = g.tw + d.kw - 1;
    g.npos = g.rows * g.cols;
    g.xi = fsc::ceil_div(g.npos, 64);
    // ci tiles per wave.  1x1 on three limbs: 4, else 2, else 1 -- six bytes per element leave room for fewer staged octets
    const int ct_opts[3] = {taps == 1 ? 4 : 1, 2, 1};
    const int n_opts = (taps == 1 && bf3) ? 3 : 1;
    int ct = ct_opts[0];
    p.nl = nl;
    p.nprod = bf3 ? (d.arith == 6 || l16::is_f3(d.arith) ? 6 : d.arith == 8 ? 8 : 9) : 3;
    p.f16 = !l16::is_bf3(d.arith);
    const int tiles_co = fsc::ceil_div(d.c_out, 16), tiles_ci = fsc::ceil_div(d.c_in, 16);
    double best_eff = -1.0;
    for (int opt = 0; opt < n_opts && best_eff < 0.4; ++opt)
    for (int pad = 1; pad >= 0 && best_eff < 0.4; --pad) {       // bank-conflict padding first; without it when nothing fits
        ct = ct_opts[opt];
        // (the two octets a transpose read touches must lie 64 or 192 bytes apart modulo 256: octet pitch = nl planes of 16-byte
        // positions.  Two limbs: pitch == 2 (mod 8) positions; three limbs: 3 * pitch == 4 or 12 (mod 16), i.e. pitch == 4 (mod 8) --
        // with the two-limb pitches half of the LDS cycles of the three-limb kernel were bank conflicts, profiles/r05_pmc_conv_l3_wgrad.txt)
        const int want = nl == 3 ? 4 : 2;
        int plane = g.npos;
        if (pad) while (plane % 8 != want) ++plane;
        const int du = pad ? 64 + want : 64;
        for (int nt = 1; nt <= 8; nt *= 2) {
            const int ng = kWaves / nt;
            const int ci_blocks = fsc::ceil_div(tiles_ci, nt * ct);
            const int co_blocks = fsc::ceil_div(tiles_co, ng * max_tpg);
            const int tpb = fsc::ceil_div(tiles_co, co_blocks);
            const int tpg = fsc::ceil_div(tpb, ng);
            const size_t lds = 2 * 16 * ((size_t)ng * tpg * 2 * nl * du + (size_t)nt * ct * 2 * nl * plane);
            if (lds > 160 * 1024) continue;
            static const double kTileWeight[5] = {0.0, 0.6, 0.8, 0.93, 1.0};
            long busiest = 0;
            for (int cb = 0; cb < co_blocks; ++cb)
                for (int ib = 0; ib < ci_blocks; ++ib) {
                    int worst = 0;
                    for (int sd = 0; sd < 4; ++sd) {
                        int load = 0;
                        for (int wv = sd; wv < kWaves; wv += 4) {
                            const int cog = wv / nt, cig = wv % nt;
                            const int blk_live = cb * tpb + tpb < tiles_co ? tpb : tiles_co - cb * tpb;
                            int first, live;
                            w_group(blk_live > 0 ? blk_live : 0, ng, cog, &first, &live);
                            int ci_live = tiles_ci - (ib * nt + cig) * ct;
                            ci_live = ci_live < 0 ? 0 : ci_live > ct ? ct : ci_live;
                            load += live * (ci_live > 0 ? ct : 0);           // (dead ci tiles of a live group still run)
                        }
                        if (load > worst) worst = load;
                    }
                    busiest += worst;
                }
            const double eff = (double)tiles_co * tiles_ci / (4.0 * (double)busiest) * kTileWeight[tpg];
            if (eff > best_eff) {
                best_eff = eff;
                g.ng = ng; g.nt = nt; g.tpb = tpb; g.tpg = tpg; g.co_blocks = co_blocks; g.ci_blocks = ci_blocks;
                g.plane = plane; g.du = du;
                p.lds_bytes = lds;
            }
        }
    }
    if (best_eff < 0.4) return false;
    p.ct = ct;
    p.mt = g.tpg;
    g.co_pad = g.co_blocks * g.tpb * 16;
    g.ci_pad = g.ci_blocks * g.nt * ct * 16;
    const long base = (long)g.co_blocks * g.ci_blocks;
    const long part_bytes_per_split = (long)taps * g.ci_pad * g.co_pad * 4;
    long ns = base >= 256 ? 1 : 256 / base;
    if (ns > g.units / 4) ns = g.units / 4;
    if (ns < 1) ns = 1;
    while (ns > 1 && ns * part_bytes_per_split > (256L << 20)) --ns;
    g.nsplit = (int)ns;
    *out = p;
    return true;
}

template <int KH, int KW, int MT, int CT, int NL = 2, int NPROD = 3, bool F16 = (NL == 2)>
void launch_k(const WPlan& p, const uint4* in, const uint4* dout, float* part, const float* in_amax, const float* dout_amax,
              hipStream_t st) {
    auto kern = conv_l16_wgrad_kernel<KH, KW, MT, CT, NL, NPROD, F16>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    dim3 grid(p.g.co_blocks * p.g.ci_blocks, p.g.nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), p.lds_bytes, st, p.g, in, dout, part, in_amax, dout_amax);
}

bool valid_desc(const fsc_conv_desc* d) {
    if (!d || d->n <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->h <= 0 || d->w <= 0) return false;
    const long big = 1L << 31;
    return (long)d->n * d->c_in * d->h * d->w < big && (long)d->n * d->c_out * d->h * d->w < big;
}

}  // namespace

extern "C" {

int fsc_conv_l16_wgrad_supported(const fsc_conv_desc* d) {
    FSC_RESOLVE_DESC(d)
    WPlan p;
    return valid_desc(d) && plan_l16_wgrad(*d, &p) ? 1 : 0;
}

size_t fsc_conv_l16_wgrad_workspace_bytes(const fsc_conv_desc* d) {
    FSC_RESOLVE_DESC(d)
    WPlan p;
    if (!valid_desc(d) || !plan_l16_wgrad(*d, &p)) return 0;
    return (size_t)p.g.nsplit * d->kh * d->kw * p.g.ci_pad * p.g.co_pad * sizeof(float);
}

int fsc_conv_l16_wgrad(const fsc_conv_desc* d, const void* in_l16, const float* in_amax, const void* dout_l16,
                       const float* dout_amax, float* dweight, void* workspace, fsc_stream_t stream) {
    FSC_RESOLVE_DESC(d)
    WPlan p;
    FSC_CHECK_ARG(valid_desc(d) && in_l16 && dout_l16 && dweight && workspace, "fsc_conv_l16_wgrad: bad descriptor or null pointer");
    FSC_CHECK_ARG(l16::is_bf3(d->arith) || (in_amax && dout_amax), "fsc_conv_l16_wgrad: the scaled fp16 formats need both operand maxima");
    FSC_CHECK_ARG(plan_l16_wgrad(*d, &p), "fsc_conv_l16_wgrad: unsupported shape (see fsc_conv_l16_wgrad_supported)");
    hipStream_t st = fsc::as_stream(stream);
    const uint4* in = reinterpret_cast<const uint4*>(in_l16);
    const uint4* dout = reinterpret_cast<const uint4*>(dout_l16);
    float* part = reinterpret_cast<float*>(workspace);
#define FSC_WL(MT_)                                                                               \
    if (d->kh == 3) launch_k<3, 3, MT_, 1>(p, in, dout, part, in_amax, dout_amax, st);            \
    else launch_k<1, 1, MT_, 4>(p, in, dout, part, in_amax, dout_amax, st);                       \
    break;
#define FSC_WL3(MT_, NP_, F16_)                                                                                \
    if (d->kh == 3) launch_k<3, 3, MT_, 1, 3, NP_, F16_>(p, in, dout, part, in_amax, dout_amax, st);           \
    else if (p.ct == 4) launch_k<1, 1, MT_, 4, 3, NP_, F16_>(p, in, dout, part, in_amax, dout_amax, st);       \
    else if (p.ct == 2) launch_k<1, 1, MT_, 2, 3, NP_, F16_>(p, in, dout, part, in_amax, dout_amax, st);       \
    else launch_k<1, 1, MT_, 1, 3, NP_, F16_>(p, in, dout, part, in_amax, dout_amax, st);                      \
    break;
    if (p.nl == 3 && p.f16) {
        switch (p.mt) {
            case 1: FSC_WL3(1, 6, true)
            case 2: FSC_WL3(2, 6, true)
            default: FSC_WL3(3, 6, true)
        }
    } else if (p.nl == 3) {
        if (p.nprod != 9) {
            fsc::set_error("fsc_conv_l16_wgrad: this build has no bf16-limb kernels with %d products", p.nprod);
            return 22;
        }
        switch (p.mt) {
            case 1: FSC_WL3(1, 9, false)
            case 2: FSC_WL3(2, 9, false)
            default: FSC_WL3(3, 9, false)
        }
    } else {
        switch (p.mt) {
            case 1: FSC_WL(1)
            case 2: FSC_WL(2)
            case 3: FSC_WL(3)
            default: FSC_WL(4)
        }
    }
#undef FSC_WL3
#undef FSC_WL
    FSC_LAUNCH_CHECK("fsc_conv_l16_wgrad");
    const int taps = d->kh * d->kw;
    hipLaunchKernelGGL(l16_wgrad_reduce_kernel, dim3(fsc::ceil_div(d->c_out, kRedX), taps * d->c_in), dim3(kRedX, kRedY), 0, st,
                       part, dweight, d->c_out, d->c_in, taps, p.g.ci_pad, p.g.co_pad, p.g.nsplit);
    FSC_LAUNCH_CHECK("fsc_conv_l16_wgrad(reduce)");
    return 0;
}

/* which = 0: the last fsc_conv_l16_fwd / _pool_fwd (_stats) launch, 1: the last fsc_conv_l16_wgrad launch.  Synchronises the
 * device (a measurement aid: bench.py reads it after re-running the dominant layer, outside any timed region). */
int fsc_conv_l16_last_clock(int which, double* shader_mhz) {
    FSC_CHECK_ARG(shader_mhz && (which == 0 || which == 1 || which == 2), "fsc_conv_l16_last_clock: bad arguments");
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { fsc::set_error("fsc_conv_l16_last_clock: %s", hipGetErrorString(e)); return (int)e; }
    if (which == 0) return fsc::l16_fwd_clock(shader_mhz);
    if (which == 2) return fsc::l3::last_clock(shader_mhz);
    unsigned long long v[2] = {0, 0};
    e = hipMemcpyFromSymbol(v, HIP_SYMBOL(g_l16w_clock), sizeof(v));
    if (e != hipSuccess) { fsc::set_error("fsc_conv_l16_last_clock: %s", hipGetErrorString(e)); return (int)e; }
    *shader_mhz = v[1] ? 100.0 * (double)v[0] / (double)v[1] : 0.0;
    return 0;
}

#ifdef FSC_L16_PROFILE
int fsc_debug_l16w_prof(unsigned long long* out64) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_l16w_prof), sizeof(unsigned long long) * 64);
    unsigned long long z[64] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_l16w_prof), z, sizeof(z));
    return 0;
}
#endif

int fsc_conv_l16_wgrad_plan_describe(const fsc_conv_desc* d, char* buf, size_t buf_len) {
    FSC_RESOLVE_DESC(d)
    WPlan p;
    FSC_CHECK_ARG(valid_desc(d) && buf && buf_len > 0 && plan_l16_wgrad(*d, &p), "fsc_conv_l16_wgrad_plan_describe: unsupported shape");
    char name[64];
    if (p.nl == 3) snprintf(name, sizeof(name), "conv_l3_wgrad_kernel<%d,%d,%d,%d,%d%s>", d->kh, d->kw, p.mt, p.ct, p.nprod, p.f16 ? ",f16" : "");
    else snprintf(name, sizeof(name), "conv_l16_wgrad_kernel<%d,%d,%d,%d>", d->kh, d->kw, p.mt, p.ct);
    snprintf(buf, buf_len, "%s box=%dx%d groups=%dx%d tiles/block=%d units=%d split=%d grid=%dx%d lds=%zu", name, p.g.th, p.g.tw, p.g.ng,
             p.g.nt, p.g.tpb, p.g.units, p.g.nsplit, p.g.co_blocks * p.g.ci_blocks, p.g.nsplit, p.lds_bytes);
    return 0;
}

}  // extern "C"
