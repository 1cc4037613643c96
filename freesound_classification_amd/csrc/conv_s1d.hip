// Small 1-d convolutions in the bf16 arithmetic (arith 1) for gfx950 -- the LATE blocks of the hierarchical model of
// BASELINE.json configs[2] (reference networks/classifiers.py:147-163, 37-69: Conv1d k = 3 / k = 1 on 156 ... 476 channels over
// rows of 107 ... 3 frames, batch 128: 13.7 k ... 384 positions per layer, tensors of 8 MB ... 0.7 MB).
//
// conv_fwd_x3_kernel (conv.hip) is built for layers that fill the chip with 128-position x 128-channel tiles and walks K in
// barrier-paced steps behind an LDS ring: on these layers it launches 12 ... 100 workgroups of 36 ... 45 serial steps at ~0.9 us a
// step -- 20 ... 50 us per convolution, 1.9 ms of the 8.5 ms cfg-3 step (profiles/r06_step_timeline_cfg3.txt).  They are tiny GEMMs
// (M = output channels, N = positions, K = taps x input channels; <= 0.9 GFLOP), so the kernels here are organised for LATENCY:
//   * forward / input gradient: a workgroup owns 32 channels x 32 positions and ALL of K; its 8 waves take every 8th K step
//     (one step = one tap x 32 channels = one v_mfma_f32_16x16x32_bf16 per 16 x 16 tile) and add their partial tiles through LDS
//     at the end: 5 - 6 steps per wave instead of 45 in a row, hundreds to thousands of workgroups instead of a dozen;
//   * no LDS staging, no barrier inside the K loop: a wave loads its A fragments (the packed bf16 weight fragments of
//     fsc_conv_pack_weights, conv.hip pack_x3_items: 1 KB per tile and step, L2-resident) and gathers its B fragments straight from
//     the fp32 NCL activation (8 dwords per lane: the 8 channels of its octet at its position and tap; rounded to bf16 with
//     v_cvt_pk_bf16_f32 -- round to nearest even, the contract of arith 1), the loads of step s + 8 in flight behind the MFMAs of s;
//   * weight gradient: D[ci][co] per tap, K = positions: a workgroup owns 32 x 32 (ci, co) x taps, its 4 waves every 4th
//     32-position K step of its split; the three taps of a k = 3 layer share ONE window of 10 positions per lane.  Slices go to
//     the caller's workspace in the [split][tap][ci][co] layout of conv.hip, reduced by the library's wgrad_reduce kernels.
// Same arithmetic contract as the matrix-core path of arith 1 (operands rounded once to bf16, exact products, fp32 sums).
#include "s1d.h"

#include <stdlib.h>

namespace {

using fsc::s1d::kCot;
using fsc::s1d::kPt;
using fsc::s1d::kWaves;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ float g_zero_s1d[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {      // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}
__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
    return (u32x4){cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7])};
}

struct FwdArgs {
    const float* in;
    const u32x4* packed;
    const float* bias;
    float* out;
    int cin, cout, h, w, n;       // plane = h x w positions (1-d: h = 1)
    long hw, npix;
    int nfull, tail_oct, steps, accumulate;
};

// the three exact bf16 limbs of two fp32 values (l16.h split3_pair): x = h + m + l
__device__ __forceinline__ void split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned hp = cvt_pk_bf16(x0, x1);
    float r0, r1;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(__uint_as_float(hp << 16)));
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(__uint_as_float(hp & 0xffff0000u)));
    const unsigned mp = cvt_pk_bf16(r0, r1);
    float s0, s1;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(s0) : "v"(r0), "v"(__uint_as_float(mp << 16)));
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(s1) : "v"(r1), "v"(__uint_as_float(mp & 0xffff0000u)));
    h = hp;
    m = mp;
    l = cvt_pk_bf16(s0, s1);
}

// grid: x = groups of kPt * 16 positions of the flattened (image, row, column) sequence, y = blocks of COT * 16 output channels.
// NP = 1: operands rounded once to bf16 (arith 1); NP = 9: three exact bf16 limbs per operand, all nine limb products summed in a
// step accumulator (smallest terms first) that reaches the running sum with ONE fp32 addition (arith 9; conv_l3.hip FSC_L3_TMPACC:
// the product of the two fp32 operands is exact, one rounding per 32-channel tap).  The A fragments are those of
// fsc_conv_pack_weights (conv.hip pack_x3_items: [block][step][tile][limb][lane]).
template <int KH, int KW, int NP, int COT>
__global__ __launch_bounds__(kWaves * 64) void s1d_fwd_kernel(FwdArgs a) {
    constexpr int TAPS = KH * KW, PH = KH / 2, PW = KW / 2;
    constexpr int NL = NP == 1 ? 1 : 3;
    constexpr int NT = COT * kPt;
    __shared__ f32x4 red[kWaves][NT][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kq = lane >> 4;
    const long p0 = (long)blockIdx.x * (kPt * 16);
    const int cb = blockIdx.y;

    long base[kPt];               // element offset of (image, channel 0, row, column) of this lane's position in tile j
    int py[kPt], px[kPt];
    bool live[kPt];
#pragma unroll
    for (int j = 0; j < kPt; ++j) {
        const long p = p0 + j * 16 + lm;
        live[j] = p < a.npix;
        const long pc = live[j] ? p : 0;
        const int img = (int)(pc / a.hw);
        const int rem = (int)(pc - (long)img * a.hw);
        py[j] = KH == 1 ? 0 : rem / a.w;
        px[j] = rem - py[j] * a.w;
        base[j] = (long)img * a.cin * a.hw + rem;
    }
    f32x4 acc[COT][kPt];
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < kPt; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const u32x4* const a_base = a.packed + (long)cb * a.steps * (COT * NL * 64) + lane;
    const float* const zero = g_zero_s1d;

    // operands of one K step: A fragments of the block's tiles, raw B values of this lane's two positions
    struct Ops {
        u32x4 A[COT][NL];
        float v[kPt][8];
    };
    auto load = [&](int S, Ops& o) {
        int c, s;
        if (S < a.nfull * TAPS) {
            c = S / TAPS;
            s = S - c * TAPS;
        } else {
            c = a.nfull;
            s = S - a.nfull * TAPS;
        }
        int tap, ch0;
        bool kvalid = true;
        if (c < a.nfull) {                     // a full chunk: step = tap, lane group = octet
            tap = s;
            ch0 = c * 32 + kq * 8;
        } else {                               // the remainder chunk: (tap, octet) flattened over the lane groups
            const int gi = 4 * s + kq;
            kvalid = gi < TAPS * a.tail_oct;
            tap = kvalid ? gi / a.tail_oct : 0;
            ch0 = c * 32 + (kvalid ? gi - tap * a.tail_oct : 0) * 8;
        }
        const int ty = KW == 1 ? 0 : tap / KW, tx = tap - ty * KW;
        const int dy = ty - PH, dx = tx - PW;
#pragma unroll
        for (int i = 0; i < COT; ++i)
#pragma unroll
            for (int l = 0; l < NL; ++l) o.A[i][l] = a_base[(((long)S * COT + i) * NL + l) * 64];
#pragma unroll
        for (int j = 0; j < kPt; ++j) {
            const int ys = py[j] + dy, xs = px[j] + dx;
            const bool ok = live[j] && kvalid && ys >= 0 && ys < a.h && xs >= 0 && xs < (KH == 1 && KW == 1 ? a.hw : a.w);   // 1 x 1: the plane is one row
            const float* src = a.in + base[j] + (long)ch0 * a.hw + dy * a.w + dx;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float* q = (ok && ch0 + e < a.cin) ? src + (long)e * a.hw : zero;
                o.v[j][e] = *q;
            }
        }
    };
    auto compute = [&](const Ops& o) {
        if constexpr (NP == 1) {
            u32x4 B[kPt];
#pragma unroll
            for (int j = 0; j < kPt; ++j) B[j] = pack8(o.v[j]);
#pragma unroll
            for (int i = 0; i < COT; ++i)
#pragma unroll
                for (int j = 0; j < kPt; ++j) acc[i][j] = mfma_bf(o.A[i][0], B[j], acc[i][j]);
        } else {
            u32x4 B[kPt][3];
#pragma unroll
            for (int j = 0; j < kPt; ++j) {
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) split3(o.v[j][2 * q], o.v[j][2 * q + 1], h[q], m[q], l[q]);
                B[j][0] = (u32x4){h[0], h[1], h[2], h[3]};
                B[j][1] = (u32x4){m[0], m[1], m[2], m[3]};
                B[j][2] = (u32x4){l[0], l[1], l[2], l[3]};
            }
            // (A limb, B limb) smallest terms first: l l, m l, l m, h l, m m, l h, h m, m h, h h
            constexpr int la[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0}, lb[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int i = 0; i < COT; ++i)
#pragma unroll
                for (int j = 0; j < kPt; ++j) {
                    f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 9; ++q) t = mfma_bf(o.A[i][la[q]], B[j][lb[q]], t);
                    acc[i][j] += t;
                }
        }
    };

    // this wave's steps: wid, wid + 8, ...; the loads of the next one are issued before the MFMAs of the current one.  (Two steps
    // ahead -- pairs of steps, 136 registers -- measured SLOWER: 15.4 -> 21.3 us on 156 -> 156 @ 107 frames; the loop is bound by the
    // cache lines its gathers touch, not by the latency of one step.)
    int S = wid;
    if (S < a.steps) {
        Ops cur;
        load(S, cur);
        for (S += kWaves; S < a.steps; S += kWaves) {
            Ops nxt;
            load(S, nxt);
            compute(cur);
            cur = nxt;
        }
        compute(cur);
    }
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < kPt; ++j) red[wid][i * kPt + j][lane] = acc[i][j];
    __syncthreads();
    for (int t = wid; t < NT; t += kWaves) {
        const int i = t / kPt, j = t - i * kPt;
        f32x4 s = red[0][t][lane];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) s += red[w][t][lane];
        const long p = p0 + j * 16 + lm;
        if (p < a.npix) {
            const int img = (int)(p / a.hw);
            const long rem = p - (long)img * a.hw;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (cb * COT + i) * 16 + kq * 4 + r;
                if (co < a.cout) {
                    const long idx = ((long)img * a.cout + co) * a.hw + rem;
                    float v = s[r];
                    if (a.bias != nullptr) v += a.bias[co];
                    if (a.accumulate) v += a.out[idx];
                    a.out[idx] = v;
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------------
// weight gradient: part[split][tap][ci][co] = sum over the split's positions of x[ci][p + tap - pad] * dout[co][p]
struct WArgs {
    const float* in;
    const float* dout;
    float* part;
    int cin, cout, len, n;
    long npix;
    int ksteps, nsplit, ci_pad, co_pad;
};
constexpr int kWWaves = 4;

// grid: x = 32-channel blocks of ci, y = 32-channel blocks of co, z = split.  D[M = ci][N = co]: the A operand is the input window
// (lane (kq, m): 8 consecutive positions of channel m), the B operand the output gradient (lane (kq, n): the same 8 positions of
// channel n): an accumulator row is 4 consecutive ci of one co, so a lane group stores 16 consecutive co of a ci row.
template <int TAPS>
__global__ __launch_bounds__(kWWaves * 64) void s1d_wgrad_kernel(WArgs a) {
    constexpr int PAD = TAPS / 2;
    constexpr int NW = 8 + 2 * PAD;            // window positions per lane
    constexpr int NT = 2 * 2 * TAPS;
    __shared__ f32x4 red[kWWaves][NT][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kq = lane >> 4;
    const int cib = blockIdx.x, cob = blockIdx.y, z = blockIdx.z;
    f32x4 acc[2][2][TAPS];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[i][j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* const zero = g_zero_s1d;

    struct Ops {
        float x[2][NW];           // input window of this lane's ci (two ci tiles)
        float g[2][8];            // output gradient of this lane's co (two co tiles)
        unsigned first, last;     // bit e: position e of the lane's 8 is the first / last of its row
    };
    auto load = [&](int ks, Ops& o) {
        const long pk = (long)ks * 32 + kq * 8;                   // this lane group's first position
        // walk the window pk - PAD ... pk + 7 + PAD of the flattened (image, position) sequence
        long ps = pk - PAD;
        const bool neg = ps < 0;
        if (neg) ps = 0;
        int img = (int)(ps / a.len);
        int l = (int)(ps - (long)img * a.len);
        o.first = 0;
        o.last = 0;
        long off_x[2], off_g[2];
        bool ci_ok[2], co_ok[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int ci = (cib * 2 + t2) * 16 + lm, co = (cob * 2 + t2) * 16 + lm;
            ci_ok[t2] = ci < a.cin;
            co_ok[t2] = co < a.cout;
            off_x[t2] = ((long)img * a.cin + (ci_ok[t2] ? ci : 0)) * a.len + l;
            off_g[t2] = ((long)img * a.cout + (co_ok[t2] ? co : 0)) * a.len + l;
        }
        long p = pk - PAD;
#pragma unroll
        for (int q = 0; q < NW; ++q, ++p) {
            const bool in_range = p >= 0 && p < a.npix;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const float* px = (in_range && ci_ok[t2]) ? a.in + off_x[t2] : zero;
                o.x[t2][q] = *px;
            }
            const int e = q - PAD;
            if (e >= 0 && e < 8) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    const float* pg = (in_range && co_ok[t2]) ? a.dout + off_g[t2] : zero;
                    o.g[t2][e] = *pg;
                }
                if (l == 0) o.first |= 1u << e;
                if (l == a.len - 1) o.last |= 1u << e;
            }
            if (p >= 0) {                                          // advance (p < 0: the walk has not started: ps was clamped to 0)
                ++l;
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) { ++off_x[t2]; ++off_g[t2]; }
                if (l == a.len) {
                    l = 0;
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        off_x[t2] += (long)(a.cin - 1) * a.len;
                        off_g[t2] += (long)(a.cout - 1) * a.len;
                    }
                }
            }
        }
    };
    auto compute = [&](const Ops& o) {
        u32x4 B[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) B[j] = pack8(o.g[j]);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xv = o.x[i][e + t];
                    if (TAPS == 3) {
                        // the tap's neighbour lies in another row (the flattened sequence continues into the next image): zero padding
                        if (t == 0 && ((o.first >> e) & 1u)) xv = 0.f;
                        if (t == 2 && ((o.last >> e) & 1u)) xv = 0.f;
                    }
                    v[e] = xv;
                }
                const u32x4 A = pack8(v);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j][t] = mfma_bf(A, B[j], acc[i][j][t]);
            }
        }
    };
    const int stride = a.nsplit * kWWaves;
    int ks = z * kWWaves + wid;
    if (ks < a.ksteps) {
        Ops cur;
        load(ks, cur);
        for (ks += stride; ks < a.ksteps; ks += stride) {
            Ops nxt;
            load(ks, nxt);
            compute(cur);
            cur = nxt;
        }
        compute(cur);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) red[wid][(i * 2 + j) * TAPS + t][lane] = acc[i][j][t];
    __syncthreads();
    for (int tt = wid; tt < NT; tt += kWWaves) {
        f32x4 s = red[0][tt][lane];
#pragma unroll
        for (int w = 1; w < kWWaves; ++w) s += red[w][tt][lane];
        const int t = tt % TAPS, ij = tt / TAPS, i = ij >> 1, j = ij & 1;
        const int co = (cob * 2 + j) * 16 + lm;
        float* dst = a.part + ((long)z * TAPS + t) * a.ci_pad * a.co_pad + co;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = (cib * 2 + i) * 16 + kq * 4 + r;
            dst[(long)ci * a.co_pad] = s[r];
        }
    }
}

// -------------------------------------------------------------------------------------------
// weight gradient, SEGMENT form (rows of >= 8 positions): the K dimension runs over segments of 8 consecutive positions of ONE row
// (a row of L positions has ceil(L / 8) segments, the last one masked), four segments per K step, one per lane group.  A lane's
// operand values are then contiguous in memory -- the window l0 - 1 ... l0 + 8 of its input channel, the 8 positions of its output
// channel -- and arrive as three / two vector loads instead of ten / eight gathered dwords (lane = channel: every load instruction
// of the flattened form touched 64 cache lines); positions outside the row are masked to zero, which IS the convolution's padding.
struct f4u { float v[4]; } __attribute__((packed, aligned(4)));
struct f2u { float v[2]; } __attribute__((packed, aligned(4)));

template <int TAPS>
__global__ __launch_bounds__(kWWaves * 64) void s1d_wgrad_seg_kernel(WArgs a, int spr, long nseg, long total_x, long total_g) {
    constexpr int PAD = TAPS / 2;
    constexpr int NW = 8 + 2 * PAD;
    constexpr int NT = 2 * 2 * TAPS;
    __shared__ f32x4 red[kWWaves][NT][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kq = lane >> 4;
    const int cib = blockIdx.x, cob = blockIdx.y, z = blockIdx.z;
    f32x4 acc[2][2][TAPS];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[i][j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    struct Ops {
        float x[2][NW];
        float g[2][8];
    };
    auto load = [&](int ks, Ops& o) {
        const long seg = (long)ks * 4 + kq;
        const bool seg_ok = seg < nseg;
        const long sc = seg_ok ? seg : 0;
        const int img = (int)(sc / spr);
        const int l0 = (int)(sc - (long)img * spr) * 8;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int ci = (cib * 2 + t2) * 16 + lm, co = (cob * 2 + t2) * 16 + lm;
            const bool ci_ok = seg_ok && ci < a.cin, co_ok = seg_ok && co < a.cout;
            // ---- input window l0 - PAD ... l0 + 7 + PAD of channel ci
            const long bx = ((long)img * a.cin + (ci_ok ? ci : 0)) * a.len + l0 - PAD;
            float xv[NW];
            if (bx >= 0 && bx + NW <= total_x) {              // (all but the first / last rows of the tensor: vector loads)
                const f4u q0 = *reinterpret_cast<const f4u*>(a.in + bx), q1 = *reinterpret_cast<const f4u*>(a.in + bx + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { xv[e] = q0.v[e]; xv[4 + e] = q1.v[e]; }
                if (PAD) {
                    const f2u q2 = *reinterpret_cast<const f2u*>(a.in + bx + 8);
                    xv[8] = q2.v[0]; xv[NW - 1] = q2.v[1];
                }
            } else {
#pragma unroll
                for (int q = 0; q < NW; ++q) xv[q] = (bx + q >= 0 && bx + q < total_x) ? a.in[bx + q] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const int l = l0 - PAD + q;
                o.x[t2][q] = (ci_ok && l >= 0 && l < a.len) ? xv[q] : 0.f;
            }
            // ---- output gradient l0 ... l0 + 7 of channel co
            const long bg = ((long)img * a.cout + (co_ok ? co : 0)) * a.len + l0;
            float gv[8];
            if (bg + 8 <= total_g) {
                const f4u q0 = *reinterpret_cast<const f4u*>(a.dout + bg), q1 = *reinterpret_cast<const f4u*>(a.dout + bg + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { gv[e] = q0.v[e]; gv[4 + e] = q1.v[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) gv[e] = bg + e < total_g ? a.dout[bg + e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o.g[t2][e] = (co_ok && l0 + e < a.len) ? gv[e] : 0.f;
        }
    };
    auto compute = [&](const Ops& o) {
        u32x4 B[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) B[j] = pack8(o.g[j]);
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = o.x[i][e + t];
                const u32x4 A = pack8(v);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j][t] = mfma_bf(A, B[j], acc[i][j][t]);
            }
    };
    const int stride = a.nsplit * kWWaves;
    int ks = z * kWWaves + wid;
    if (ks < a.ksteps) {
        Ops cur;
        load(ks, cur);
        for (ks += stride; ks < a.ksteps; ks += stride) {
            Ops nxt;
            load(ks, nxt);
            compute(cur);
            cur = nxt;
        }
        compute(cur);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) red[wid][(i * 2 + j) * TAPS + t][lane] = acc[i][j][t];
    __syncthreads();
    for (int tt = wid; tt < NT; tt += kWWaves) {
        f32x4 s = red[0][tt][lane];
#pragma unroll
        for (int w = 1; w < kWWaves; ++w) s += red[w][tt][lane];
        const int t = tt % TAPS, ij = tt / TAPS, i = ij >> 1, j = ij & 1;
        const int co = (cob * 2 + j) * 16 + lm;
        float* dst = a.part + ((long)z * TAPS + t) * a.ci_pad * a.co_pad + co;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = (cib * 2 + i) * 16 + kq * 4 + r;
            dst[(long)ci * a.co_pad] = s[r];
        }
    }
}

// positions (n * length) up to which these kernels take a layer (development: FSC_S1D_MAXPX / FSC_S1D_WMAXPX).  Measured per layer
// of cfg 3 at batch 128 (tools/s1d_bench.py, profiles/r06_s1d_layers.txt): forward / input gradient win from rows of 107 frames
// down (13.7 k positions: 27 -> 20 us, 35 -> 7.6 us on the last block) and lose above (27.5 k positions: gathered loads, the
// activation re-read by every channel block); the weight gradient in its flattened form -- lane = channel, so a wave's gathers
// touch 64 lines per instruction -- wins from rows of 26 frames down (4096 positions), in its segment form (rows of >= 13 frames)
// over the whole range: 445 -> 358 us over blocks 4 - 9 (30 -> 21 us on 195 -> 195 k1 @ 53, 38 -> 17 on 244 -> 305 @ 26).
// (re-measured with 64-channel workgroups, profiles/r06_s1d_layers.txt last table: k = 3 wins up to rows of 215 frames (27.5 k
// positions: 27 -> 19 us) and loses at 430 (29 against 27 us); k = 1 -- a third of the gathers -- still wins there (27 -> 24 us)
// and loses at 861; the weight gradient's segment form wins or ties up to 55 k positions (38 -> 28 us on 100 -> 100 k1 @ 430))
// (the FORWARD of a k = 1 layer at 55 k positions goes back to the ring kernel since that has the statistics epilogue: 27 us with
// the BatchNorm statistics against 24 + a 16 us statistics pass; the input gradient keeps the gathers)
int max_px(int taps, int dgrad) {
    static const int v = [] {
        const char* e = getenv("FSC_S1D_MAXPX");
        return e ? atoi(e) : -1;
    }();
    return v >= 0 ? v : ((taps == 1 && dgrad) ? 65536 : 32768);
}
int max_px_small() {
    static const int v = [] {
        const char* e = getenv("FSC_S1D_SMALLPX");         // 2-d planes / nine products: positions up to which these kernels take a layer
        return e ? atoi(e) : 4096;
    }();
    return v;
}
int max_px_wgrad_seg() {
    static const int v = [] {
        const char* e = getenv("FSC_S1D_WSEGMAXPX");
        return e ? atoi(e) : 65536;
    }();
    return v;
}
int max_px_wgrad() {
    static const int v = [] {
        const char* e = getenv("FSC_S1D_WMAXPX");
        return e ? atoi(e) : 4096;
    }();
    return v;
}

}  // namespace

namespace fsc {
namespace s1d {

bool plan_fwd(const fsc_conv_desc& d, int dgrad, Plan* out) {
    const bool k11 = d.kh == 1 && d.kw == 1, k13 = d.kh == 1 && d.kw == 3, k33 = d.kh == 3 && d.kw == 3;
    if (!(d.arith == 1 || d.arith == 9) || !(k11 || k13 || k33)) return false;
    if (k13 && d.h != 1) return false;
    // Shipped: the bf16 arithmetic on 1-d rows (cfg 3).  The kernel itself is general -- 2-d planes, 3 x 3 taps, nine exact limb
    // products (arith 9) -- and was measured on cfg 2's last block (759 channels on 2 x 6 pixels, batch 128, bf16x9): 1 x 1 layers
    // 70 -> 42 us, 3 x 3 layers 288 -> 269 us (a 32-channel x 32-position workgroup re-reads 1.3 MB of weight fragments: 0.7 GB of
    // L2 traffic per launch): 0.15 ms of a 46 ms step.  Not worth a second route through the fp32-exact arithmetics;
    // FSC_S1D_GENERAL=1 enables it (development).
    static const bool general = [] { const char* e = getenv("FSC_S1D_GENERAL"); return e && e[0] == '1'; }();
    if (!general && (d.arith != 1 || d.h != 1)) return false;
    Plan p{};
    p.n = d.n; p.h = d.h; p.w = d.w; p.kh = d.kh; p.kw = d.kw; p.taps = d.kh * d.kw;
    p.nprod = d.arith == 9 ? 9 : 1;
    p.cot = p.nprod == 9 ? 2 : kCot;            // (nine products: three limb fragments per tile -- 32 channels per workgroup)
    p.cin = dgrad ? d.c_out : d.c_in;
    p.cout = dgrad ? d.c_in : d.c_out;
    p.npix = (long)d.n * d.h * d.w;
    // ranges: the bf16 kernels on 1-d rows as measured (max_px); 2-d planes and the nine-product arithmetic take the layers the
    // persistent kernels cannot fill (cfg 2's last block: 759 channels on 2 x 6 pixels, 1536 positions at batch 128)
    const long cap = (p.nprod == 1 && d.h == 1) ? max_px(p.taps, dgrad) : max_px_small();
    if (p.npix > cap || p.cin < 32 || p.cout < 48) return false;
    const int rem = p.cin % 32;
    p.nfull = p.cin / 32 + (rem > 24 ? 1 : 0);
    p.tail_oct = (rem > 0 && rem <= 24) ? ceil_div(rem, 8) : 0;
    p.steps = p.nfull * p.taps + ceil_div(p.taps * p.tail_oct, 4);
    p.co_blocks = ceil_div(ceil_div(p.cout, 16), p.cot);
    p.px_groups = ceil_div(p.npix, kPt * 16);
    *out = p;
    return true;
}

int launch_fwd(const Plan& p, const float* in, const unsigned short* packed, const float* bias, float* out, int accumulate,
               hipStream_t st) {
    FwdArgs a{in, reinterpret_cast<const u32x4*>(packed), bias, out, p.cin, p.cout, p.h, p.w, p.n, (long)p.h * p.w, p.npix, p.nfull,
              p.tail_oct, p.steps, accumulate};
    const dim3 grid((unsigned)p.px_groups, (unsigned)p.co_blocks), block(kWaves * 64);
#define FSC_S1D_LAUNCH(KH_, KW_)                                                                              \
    do {                                                                                                      \
        if (p.nprod == 9) hipLaunchKernelGGL((s1d_fwd_kernel<KH_, KW_, 9, 2>), grid, block, 0, st, a);        \
        else hipLaunchKernelGGL((s1d_fwd_kernel<KH_, KW_, 1, kCot>), grid, block, 0, st, a);                  \
    } while (0)
    if (p.kh == 3) FSC_S1D_LAUNCH(3, 3);
    else if (p.kw == 3) FSC_S1D_LAUNCH(1, 3);
    else FSC_S1D_LAUNCH(1, 1);
#undef FSC_S1D_LAUNCH
    FSC_LAUNCH_CHECK("fsc_conv_fwd(s1d)");
    return 0;
}

bool plan_wgrad(const fsc_conv_desc& d, WPlan* out) {
    if (d.arith != 1 || d.h != 1 || d.kh != 1 || !(d.kw == 1 || d.kw == 3)) return false;
    WPlan p{};
    p.n = d.n; p.len = d.w; p.taps = d.kw; p.cin = d.c_in; p.cout = d.c_out;
    p.npix = (long)d.n * d.w;
    // rows of >= 13 positions: the segment form (K step = four 8-position segments of single rows: vector loads); shorter rows
    // (6 and 3 frames fill 75 % / 37 % of a segment) keep the flattened form
    p.seg = d.w >= 13 ? 1 : 0;
    if (p.npix > (p.seg ? max_px_wgrad_seg() : max_px_wgrad()) || p.cin < 32 || p.cout < 32) return false;
    p.spr = ceil_div(d.w, 8);
    p.ksteps = p.seg ? ceil_div((long)d.n * p.spr, 4) : ceil_div(p.npix, 32);
    p.ci_blocks = ceil_div(p.cin, 32);
    p.co_blocks = ceil_div(p.cout, 32);
    p.ci_pad = p.ci_blocks * 32;
    p.co_pad = p.co_blocks * 32;
    // split K until the grid has ~2 workgroups per CU, with >= 2 K steps per wave
    const long blocks = (long)p.ci_blocks * p.co_blocks;
    long ns = (512 + blocks - 1) / blocks;
    const long cap = p.ksteps / (2 * kWWaves);
    if (ns > cap) ns = cap;
    if (ns > 64) ns = 64;
    if (ns < 1) ns = 1;
    p.nsplit = (int)ns;
    *out = p;
    return true;
}

int launch_wgrad(const WPlan& p, const float* in, const float* dout, float* part, hipStream_t st) {
    WArgs a{in, dout, part, p.cin, p.cout, p.len, p.n, p.npix, p.ksteps, p.nsplit, p.ci_pad, p.co_pad};
    const dim3 grid((unsigned)p.ci_blocks, (unsigned)p.co_blocks, (unsigned)p.nsplit);
    if (p.seg) {
        const long nseg = (long)p.n * p.spr, tx = p.npix * p.cin, tg = p.npix * p.cout;
        if (p.taps == 3) hipLaunchKernelGGL(s1d_wgrad_seg_kernel<3>, grid, dim3(kWWaves * 64), 0, st, a, p.spr, nseg, tx, tg);
        else hipLaunchKernelGGL(s1d_wgrad_seg_kernel<1>, grid, dim3(kWWaves * 64), 0, st, a, p.spr, nseg, tx, tg);
        FSC_LAUNCH_CHECK("fsc_conv_wgrad(s1d, segments)");
        return 0;
    }
    if (p.taps == 3) hipLaunchKernelGGL(s1d_wgrad_kernel<3>, grid, dim3(kWWaves * 64), 0, st, a);
    else hipLaunchKernelGGL(s1d_wgrad_kernel<1>, grid, dim3(kWWaves * 64), 0, st, a);
    FSC_LAUNCH_CHECK("fsc_conv_wgrad(s1d)");
    return 0;
}

}  // namespace s1d
}  // namespace fsc
