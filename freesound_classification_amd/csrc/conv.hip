// Convolutions for gfx950: split-bf16 kernels on v_mfma_f32_16x16x32_bf16 (fp32 results, see conv_fwd_x3_kernel /
// conv_wgrad_x3_kernel below), native fp32 implicit GEMM on v_mfma_f32_16x16x4_f32, direct stem kernels.
// Replaces nn.Conv2d 3x3/1x1 and nn.Conv1d k3/k1 forward, input-gradient and weight-gradient
// (reference networks/classifiers.py:526-531, 77-81, 149-154, 42-46), NCHW, stride 1, "same"
// padding.  These kernels carry ~99 % of the step's FLOPs and are MFMA-bound.
//
// Forward / dgrad (one kernel; dgrad = forward with the flipped, transposed packed weights):
//   D[m = out channel][n = pixel] += A[m][k] * B[k][n],  k = (tap, in channel)
//   * workgroup = 4 waves; output tile = COT*16 channels x PT*64 pixels; the pixel tile is a
//     (nb images) x (th rows) x (tw cols) box chosen per layer on the host so odd widths
//     (431, 215, 107, 53, ...) waste few MFMA columns;
//   * per K-chunk of KC input channels the halo'd input box [KC][rows][cols] and the weight
//     slab [tap][KC][COT*16] are staged in LDS; plane strides are == 16 (mod 32) so the
//     four k-planes an MFMA operand read touches sit on disjoint banks;
//   * global loads of chunk c+1 are issued into registers before the MFMAs of chunk c
//     (issue-early / write-late), one LDS buffer.
// Wgrad:
//   D[m = out channel][n = (in channel, tap)] += dOut[m][pixel] * In[n][pixel + tap]
//   * split-K over pixel tiles; partials [split][tap][ci][co] reduced by a second kernel that
//     also transposes into the (c_out, c_in, kh, kw) layout of the state dict.
#include "common.h"
#include "s1d.h"
#include <stdlib.h>
#include <string.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;

// -------------------------------------------------------------------------------------------
// geometry shared by host and device
struct Geom {
    int n, cin, cout, h, w;   // cin: channels of the tensor read, cout: channels of the tensor written
    long hw;
    int nb, th, tw;           // pixel box: images x rows x cols
    int tiles_n, tiles_h, tiles_w;
    int rows, cols;           // staged box incl. halo
    int plane;                // LDS stride between staged channels (== 16 mod 32 for fwd, == 2 mod 32 for wgrad)
    int npos;                 // nb * rows * cols
    int npix;                 // nb * th * tw   (<= tile capacity)
    int k_pad, m_pad;         // packed weight extents
    int flat;                 // 1x1: pixels are a flat (n, hw) range, box = [p0, p0 + npix)
    long flat_total;          // n * hw
    int ksplit;               // > 1: blockIdx.z owns a slice of the K chunks and adds into a zeroed output
    // split-bf16 kernel (conv_fwd_x3_kernel): K runs over 32-channel chunks, one MFMA step per tap
    int x_nfull;              // chunks of 32 channels (a remainder of 25..31 channels counts as full, zero-filled)
    int x_tail_oct;           // 8-channel octets of the remainder chunk (0 = none, else 1..3)
    int x_tail_steps;         // ceil(taps * x_tail_oct / 4)
    int x_steps;              // x_nfull * taps + x_tail_steps
    int x_npt;                // input DMA instructions per staged channel = ceil(plane / 64)
    int x_coblk;              // channel blocks (work items = pixel tiles x channel blocks, walked by persistent workgroups)
};

// Forward / dgrad tile configuration per kernel size: 128 pixels per workgroup (4 waves x 2 pixel
// tiles), K chunk of 8 channels x 9 taps (3x3) = 252 MFMAs per wave between barriers; two LDS stages
// of (weight slab + input box) = 78 KB at COT = 7, i.e. two workgroups per CU.
// COT = 8 with 8-channel chunks needs 2 x 47 KB of LDS (one workgroup per CU, measured 73-85 TF);
// 4-channel chunks bring three workgroups back onto the CU.
__host__ __device__ constexpr int fwd_kc(int taps) { return taps == 1 ? 32 : 8; }   // 3x3 may also run with 4
__host__ __device__ constexpr int fwd_pt(int taps) { return 2; }   // PT = 4 needs > 256 VGPR+AGPR (1 wave/SIMD)
__host__ __device__ constexpr int fwd_nxi_max(int taps) { return taps == 1 ? 20 : 9; }

// 16 bytes of zeros: the source of every lane whose element lies outside the image / channel range
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(1))) const void* glb_ptr;

// x / d for 0 <= x < 2^20 and 1 <= d < 2^20 through the fp32 reciprocal `inv` = 1.0f / d: exact at
// these sizes ((x + 0.5) / d is never within 2^-21 of an integer), three instructions instead
// of the ~35 of an integer division.  Used for the per-lane index decodes of the conv prologues.
__device__ __forceinline__ int fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// LDS-DMA: each lane copies 16 / 4 bytes from its own global address to (wave-uniform LDS base +
// lane * size).  No VGPR staging; completion is tracked by vmcnt (hipcc drains it at __syncthreads).
__device__ __forceinline__ void glds16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)lds_wave_base, 4, 0, 0);
}

// -------------------------------------------------------------------------------------------
template <int KH, int KW, int COT, int PT, int KC>
__global__ __launch_bounds__(kThreads) void conv_fwd_kernel(Geom g, const float* __restrict__ in,
                                                            const float* __restrict__ packed,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out, int accumulate) {
    constexpr int TAPS = KH * KW;
    constexpr int CO_BLK = COT * 16;
    constexpr int COS = CO_BLK + ((CO_BLK % 32 == 16) ? 0 : 16);   // == 16 (mod 32)
    constexpr int WROW4 = COS / 4;                                  // float4 slots per LDS weight row
    constexpr int W4_TOTAL = TAPS * KC * WROW4;
    constexpr int NWI = (W4_TOTAL + kThreads - 1) / kThreads;       // weight DMA instructions per wave
    constexpr int WL_FLOATS = TAPS * KC * COS;
    constexpr int PADH = KH / 2, PADW = KW / 2;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int il_floats = KC * g.plane;
    const int buf_floats = WL_FLOATS + ((il_floats + 3) & ~3);      // one stage: weights then input box

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kq = lane >> 4;

    // ---- which box
    int t = blockIdx.x;
    const int twi = t % g.tiles_w; t /= g.tiles_w;
    const int thi = t % g.tiles_h; t /= g.tiles_h;
    const int n0 = t * g.nb;
    const int h0 = thi * g.th, w0 = twi * g.tw;
    const long p0 = (long)blockIdx.x * g.npix;          // flat mode
    const int co0 = blockIdx.y * CO_BLK;

    // ---- per-lane DMA plan (chunk-invariant).  Weight slab: LDS image [TAPS*KC][COS] is filled
    //      linearly, 16 B per lane; instruction iq = q*4 + wid covers float4 slots [iq*64, iq*64+64).
    int w_off[NWI];       // source offset (floats) relative to packed + ci0*m_pad + co0; -1 = skip (row padding)
#pragma unroll
    for (int q = 0; q < NWI; ++q) {
        const int f4 = (q * 4 + wid) * 64 + lane;
        w_off[q] = -1;
        if (f4 < W4_TOTAL) {
            const int row = f4 / WROW4, c4 = f4 - row * WROW4;
            const int tap = row / KC, k = row - tap * KC;
            if (c4 * 4 < CO_BLK) w_off[q] = (tap * g.k_pad + k) * g.m_pad + c4 * 4;
        }
    }
    //      Input box: LDS image [KC][plane], 4 B per lane; instruction iq covers floats [iq*64, iq*64+64).
    const float inv_plane = 1.0f / (float)g.plane, inv_per = 1.0f / (float)(g.rows * g.cols);
    const float inv_cols = 1.0f / (float)g.cols, inv_tw = 1.0f / (float)g.tw, inv_thw = 1.0f / (float)(g.th * g.tw);
    // flat mode: a box of npix consecutive (n, hw) pixels touches image img0 and possibly the next ones
    const unsigned img0 = g.flat ? (unsigned)(p0 / g.hw) : 0u;
    const unsigned rem0 = g.flat ? (unsigned)(p0 - (long)img0 * g.hw) : 0u;
    constexpr int NXI_MAX = fwd_nxi_max(TAPS);
    const int nxi = (il_floats + kThreads - 1) / kThreads;          // <= NXI_MAX (checked on the host)
    int x_off[NXI_MAX];   // offset inside the input tensor relative to channel ci0; -1 = zero fill; -2 = skip
    int x_k[NXI_MAX];
#pragma unroll
    for (int q = 0; q < NXI_MAX; ++q) {
        const int f = (q * 4 + wid) * 64 + lane;
        x_off[q] = -2;
        x_k[q] = 0;
        if (q < nxi && f < il_floats) {
            const int k = fdiv(f, inv_plane), pos = f - k * g.plane;
            x_k[q] = k;
            if (pos < g.npos) {
                long goff = -1;
                if (g.flat) {
                    const long pg = p0 + pos;
                    if (pos < g.npix && pg < g.flat_total) {
                        unsigned img = img0, i = rem0 + (unsigned)pos;
                        while (i >= (unsigned)g.hw) { i -= (unsigned)g.hw; ++img; }
                        goff = ((long)img * g.cin + k) * g.hw + i;
                    }
                } else {
                    const int per = g.rows * g.cols;
                    const int b = fdiv(pos, inv_per), rem = pos - b * per;
                    const int rr = fdiv(rem, inv_cols), cc = rem - rr * g.cols;
                    const int gh = h0 + rr - PADH, gw = w0 + cc - PADW;
                    if (n0 + b < g.n && gh >= 0 && gh < g.h && gw >= 0 && gw < g.w)
                        goff = ((long)(n0 + b) * g.cin + k) * g.hw + (long)gh * g.w + gw;
                }
                x_off[q] = (int)goff;      // tensors on this path stay below 2^31 elements (checked on host)
            }
        }
    }

    // ---- this lane's output pixels
    int pix_l[PT];        // LDS offset of the pixel inside one staged channel (tap (0,0))
    long pix_g[PT];       // offset inside the output tensor for channel 0, or -1
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wid * PT + pt) * 16 + lm;
        pix_l[pt] = 0;
        pix_g[pt] = -1;
        if (p < g.npix) {
            if (g.flat) {
                const long pg = p0 + p;
                pix_l[pt] = p;
                if (pg < g.flat_total) {
                    unsigned img = img0, i = rem0 + (unsigned)p;
                    while (i >= (unsigned)g.hw) { i -= (unsigned)g.hw; ++img; }
                    pix_g[pt] = (long)img * g.cout * g.hw + i;
                }
            } else {
                const int per = g.th * g.tw;
                const int b = fdiv(p, inv_thw), rem = p - b * per;
                const int r = fdiv(rem, inv_tw), c = rem - r * g.tw;
                pix_l[pt] = (b * g.rows + r) * g.cols + c;
                if (n0 + b < g.n && h0 + r < g.h && w0 + c < g.w)
                    pix_g[pt] = (long)(n0 + b) * g.cout * g.hw + (long)(h0 + r) * g.w + (w0 + c);
            }
        }
    }

    f32x4 acc[COT][PT];
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float* zero = g_zero16;
    auto issue_chunk = [&](int ci0, float* buf) {
        const float* wsrc = packed + (long)ci0 * g.m_pad + co0;
#pragma unroll
        for (int q = 0; q < NWI; ++q) {
            if (w_off[q] >= 0) glds16(wsrc + w_off[q], buf + (q * 4 + wid) * 256);
        }
        const float* xsrc = in + (long)ci0 * g.hw;
        float* il = buf + WL_FLOATS;
#pragma unroll
        for (int q = 0; q < NXI_MAX; ++q) {
            if (x_off[q] != -2) {
                const bool live = x_off[q] >= 0 && ci0 + x_k[q] < g.cin;
                glds4(live ? xsrc + x_off[q] : zero, il + (q * 4 + wid) * 64);
            }
        }
    };

    const int nchunks_all = g.k_pad / KC;
    // split-K (late layers whose pixel grid cannot fill the chip): this workgroup owns chunks [c_lo, c_hi)
    const int c_lo = (int)((long)nchunks_all * blockIdx.z / g.ksplit);
    const int c_hi = (int)((long)nchunks_all * (blockIdx.z + 1) / g.ksplit);
    const int k_live = (g.cin + 3) & ~3;       // k-steps beyond the real channels are skipped
    constexpr int NSTEP = TAPS * (KC / 4);

    // one MFMA k-step: 4 input channels (kq picks this lane's) of one tap
    auto load_ab = [&](const float* wl, const float* il, int step, float (&a)[COT], float (&b)[PT]) {
        const int tap = step / (KC / 4), ks = step - tap * (KC / 4);
        const int tapoff = (tap / KW) * g.cols + (tap % KW);
        const float* wrow = wl + (tap * KC + ks * 4 + kq) * COS + lm;
#pragma unroll
        for (int i = 0; i < COT; ++i) a[i] = wrow[i * 16];
        const float* irow = il + (ks * 4 + kq) * g.plane + tapoff;
#pragma unroll
        for (int j = 0; j < PT; ++j) b[j] = irow[pix_l[j]];
    };
    auto mfma_all = [&](const float (&a)[COT], const float (&b)[PT]) {
#pragma unroll
        for (int i = 0; i < COT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    };

    if (c_lo < c_hi) issue_chunk(c_lo * KC, smem + (c_lo & 1) * buf_floats);
    for (int c = c_lo; c < c_hi; ++c) {
        // the DMA of chunk c has landed (vmcnt drained at the barrier) and every wave has finished
        // reading the other stage, which chunk c+1 now overwrites while this chunk is multiplied
        __syncthreads();
        float* cur = smem + (c & 1) * buf_floats;
        if (c + 1 < c_hi) issue_chunk((c + 1) * KC, smem + ((c + 1) & 1) * buf_floats);
        const float* wl = cur;
        const float* il = cur + WL_FLOATS;
        const int ks_live = (k_live - c * KC) >> 2;          // >= 1
        if (ks_live >= KC / 4) {
            // full chunk: branch-free, operands of step s+1 are fetched from LDS while the MFMAs
            // of step s run (register double buffer)
            float a[2][COT], b[2][PT];
            load_ab(wl, il, 0, a[0], b[0]);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (s + 1 < NSTEP) load_ab(wl, il, s + 1, a[(s + 1) & 1], b[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of this step's MFMAs
                mfma_all(a[s & 1], b[s & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // channel tail (c_in not a multiple of KC): only the live k-steps of every tap
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                for (int ks = 0; ks < ks_live; ++ks) {
                    float a[COT], b[PT];
                    load_ab(wl, il, tap * (KC / 4) + ks, a, b);
                    mfma_all(a, b);
                }
            }
        }
    }

    // ---- epilogue: D row = channel (kq*4 + r), column = pixel (lm)
#pragma unroll
    for (int i = 0; i < COT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + i * 16 + kq * 4 + r;
            if (co >= g.cout) continue;
            const float bv = (bias && blockIdx.z == 0) ? bias[co] : 0.f;
#pragma unroll
            for (int j = 0; j < PT; ++j)
                if (pix_g[j] >= 0) {
                    float* o = out + pix_g[j] + (long)co * g.hw;
                    const float v = acc[i][j][r] + bv;
                    if (g.ksplit > 1) atomicAdd(o, v);           // output zeroed by the host (or accumulating)
                    else *o = accumulate ? *o + v : v;
                }
        }
    }
}

// -------------------------------------------------------------------------------------------
// weight (c_out, c_in, kh, kw) -> packed[tap][k][m]  (fwd: k = c_in, m = c_out;
// dgrad: k = c_out, m = c_in, taps mirrored)
__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int c_out, int c_in, int kh,
                            int kw, int k_pad, int m_pad, int dgrad) {
    const int taps = kh * kw;
    const long total = (long)taps * k_pad * m_pad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i % m_pad);
        const long r = i / m_pad;
        const int k = (int)(r % k_pad);
        const int tap = (int)(r / k_pad);
        const int co = dgrad ? k : m, ci = dgrad ? m : k;
        float v = 0.f;
        if (co < c_out && ci < c_in) {
            const int src_tap = dgrad ? (taps - 1 - tap) : tap;
            v = w[((long)co * c_in + ci) * taps + src_tap];
        }
        packed[i] = v;
    }
}

// -------------------------------------------------------------------------------------------
// Forward / dgrad on the bf16 matrix pipe with fp32 results ("x3" = three bf16 limbs per fp32 operand).
//
// gfx950 has no tf32/xf32 MFMA and its fp32 MFMA runs at the vector rate (157 TF), 1/16 of the bf16
// rate.  An fp32 value splits EXACTLY into three bf16 limbs, x = h + m + l (8 + 8 + 8 significant
// bits, round-to-nearest at each level, the residuals are exact fp32 subtractions), so
//     a * b = ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh) + (am*bl + al*bm) + al*bl
// where every limb product is exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16.  NPROD = 9
// keeps all terms (the product a*b is then exact before accumulation, i.e. at least as accurate as
// the fp32 FMA chain of the native kernel); NPROD = 6 drops the last three, whose sum is below
// 2^-23 |a*b| - the size of one fp32 rounding.  6 (9) bf16 MFMAs of K = 32 replace 8 fp32 MFMAs of
// K = 4, a 2.7x (1.8x) higher MFMA ceiling (2075 TF bf16 16x16x32 / 6 = 346 TF).
//
// Data flow per workgroup (4 waves, one per SIMD; COT*16 channels x PT*64 pixels):
//   * weights arrive pre-split from pack_x3_kernel as ready-made A fragments (1 KB = 64 lanes x
//     8 bf16 per (step, channel tile, limb)), streamed by LDS-DMA through a ring of three step slots;
//   * the fp32 input box of a 32-channel chunk [32][plane] is staged by LDS-DMA in two stages;
//   * one MFMA step = one tap x 32 channels; lane group kq owns channel octet kq (its 8 k-values);
//     the B operand (8 channels of one pixel) is read as fp32 from LDS and split in registers
//     while the MFMAs of the previous step run;
//   * DMAs stay in flight across the raw s_barrier of every step: each wave counts its own DMA
//     instructions and waits with s_waitcnt vmcnt(N) only for the slot / stage the next step reads.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kXChunk = 32;        // channels per K chunk
constexpr int kXNptMax = 6;        // input DMA instructions per staged channel (plane <= 384 floats)
constexpr int kXWaves = 8;         // two waves per SIMD: one wave's split / LDS / scalar work hides behind the other's MFMAs

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {      // v_cvt_pk_bf16_f32 (RNE)
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}

// a - b as one v_sub_f32: hipcc otherwise pairs the residual subtractions into v_pk_add_f32, which costs
// ~15 cycles beside MFMAs on gfx950 (measured) against ~4 for the plain instruction
__device__ __forceinline__ float fsub(float a, float b) {
    float r;
    asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// x[0..7] -> three packed bf16x8 limbs with x = h + m + l exactly
__device__ __forceinline__ void split3(const float (&x)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = x[2 * q], x1 = x[2 * q + 1];
        const unsigned hp = cvt_pk_bf16(x0, x1);
        const float r0 = fsub(x0, __uint_as_float(hp << 16)), r1 = fsub(x1, __uint_as_float(hp & 0xffff0000u));
        const unsigned mp = cvt_pk_bf16(r0, r1);
        const float s0 = fsub(r0, __uint_as_float(mp << 16)), s1 = fsub(r1, __uint_as_float(mp & 0xffff0000u));
        h[q] = hp;
        m[q] = mp;
        l[q] = cvt_pk_bf16(s0, s1);
    }
}

// fp16 two-limb split with a power-of-two scale: h = rne16(x * s), l = rne16(x * s - h), one v_fma_mix each
// (the mixed-precision FMA reads the fp32 source and the fp16 half directly and rounds once), so a pair of
// values costs four VALU instructions including the scaling.  |x * s - h - l| <= 2^-24 |x * s|: the two limbs
// carry the fp32 value to within half an fp32 ulp (11 + 11 significand bits plus the sign of the residual).
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
// Inverse of the scale for the epilogue.  A declared maximum of +Inf (an operand holding an infinity; NaNs never
// enter the maximum, they travel through the limbs on their own) leaves no usable scale for the finite elements:
// the inverse is NaN then, so EVERY output of the launch is non-finite instead of silently losing the finite part.
__device__ __forceinline__ float inv_scale(int f, float amax) {
    return ((__float_as_uint(amax) >> 23) & 0xffu) == 0xffu ? __uint_as_float(0x7fc00000u) : field_to_float(254 - f);
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x4 mfma_k32(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// workgroup barrier that neither drains the DMAs in flight nor lets the compiler move LDS accesses across it
__device__ __forceinline__ void raw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// STATS (plain bf16 arithmetic on 1-d rows: the 1-d model's early blocks; the pre-split kernels of conv_l16.hip / conv_l3.hip have
// had it since round 3): the forward of a convolution whose output goes into a BatchNorm (classifiers.py:78-101) also accumulates,
// per lane and output channel, sum (y - pivot), sum (y - pivot)^2, min y, max y of what it stores; one float4 record per (worker,
// wave, channel) at the end -- conv_l16.hip's record format, folded and finalised by fsc_bn_train_stats_conv: the statistics pass
// over the output (21 - 25 us per convolution on cfg 3's 14 - 56 MB tensors) disappears.  The host launches it only where a
// worker keeps ONE channel block (one block in all, or a worker count that is a multiple), no K split, no accumulation.
// POOL (with STATS, single rows): the block's entry convolution followed by MaxPool1d(2) (classifiers.py:149-155): a lane's four
// consecutive positions are two pooling windows -- `out` receives the POOLED tensor (n, c_out, w / 2), `pool_idx` the window
// positions (fsc_maxpool_fwd's format and tie / NaN rule), the statistics are those of the pooled values; the full-resolution
// output is never written (113 MB at cfg 3's first block) and the max-pool launch and the statistics pass behind it disappear.
template <int KH, int KW, int COT, int PT, int NPROD, bool STATS = false, bool POOL = false>
__global__ __launch_bounds__(kXWaves * 64) void conv_fwd_x3_kernel(Geom g, const float* __restrict__ in,
                                                               const float* __restrict__ packed,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ out, int accumulate,
                                                               const float* __restrict__ in_amax,
                                                               const float* __restrict__ w_amax,
                                                               const float* __restrict__ stat_pivot = nullptr,
                                                               float4* __restrict__ stat_rec = nullptr,
                                                               uint8_t* __restrict__ pool_idx = nullptr) {
    static_assert(!POOL || (STATS && KH == 1), "POOL: single rows, with the statistics epilogue");
    constexpr int TAPS = KH * KW;
    constexpr int CO_BLK = COT * 16;
    constexpr int PADH = KH / 2, PADW = KW / 2;
    constexpr bool F16 = NPROD == 3;                // two scaled fp16 limbs, 3 products; else three bf16 limbs, 6 / 9 ...
    constexpr int NL = NPROD == 1 ? 1 : F16 ? 2 : 3;   // ... or ONE bf16 limb, one product: plain bf16 arithmetic (NPROD = 1)
    constexpr int KCH = kXChunk;                    // channels per K chunk
    constexpr int SPC = TAPS;                       // steps per full chunk (one per tap)
    constexpr int NSTG = TAPS == 1 ? (PT == 2 ? 3 : 4) : 2;   // input stages: the input DMA runs NSTG - 1 chunks ahead
                                                    // (a 1x1 chunk is a single step)
    constexpr int WUNITS = COT * NL;                // 1 KB fragment images per step
    constexpr int WSLOT_F = WUNITS * 256;           // floats per ring slot
    constexpr int NWQ = (WUNITS + kXWaves - 1) / kXWaves;
    constexpr int RING = (COT > 8 || (TAPS == 1 && PT == 2)) ? 2 : 3;   // weight slots: W runs RING - 1 steps ahead
    constexpr int AHEAD = RING - 1;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wring = smem;                                  // RING step slots
    float* const ibase = smem + RING * WSLOT_F;                 // NSTG stages of [KCH][plane]
    const int istage = KCH * g.plane;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kq = lane >> 4;

    // ---- tile-invariant decode.  The workgroup is persistent: it walks the work items (pixel tile, channel
    //      block) = divmod(item, co_blocks) for item = blockIdx.x, blockIdx.x + gridDim.x, ... and keeps ONE
    //      stream of MFMA steps and DMAs running across them (the next item's weights and first input box are
    //      fetched during the current item's last steps), so only the output stores interrupt the MFMA stream.
    //      Neighbouring workgroups hold the channel blocks of the same pixel tile at the same time (L2 reuse).

    const float inv_per = 1.0f / (float)(g.rows * g.cols), inv_cols = 1.0f / (float)g.cols;
    const float inv_tw = 1.0f / (float)g.tw, inv_thw = 1.0f / (float)(g.th * g.tw);
    int pos_dec[kXNptMax];       // staged position q*64 + lane as (image << 20 | row << 10 | col); -1 = outside the box
#pragma unroll
    for (int q = 0; q < kXNptMax; ++q) {
        const int pos = q * 64 + lane;
        pos_dec[q] = -1;
        if (pos < g.npos) {
            const int per = g.rows * g.cols;
            const int b = fdiv(pos, inv_per), rem = pos - b * per;
            const int rr = fdiv(rem, inv_cols), cc = rem - rr * g.cols;
            pos_dec[q] = (b << 20) | (rr << 10) | cc;
        }
    }
    int pix_l[PT], pix_dec[PT];  // LDS offset of this lane's pixels inside a staged channel; (image, row, col) in the box
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wid * PT + pt) * 16 + lm;
        pix_l[pt] = 0;
        pix_dec[pt] = -1;
        if (p < g.npix) {
            const int per = g.th * g.tw;
            const int b = fdiv(p, inv_thw), rem = p - b * per;
            const int r = fdiv(rem, inv_tw), c = rem - r * g.tw;
            pix_l[pt] = (b * g.rows + r) * g.cols + c;
            pix_dec[pt] = (b << 20) | (r << 10) | c;
        }
    }
    const int ntiles = g.tiles_n * g.tiles_h * g.tiles_w;
    // per tile: input offsets of the staged positions (relative to channel 0; -1 = zero fill) ...
    int pos_off[kXNptMax];
    // ... 1x1 kernels (every step opens a chunk, so the copy issue is on the critical path of every step) keep running
    // source pointers instead: this lane's pointer for each staged position at the NEXT channel this wave copies
    // (channels wid, wid + 8, ... of chunk after chunk), advanced by `istep` after every copy; positions outside the
    // image point at a zero word with step 0, so a copy costs one 64-bit increment and no selects (+8 % on the
    // 100-channel layer; on the 3x3 kernels the extra registers cost more than the instructions saved)
    constexpr bool RUNPTR = TAPS == 1;
    const char* iptr[RUNPTR ? kXNptMax : 1];
    int istep[RUNPTR ? kXNptMax : 1];
    const int c_lo_first = (int)((long)(g.x_nfull + (g.x_tail_oct ? 1 : 0)) * blockIdx.z / g.ksplit) * KCH + wid;
    const int chan_step = (int)(8 * g.hw * (long)sizeof(float));          // (host: hw < 2^26)
    auto plan_input = [&](int tile) {
        int t = tile;
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h; t /= g.tiles_h;
        const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
#pragma unroll
        for (int q = 0; q < kXNptMax; ++q) {
            pos_off[q] = -1;
            if (pos_dec[q] >= 0) {
                const int b = pos_dec[q] >> 20, rr = (pos_dec[q] >> 10) & 1023, cc = pos_dec[q] & 1023;
                const int gh = h0 + rr - PADH, gw = w0 + cc - PADW;
                if (n0 + b < g.n && gh >= 0 && gh < g.h && gw >= 0 && gw < g.w)
                    pos_off[q] = (int)(((long)(n0 + b) * g.cin) * g.hw + (long)gh * g.w + gw);
            }
            if constexpr (RUNPTR) {
                iptr[q] = pos_off[q] >= 0 ? reinterpret_cast<const char*>(in + pos_off[q] + (long)c_lo_first * g.hw)
                                          : reinterpret_cast<const char*>(g_zero16);
                istep[q] = pos_off[q] >= 0 ? chan_step : 0;
            }
        }
    };
    // ... and the output side.  The epilogue transposes each 16-channel x 16-pixel accumulator tile through a
    // per-wave LDS scratch so that a lane owns FOUR CONSECUTIVE PIXELS of one channel (box widths are multiples
    // of 4) and writes them with one 16-byte store: 64 lanes x 4 bytes cost the same 16 address cycles as
    // 64 lanes x 16 bytes, and the dword version of this epilogue took 18 k cycles per tile.
    // Lane -> channel (lane >> 2) of the tile, pixels 4 * (lane & 3) .. + 3.
    constexpr int SCR = 20;                       // scratch row stride in floats (16 pixels + pad, 16-byte rows)
    float* const scratch = ibase + NSTG * istage + wid * (16 * SCR);
    int quad_dec[PT];                             // (image, row, col) of this lane's first quad pixel; -1 = outside the box
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wid * PT + pt) * 16 + (lane & 3) * 4;
        quad_dec[pt] = -1;
        if (p < g.npix) {
            const int per = g.th * g.tw;
            const int b = fdiv(p, inv_thw), rem = p - b * per;
            const int r = fdiv(rem, inv_tw), c = rem - r * g.tw;
            quad_dec[pt] = (b << 20) | (r << 10) | c;
        }
    }
    long quad_g[PT];                              // offset of the quad's first pixel in the output tensor (channel 0)
    int quad_ok[PT];                              // bit k: pixel k of the quad lies inside the image
    const int pool_ow = g.w >> 1;                 // (POOL) windows per row
    auto plan_output = [&](int tile) {
        int t = tile;
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h; t /= g.tiles_h;
        const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            quad_g[pt] = 0;
            quad_ok[pt] = 0;
            if (quad_dec[pt] >= 0) {
                const int b = quad_dec[pt] >> 20, r = (quad_dec[pt] >> 10) & 1023, c = quad_dec[pt] & 1023;
                if (n0 + b < g.n && h0 + r < g.h) {
                    quad_g[pt] = POOL ? (long)(n0 + b) * g.cout * pool_ow + ((w0 + c) >> 1)        // (columns are multiples of 4)
                                      : (long)(n0 + b) * g.cout * g.hw + (long)(h0 + r) * g.w + (w0 + c);
                    const int left = g.w - (w0 + c), inbox = g.npix - ((wid * PT + pt) * 16 + (lane & 3) * 4);
                    const int nv = left < inbox ? left : inbox;       // valid pixels of the quad
                    quad_ok[pt] = nv >= 4 ? 15 : nv <= 0 ? 0 : (1 << nv) - 1;
                }
            }
        }
    };

    f32x4 acc[COT][PT];
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NST = STATS ? COT : 1;
    float st_s1[NST], st_s2[NST], st_mn[NST], st_mx[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) { st_s1[i] = 0.f; st_s2[i] = 0.f; st_mn[i] = INFINITY; st_mx[i] = -INFINITY; }

    // ---- fp16 limbs: activations are scaled to [2^14, 2^15) by a power of two from the tensor's largest magnitude
    //      (the weights were scaled the same way when they were packed); the epilogue undoes both exactly
    float sx = 1.f, inv_x = 1.f, inv_w = 1.f;
    if constexpr (F16) {
        static_assert(kXWaves * 64 == fsc::kAmaxFloats, "one amax slot float per thread");
        float* const red = smem;                       // (before any DMA lands in the weight ring)
        const float mw = fsc::wave_max(in_amax[tid]);
        if (lane == 0) red[wid] = mw;
        __syncthreads();
        float ax = red[0];
#pragma unroll
        for (int i = 1; i < kXWaves; ++i) ax = fmaxf(ax, red[i]);
        ax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ax)));
        __syncthreads();
        const float aw = *w_amax;
        const int fx = scale_field(ax), fw = scale_field(aw);
        sx = field_to_float(fx);
        inv_x = inv_scale(fx, ax);
        inv_w = inv_scale(fw, aw);
    }

    // ---- K range of this workgroup (split-K over chunks through blockIdx.z)
    const int nchunks = g.x_nfull + (g.x_tail_oct ? 1 : 0);
    const int c_lo = (int)((long)nchunks * blockIdx.z / g.ksplit);
    const int c_hi = (int)((long)nchunks * (blockIdx.z + 1) / g.ksplit);
    const int s_lo = c_lo * SPC;                                     // chunks below the tail are all full
    const int s_hi = c_hi == nchunks ? g.x_steps : c_hi * SPC;       // >= s_lo + 2 (host)

    // ---- DMA issue.  Weights: unit u of a step goes to wave u % 8.  Input: channel kk of a chunk to wave kk % 8.
    const float* zero = g_zero16;
    const float* const wbase0 = packed + (long)s_lo * WSLOT_F + lane * 4;    // W(s_lo) of channel block 0, this lane
    const long wblk = (long)g.x_steps * WSLOT_F;                              // stride between channel blocks
    const float* wsrc = wbase0;                                              // W(S) of the current item
    const float* wnext = wbase0;                                             // W(s_lo) of the next item
    auto issue_w = [&](const float* src, int slot) {
        float* dst = wring + slot * WSLOT_F;
#pragma unroll
        for (int q = 0; q < NWQ; ++q) {
            const int u = q * kXWaves + wid;
            if ((q + 1) * kXWaves <= WUNITS || u < WUNITS) glds16(src + u * 256, dst + u * 256);   // only the last round is partial
        }
    };
    auto issue_i = [&](int stage, int c) {                          // uses the current pos_off[]
        const int ci0 = c * KCH;
        const int nch = c < g.x_nfull ? KCH : g.x_tail_oct * 8;         // staged channels of this chunk
        float* dst = ibase + stage * istage;
        if constexpr (RUNPTR) {
            float* dw = dst + wid * g.plane;
#pragma unroll 1
            for (int kk = wid; kk < nch; kk += kXWaves) {
                if (ci0 + kk < g.cin) {                                 // (wave-uniform)
#pragma unroll
                    for (int q = 0; q < kXNptMax; ++q)
                        if (q < g.x_npt && q * 64 + lane < g.plane) {
                            glds4(reinterpret_cast<const float*>(iptr[q]), dw + q * 64);
                            iptr[q] += istep[q];
                        }
                } else {                                                // channels beyond c_in (last chunk only): zeros
#pragma unroll
                    for (int q = 0; q < kXNptMax; ++q)
                        if (q < g.x_npt && q * 64 + lane < g.plane) glds4(zero, dw + q * 64);
                }
                dw += kXWaves * g.plane;
            }
            return;
        }
        for (int kk = wid; kk < nch; kk += kXWaves) {
            const float* src = in + (long)(ci0 + kk) * g.hw;
            const bool ch_live = ci0 + kk < g.cin;
#pragma unroll
            for (int q = 0; q < kXNptMax; ++q) {
                if (q < g.x_npt) {
                    if (q * 64 + lane < g.plane) {
                        const bool live = ch_live && pos_off[q] >= 0;
                        glds4(live ? src + pos_off[q] : zero, dst + kk * g.plane + q * 64);
                    }
                }
            }
        }
    };
    // Outstanding-DMA bookkeeping: a wave issues >= NWLO weight units per step and exactly (KCH / 8) * x_npt input
    // instructions per full chunk.  At the barrier of step S the weights W(S) (issued two steps ago) must have
    // landed; younger and allowed to stay in flight are W(S+1) and an input box issued in the last two steps that
    // the coming step does not read yet.  Counting less than what is really in flight only waits longer (the
    // output stores of the previous tile share the counter: the first barriers of a tile also wait for them).
    constexpr int NWLO = RING == 3 ? WUNITS / kXWaves : 0;     // two slots: W(S) was issued in step S-1, nothing newer
    // `w_young`: the previous step issued weights (W(S+1)); without them (last steps of a K slice with no item to
    // follow) the youngest outstanding units are W(S) itself and nothing may be left in flight.
    auto wait_weights = [&](bool w_young, bool input_in_flight) {
#define FSC_VMW(k) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(k) : "memory")
        if (input_in_flight && w_young) {
            switch (g.x_npt) {
                case 3: FSC_VMW(NWLO + 3 * (KCH / 8)); break;
                case 4: FSC_VMW(NWLO + 4 * (KCH / 8)); break;
                case 5: FSC_VMW(NWLO + 5 * (KCH / 8)); break;
                case 6: FSC_VMW(NWLO + 6 * (KCH / 8)); break;
                default: FSC_VMW(NWLO); break;
            }
        } else if (input_in_flight) {
            switch (g.x_npt) {
                case 3: FSC_VMW(3 * (KCH / 8)); break;
                case 4: FSC_VMW(4 * (KCH / 8)); break;
                case 5: FSC_VMW(5 * (KCH / 8)); break;
                case 6: FSC_VMW(6 * (KCH / 8)); break;
                default: FSC_VMW(0); break;
            }
        } else if (w_young) {
            FSC_VMW(NWLO);
        } else {
            FSC_VMW(0);
        }
#undef FSC_VMW
    };

    // ---- B operand: 8 channels (this lane group's octet) of one pixel at the tap of step s of chunk c.
    //      b_base is this lane's pointer to (octet, tap) inside the staged box of `stage`.
    auto b_base = [&](int stage, int c, int s) -> const float* {
        const float* st = ibase + stage * istage;
        if (c < g.x_nfull) {                               // full chunk: step = (tap, 32-channel group), lane group = octet
            const int ty = s / KW, tx = s - ty * KW;       // wave-uniform
            return st + kq * 8 * g.plane + ty * g.cols + tx;
        }
        const int noct = g.x_tail_oct;
        int gi = 4 * s + kq;
        if (gi >= TAPS * noct) gi = 0;                     // its weights are zero
        const int tap = TAPS == 1 ? 0 : fdiv(gi, 1.0f / (float)noct);
        const int oct = gi - tap * noct;
        const int ty = fdiv(tap, 1.0f / (float)KW), tx = tap - ty * KW;
        return st + oct * 8 * g.plane + ty * g.cols + tx;
    };

    struct Limbs { u32x4 v[NL][PT]; };            // v[0] = high limb ... v[NL - 1] = low limb
    Limbs lb0, lb1;
    auto split_all = [&](const float (&x)[8], Limbs& dst, int j) {
        if constexpr (F16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned hp, lp;
                split2_pair(x[2 * q], x[2 * q + 1], sx, hp, lp);
                dst.v[0][j][q] = hp;
                dst.v[1][j][q] = lp;
            }
        } else if constexpr (NL == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dst.v[0][j][q] = cvt_pk_bf16(x[2 * q], x[2 * q + 1]);
        } else {
            split3(x, dst.v[0][j], dst.v[1][j], dst.v[2][j]);
        }
    };
    const int nitems = ntiles * g.x_coblk;
    int item = blockIdx.x;
    // Input producer: a cursor (item, chunk, stage) that runs NSTG - 1 chunks ahead of the MFMA steps, across items;
    // pos_off[] always belongs to the producer's pixel tile.
    int p_item = blockIdx.x, p_c = c_lo, p_stg = 0;
    auto produce = [&]() -> bool {                                   // returns: a full 32-channel chunk was issued
        if (p_item >= nitems) return false;
        issue_i(p_stg, p_c);
        const bool full = p_c < g.x_nfull;
        p_stg = p_stg == NSTG - 1 ? 0 : p_stg + 1;
        if (++p_c == c_hi) {
            p_c = c_lo;
            p_item += gridDim.x;
            if (p_item < nitems) plan_input(p_item / g.x_coblk);
        }
        return full;
    };
    if (item < nitems) {
        const int tile = item / g.x_coblk;
        wsrc = wbase0 + (item - tile * g.x_coblk) * wblk;
        plan_input(tile);
#pragma unroll
        for (int d = 0; d < NSTG - 1; ++d) produce();
        issue_w(wsrc, 0);
        if (RING == 3) issue_w(wsrc + WSLOT_F, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_barrier();
        const float* il = b_base(0, c_lo, 0);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            float raw[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) raw[e] = il[e * g.plane + pix_l[j]];
            split_all(raw, lb0, j);
        }
    }

    // incremental step state (all wave-uniform)
    int c = c_lo, sc = 0;                                            // chunk of step S and step inside it
    int nst = c < g.x_nfull ? SPC : g.x_tail_steps;
    int slot = 0;                                                    // ring slot of W(S)
    int stg = 0;                                                     // input stage of chunk c
    int input_age = 99;                                              // steps since a full-chunk input box was issued
    bool first_step = true;                                          // the prologue has already synchronised for it
    bool has_next = false;                                           // another item follows the current one
    bool stores_pending = false;                                     // the previous tile's output stores may be in flight
    bool w_prev = false;                                             // the previous step issued weight DMAs

    // One MFMA step.  `cur` holds the split B operand of step S; the B operand of the next step (of the next
    // tile after a tile's last step) is read from LDS and split into `nxt` between the MFMAs.  Phase i = the
    // PT*NPROD MFMAs of channel tile i in NPROD groups of PT independent MFMAs (one limb pair each); the A
    // fragments of tile i+1, the raw B reads and the split arithmetic are placed between the groups at
    // compile time, and the DMA issue after phase 0, so that the second wave of the SIMD always finds MFMAs
    // of this wave to overlap with.
    // The MFMA / side-work body of one step (see `step` below): `il` = this lane's B pointer of the next step,
    // `wl` = this lane's A fragments of this step; `dma` is called once after phase 0.
    auto phases = [&](const float* il, const u32x4* wl, const Limbs& cur, Limbs& nxt, auto&& dma) {
        float raw[PT][8];
        u32x4 a[2][NL];
#pragma unroll
        for (int p = 0; p < NL; ++p) a[0][p] = wl[p * 64];
#pragma unroll
        for (int i = 0; i < COT; ++i) {
#pragma unroll
            for (int gq = 0; gq < NPROD; ++gq) {
                // ---- the side work of this MFMA group (all compile-time placement)
                if (gq == 0 && i + 1 < COT) {
#pragma unroll
                    for (int p = 0; p < NL; ++p) a[(i + 1) & 1][p] = wl[((i + 1) * NL + p) * 64];
                }
#pragma unroll
                for (int j = 0; j < PT; ++j) {
                    const int rp = (j >> 1) < COT - 1 ? (j >> 1) : COT - 1;       // phase that reads raw[j]
                    int sp = rp + 1 > COT - PT + j ? rp + 1 : COT - PT + j;        // phase that splits it
                    if (sp > COT - 1) sp = COT - 1;
                    if (rp == i) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (((j & 1) * 8 + e) % NPROD == gq && (rp != sp || gq < 2))
                                raw[j][e] = il[e * g.plane + pix_l[j]];
                        if (rp == sp && gq == 1) {      // single-phase tiles: read everything up front
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (((j & 1) * 8 + e) % NPROD >= 2) raw[j][e] = il[e * g.plane + pix_l[j]];
                        }
                    }
                    if constexpr (F16) {
                        if (sp == i) {
                            // four pair splits of four instructions: groups 0, 1, 1, 2 (all in the last group when
                            // the raw values are read in this same phase)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int gsel = rp == sp ? 2 : (q == 0 ? 0 : q == 3 ? 2 : 1);
                                if (gsel == gq) {
                                    unsigned hp, lp;
                                    split2_pair(raw[j][2 * q], raw[j][2 * q + 1], sx, hp, lp);
                                    asm volatile("" : "+v"(hp), "+v"(lp));
                                    nxt.v[0][j][q] = hp;
                                    nxt.v[1][j][q] = lp;
                                }
                            }
                        }
                    } else if constexpr (NL == 1) {
                        if (sp == i) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                unsigned hp = cvt_pk_bf16(raw[j][2 * q], raw[j][2 * q + 1]);
                                asm volatile("" : "+v"(hp));
                                nxt.v[0][j][q] = hp;
                            }
                        }
                    } else if (sp == i && gq >= 2 && gq < 6) {
                        const int q = gq - 2;
                        const float x0 = raw[j][2 * q], x1 = raw[j][2 * q + 1];
                        unsigned hp = cvt_pk_bf16(x0, x1);
                        const float r0 = fsub(x0, __uint_as_float(hp << 16)), r1 = fsub(x1, __uint_as_float(hp & 0xffff0000u));
                        unsigned mp = cvt_pk_bf16(r0, r1);
                        const float s0 = fsub(r0, __uint_as_float(mp << 16)), s1 = fsub(r1, __uint_as_float(mp & 0xffff0000u));
                        unsigned lp = cvt_pk_bf16(s0, s1);
                        // pin the split to this group (otherwise it is sunk to its first use in the next step)
                        asm volatile("" : "+v"(hp), "+v"(mp), "+v"(lp));
                        nxt.v[0][j][q] = hp;
                        nxt.v[1][j][q] = mp;
                        nxt.v[NL - 1][j][q] = lp;
                    }
                }
                // ---- PT independent MFMAs (one limb pair, every pixel tile), smallest products first
                constexpr int kPairA[9] = {1, 0, 2, 0, 1, 0, 2, 1, 2};      // bf16: 0 = h, 1 = m, 2 = l
                constexpr int kPairB[9] = {1, 2, 0, 1, 0, 0, 2, 2, 1};      // x6 uses the first six
                constexpr int kPairA2[3] = {1, 0, 0}, kPairB2[3] = {0, 1, 0};   // fp16: 0 = h, 1 = l
                const int pa = NL == 1 ? 0 : F16 ? kPairA2[gq % 3] : NPROD == 9 ? kPairA[(gq + 6) % 9] : kPairA[gq % 9];
                const int pb = NL == 1 ? 0 : F16 ? kPairB2[gq % 3] : NPROD == 9 ? kPairB[(gq + 6) % 9] : kPairB[gq % 9];
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = mfma_k32<F16>(a[i & 1][pa], cur.v[pb][j], acc[i][j]);
#pragma unroll
                for (int k = 0; k < PT; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // up to two LDS reads
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // up to four VALU
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (i == 0) {
                dma();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // One MFMA step.  `cur` holds the split B operand of step S; the B operand of the next step (of the next
    // item after an item's last step) is read from LDS and split into `nxt` between the MFMAs.  Phase i = the
    // PT*NPROD MFMAs of channel tile i in NPROD groups of PT independent MFMAs (one limb pair each); the A
    // fragments of tile i+1, the raw B reads and the split arithmetic are placed between the groups at
    // compile time, and the DMA issue after phase 0, so that the second wave of the SIMD always finds MFMAs
    // of this wave to overlap with.
    // Interior steps of a full chunk (7 of 9 for 3x3) take `step_fast`: no input box to issue, no chunk / item /
    // ring wrap, so the scalar bookkeeping shrinks to a few instructions (every instruction beside the MFMAs
    // costs: about two per MFMA are free, measured).
    auto step_fast = [&](const Limbs& cur, Limbs& nxt) {
        // the next step: the next tap of this chunk, or tap 0 of the next (full) chunk in the next stage -- then its
        // box (issued at this chunk's first step, possibly only AHEAD steps ago: k3 kernels) must have landed too
        const bool wrap = sc + 1 == SPC;
        wait_weights(true, input_age <= AHEAD - 1 && !wrap);      // (the previous step had S + AHEAD < s_hi as well)
        raw_barrier();
        const int scn = wrap ? 0 : sc + 1;
        const int stgn = wrap ? (stg == NSTG - 1 ? 0 : stg + 1) : stg;
        const int ty = scn / KW, tx = scn - ty * KW;
        const float* il = ibase + stgn * istage + kq * 8 * g.plane + ty * g.cols + tx;
        const u32x4* wl = reinterpret_cast<const u32x4*>(wring + slot * WSLOT_F) + lane;
        phases(il, wl, cur, nxt, [&] {
            issue_w(wsrc + AHEAD * WSLOT_F, slot == 0 ? RING - 1 : slot - 1);
            w_prev = true;
            ++input_age;
        });
        wsrc += WSLOT_F;
        slot = slot == RING - 1 ? 0 : slot + 1;
        sc = scn;
        stg = stgn;
        c += wrap ? 1 : 0;
    };
    auto step_slow = [&](int S, const Limbs& cur, Limbs& nxt) {
        const bool tile_end = S + 1 >= s_hi;
        const bool last = tile_end && !has_next;                     // nothing follows: no prefetch
        int cn = c, sn = sc + 1, stg_n = stg;                        // coordinates of the next step
        const int stg_up = stg == NSTG - 1 ? 0 : stg + 1;
        if (sn == nst) { cn = c + 1; sn = 0; stg_n = stg_up; }
        if (tile_end) { cn = c_lo; sn = 0; stg_n = stg_up; }
        if (last) { cn = c; sn = sc; stg_n = stg; }
        if (!first_step) {
            // stores share the VM counter with the DMAs and may retire out of order with them: the first
            // barrier after a tile's output stores drains everything
            if (stores_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (TAPS == 1) wait_weights(w_prev, NSTG == 4 && input_age == 0);   // every step opens a chunk: with four
                                                                      // stages the box issued in the previous step (younger than
                                                                      // the one needed now) may stay in flight
            else wait_weights(w_prev, input_age <= AHEAD - 1 && stg_n == stg);
            raw_barrier();
        }
        first_step = false;
        stores_pending = false;
        const float* il = b_base(stg_n, cn, sn);
        const u32x4* wl = reinterpret_cast<const u32x4*>(wring + slot * WSLOT_F) + lane;
        phases(il, wl, cur, nxt, [&] {
                // DMA issue for the step after next (its slot was read during the previous step; the weight
                // stream wraps around at an item end) and, at the first step of a chunk, of the input box
                // NSTG - 1 chunks ahead (its stage was last read during the previous step)
                // (weights FIRST, then the box: the weights of step S + AHEAD must be OLDER than a box issued in the same step -- the
                // counted wait of step S + AHEAD leaves "the youngest weights + one box" in flight and needs these weights landed.
                // Round 5 tried the other order: the batch-32 model test turned flaky.)
                w_prev = true;
                if (S + AHEAD < s_hi) issue_w(wsrc + AHEAD * WSLOT_F, slot == 0 ? RING - 1 : slot - 1);
                else if (has_next) issue_w(wnext + (S + AHEAD - s_hi) * WSLOT_F, slot == 0 ? RING - 1 : slot - 1);
                else w_prev = false;
                ++input_age;
                if (sc == 0 && produce()) input_age = 0;
        });
        // advance
        wsrc = tile_end ? wnext : wsrc + WSLOT_F;
        slot = slot == RING - 1 ? 0 : slot + 1;
        if (cn != c || tile_end) nst = cn < g.x_nfull ? SPC : g.x_tail_steps;
        c = cn;
        sc = sn;
        stg = stg_n;
    };
    // dispatch: the slow step that opens a full chunk counts the interior steps that may take the lean path
    int fast_left = 0;
    auto step = [&](int S, const Limbs& cur, Limbs& nxt) {
        if (fast_left > 0) {
            --fast_left;
            step_fast(cur, nxt);
        } else {
            const bool opens_full_chunk = sc == 0 && c < g.x_nfull;
            step_slow(S, cur, nxt);
            if (opens_full_chunk) {
                // steps S+1 .. of this chunk with sc >= 1 and S' + AHEAD < s_hi: the interior ones (sc + 1 < SPC) and,
                // when another full chunk of this K slice follows, the last one
                int nf = SPC - 2 + ((SPC > 1 && c + 1 < g.x_nfull && c + 1 < c_hi) ? 1 : 0);
                const int cap = s_hi - AHEAD - (S + 1);
                if (nf > cap) nf = cap;
                fast_left = nf > 0 ? nf : 0;
            }
        }
    };

    const bool add_bias = bias != nullptr && blockIdx.z == 0;
    for (; item < nitems; item += gridDim.x) {
        const int tile = item / g.x_coblk;
        const int co0 = (item - tile * g.x_coblk) * CO_BLK;
        has_next = item + (int)gridDim.x < nitems;
        if (has_next) {
            const int inext = item + gridDim.x;
            wnext = wbase0 + (inext - (inext / g.x_coblk) * g.x_coblk) * wblk;
        }
        plan_output(tile);
        for (int S = s_lo; S < s_hi; S += 2) {
            step(S, lb0, lb1);
            if (S + 1 < s_hi) step(S + 1, lb1, lb0);
            else lb0 = lb1;                       // odd number of steps: the next tile starts from lb0 again
        }

        // ---- epilogue (see plan_output): D row = channel (kq*4 + r), column = pixel (lm) -> scratch[ch][px] ->
        //      lane = (channel, pixel quad).  The plane size and the bias pointer are made opaque per item:
        //      otherwise hipcc hoists the channel offsets and bias values out of the item loop and spills.
        long hw_t = g.hw;
        const float* bias_t = bias;
        asm volatile("" : "+s"(hw_t), "+s"(bias_t));
        const bool plain = g.ksplit == 1 && !accumulate;
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
                for (int r = 0; r < 4; ++r)
                    scratch[(kq * 4 + r) * SCR + lm] = F16 ? fmaf(acc[i][j][r] * inv_x, inv_w, bv[r]) : acc[i][j][r] + bv[r];
                const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + ch * SCR + (lane & 3) * 4);
                if constexpr (POOL) {
                    if (co < g.cout) {
                        float* o = out + quad_g[j] + (long)co * pool_ow;
                        uint8_t* oi = pool_idx + quad_g[j] + (long)co * pool_ow;
#pragma unroll
                        for (int wdw = 0; wdw < 2; ++wdw) {
                            if ((quad_ok[j] & (3 << (2 * wdw))) != (3 << (2 * wdw))) continue;     // both positions inside the row
                            float best = v[2 * wdw];
                            int bi = 0;
                            const float u = v[2 * wdw + 1];
                            if (u > best || u != u) { best = u; bi = 1; }
                            o[wdw] = best;
                            oi[wdw] = (uint8_t)bi;
                            const float a = best - pv;
                            st_s1[i] += a;
                            st_s2[i] = fmaf(a, a, st_s2[i]);
                            st_mn[i] = fminf(st_mn[i], best);
                            st_mx[i] = fmaxf(st_mx[i], best);
                        }
                    }
                    continue;
                }
                if constexpr (STATS) {
                    if (co < g.cout && quad_ok[j]) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (quad_ok[j] & (1 << k)) {
                                const float a = v[k] - pv;
                                st_s1[i] += a;
                                st_s2[i] = fmaf(a, a, st_s2[i]);
                                st_mn[i] = fminf(st_mn[i], v[k]);
                                st_mx[i] = fmaxf(st_mx[i], v[k]);
                            }
                    }
                }
                if (co < g.cout && quad_ok[j]) {
                    float* o = out + quad_g[j] + (long)co * hw_t;
                    if (plain && quad_ok[j] == 15) {
                        *reinterpret_cast<f32x4*>(o) = v;            // 4-byte aligned 16-byte store
                    } else if (g.ksplit == 1 && quad_ok[j] == 15) {  // accumulate: 16-byte read-modify-write
                        const f32x4 old = *reinterpret_cast<const f32x4*>(o);
                        *reinterpret_cast<f32x4*>(o) = old + v;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (quad_ok[j] & (1 << k)) {
                                if (g.ksplit > 1) atomicAdd(o + k, v[k]);
                                else o[k] = accumulate ? o[k] + v[k] : v[k];
                            }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < COT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        stores_pending = true;
    }
    if constexpr (STATS) {
        // the four lanes of a quad hold the same channel: fold them, lane (lane & 3) == 0 writes the record
        constexpr int CO_BLK_ = COT * 16;
#pragma unroll
        for (int i = 0; i < COT; ++i) {
            float a = st_s1[i], b = st_s2[i], mn = st_mn[i], mx = st_mx[i];
            a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64);
            b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64);
            mn = fminf(mn, __shfl_xor(mn, 1, 64)); mn = fminf(mn, __shfl_xor(mn, 2, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
            // (transposed: [channel][slot] -- the fold's workgroup of a channel reads its slots contiguously; layout order 2)
            if ((lane & 3) == 0) {
                const int chn = (int)(blockIdx.x % g.x_coblk) * CO_BLK_ + i * 16 + (lane >> 2);
                stat_rec[(long)chn * ((long)gridDim.x * kXWaves) + (long)blockIdx.x * kXWaves + wid] = make_float4(a, b, mn, mx);
            }
        }
    }
}

// max |x| of a tensor as the bit pattern of a non-negative float (ordered like an unsigned integer).
// `out` must hold 0 before a multi-block launch; with one block (`direct`) the result is stored.
// `slots` > 1: the fsc::publish_amax buffer format (kAmaxFloats floats, the result is the maximum over all of them).
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long n, float* __restrict__ out, int direct,
                                                   int slots) {
    __shared__ float sm[4];
    float m = 0.f;
    const long n4 = n >> 2, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fabsf(v[1]), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        if (direct) *out = m;
        else atomicMax(reinterpret_cast<unsigned*>(out + (blockIdx.x % slots) * fsc::kAmaxStride), __float_as_uint(m));
    }
}

// weight (c_out, c_in, kh, kw) -> A fragments of conv_fwd_x3_kernel:
// packed[co block][step][channel tile][limb][lane][8 bf16]; lane = (kq, m), its 8 values are the channels
// of octet `oct` at tap `tap` where (tap, oct) = divmod(4 * step_in_chunk + kq, octets of the chunk)
// `w_amax` != null: two fp16 limbs of the weights scaled by scale_field(*w_amax) instead of three bf16 limbs
__device__ __forceinline__ void pack_x3_items(const float* __restrict__ w, unsigned short* __restrict__ packed, int c_out, int c_in,
                                              int taps, int cot, int co_blocks, int nfull, int tail_oct, int steps, int nch, int dgrad,
                                              const float* __restrict__ w_amax, int nl, long first, long stride) {
    const long total = (long)co_blocks * steps * cot * 512;
    const float sw = w_amax ? field_to_float(scale_field(*w_amax)) : 1.f;
    for (long idx = first; idx < total; idx += stride) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        long rest = idx >> 9;
        const int i = (int)(rest % cot); rest /= cot;
        const int S = (int)(rest % steps);
        const int cb = (int)(rest / steps);
        const int spc = taps * nch;                          // steps per full chunk of 32 * nch channels
        const int c = S < nfull * spc ? S / spc : nfull;
        const int s = S - c * spc;
        const int noct = c < nfull ? 4 * nch : tail_oct;
        const int gi = 4 * s + (lane >> 4);
        float v = 0.f;
        if (gi < taps * noct) {
            const int tap = gi / noct, oct = gi - tap * noct;
            const int k = c * kXChunk * nch + oct * 8 + e, m = (cb * cot + i) * 16 + (lane & 15);
            const int co = dgrad ? k : m, ci = dgrad ? m : k;
            if (co < c_out && ci < c_in) v = w[((long)co * c_in + ci) * taps + (dgrad ? taps - 1 - tap : tap)];
        }
        if (w_amax) {
            unsigned h2, l2;
            split2_pair(v, 0.f, sw, h2, l2);
            const long base2 = ((((long)cb * steps + S) * cot + i) * 2) * 512 + lane * 8 + e;
            packed[base2] = (unsigned short)h2;
            packed[base2 + 512] = (unsigned short)l2;
            continue;
        }
        const unsigned hp = cvt_pk_bf16(v, 0.f);
        const long base = ((((long)cb * steps + S) * cot + i) * nl) * 512 + lane * 8 + e;
        packed[base] = (unsigned short)hp;
        if (nl == 1) continue;                               // plain bf16: one limb
        const float r = v - __uint_as_float(hp << 16);
        const unsigned mp = cvt_pk_bf16(r, 0.f);
        const float r2 = r - __uint_as_float(mp << 16);
        const unsigned lp = cvt_pk_bf16(r2, 0.f);
        packed[base + 512] = (unsigned short)mp;
        packed[base + 1024] = (unsigned short)lp;
    }
}

__global__ void pack_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed, int c_out, int c_in,
                               int taps, int cot, int co_blocks, int nfull, int tail_oct, int steps, int nch, int dgrad,
                               const float* __restrict__ w_amax, int nl) {
    pack_x3_items(w, packed, c_out, c_in, taps, cot, co_blocks, nfull, tail_oct, steps, nch, dgrad, w_amax, nl,
                  (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// Many weights in one launch (fsc_conv_pack_weights_multi: the 1-d model packs 80 small weights per step, 4.6 us each on their
// own): the jobs travel in the kernel arguments, job j owns the workgroups [block_start[j], block_start[j + 1]).
constexpr int kPackJobs = 48;
struct X3PackJob {
    const float* w;
    unsigned short* packed;
    int c_out, c_in, taps, cot, co_blocks, nfull, tail_oct, steps, dgrad, nl;
};
struct X3PackJobs {
    X3PackJob j[kPackJobs];
    int block_start[kPackJobs + 1];
    int n;
};

__global__ void pack_x3_multi_kernel(X3PackJobs jobs) {
    int lo = 0, hi = jobs.n - 1;
    while (lo < hi) {                                   // the last job whose first workgroup is <= this one
        const int mid = (lo + hi + 1) >> 1;
        if ((int)blockIdx.x >= jobs.block_start[mid]) lo = mid; else hi = mid - 1;
    }
    const X3PackJob j = jobs.j[lo];
    const int nb = jobs.block_start[lo + 1] - jobs.block_start[lo], b = (int)blockIdx.x - jobs.block_start[lo];
    pack_x3_items(j.w, j.packed, j.c_out, j.c_in, j.taps, j.cot, j.co_blocks, j.nfull, j.tail_oct, j.steps, 1, j.dgrad, nullptr, j.nl,
                  (long)b * blockDim.x + threadIdx.x, (long)nb * blockDim.x);
}

// -------------------------------------------------------------------------------------------
// Stem layer (c_in <= 4, 3x3; classifiers.py:526-531 with the 2-channel log-mel + frequency input): direct
// fp32 convolution on the vector ALUs.  With K = c_in * 9 = 18 the matrix tiles are 8-16x padded and
// the layer is bound by its 100-channel side (2.8 GB at cfg 2), so these kernels are organised around that
// tensor: a thread owns four consecutive pixels of a row and streams the wide tensor once with 16-byte
// accesses, the narrow tensor and the weights (LDS, broadcast reads) are cheap.
//   forward: out[co][4 px] = bias + sum_{ci,tap} W * in[ci][px + tap]      loop over co, 72 FMAs per store
//   dgrad:   dx[ci][4 px]  = sum_{co,tap} Wm * dout[co][px + tap]           loop over co, 9 loads per 72 FMAs
// `packed` is the [tap][k][m] layout of fsc_conv_pack_weights (dgrad: mirrored transpose).
constexpr int kStemThreads = 256;

template <int CIN>
__global__ __launch_bounds__(kStemThreads) void conv_stem_fwd_kernel(const float* __restrict__ in, const float* __restrict__ packed,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     int cout, int h, int w, int k_pad, int m_pad, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [cout][CIN*9 padded to 20 (CIN <= 2) or 36]
    constexpr int WROW = CIN <= 2 ? 20 : 36;
    for (int i = threadIdx.x; i < cout * CIN * 9; i += kStemThreads) {
        const int co = i / (CIN * 9), r = i - co * (CIN * 9);
        const int ci = r / 9, tap = r - ci * 9;
        smem[co * WROW + r] = packed[((long)tap * k_pad + ci) * m_pad + co];
    }
    __syncthreads();
    const int qpr = (w + 3) >> 2;                                        // pixel quads per row
    const int q = blockIdx.x * kStemThreads + threadIdx.x;
    if (q >= h * qpr) return;
    const int r = q / qpr, c0 = (q - r * qpr) * 4;
    const long hw = (long)h * w;
    const float* xin = in + (long)blockIdx.y * CIN * hw;
    float x[CIN][3][6];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const int rr = r + ty - 1;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int cc = c0 + k - 1;
                x[ci][ty][k] = (rr >= 0 && rr < h && cc >= 0 && cc < w) ? xin[ci * hw + (long)rr * w + cc] : 0.f;
            }
        }
    float* o = out + (long)blockIdx.y * cout * hw + (long)r * w + c0;
    const int nv = w - c0;                                               // valid pixels of the quad (>= 1)
    for (int co = 0; co < cout; ++co) {
        const float* wr = smem + co * WROW;
        const float b = bias ? bias[co] : 0.f;
        float a[4] = {b, b, b, b};
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const float wv = wr[ci * 9 + ty * 3 + tx];
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[k] = fmaf(wv, x[ci][ty][tx + k], a[k]);
                }
        float* oc = o + (long)co * hw;
        if (nv >= 4 && !accumulate) {
            *reinterpret_cast<f32x4*>(oc) = (f32x4){a[0], a[1], a[2], a[3]};
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nv) oc[k] = accumulate ? oc[k] + a[k] : a[k];
        }
    }
}

// forward fused with the 2x2 max-pool that follows the stem conv (classifiers.py:526-532): the full-resolution
// conv output (2.8 GB at cfg 2) is never written.  A thread owns four pooled pixels of a row = 2 x 8 conv
// outputs; tie rule and window index as fsc_maxpool_fwd (first maximum, NaN propagates).
template <int CIN>
__global__ __launch_bounds__(kStemThreads) void conv_stem_pool_fwd_kernel(const float* __restrict__ in, const float* __restrict__ packed,
                                                                          const float* __restrict__ bias, float* __restrict__ pooled,
                                                                          uint8_t* __restrict__ idx, int cout, int h, int w,
                                                                          int k_pad, int m_pad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WROW = 20;
    for (int i = threadIdx.x; i < cout * CIN * 9; i += kStemThreads) {
        const int co = i / (CIN * 9), r = i - co * (CIN * 9);
        const int ci = r / 9, tap = r - ci * 9;
        smem[co * WROW + r] = packed[((long)tap * k_pad + ci) * m_pad + co];
    }
    __syncthreads();
    const int oh = h >> 1, ow = w >> 1;
    const int qpr = (ow + 3) >> 2;
    const int q = blockIdx.x * kStemThreads + threadIdx.x;
    if (q >= oh * qpr) return;
    const int oy = q / qpr, ox0 = (q - oy * qpr) * 4;
    const long hw = (long)h * w, ohw = (long)oh * ow;
    const float* xin = in + (long)blockIdx.y * CIN * hw;
    float x[CIN][4][10];                                   // rows 2oy-1 .. 2oy+2, columns 2ox0-1 .. 2ox0+8
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int ty = 0; ty < 4; ++ty) {
            const int rr = 2 * oy + ty - 1;
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const int cc = 2 * ox0 + k - 1;
                x[ci][ty][k] = (rr >= 0 && rr < h && cc >= 0 && cc < w) ? xin[ci * hw + (long)rr * w + cc] : 0.f;
            }
        }
    const int nv = ow - ox0;                               // valid pooled pixels of the quad (>= 1)
    const long obase = (long)blockIdx.y * cout * ohw + (long)oy * ow + ox0;
    for (int co = 0; co < cout; ++co) {
        const float* wr = smem + co * WROW;
        const float b = bias ? bias[co] : 0.f;
        float a[2][8];
#pragma unroll
        for (int ry = 0; ry < 2; ++ry)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[ry][k] = b;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const float wv = wr[ci * 9 + ty * 3 + tx];
#pragma unroll
                    for (int ry = 0; ry < 2; ++ry)
#pragma unroll
                        for (int k = 0; k < 8; ++k) a[ry][k] = fmaf(wv, x[ci][ty + ry][tx + k], a[ry][k]);
                }
        float best[4];
        unsigned bidx = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float bv = a[0][2 * k];
            unsigned bi = 0;
            float v = a[0][2 * k + 1];
            if (v > bv || v != v) { bv = v; bi = 1; }
            v = a[1][2 * k];
            if ((v > bv || v != v) && bv == bv) { bv = v; bi = 2; }
            v = a[1][2 * k + 1];
            if ((v > bv || v != v) && bv == bv) { bv = v; bi = 3; }
            best[k] = bv;
            bidx |= bi << (8 * k);
        }
        float* po = pooled + obase + (long)co * ohw;
        uint8_t* pi = idx + obase + (long)co * ohw;
        if (nv >= 4) {
            *reinterpret_cast<f32x4*>(po) = (f32x4){best[0], best[1], best[2], best[3]};
            if ((reinterpret_cast<uintptr_t>(pi) & 3) == 0) {
                *reinterpret_cast<unsigned*>(pi) = bidx;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) pi[k] = (uint8_t)(bidx >> (8 * k));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nv) { po[k] = best[k]; pi[k] = (uint8_t)(bidx >> (8 * k)); }
        }
    }
}

// dgrad of the stem: COUT = channels written (the conv's c_in), kin = channels read (the conv's c_out)
template <int COUT>
__global__ __launch_bounds__(kStemThreads) void conv_stem_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ packed,
                                                                       float* __restrict__ dx, int kin, int h, int w, int k_pad,
                                                                       int m_pad, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [kin][9 taps][COUT] padded to a multiple of 4
    constexpr int WROW = (9 * COUT + 3) & ~3;
    for (int i = threadIdx.x; i < kin * 9 * COUT; i += kStemThreads) {
        const int k = i / (9 * COUT), r = i - k * (9 * COUT);
        const int tap = r / COUT, ci = r - tap * COUT;
        smem[k * WROW + r] = packed[((long)tap * k_pad + k) * m_pad + ci];
    }
    __syncthreads();
    // block = 8 rows x 32 quads so that the three input rows a thread reads are shared through L1
    const int qpr = (w + 3) >> 2;
    const int bq = blockIdx.x % ((qpr + 31) >> 5), br = blockIdx.x / ((qpr + 31) >> 5);
    const int r = br * 8 + (threadIdx.x >> 5), qc = bq * 32 + (threadIdx.x & 31);
    if (r >= h || qc >= qpr) return;
    const int c0 = qc * 4;
    const long hw = (long)h * w;
    const float* src = dout + (long)blockIdx.y * kin * hw;
    float a[COUT][4];
#pragma unroll
    for (int ci = 0; ci < COUT; ++ci)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[ci][k] = 0.f;
    const bool interior = c0 >= 1 && c0 + 5 <= w;                     // the six columns c0-1 .. c0+4 exist
    for (int k = 0; k < kin; ++k) {
        const float* wr = smem + k * WROW;
        const float* pk = src + (long)k * hw;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const int rr = r + ty - 1;
            float v[6];
            if (rr >= 0 && rr < h) {
                const float* row = pk + (long)rr * w + c0;
                if (interior) {
                    v[0] = row[-1];
                    const f32x4 m = *reinterpret_cast<const f32x4*>(row);   // 4-byte aligned 16-byte load
                    v[1] = m[0]; v[2] = m[1]; v[3] = m[2]; v[4] = m[3];
                    v[5] = row[4];
                } else {
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int cc = c0 + j - 1;
                        v[j] = (cc >= 0 && cc < w) ? row[j - 1] : 0.f;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 6; ++j) v[j] = 0.f;
            }
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int ci = 0; ci < COUT; ++ci) {
                    const float wv = wr[(ty * 3 + tx) * COUT + ci];
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[ci][j] = fmaf(wv, v[tx + j], a[ci][j]);
                }
        }
    }
    const int nv = w - c0;
#pragma unroll
    for (int ci = 0; ci < COUT; ++ci) {
        float* o = dx + ((long)blockIdx.y * COUT + ci) * hw + (long)r * w + c0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nv) o[k] = accumulate ? o[k] + a[ci][k] : a[ci][k];
    }
}

// -------------------------------------------------------------------------------------------
// weight gradient
// XCD-aware block order for grids of (blocks, splits): the hardware deals workgroups to the 8 XCDs round-robin in
// dispatch order (x fastest), so the `blocks` workgroups of one split -- which read the SAME pixels of both operands --
// would land on different XCDs and each fetch its operands from HBM into its own L2.  Re-index so that every XCD owns
// a contiguous range of the (split, block) sequence: the blocks of a split then run on one XCD, back to back, and the
// re-reads hit its L2.  Returns the logical (block, split) of this workgroup; a bijection for any grid size.
__device__ __forceinline__ void xcd_block_order(int* block, int* split) {
    const unsigned bx = gridDim.x, total = gridDim.x * gridDim.y;
    const unsigned id = blockIdx.x + bx * blockIdx.y;
    const unsigned xcd = id & 7u, slot = id >> 3;
    const unsigned chunk = total >> 3, rem = total & 7u;
    const unsigned v = xcd < rem ? xcd * (chunk + 1) + slot : rem * (chunk + 1) + (xcd - rem) * chunk + slot;
    *split = (int)(v / bx);
    *block = (int)(v - (unsigned)*split * bx);
}

template <int KH, int KW>
struct WgCfg {
    static constexpr int TAPS = KH * KW;
    static constexpr int WAVES = 4;                               // one wave per SIMD: two workgroups keep a CU balanced
    static constexpr int CIT = TAPS == 9 ? 2 : (TAPS == 3 ? 4 : 8);   // ci tiles (16 channels) per workgroup
    static constexpr int NT = CIT * TAPS;                         // (ci-tile, tap) pairs per workgroup
    static constexpr int NPW = (NT + WAVES - 1) / WAVES;          // pairs per wave (3x3: 5,5,4,4)
    static constexpr int PIXC = 64;                               // pixels per unit (K of the GEMM) = one wave
    static constexpr int MAXPOS64 = TAPS == 1 ? 1 : (TAPS == 3 ? 2 : 4);   // staged positions per lane (x 64)
};

struct WgGeom {
    int n, cin, cout, h, w;
    long hw;
    int nb, th, tw, tiles_n, tiles_h, tiles_w;
    int rows, cols, plane, npos, npix;
    int ci_pad, co_pad;
    int units;            // pixel boxes in total
    int nsplit;
    int ci_blocks;
};

template <int KH, int KW>
constexpr int wg_threads() { return WgCfg<KH, KW>::WAVES * 64; }

// One workgroup owns a (co block, ci block) pair and walks the pixel boxes split, split+nsplit, ...
// Per box ("unit" = up to 64 pixels) the dOut tile [CO_BLK][66] and the halo'd input tile
// [CI_BLK][plane] are copied by LDS-DMA (lane = pixel / staged position, so the per-lane source
// offset is computed once per unit; each DMA instruction is one channel row).  No VGPR staging:
// the kernel fits two workgroups (12 waves, 3 per SIMD) on a CU, which a 6-wave workgroup needs
// to keep all four SIMDs evenly loaded.  Row strides are == 2 (mod 32) so the 16-row x 2-pixel
// operand reads of an MFMA hit 32 distinct banks.
//
// PACKED (stem layers, c_in * taps <= 32): the GEMM's N axis enumerates (ci, tap) combinations
// directly (2 column tiles instead of 18 for c_in = 2), every wave owns all tiles, and the four waves
// split the unit's k-steps; each wave writes its own split-K slice (split * 4 + wave).
template <int KH, int KW, int MT, bool PACKED>
__global__ __launch_bounds__((wg_threads<KH, KW>())) void conv_wgrad_kernel(WgGeom g, const float* __restrict__ in,
                                                                           const float* __restrict__ dout,
                                                                           float* __restrict__ part) {
    using C = WgCfg<KH, KW>;
    constexpr int TAPS = C::TAPS, WAVES = C::WAVES, CIT = C::CIT, PIXC = C::PIXC;
    constexpr int NPW = PACKED ? 2 : C::NPW;
    constexpr int NT = PACKED ? 2 : C::NT;
    constexpr int MAXPOS64 = C::MAXPOS64;
    constexpr int CO_BLK = MT * 16, CI_BLK = CIT * 16;
    constexpr int DS = PIXC + 2;                                   // == 2 (mod 32)
    constexpr int NKS = PIXC / 4;
    constexpr int PADH = KH / 2, PADW = KW / 2;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dl = smem;                              // [CO_BLK][DS]
    float* il = smem + CO_BLK * DS;                // [CI_BLK][plane]
    constexpr int CI_LDS = PACKED ? 16 : CI_BLK;   // staged input channels (packed stem layers have c_in <= 16)
    int* ptab = reinterpret_cast<int*>(il + CI_LDS * g.plane);   // [PIXC] LDS offset of pixel p inside a staged channel

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kq = lane >> 4;
    int blk, split;
    xcd_block_order(&blk, &split);
    const int co0 = (blk / g.ci_blocks) * CO_BLK;
    const int ci0 = (blk % g.ci_blocks) * CI_BLK;

    // ---- unit-invariant decode: lane = pixel (dOut) and lane + 64 j = staged position (input)
    const int per_pix = g.th * g.tw, per_pos = g.rows * g.cols;
    int pb = -1, pr = 0, pc = 0;
    if (lane < g.npix) {
        pb = lane / per_pix;
        const int rem = lane - pb * per_pix;
        pr = rem / g.tw;
        pc = rem - pr * g.tw;
    }
    if (tid < PIXC) ptab[tid] = pb >= 0 ? (pb * g.rows + pr) * g.cols + pc : 0;
    int qb[MAXPOS64], qr[MAXPOS64], qc[MAXPOS64];
#pragma unroll
    for (int j = 0; j < MAXPOS64; ++j) {
        const int pos = lane + 64 * j;
        qb[j] = -1; qr[j] = 0; qc[j] = 0;
        if (pos < g.npos) {
            qb[j] = pos / per_pos;
            const int rem = pos - qb[j] * per_pos;
            qr[j] = rem / g.cols;
            qc[j] = rem - qr[j] * g.cols;
        }
    }

    f32x4 acc[NPW][MT];
#pragma unroll
    for (int s = 0; s < NPW; ++s)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[s][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Rows beyond c_out / c_in are never copied: zero them once.  The copy loops walk the live rows only with one
    // pointer increment per instruction (lanes outside the image keep pointing at a zero word with stride 0);
    // the earlier version recomputed a 64-bit row address per instruction: 6 scalar instructions per MFMA (PMC).
    constexpr int CI_STAGE_MAX = PACKED ? 16 : CI_BLK;
    const int co_live = min(CO_BLK, g.cout - co0);
    const int ci_live = PACKED ? g.cin : min(CI_BLK, g.cin - ci0);
    for (int i = (co_live > 0 ? co_live : 0) * DS + tid; i < CO_BLK * DS; i += WAVES * 64) dl[i] = 0.f;
    for (int i = (ci_live > 0 ? ci_live : 0) * g.plane + tid; i < CI_STAGE_MAX * g.plane; i += WAVES * 64) il[i] = 0.f;
    const char* zero = reinterpret_cast<const char*>(g_zero16);
    const long row_bytes = (long)WAVES * g.hw * (long)sizeof(float);       // this wave copies every WAVES-th row
    auto issue_unit = [&](int u) {
        int t = u;
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h; t /= g.tiles_h;
        const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
        {
            const bool live = pb >= 0 && n0 + pb < g.n && h0 + pr < g.h && w0 + pc < g.w;
            const char* src = live ? reinterpret_cast<const char*>(dout + (long)(n0 + pb) * g.cout * g.hw + (long)(h0 + pr) * g.w +
                                                                   (w0 + pc) + (long)(co0 + wid) * g.hw)
                                   : zero;
            const long step = live ? row_bytes : 0;
            float* dst = dl + wid * DS;
#pragma unroll 1
            for (int row = wid; row < co_live; row += WAVES) {
                glds4(reinterpret_cast<const float*>(src), dst);
                src += step;
                dst += WAVES * DS;
            }
        }
#pragma unroll
        for (int j = 0; j < MAXPOS64; ++j) {
            if (j * 64 < g.npos) {                       // uniform
                const int gh = h0 + qr[j] - PADH, gw = w0 + qc[j] - PADW;
                const bool live = qb[j] >= 0 && n0 + qb[j] < g.n && gh >= 0 && gh < g.h && gw >= 0 && gw < g.w;
                if (qb[j] >= 0) {                        // lanes past npos stay out of the DMA
                    const char* src = live ? reinterpret_cast<const char*>(in + (long)(n0 + qb[j]) * g.cin * g.hw + (long)gh * g.w + gw +
                                                                           (long)(ci0 + wid) * g.hw)
                                           : zero;
                    const long step = live ? row_bytes : 0;
                    float* dst = il + wid * g.plane + j * 64;
#pragma unroll 1
                    for (int cl = wid; cl < ci_live; cl += WAVES) {
                        glds4(reinterpret_cast<const float*>(src), dst);
                        src += step;
                        dst += WAVES * g.plane;
                    }
                }
            }
        }
    };

    // this wave's (ci tile, tap) pairs; pairs whose ci tile lies wholly beyond c_in are skipped
    int b_base[NPW];
    bool pair_live[NPW];
#pragma unroll
    for (int s = 0; s < NPW; ++s) {
        if (PACKED) {
            const int q = s * 16 + lm;                       // (ci, tap) combination of this lane
            const int ci = q / TAPS, tap = q - ci * TAPS;
            const bool ok = q < g.cin * TAPS;
            b_base[s] = ok ? ci * g.plane + (tap / KW) * g.cols + (tap % KW) : 0;
            pair_live[s] = s * 16 < g.cin * TAPS;
        } else {
            const int nt = wid * NPW + s;
            const int cit = nt / TAPS, tap = nt - cit * TAPS;
            b_base[s] = (cit * 16 + lm) * g.plane + (tap / KW) * g.cols + (tap % KW);
            pair_live[s] = nt < NT && ci0 + cit * 16 < g.cin;
            if (nt >= NT) b_base[s] = 0;
        }
    }
    const int a_base = lm * DS + kq;

    auto load_ab = [&](int ks, float (&a)[MT], float (&b)[NPW]) {
        const float* arow = dl + a_base + ks * 4;
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = arow[i * 16 * DS];
        const int pk = ptab[ks * 4 + kq];
#pragma unroll
        for (int s = 0; s < NPW; ++s) b[s] = il[b_base[s] + pk];
    };

    for (int u = split; u < g.units; u += g.nsplit) {
        __syncthreads();              // every wave is done reading the previous unit (and ptab is visible)
        issue_unit(u);
        __syncthreads();              // hipcc drains vmcnt here: the unit has landed in LDS
        if (PACKED) {
#pragma unroll 1
            for (int ks = wid; ks < NKS; ks += WAVES) {
                float a1[MT], b1[NPW];
                load_ab(ks, a1, b1);
#pragma unroll
                for (int s = 0; s < NPW; ++s)
                    if (pair_live[s]) {
#pragma unroll
                        for (int i = 0; i < MT; ++i)
                            acc[s][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], b1[s], acc[s][i], 0, 0, 0);
                    }
            }
            continue;
        }
        float a[2][MT], b[2][NPW];
        load_ab(0, a[0], b[0]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 1 < NKS) load_ab(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NPW; ++s) {
                if (pair_live[s]) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[s][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks & 1][i], b[ks & 1][s], acc[s][i], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (PACKED) {
        // the four waves hold partial sums over disjoint k-steps: tree-reduce them through LDS
        // (the dOut stage is free now) so the workgroup writes ONE split-K slice
        f32x4* red = reinterpret_cast<f32x4*>(smem);                 // [2 waves][NPW][MT][64 lanes]
        __syncthreads();
        if (wid >= 2) {
#pragma unroll
            for (int s = 0; s < NPW; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i) red[(((wid - 2) * NPW + s) * MT + i) * 64 + lane] = acc[s][i];
        }
        __syncthreads();
        if (wid < 2) {
#pragma unroll
            for (int s = 0; s < NPW; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[s][i] += red[((wid * NPW + s) * MT + i) * 64 + lane];
        }
        __syncthreads();
        if (wid == 1) {
#pragma unroll
            for (int s = 0; s < NPW; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i) red[(s * MT + i) * 64 + lane] = acc[s][i];
        }
        __syncthreads();
        if (wid != 0) return;
#pragma unroll
        for (int s = 0; s < NPW; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[s][i] += red[(s * MT + i) * 64 + lane];
    }

    // partial[split][tap][ci][co]; D row = co (kq*4 + r), column = ci (lm)
#pragma unroll
    for (int s = 0; s < NPW; ++s) {
        long row;
        if (PACKED) {
            const int q = s * 16 + lm;
            if (q >= g.cin * TAPS) continue;
            const int ci = q / TAPS, tap = q - ci * TAPS;
            row = ((long)split * TAPS + tap) * g.ci_pad + ci;
        } else {
            const int nt = wid * NPW + s;
            if (nt >= NT) continue;
            const int cit = nt / TAPS, tap = nt - cit * TAPS;
            row = ((long)split * TAPS + tap) * g.ci_pad + ci0 + cit * 16 + lm;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float4 v = make_float4(acc[s][i][0], acc[s][i][1], acc[s][i][2], acc[s][i][3]);
            *reinterpret_cast<float4*>(part + row * g.co_pad + co0 + i * 16 + kq * 4) = v;
        }
    }
}

// -------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix pipe with fp32 results (same three-limb arithmetic as
// conv_fwd_x3_kernel; here BOTH operands are activations and are split in registers).
//
//   dW[co][ci][tap] = sum over pixels  dOut[co][p] * In[ci][p + tap]        K = pixels
//
// One MFMA k-step covers 32 pixels: lane group kq owns a run of 8 consecutive pixels of one box row.
// Workgroup = 8 waves = ng co-groups x nt ci-tiles; a wave owns <= MT co tiles x one ci tile x all taps.
// Per k-step it splits MT dOut fragments (8 values each) and, per kernel row ty, ONE 10-pixel window
// of the input row r + ty: the three horizontal taps tx = -1, 0, +1 are the same bf16 values packed
// in two pairings (odd pairs serve tx = -1 and +1, even pairs tx = 0), so the split costs 67 VALU per
// three taps instead of 132.  Per k-step: 54 * MT MFMAs for 44 * MT + 201 VALU.
// The unit of work is a box of 64 pixels (th rows x tw columns, tw in {8,16,32,64}) staged by LDS-DMA:
// dOut [co][64] in run order and the halo'd input rows [ci][th + 2][tw + 8] with column 0 at a 16-byte
// aligned offset, so every operand read is an aligned ds_read_b128.  Two stages; partial sums
// [split][tap][ci][co] go to the same workspace and reduce kernel as the native wgrad.
constexpr int kWgxWaves = 8;
constexpr int kWgxDso = 68;        // dOut row stride in floats: == 4 (mod 64), 16 b128 lanes on 64 distinct banks

struct WgxGeom {
    int n, cin, cout, h, w;
    long hw;
    int th, tw, tiles_h, tiles_w;   // 64-pixel box and boxes per image
    int units, nsplit;
    int ng, nt;                     // co groups x ci tiles of a workgroup (ng * nt == 8)
    int tpb, tpg;                   // co tiles per block / per group (tpg <= MT)
    int co_blocks, ci_blocks, co_pad, ci_pad;
    int rowp, plane;                // input row pitch (tw + 8) and channel plane in floats
    int in_instr;                   // input DMA instructions per channel = ceil((th + 2) * rowp / 64)
};

// The `live` co tiles of a block are spread over its ng groups as evenly as possible; the groups that take one tile
// more are chosen so that the SIMD pairs (group q and q + ng/2 share the SIMDs) carry equal loads: 10 tiles on
// 4 groups -> 3, 2, 2, 3 (not 3, 3, 3, 1).  Returns the first tile of group q and its tile count.
__host__ __device__ inline void wgx_group(int live, int ng, int q, int* start, int* count) {
    const int base = live / ng, extra = live - base * ng;
    int st = 0, cnt = 0;
    for (int k = 0; k <= q; ++k) {
        const int rank = k < (ng + 1) / 2 ? 2 * k : 2 * (ng - 1 - k) + 1;     // order 0, ng-1, 1, ng-2, ...
        cnt = base + (rank < extra ? 1 : 0);
        if (k < q) st += cnt;
    }
    *start = st;
    *count = cnt;
}

template <int KH, int KW, int MT, int NPROD>
__global__ __launch_bounds__(kWgxWaves * 64) void conv_wgrad_x3_kernel(WgxGeom g, const float* __restrict__ in,
                                                                       const float* __restrict__ dout,
                                                                       float* __restrict__ part,
                                                                       const float* __restrict__ in_amax,
                                                                       const float* __restrict__ dout_amax) {
    constexpr int TAPS = KH * KW;
    constexpr int PADH = KH / 2;
    constexpr int DSO = kWgxDso;
    constexpr bool F16 = NPROD == 3;                         // two scaled fp16 limbs, 3 products (see conv_fwd_x3_kernel)
    constexpr int NL = NPROD == 1 ? 1 : F16 ? 2 : 3;         // NPROD = 1: one bf16 limb (plain bf16 arithmetic)
    constexpr int MAXI = 4;                                  // input DMA instructions per channel (<= 256 staged floats)

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int co_rows = g.ng * g.tpg * 16, ci_rows = g.nt * 16;
    const int stage_floats = co_rows * DSO + ci_rows * g.plane;
    // stage s: dOut rows at smem + s * stage_floats, input planes behind them

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, kq = lane >> 4;
    const int cog = wid / g.nt, cit = wid - cog * g.nt;                     // this wave's co group and ci tile
    int blk, split;
    xcd_block_order(&blk, &split);
    const int cb = blk / g.ci_blocks, ib = blk - cb * g.ci_blocks;
    const int tile0 = cb * g.tpb;                                          // first co tile of the block
    const int ci0 = ib * g.nt * 16;
    const int total_tiles = (g.cout + 15) >> 4;
    // live co tiles of this wave: i < mt_live
    int mt_live, g_start;
    {
        const int blk_live = tile0 + g.tpb < total_tiles ? g.tpb : total_tiles - tile0;      // live co tiles of this block
        wgx_group(blk_live > 0 ? blk_live : 0, g.ng, cog, &g_start, &mt_live);
        if (ci0 + cit * 16 >= g.cin) mt_live = 0;                          // ci tile wholly beyond c_in
    }

    // ---- unit-invariant DMA plans.  dOut: lane = pixel of the box in row-major (= run) order.
    const int pr = lane / g.tw, pc = lane - pr * g.tw;
    // input: lane + 64 j = offset inside the pitched rows; column = offset - 4
    int qr[MAXI], qc[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int o = lane + 64 * j;
        qr[j] = o / g.rowp;
        qc[j] = o - qr[j] * g.rowp - 4;
        if (qr[j] >= g.th + KH - 1 || qc[j] < -1 || qc[j] > g.tw) qr[j] = -1;      // outside the staged window
    }

    // fp16 limbs: both operands are scaled to [2^14, 2^15) by powers of two; the partial sums are unscaled on store
    float sa = 1.f, sb = 1.f, inv_a = 1.f, inv_b = 1.f;
    if constexpr (F16) {
        static_assert(kWgxWaves * 64 == fsc::kAmaxFloats, "one amax slot float per thread");
        const float ma = fsc::wave_max(dout_amax[tid]), mb = fsc::wave_max(in_amax[tid]);
        if (lane == 0) { smem[wid] = ma; smem[kWgxWaves + wid] = mb; }
        __syncthreads();
        float xa = smem[0], xb = smem[kWgxWaves];
#pragma unroll
        for (int i = 1; i < kWgxWaves; ++i) { xa = fmaxf(xa, smem[i]); xb = fmaxf(xb, smem[kWgxWaves + i]); }
        xa = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xa)));
        xb = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xb)));
        __syncthreads();
        const int fa = scale_field(xa), fb = scale_field(xb);
        sa = field_to_float(fa); inv_a = inv_scale(fa, xa);
        sb = field_to_float(fb); inv_b = inv_scale(fb, xb);
    }

    // Rows beyond c_out / c_in are never copied: zero them once in both stages.  The copy loops then walk the
    // live rows only with one pointer increment per instruction (lanes outside the image keep pointing at a
    // zero word with stride 0), which keeps the per-unit DMA issue at a few instructions per row.
    const int co_live = min(co_rows, g.cout - tile0 * 16), ci_live = min(ci_rows, g.cin - ci0);
    for (int st = 0; st < 2; ++st) {
        float* dl = smem + st * stage_floats;
        float* il = dl + co_rows * DSO;
        for (int i = (co_live > 0 ? co_live : 0) * DSO + tid; i < co_rows * DSO; i += kWgxWaves * 64) dl[i] = 0.f;
        for (int i = (ci_live > 0 ? ci_live : 0) * g.plane + tid; i < ci_rows * g.plane; i += kWgxWaves * 64) il[i] = 0.f;
    }
    const char* zero = reinterpret_cast<const char*>(g_zero16);
    const long row_bytes = 8 * g.hw * (long)sizeof(float);          // this wave copies every 8th row
    auto issue_unit = [&](int u, int stage) {
        float* dl = smem + stage * stage_floats;
        float* il = dl + co_rows * DSO;
        int t = u;
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h; t /= g.tiles_h;
        const int n0 = t, h0 = thi * g.th, w0 = twi * g.tw;
        {
            const bool live = h0 + pr < g.h && w0 + pc < g.w;
            const char* src = live ? reinterpret_cast<const char*>(dout + (long)n0 * g.cout * g.hw + (long)(h0 + pr) * g.w +
                                                                   (w0 + pc) + (long)(tile0 * 16 + wid) * g.hw)
                                   : zero;
            const long step = live ? row_bytes : 0;
            float* dst = dl + wid * DSO;
#pragma unroll 1
            for (int row = wid; row < co_live; row += kWgxWaves) {
                glds4(reinterpret_cast<const float*>(src), dst);
                src += step;
                dst += kWgxWaves * DSO;
            }
        }
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            if (j < g.in_instr) {                                           // uniform
                const int gh = h0 + qr[j] - PADH, gw = w0 + qc[j];
                const bool live = qr[j] >= 0 && gh >= 0 && gh < g.h && gw >= 0 && gw < g.w;
                if (lane + 64 * j < g.plane) {                              // lanes past the plane stay out of the DMA
                    const char* src = live ? reinterpret_cast<const char*>(in + (long)n0 * g.cin * g.hw + (long)gh * g.w + gw +
                                                                           (long)(ci0 + wid) * g.hw)
                                           : zero;
                    const long step = live ? row_bytes : 0;
                    float* dst = il + wid * g.plane + j * 64;
#pragma unroll 1
                    for (int cl = wid; cl < ci_live; cl += kWgxWaves) {
                        glds4(reinterpret_cast<const float*>(src), dst);
                        src += step;
                        dst += kWgxWaves * g.plane;
                    }
                }
            }
        }
    };

    f32x4 acc[KH][KW][MT];
#pragma unroll
    for (int ty = 0; ty < KH; ++ty)
#pragma unroll
        for (int tx = 0; tx < KW; ++tx)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[ty][tx][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane's runs in the two k-steps of a unit: run = 4 st + kq -> (row, first column)
    const int runs_shift = g.tw == 8 ? 0 : g.tw == 16 ? 1 : g.tw == 32 ? 2 : 3;      // log2(runs per box row)
    const int a_row = (g_start * 16 + lm) * DSO;
    const int b_row = (cit * 16 + lm) * g.plane;

    constexpr int kPairA[9] = {1, 0, 2, 0, 1, 0, 2, 1, 2};      // limb pairs, 0 = h, 1 = m, 2 = l; x6 uses the first six
    constexpr int kPairB[9] = {1, 2, 0, 1, 0, 0, 2, 2, 1};
    constexpr int kPairA2[3] = {1, 0, 0}, kPairB2[3] = {0, 1, 0};   // fp16: 0 = h, 1 = l

    int stage = 0;
    if (split < g.units) issue_unit(split, 0);
    for (int u = split; u < g.units; u += g.nsplit) {
        __syncthreads();                      // unit u has landed (vmcnt drained); everyone is done with the other stage
        if (u + g.nsplit < g.units) issue_unit(u + g.nsplit, stage ^ 1);
        const float* dl = smem + stage * stage_floats;
        const float* il = dl + co_rows * DSO;
        // Every wave runs all of its MT co tiles: tiles beyond c_out read zero-filled dOut rows, tiles beyond
        // the block are recomputed and not stored.  (A second, predicated code path doubles the accumulator
        // registers in hipcc's allocation and spills.)
        if (mt_live > 0) {
#pragma unroll 1
            for (int st = 0; st < 2; ++st) {
                const int run = 4 * st + kq;
                const int r = run >> runs_shift, c0 = (run - (r << runs_shift)) * 8;
                const int a_off = a_row + run * 8;
                const int b_off = b_row + r * g.rowp + c0;                 // floats c0-4 .. c0+11 of row r + ty
                // ---- issue every LDS read of the k-step first: the first input row's window, then the dOut
                //      fragments of all co tiles (8 values each); the splits below consume them as they arrive
                const float* brow = il + b_off;
                f32x4 w1 = *reinterpret_cast<const f32x4*>(brow + 4), w2 = *reinterpret_cast<const f32x4*>(brow + 8);
                float wl = brow[3], wr = brow[12];                          // pixels c0-1 and c0+8
                f32x4 ar[MT][2];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const f32x4* src = reinterpret_cast<const f32x4*>(dl + a_off + i * 16 * DSO);
                    ar[i][0] = src[0];
                    ar[i][1] = src[1];
                }
                __builtin_amdgcn_sched_barrier(0);
                u32x4 al[MT][NL];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    if (i < mt_live) {
                        const float x[8] = {ar[i][0][0], ar[i][0][1], ar[i][0][2], ar[i][0][3],
                                            ar[i][1][0], ar[i][1][1], ar[i][1][2], ar[i][1][3]};
                        if constexpr (F16) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                unsigned hp, lp;
                                split2_pair(x[2 * q], x[2 * q + 1], sa, hp, lp);
                                al[i][0][q] = hp;
                                al[i][1][q] = lp;
                            }
                        } else if constexpr (NL == 1) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) al[i][0][q] = cvt_pk_bf16(x[2 * q], x[2 * q + 1]);
                        } else {
                            split3(x, al[i][0], al[i][1], al[i][NL - 1]);
                        }
                    }
                }
#pragma unroll
                for (int ty = 0; ty < KH; ++ty) {
                    // ---- B: the 10-pixel window c0-1 .. c0+8 of input row r + ty, split once for the three tx
                    const float e[10] = {wl, w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3], wr};   // e[k] = pixel c0-1+k
                    // B operands as ready register quads: bq[tx][limb]; with e[k] packed in pairs, tx = 0 takes the
                    // odd pairing P_0..P_3 = (e0,e1)..(e6,e7), tx = 2 its shift P_1..P_4, tx = 1 the even pairing
                    // Q_0..Q_3 = (e1,e2)..(e7,e8)
                    u32x4 bq[3][NL];
                    float res[10];
#pragma unroll
                    for (int k = 0; k < 10; ++k) res[k] = e[k];
                    if constexpr (F16) {
                        // five pair splits (four instructions each) give P; Q_k = (high half of P_k, low half of P_k+1)
                        unsigned P[2][5];
#pragma unroll
                        for (int k = 0; k < 5; ++k) split2_pair(res[2 * k], res[2 * k + 1], sb, P[0][k], P[1][k]);
#pragma unroll
                        for (int lv = 0; lv < 2; ++lv) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) bq[1][lv][k] = __builtin_amdgcn_alignbit(P[lv][k + 1], P[lv][k], 16);
                            bq[0][lv] = (u32x4){P[lv][0], P[lv][1], P[lv][2], P[lv][3]};
                            bq[2][lv] = (u32x4){P[lv][1], P[lv][2], P[lv][3], P[lv][4]};
                        }
                    }
#pragma unroll
                    for (int lv = 0; lv < (F16 ? 0 : NL); ++lv) {
                        unsigned P[5];
#pragma unroll
                        for (int k = 0; k < 5; ++k) P[k] = cvt_pk_bf16(res[2 * k], res[2 * k + 1]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) bq[1][lv][k] = cvt_pk_bf16(res[2 * k + 1], res[2 * k + 2]);
                        bq[0][lv] = (u32x4){P[0], P[1], P[2], P[3]};
                        bq[2][lv] = (u32x4){P[1], P[2], P[3], P[4]};
                        if (lv + 1 < NL) {
#pragma unroll
                            for (int k = 0; k < 5; ++k) {
                                res[2 * k] = fsub(res[2 * k], __uint_as_float(P[k] << 16));
                                res[2 * k + 1] = fsub(res[2 * k + 1], __uint_as_float(P[k] & 0xffff0000u));
                            }
                        }
                    }
                    // pin the operands here: otherwise the quads are re-assembled in front of every MFMA group
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int lv = 0; lv < NL; ++lv) asm volatile("" : "+v"(bq[tx][lv]));
                    // the next row's window is fetched while this row's MFMAs run (its registers are free now)
                    if (ty + 1 < KH) {
                        const float* nrow = brow + (ty + 1) * g.rowp;
                        w1 = *reinterpret_cast<const f32x4*>(nrow + 4);
                        w2 = *reinterpret_cast<const f32x4*>(nrow + 8);
                        wl = nrow[3];
                        wr = nrow[12];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // channel tile outermost, under ONE uniform branch per tile: a wave whose group holds fewer than MT
                    // live tiles skips the dead ones (consecutive MFMAs still alternate over the three tx accumulators)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        if (i < mt_live) {
#pragma unroll
                            for (int gq = 0; gq < NPROD; ++gq) {
                                const int pa = NL == 1 ? 0 : F16 ? kPairA2[gq % 3] : NPROD == 9 ? kPairA[(gq + 6) % 9] : kPairA[gq % 9];
                                const int pb = NL == 1 ? 0 : F16 ? kPairB2[gq % 3] : NPROD == 9 ? kPairB[(gq + 6) % 9] : kPairB[gq % 9];
#pragma unroll
                                for (int tx = 0; tx < KW; ++tx)
                                    acc[ty][tx][i] = mfma_k32<F16>(al[i][pa], bq[KW == 1 ? 1 : tx][pb], acc[ty][tx][i]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);        // keep the next row's loads and split behind these MFMAs
                }
            }
        }
        stage ^= 1;
    }

    // partial[split][tap][ci][co]; D row = co (kq*4 + r), column = ci (lm)
#pragma unroll
    for (int ty = 0; ty < KH; ++ty)
#pragma unroll
        for (int tx = 0; tx < KW; ++tx) {
            const int tap = ty * KW + tx;
            const long row = ((long)split * TAPS + tap) * g.ci_pad + ci0 + cit * 16 + lm;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (i < g.tpg) {
                    const int tile = tile0 + g_start + i;
                    if (i < mt_live) {          // (tiles beyond c_out and ci rows beyond c_in are never read by the reduce)
                        f32x4 a = acc[ty][tx][i];
                        if constexpr (F16) a = a * inv_a * inv_b;
                        *reinterpret_cast<float4*>(part + row * g.co_pad + tile * 16 + kq * 4) = make_float4(a[0], a[1], a[2], a[3]);
                    }
                }
            }
        }
}

// blockIdx.y = (tap, ci) row of the partial slices; threadIdx.x runs along co (coalesced reads), threadIdx.y over four interleaved
// groups of split-K slices (as l16_wgrad_reduce_kernel: one thread walking every slice was a chain of dependent loads).
constexpr int kRedX = 64, kRedY = 4;
__global__ __launch_bounds__(kRedX * kRedY) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int c_out,
                                                                      int c_in, int taps, int ci_pad, int co_pad, int nsplit) {
    __shared__ float red[kRedY][kRedX];
    const int co = blockIdx.x * kRedX + threadIdx.x;
    const int tap = blockIdx.y / c_in, ci = blockIdx.y - tap * c_in;
    const long slice = (long)taps * ci_pad * co_pad;
    float s0 = 0.f, s1 = 0.f;
    if (co < c_out) {
        const float* p = part + ((long)tap * ci_pad + ci) * co_pad + co;
        int sp = threadIdx.y;
        float s2 = 0.f, s3 = 0.f;                 // (four loads in flight: 64 slices are 16 per thread)
        for (; sp + 3 * kRedY < nsplit; sp += 4 * kRedY) {
            s0 += p[(long)sp * slice];
            s1 += p[(long)(sp + kRedY) * slice];
            s2 += p[(long)(sp + 2 * kRedY) * slice];
            s3 += p[(long)(sp + 3 * kRedY) * slice];
        }
        for (; sp < nsplit; sp += kRedY) s0 += p[(long)sp * slice];
        s0 += s2;
        s1 += s3;
    }
    red[threadIdx.y][threadIdx.x] = s0 + s1;
    __syncthreads();
    if (threadIdx.y == 0 && co < c_out)
        dw[((long)co * c_in + ci) * taps + tap] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// The reduces of several weight gradients in one launch (fsc_conv_wgrad_reduce_multi: the four convolutions of a block of the 1-d
// model reduced their slices in four launches of ~8 us): job j owns the workgroups [block_start[j], block_start[j + 1]).
constexpr int kRedJobs = 16;
struct RedJob {
    const float* part;
    float* dw;
    int c_out, c_in, taps, ci_pad, co_pad, nsplit, bx;      // bx = workgroups along co
};
struct RedJobs {
    RedJob j[kRedJobs];
    int block_start[kRedJobs + 1];
    int n;
};
__global__ __launch_bounds__(kRedX * kRedY) void wgrad_reduce_multi_kernel(RedJobs jobs) {
    __shared__ float red[kRedY][kRedX];
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.block_start[ji + 1]) ++ji;
    const RedJob q = jobs.j[ji];
    const int b = (int)blockIdx.x - jobs.block_start[ji];
    const int row = b / q.bx, co = (b - row * q.bx) * kRedX + threadIdx.x;       // row = (tap, ci)
    const int tap = row / q.c_in, ci = row - tap * q.c_in;
    const long slice = (long)q.taps * q.ci_pad * q.co_pad;
    float s0 = 0.f, s1 = 0.f;
    if (co < q.c_out) {
        const float* p = q.part + ((long)tap * q.ci_pad + ci) * q.co_pad + co;
        int sp = threadIdx.y;
        float s2 = 0.f, s3 = 0.f;
        for (; sp + 3 * kRedY < q.nsplit; sp += 4 * kRedY) {
            s0 += p[(long)sp * slice];
            s1 += p[(long)(sp + kRedY) * slice];
            s2 += p[(long)(sp + 2 * kRedY) * slice];
            s3 += p[(long)(sp + 3 * kRedY) * slice];
        }
        for (; sp < q.nsplit; sp += kRedY) s0 += p[(long)sp * slice];
        s0 += s2;
        s1 += s3;
    }
    red[threadIdx.y][threadIdx.x] = s0 + s1;
    __syncthreads();
    if (threadIdx.y == 0 && co < q.c_out)
        q.dw[((long)co * q.c_in + ci) * q.taps + tap] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// -------------------------------------------------------------------------------------------
// host-side planning
struct FwdPlan {
    Geom g;
    int cot;          // channel tiles per workgroup
    int co_blocks;
    int kc;
    size_t lds_bytes;
    long grid_x;
    int x3;           // 0: native fp32 MFMA kernel; 6 / 9: conv_fwd_x3_kernel with that many limb products
    int pt;           // pixel tiles per wave
    long launch_x;    // x3: persistent workgroups per (channel block, K slice)
    int s1d;          // 1: the small 1-d bf16 kernels of conv_s1d.hip (x3 = 1: same packed fragments, cot / co_blocks / x_* describe them)
    fsc::s1d::Plan sp;
};

// floats of the packed limb fragments of an x3 plan (fp16 limbs: followed by 4 floats holding the weights' amax)
int limbs_of(int nprod) { return nprod == 1 ? 1 : nprod == 3 ? 2 : 3; }
size_t x3_limb_floats(const FwdPlan& p) {
    return (size_t)p.co_blocks * p.g.x_steps * p.cot * limbs_of(p.x3) * 256;
}

// Arithmetic of the conv kernels: 0 = native fp32 MFMA everywhere; 3 = scaled split-fp16 (two limbs, three
// products: 22-bit products, the opt-in FAST mode), 6 / 9 = split-bf16 (three limbs, that many products) wherever the x3
// tilings fit, 10 = three scaled fp16 limbs, six products (products to 2^-32: the reference's fp32 nn.Conv2d precision,
// classifiers.py:526-531, 77-81 -- THE DEFAULT since round 6: what bench.py measures is what the library ships).
// A PER-CALL property: fsc_conv_desc.arith; FSC_ARITH_DEFAULT (-1) there means the process default, which is
// read once from the environment (FSC_CONV_ARITH=f32|f16x3|bf16x6|bf16x9|f16x6, else 10) and never changes afterwards.
int default_arith() {
    static const int mode = [] {
        const char* e = getenv("FSC_CONV_ARITH");
        if (e && !strcmp(e, "f32")) return 0;
        if (e && !strcmp(e, "bf16")) return 1;
        if (e && !strcmp(e, "bf16x6")) return 6;
        if (e && !strcmp(e, "bf16x9")) return 9;
        if (e && !strcmp(e, "f16x3")) return 3;
        if (e && *e && strcmp(e, "f16x6")) fprintf(stderr, "libfsc_hip: unknown FSC_CONV_ARITH=%s, using f16x6\n", e);
        return 10;
    }();
    return mode;
}
// (10 -- three SCALED fp16 limbs, six products -- exists on pre-split L16 tensors only: the fp32-input kernels of this file, which
// split beside the MFMAs, serve it with the exact nine-product bf16 split)
int arith_of(const fsc_conv_desc& d) {
    const int a = d.arith < 0 ? default_arith() : d.arith;
    return a == 10 ? 9 : a;
}

int pad_plane(int floats, int want) {   // smallest p >= floats with p % 32 == want
    int p = floats;
    while (p % 32 != want) ++p;
    return p;
}

// Narrow boxes cost global-memory efficiency (row segments shorter than a 64/128-byte line), so
// the box search trades a few percent of MFMA columns for wider rows.  Returned in 1/100.
long box_penalty(int tw, int w) {
    if (tw >= 32 || tw == w) return 100;
    if (tw >= 16) return 102;
    if (tw >= 8) return 108;
    return 125;
}

// Few channel tiles per workgroup mean little operand reuse per LDS read; weigh padded tiles
// by this factor (1/100) when splitting the output channels into blocks.
long tile_penalty(int tiles_per_block) {
    switch (tiles_per_block) {
        case 1: return 140;
        case 2: return 120;
        case 3: return 110;
        case 4: return 105;
        case 5: return 102;
        default: return 100;
    }
}


// tiling of conv_fwd_x3_kernel with PT pixel tiles per wave; false when the shape does not suit it
bool plan_fwd_x3_pt(const fsc_conv_desc& d_in, int dgrad, int nprod, int pt, int max_cot, FwdPlan* out) {
    FwdPlan p{};
    Geom& g = p.g;
    fsc_conv_desc d = d_in;
    const int taps = d.kh * d.kw;
    if (taps == 1) {                 // no halo: every (n, c) plane is one row of h*w pixels
        d.w = d.h * d.w;
        d.h = 1;
    }
    const int nch = 1;
    const int kch = kXChunk;
    const int nstg = taps == 1 ? (pt == 2 ? 3 : 4) : 2;  // input stages (kernel NSTG)
    g.n = d.n; g.h = d.h; g.w = d.w; g.hw = (long)d.h * d.w;
    g.cin = dgrad ? d.c_out : d.c_in;
    g.cout = dgrad ? d.c_in : d.c_out;
    if (g.cin < 32 || g.cout < 48) return false;         // stem layers: HBM-bound, K or M too small for 16x16x32 tiles
    if (g.hw >= (1L << 26)) return false;                // the 1x1 input copies step their pointers by 8 planes in 32 bits
    // channel tiles per workgroup: a step costs its MFMAs (proportional to the tiles) plus the side work of
    // splitting the activations, about 3.5 tiles' worth (measured); more than 8 tiles use a two-slot weight ring
    const int tiles = fsc::ceil_div(g.cout, 16);
    int best_cot = 1, best_blocks = tiles;
    long best_tile_cost = -1;
    for (int cot = 1; cot <= max_cot; ++cot) {
        const int blocks = fsc::ceil_div(tiles, cot);
        const long cost = (long)blocks * (2 * cot + 7);
        if (best_tile_cost < 0 || cost < best_tile_cost || (cost == best_tile_cost && blocks < best_blocks)) {
            best_cot = cot; best_blocks = blocks; best_tile_cost = cost;
        }
    }
    p.cot = best_cot;
    p.co_blocks = best_blocks;
    p.pt = pt;
    p.x3 = nprod;
    g.m_pad = best_blocks * best_cot * 16;
    g.flat = 0;
    const int pix_cap = kXWaves * p.pt * 16;
    const size_t lds_total = 160 * 1024;
    const size_t ring = (size_t)((p.cot > 8 || (taps == 1 && pt == 2)) ? 2 : 3) * p.cot * limbs_of(nprod) * 1024;
    const size_t scratch = (size_t)kXWaves * 16 * 20 * sizeof(float);      // per-wave epilogue transpose tiles
    int cap_pos = (int)((lds_total - ring - scratch) / (nstg * kch * sizeof(float))) - 4;
    if (cap_pos > 64 * kXNptMax - 4) cap_pos = 64 * kXNptMax - 4;
    long best_cost = -1;
    int bnb = 1, bth = 1, btw = 1;
    // box widths are multiples of 4: the epilogue stores quads of consecutive pixels (the box may overhang
    // the image's right edge; several images share a box only when whole images of a multiple-of-4 width fit)
    for (int tw = 4; tw <= ((d.w + 3) & ~3) && tw <= pix_cap; tw += 4) {
        int th = pix_cap / tw;
        if (th > d.h) th = d.h;
        int nb = 1;
        if (th == d.h && tw >= d.w) {
            nb = pix_cap / (th * tw);
            if (nb > d.n) nb = d.n;
            if (nb < 1) nb = 1;
        }
        while (nb > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) --nb;
        while (th > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) --th;
        if (nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) continue;
        const long nt = (long)fsc::ceil_div(d.n, nb) * fsc::ceil_div(d.h, th) * fsc::ceil_div(d.w, tw);
        const long cost = nt * box_penalty(tw, d.w);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && tw > btw)) {
            best_cost = cost; bnb = nb; bth = th; btw = tw;
        }
    }
    if (best_cost < 0) return false;
    g.nb = bnb; g.th = bth; g.tw = btw;
    g.rows = bth + d.kh - 1; g.cols = btw + d.kw - 1;
    g.npix = bnb * bth * btw; g.npos = bnb * g.rows * g.cols;
    g.tiles_n = fsc::ceil_div(d.n, bnb); g.tiles_h = fsc::ceil_div(d.h, bth); g.tiles_w = fsc::ceil_div(d.w, btw);
    p.grid_x = (long)g.tiles_n * g.tiles_h * g.tiles_w;
    // fewer than ~70 % live MFMA columns: smaller boxes (PT = 1, then the native kernel) fit such shapes better
    if ((double)d.n * g.hw < (pt == 2 ? 0.7 : 0.55) * (double)p.grid_x * pix_cap) return false;
    g.plane = g.npos;
    while (g.plane % 4 != 2) ++g.plane;              // 8 * plane == 16 (mod 32): lane groups kq, kq+1 on disjoint banks
    g.x_npt = fsc::ceil_div(g.plane, 64);
    if (g.x_npt > kXNptMax) return false;
    const int rem = g.cin % kch;
    g.x_nfull = g.cin / kch + (rem > kch - 8 ? 1 : 0);
    g.x_tail_oct = (rem > 0 && rem <= kch - 8) ? fsc::ceil_div(rem, 8) : 0;
    g.x_tail_steps = fsc::ceil_div(taps * g.x_tail_oct, 4);
    // A remainder chunk of ONE step (k3 rows with c_in mod 32 in 1 .. 8) opens and closes in the same step: the box of the chunk
    // behind it -- the next item's first -- would be issued in the very step whose MFMA phases already read it (found by the
    // batch-128 layer tests of round 5: every item of a worker after its second came out wrong on 129 -> 64 @ 3446 frames).
    // One more (zero) octet makes it two steps; 1x1 kernels stage NSTG - 1 >= 2 chunks ahead and are not affected.
    if (taps > 1 && g.x_tail_oct > 0 && g.x_tail_steps < 2) {
        g.x_tail_oct = 2;
        g.x_tail_steps = fsc::ceil_div(taps * g.x_tail_oct, 4);
    }
    g.x_steps = g.x_nfull * taps * nch + g.x_tail_steps;
    g.k_pad = (g.x_nfull + (g.x_tail_oct ? 1 : 0)) * kch;
    p.kc = kch;
    g.ksplit = 1;
    {
        const long wgs = p.grid_x * p.co_blocks;
        const int nchunks = g.x_nfull + (g.x_tail_oct ? 1 : 0);
        if (wgs < 200 && !fsc::env().dbg_noksplit) {
            long ks = (384 + wgs - 1) / wgs;
            if (ks > 8) ks = 8;
            if (ks > nchunks / 2) ks = nchunks / 2;
            // a K slice pays a scalar-atomic epilogue (and the output is cleared first): worth it only with enough
            // MFMA steps behind every output element -- the short-K layers of the 1-d model ran 2-4x slower split
            // (cfg 3: 6746 -> 7654 clips/s; thresholds of 16 - 64 steps per slice measure the same)
            if (ks > g.x_steps / 16) ks = g.x_steps / 16;
            if (ks > 1) g.ksplit = (int)ks;
        }
    }
    p.lds_bytes = ring + (size_t)nstg * kch * g.plane * sizeof(float) + scratch;
    if (p.lds_bytes > lds_total) return false;
    {
        // every K slice needs two steps (the weight ring runs two steps ahead, across tiles as well)
        const int nchunks = g.x_nfull + (g.x_tail_oct ? 1 : 0);
        for (int z = 0; z < g.ksplit; ++z) {
            const int c_lo = (int)((long)nchunks * z / g.ksplit), c_hi = (int)((long)nchunks * (z + 1) / g.ksplit);
            const int s_lo = c_lo * taps * nch, s_hi = c_hi == nchunks ? g.x_steps : c_hi * taps * nch;
            if (s_hi - s_lo < 2) return false;
        }
    }
    // persistent workgroups: one per CU, each walks the (pixel tile, channel block) items w, w + workers, ...
    g.x_coblk = p.co_blocks;
    long workers = 256 / g.ksplit;
    if (workers < 1) workers = 1;
    const long items = p.grid_x * p.co_blocks;
    if (items <= 2 * workers) workers = items;        // small layers: one item per workgroup, the GPU balances them
    p.launch_x = workers;
    *out = p;
    return true;
}

// 256-pixel tiles (two pixel tiles per wave) where they fit, else 128-pixel tiles (1x1 convolutions stage
// 64-channel chunks and always take the small tile)
bool plan_fwd_x3(const fsc_conv_desc& d, int dgrad, int nprod, FwdPlan* out) {
    // (9-10 tiles with a two-slot ring compile, but at 256 registers they spill and gain nothing: measured)
    // 150- and 225-channel layers (10 / 15 tiles): ONE wide block on 128-pixel tiles instead of two 5- or 8-tile blocks
    // on 256-pixel tiles -- the same MFMAs per wave-step with half the activation splitting, and the input is staged
    // once (+5 ... +18 % measured; 16-tile blocks for the 506 / 759-channel layers gained nothing)
    {
        const int tiles = fsc::ceil_div(dgrad ? d.c_in : d.c_out, 16);
        // (two fp16 limbs leave registers for 10 tiles x 2 pixel tiles: half the weight-fragment reads per MFMA,
        //  +13 % measured; 15 x 2 spills and runs at half speed)
        if (d.kh * d.kw > 1 && tiles == 10 && nprod == 3 &&
            plan_fwd_x3_pt(d, dgrad, nprod, 2, tiles, out) && out->cot == tiles)
            return true;
        if (d.kh * d.kw > 1 && (tiles == 10 || tiles == 15) && plan_fwd_x3_pt(d, dgrad, nprod, 1, tiles, out) && out->cot == tiles)
            return true;

    }
    if (plan_fwd_x3_pt(d, dgrad, nprod, 2, 8, out)) {
        // 1x1 layers with few work items keep the 128-pixel tile (256-pixel tiles halve an already short grid)
        if (d.kh * d.kw > 1 || out->grid_x * out->co_blocks >= 512) return true;
    }
    return plan_fwd_x3_pt(d, dgrad, nprod, 1, 8, out);
}

bool plan_fwd_f32(const fsc_conv_desc& d, int dgrad, FwdPlan* out);

bool stem_like(const fsc_conv_desc& d) { return d.c_in <= 4; }

bool plan_fwd(const fsc_conv_desc& d, int dgrad, FwdPlan* out) {
    const int arith = arith_of(d);
    if (arith == 1 || arith == 9) {                      // few positions (the late blocks of the 1-d model; cfg 2's 2 x 6-pixel block): conv_s1d.hip
        fsc_conv_desc dd = d;
        dd.arith = arith;
        fsc::s1d::Plan sp;
        if (!stem_like(d) && fsc::s1d::plan_fwd(dd, dgrad, &sp)) {
            FwdPlan p{};
            p.x3 = arith; p.s1d = 1; p.sp = sp;
            p.cot = sp.cot; p.co_blocks = sp.co_blocks; p.pt = fsc::s1d::kPt;
            p.g.n = d.n; p.g.h = d.h; p.g.w = d.w; p.g.hw = (long)d.h * d.w; p.g.cin = sp.cin; p.g.cout = sp.cout;
            p.g.x_nfull = sp.nfull; p.g.x_tail_oct = sp.tail_oct; p.g.x_steps = sp.steps; p.g.ksplit = 1;
            p.grid_x = sp.px_groups; p.launch_x = sp.px_groups;
            *out = p;
            return true;
        }
    }
    if (arith && plan_fwd_x3(d, dgrad, arith, out)) return true;
    return plan_fwd_f32(d, dgrad, out);
}

bool plan_fwd_f32(const fsc_conv_desc& d, int dgrad, FwdPlan* out) {
    FwdPlan p{};
    Geom& g = p.g;
    const int taps = d.kh * d.kw;
    g.n = d.n; g.h = d.h; g.w = d.w; g.hw = (long)d.h * d.w;
    g.cin = dgrad ? d.c_out : d.c_in;
    g.cout = dgrad ? d.c_in : d.c_out;
    const int nxi_max = fwd_nxi_max(taps);           // input DMA instructions per wave (kernel NXI_MAX)
    const int kPixCap = 4 * fwd_pt(taps) * 16;       // pixels per workgroup
    // channel tiling: minimise padded tiles, prefer fewer blocks
    const int tiles = fsc::ceil_div(g.cout, 16);
    int best_cot = 1, best_blocks = tiles;
    long best_tile_cost = (long)tiles * tile_penalty(1);
    for (int cot = 2; cot <= 8; ++cot) {
        const int blocks = fsc::ceil_div(tiles, cot);
        const long cost = (long)blocks * cot * tile_penalty(cot);
        if (cost < best_tile_cost || (cost == best_tile_cost && blocks < best_blocks)) {
            best_cot = cot; best_blocks = blocks; best_tile_cost = cost;
        }
    }
    p.cot = best_cot;
    p.co_blocks = best_blocks;
    p.pt = fwd_pt(taps);
    int kc = fwd_kc(taps);                 // box search below assumes this chunk; may shrink to 4 afterwards
    g.m_pad = best_blocks * best_cot * 16;
    if (taps == 1) {
        g.flat = 1;
        g.flat_total = (long)d.n * g.hw;
        g.nb = 1; g.th = 1; g.tw = kPixCap; g.rows = 1; g.cols = kPixCap;
        g.npix = kPixCap; g.npos = kPixCap;
        g.tiles_n = 1; g.tiles_h = 1; g.tiles_w = 1;
        p.grid_x = (g.flat_total + kPixCap - 1) / kPixCap;
        g.tiles_w = (int)p.grid_x;      // so the box decode in the kernel stays in range
    } else {
        g.flat = 0;
        const int cap_pos = 256 * nxi_max / kc - 32;      // staged positions incl. plane padding
        long best_cost = -1;
        int bnb = 1, bth = 1, btw = 1;
        for (int tw = 1; tw <= d.w && tw <= kPixCap; ++tw) {
            int th = kPixCap / tw;
            if (th > d.h) th = d.h;
            int nb = 1;
            if (th == d.h && tw == d.w) {            // whole image fits: pack several images per box
                nb = kPixCap / (th * tw);
                if (nb > d.n) nb = d.n;
                if (nb < 1) nb = 1;
            }
            while (nb > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) --nb;
            while (th > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) --th;
            if (nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) continue;
            const long tiles = (long)fsc::ceil_div(d.n, nb) * fsc::ceil_div(d.h, th) * fsc::ceil_div(d.w, tw);
            const long cost = tiles * box_penalty(tw, d.w);
            if (best_cost < 0 || cost < best_cost || (cost == best_cost && tw > btw)) {
                best_cost = cost; bnb = nb; bth = th; btw = tw;
            }
        }
        if (best_cost < 0) return false;
        g.nb = bnb; g.th = bth; g.tw = btw;
        g.rows = bth + d.kh - 1; g.cols = btw + d.kw - 1;
        g.npix = bnb * bth * btw; g.npos = bnb * g.rows * g.cols;
        g.tiles_n = fsc::ceil_div(d.n, bnb); g.tiles_h = fsc::ceil_div(d.h, bth); g.tiles_w = fsc::ceil_div(d.w, btw);
        p.grid_x = (long)g.tiles_n * g.tiles_h * g.tiles_w;
    }
    g.plane = pad_plane(g.npos, 16);
    // 3x3 K chunk (measured, MI355X): narrow channel blocks (COT <= 6) run faster with 4-channel chunks
    // (19-29 KB of LDS per workgroup, 3-4 workgroups per CU: +3-5 %), COT = 7 with 8 (2 x 39 KB, two
    // workgroups), COT = 8 with 4 when the grid fills the chip three times over (8 would need 2 x 47 KB =
    // one workgroup per CU) and with 8 for the small late layers (fewer barriers per workgroup).
    if (taps == 9 && (p.cot <= 6 || (p.cot == 8 && p.grid_x * p.co_blocks >= 768))) kc = 4;
    p.kc = kc;
    g.k_pad = (int)fsc::round_up(g.cin, kc);
    g.ksplit = 1;
    {
        const long wgs = p.grid_x * p.co_blocks;
        const int nchunks = g.k_pad / kc;
        if (wgs < 256) {
            long ks = (512 + wgs - 1) / wgs;
            if (ks > 8) ks = 8;
            if (ks > nchunks / 4) ks = nchunks / 4;
            if (ks > 1) g.ksplit = (int)ks;
        }
    }
    const int co_blk = p.cot * 16;
    const int cos = co_blk + ((co_blk % 32 == 16) ? 0 : 16);
    if ((kc * g.plane + 255) / 256 > nxi_max) return false;
    p.lds_bytes = 2 * sizeof(float) * ((size_t)taps * kc * cos + (size_t)((kc * g.plane + 3) & ~3));
    *out = p;
    return true;
}

struct XStat { const float* pivot; float4* rec; uint8_t* pidx; };       // statistics records of a STATS forward (null: none); POOL
// where the STATS instantiations exist and a worker keeps one channel block
bool x3_stats_ok(const FwdPlan& p, const fsc_conv_desc& d) {
    return p.x3 == 1 && !p.s1d && d.kh == 1 && p.cot <= 8 && p.g.ksplit == 1 && p.launch_x > 0 &&
           (p.co_blocks == 1 || p.launch_x % p.co_blocks == 0);
}

template <int KH, int KW, int COT, int PT, int NPROD>
void launch_x3_pt(const FwdPlan& p, dim3 grid, const float* in, const float* packed, const float* bias, float* out,
                  int accumulate, const float* in_amax, hipStream_t st, XStat sa = XStat{nullptr, nullptr, nullptr}) {
    const float* w_amax = packed + x3_limb_floats(p);      // fp16 limbs: the weights' largest magnitude follows the fragments
    if constexpr (NPROD == 1 && KH == 1 && COT <= 8) {
        if (sa.rec && sa.pidx) {
            if constexpr (KW == 3) {
                auto kern = conv_fwd_x3_kernel<KH, KW, COT, PT, NPROD, true, true>;
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
                hipLaunchKernelGGL(kern, grid, dim3(kXWaves * 64), p.lds_bytes, st, p.g, in, packed, bias, out, accumulate, in_amax, w_amax,
                                   sa.pivot, sa.rec, sa.pidx);
            }
            return;
        }
        if (sa.rec) {
            auto kern = conv_fwd_x3_kernel<KH, KW, COT, PT, NPROD, true>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
            hipLaunchKernelGGL(kern, grid, dim3(kXWaves * 64), p.lds_bytes, st, p.g, in, packed, bias, out, accumulate, in_amax, w_amax,
                               sa.pivot, sa.rec, (uint8_t*)nullptr);
            return;
        }
    }
    auto kern = conv_fwd_x3_kernel<KH, KW, COT, PT, NPROD>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL(kern, grid, dim3(kXWaves * 64), p.lds_bytes, st, p.g, in, packed, bias, out, accumulate, in_amax, w_amax,
                       (const float*)nullptr, (float4*)nullptr, (uint8_t*)nullptr);
}

template <int KH, int KW, int COT, int PT>
void launch_x3_arith(const FwdPlan& p, dim3 grid, const float* in, const float* packed, const float* bias, float* out,
                     int accumulate, const float* in_amax, hipStream_t st, XStat sa = XStat{nullptr, nullptr, nullptr}) {
    if (p.x3 == 3) launch_x3_pt<KH, KW, COT, PT, 3>(p, grid, in, packed, bias, out, accumulate, in_amax, st);
    else if (p.x3 == 1) launch_x3_pt<KH, KW, COT, PT, 1>(p, grid, in, packed, bias, out, accumulate, in_amax, st, sa);
    else if (p.x3 == 6) launch_x3_pt<KH, KW, COT, PT, 6>(p, grid, in, packed, bias, out, accumulate, in_amax, st);
    else launch_x3_pt<KH, KW, COT, PT, 9>(p, grid, in, packed, bias, out, accumulate, in_amax, st);
}

template <int KH, int KW, int COT>
int launch_fwd_cot(const FwdPlan& p, const float* in, const float* packed, const float* bias, float* out,
                   int accumulate, const float* in_amax, hipStream_t st, XStat sa = XStat{nullptr, nullptr, nullptr}) {
    dim3 grid((unsigned)p.grid_x, p.co_blocks, p.g.ksplit);
    if (p.g.ksplit > 1 && !accumulate) {
        const size_t bytes = sizeof(float) * (size_t)p.g.n * p.g.cout * p.g.hw;
        hipError_t e = hipMemsetAsync(out, 0, bytes, st);
        FSC_CHECK_ARG(e == hipSuccess, "fsc_conv_fwd: memset failed: %s", hipGetErrorString(e));
    }
    if (p.x3) {
        grid.x = (unsigned)p.launch_x;
        grid.y = 1;
        if (p.pt == 2) launch_x3_arith<KH, KW, COT, 2>(p, grid, in, packed, bias, out, accumulate, in_amax, st, sa);
        else launch_x3_arith<KH, KW, COT, 1>(p, grid, in, packed, bias, out, accumulate, in_amax, st, sa);
        FSC_LAUNCH_CHECK("fsc_conv_fwd(x3)");
        return 0;
    }
    if (KH * KW == 9 && p.kc == 4) {
        auto kern = conv_fwd_kernel<KH, KW, COT, fwd_pt(KH * KW), (KH * KW == 9) ? 4 : fwd_kc(KH * KW)>;
        hipLaunchKernelGGL(kern, grid, dim3(kThreads), p.lds_bytes, st, p.g, in, packed, bias, out, accumulate);
    } else {
        auto kern = conv_fwd_kernel<KH, KW, COT, fwd_pt(KH * KW), fwd_kc(KH * KW)>;
        if (p.lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        hipLaunchKernelGGL(kern, grid, dim3(kThreads), p.lds_bytes, st, p.g, in, packed, bias, out, accumulate);
    }
    FSC_LAUNCH_CHECK("fsc_conv_fwd");
    return 0;
}

template <int KH, int KW, int COT>
int launch_fwd_wide(const FwdPlan& p, const float* in, const float* packed, const float* bias, float* out, int accumulate,
                    const float* in_amax, hipStream_t st) {
    // 10 / 15 channel tiles: split-bf16 kernel only (3x3 / k3, one pixel tile per wave, two-slot weight ring)
    if constexpr (KH * KW > 1) {
        dim3 grid((unsigned)p.launch_x, 1, p.g.ksplit);
        if (p.g.ksplit > 1 && !accumulate) {
            const size_t bytes = sizeof(float) * (size_t)p.g.n * p.g.cout * p.g.hw;
            hipError_t e = hipMemsetAsync(out, 0, bytes, st);
            FSC_CHECK_ARG(e == hipSuccess, "fsc_conv_fwd: memset failed: %s", hipGetErrorString(e));
        }
        if constexpr (COT == 10) {
            if (p.pt == 2) {
                launch_x3_pt<KH, KW, COT, 2, 3>(p, grid, in, packed, bias, out, accumulate, in_amax, st);
                FSC_LAUNCH_CHECK("fsc_conv_fwd(x3)");
                return 0;
            }
        }
        launch_x3_arith<KH, KW, COT, 1>(p, grid, in, packed, bias, out, accumulate, in_amax, st);
        FSC_LAUNCH_CHECK("fsc_conv_fwd(x3)");
        return 0;
    } else {
        fsc::set_error("fsc_conv_fwd: internal: wide channel block for a 1x1 kernel");
        return 22;
    }
}

template <int KH, int KW>
int launch_fwd(const FwdPlan& p, const float* in, const float* packed, const float* bias, float* out, int accumulate,
               const float* in_amax, hipStream_t st, XStat sa = XStat{nullptr, nullptr, nullptr}) {
    if (p.cot == 10) return launch_fwd_wide<KH, KW, 10>(p, in, packed, bias, out, accumulate, in_amax, st);
    if (p.cot == 15) return launch_fwd_wide<KH, KW, 15>(p, in, packed, bias, out, accumulate, in_amax, st);
    switch (p.cot) {
        case 1: return launch_fwd_cot<KH, KW, 1>(p, in, packed, bias, out, accumulate, in_amax, st, sa);
        case 2: return launch_fwd_cot<KH, KW, 2>(p, in, packed, bias, out, accumulate, in_amax, st, sa);
        case 3: return launch_fwd_cot<KH, KW, 3>(p, in, packed, bias, out, accumulate, in_amax, st, sa);
        case 4: return launch_fwd_cot<KH, KW, 4>(p, in, packed, bias, out, accumulate, in_amax, st, sa);
        case 5: return launch_fwd_cot<KH, KW, 5>(p, in, packed, bias, out, accumulate, in_amax, st, sa);
        case 6: return launch_fwd_cot<KH, KW, 6>(p, in, packed, bias, out, accumulate, in_amax, st, sa);
        case 7: return launch_fwd_cot<KH, KW, 7>(p, in, packed, bias, out, accumulate, in_amax, st, sa);
        default: return launch_fwd_cot<KH, KW, 8>(p, in, packed, bias, out, accumulate, in_amax, st, sa);
    }
}

// slots == 1: a single float (the packed weights' tail); slots == kAmaxSlots: an FSC_AMAX_FLOATS buffer
int launch_amax(const float* x, long n, float* out, hipStream_t st, int slots) {
    long blocks = (n / 4 + 256 * 8 - 1) / (256 * 8);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (blocks == 1 && slots == 1) {
        hipLaunchKernelGGL(amax_kernel, dim3(1), dim3(256), 0, st, x, n, out, 1, 1);
    } else {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (slots == 1 ? 1 : fsc::kAmaxFloats), st);
        FSC_CHECK_ARG(e == hipSuccess, "fsc_amax: memset failed: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, out, 0, slots);
    }
    FSC_LAUNCH_CHECK("fsc_amax");
    return 0;
}


// stem layers take the direct kernels: 3x3, at most 4 input channels, weights fit the LDS table
bool stem_shape(const fsc_conv_desc& d) {
    return d.kh == 3 && d.kw == 3 && d.c_in <= 4 && d.c_out <= 1024 && d.w >= 8;
}

bool valid_desc(const fsc_conv_desc* d) {
    if (!d || d->n <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->h <= 0 || d->w <= 0) return false;
    if (!(d->arith == FSC_ARITH_DEFAULT || d->arith == 0 || d->arith == 1 || d->arith == 3 || d->arith == 6 || d->arith == 9 || d->arith == 10)) return false;
    const bool k33 = d->kh == 3 && d->kw == 3, k11 = d->kh == 1 && d->kw == 1, k13 = d->kh == 1 && d->kw == 3;
    if (!(k33 || k11 || k13)) return false;
    const long big = 1L << 31;
    return (long)d->n * d->c_in * d->h * d->w < big && (long)d->n * d->c_out * d->h * d->w < big;
}

// ---- wgrad planning
struct WgPlan {
    WgGeom g;
    int mt, co_blocks;
    size_t lds_bytes;
    int threads;
    int packed;           // stem layers: (ci, tap) packed along N, waves split K
    int part_splits;      // split-K slices in the partial workspace
};

bool plan_wgrad(const fsc_conv_desc& d_in, WgPlan* out) {
    fsc_conv_desc d = d_in;
    if (d.kh == 1 && d.kw == 1) {       // no halo: treat each (n, c) plane as one row of h*w pixels
        d.w = d.h * d.w;
        d.h = 1;
    }
    WgPlan p{};
    WgGeom& g = p.g;
    const int taps = d.kh * d.kw;
    const int waves = 4;
    const int cit = taps == 9 ? 2 : (taps == 3 ? 4 : 8);
    const int pixc = 64, maxpos_total = 64 * (taps == 1 ? 1 : (taps == 3 ? 2 : 4));
    // stem layers (packed kernel) are HBM-bound on dOut: PMC showed 3x the tensor fetched with 16-pixel
    // rows (each 64-byte row segment straddles two sectors), so they use full 64-pixel rows
    const bool wide_rows = taps > 1 && d.c_in * taps <= 32 && d.c_in <= 16 && d.w >= 64;
    g.n = d.n; g.cin = d.c_in; g.cout = d.c_out; g.h = d.h; g.w = d.w; g.hw = (long)d.h * d.w;
    const int tiles = fsc::ceil_div(d.c_out, 16);
    int best_mt = 1, best_blocks = tiles;
    long best_tile_cost = (long)tiles * tile_penalty(1);
    for (int mt = 2; mt <= 7; ++mt) {
        const int blocks = fsc::ceil_div(tiles, mt);
        const long cost = (long)blocks * mt * tile_penalty(mt);
        if (cost < best_tile_cost || (cost == best_tile_cost && blocks < best_blocks)) {
            best_mt = mt; best_blocks = blocks; best_tile_cost = cost;
        }
    }
    p.mt = best_mt; p.co_blocks = best_blocks;
    g.co_pad = best_blocks * best_mt * 16;
    g.ci_blocks = fsc::ceil_div(d.c_in, cit * 16);
    g.ci_pad = g.ci_blocks * cit * 16;
    // pixel box of <= 64 pixels whose halo'd footprint fits maxpos*64 positions
    long best_cost = -1;
    int bnb = 1, bth = 1, btw = 1;
    for (int tw = 1; tw <= d.w && tw <= pixc; ++tw) {
        int th = pixc / tw;
        if (th > d.h) th = d.h;
        int nb = 1;
        if (th == d.h && tw == d.w) {
            nb = pixc / (th * tw);
            if (nb > d.n) nb = d.n;
            if (nb < 1) nb = 1;
        }
        while (nb > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > maxpos_total) --nb;
        while (th > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > maxpos_total) --th;
        if (nb * (th + d.kh - 1) * (tw + d.kw - 1) > maxpos_total) continue;
        const long tiles = (long)fsc::ceil_div(d.n, nb) * fsc::ceil_div(d.h, th) * fsc::ceil_div(d.w, tw);
        // halo'd positions are DMA work per unit (64 positions per instruction and channel)
        const long halo_instr = (nb * (th + d.kh - 1) * (tw + d.kw - 1) + 63) / 64;
        long cost = tiles * box_penalty(tw, d.w) * (20 + halo_instr);
        if (wide_rows && tw != pixc) cost *= 4;
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && tw > btw)) {
            best_cost = cost; bnb = nb; bth = th; btw = tw;
        }
    }
    if (best_cost < 0) return false;
    g.nb = bnb; g.th = bth; g.tw = btw;
    g.rows = bth + d.kh - 1; g.cols = btw + d.kw - 1;
    g.npix = bnb * bth * btw; g.npos = bnb * g.rows * g.cols;
    g.tiles_n = fsc::ceil_div(d.n, bnb); g.tiles_h = fsc::ceil_div(d.h, bth); g.tiles_w = fsc::ceil_div(d.w, btw);
    g.units = g.tiles_n * g.tiles_h * g.tiles_w;
    g.plane = pad_plane(g.npos, 2);
    // split-K: all workgroups do the same work, so the grid should fill the 512 resident slots
    // (256 CUs x 2 workgroups) a whole number of times -- 1030 workgroups would cost three rounds.
    // Pick the split count (>= 4 units per split, partials <= 256 MB) with the best slot utilisation.
    const long base = (long)p.co_blocks * g.ci_blocks;
    const long part_bytes_per_split = (long)taps * g.ci_pad * g.co_pad * 4;
#ifndef FSC_WGRAD_SMALL_UNITS
#define FSC_WGRAD_SMALL_UNITS 64
#endif
    // (layers of fewer than 64 units -- the late blocks of the 1-d model: 7 ... 32 boxes of 64 pixels -- split down to one unit per
    // workgroup: 20 workgroups walking 7 units each took 55 us for 0.17 GFLOP)
    long ns_max = g.units < FSC_WGRAD_SMALL_UNITS ? g.units : g.units / 4;
    if (ns_max < 1) ns_max = 1;
    while (ns_max > 1 && ns_max * part_bytes_per_split > (256L << 20)) --ns_max;
    // resident workgroups per CU: 4-wave workgroups put one wave on each SIMD, so it is the waves/SIMD
    // the register file admits (VGPR + AGPR ~ 48 + 20*MT for 3x3, measured) capped by LDS
    const int npw = (cit * taps + waves - 1) / waves;
    const int regs = 48 + 4 * npw * p.mt;
    int per_cu = 512 / (int)fsc::round_up(regs, 8);
    if (per_cu > 8) per_cu = 8;
    const size_t lds_est = sizeof(float) * ((size_t)p.mt * 16 * (pixc + 2) + (size_t)cit * 16 * g.plane + pixc);
    if ((size_t)per_cu * lds_est > 160 * 1024) per_cu = (int)(160 * 1024 / lds_est);
    if (per_cu < 1) per_cu = 1;
    const long slots = 256L * per_cu;
    long ns = 1;
    double best_util = -1.0;
    for (long cand = 1; cand <= ns_max && cand * base <= 2 * slots + base; ++cand) {
        const long wgs = cand * base;
        const double util = (double)wgs / (double)(((wgs + slots - 1) / slots) * slots);
        if (util > best_util + 1e-9) { best_util = util; ns = cand; }
    }
    g.nsplit = (int)ns;
    p.packed = (taps > 1 && d.c_in * taps <= 32 && d.c_in <= 16) ? 1 : 0;
    if (p.packed) {
        g.ci_blocks = 1;
        long want = 256L * 5 / p.co_blocks;           // ~5 resident workgroups per CU (latency-bound units)
        if (want > g.units / 4) want = g.units / 4;
        if (want < 1) want = 1;
        g.nsplit = (int)want;
    }
    p.part_splits = g.nsplit;
    p.threads = waves * 64;
    p.lds_bytes = sizeof(float) * ((size_t)p.mt * 16 * (pixc + 2) + (size_t)(p.packed ? 16 : cit * 16) * g.plane + pixc);
    if (g.npos > maxpos_total) return false;
    *out = p;
    return true;
}

// ---- split-bf16 wgrad planning
struct WgxPlan {
    WgxGeom g;
    int mt;               // kernel template: co tiles per wave
    int nprod;
    size_t lds_bytes;
};

bool plan_wgrad_x3(const fsc_conv_desc& d, int nprod, WgxPlan* out) {
    const int taps = d.kh * d.kw;
    if (taps == 1 || d.c_in < 32 || d.c_out < 32) return false;
    WgxPlan p{};
    WgxGeom& g = p.g;
    g.n = d.n; g.cin = d.c_in; g.cout = d.c_out; g.h = d.h; g.w = d.w; g.hw = (long)d.h * d.w;
    // 64-pixel box: th x tw with tw a multiple of 8 (a run of 8 pixels is one lane's K slice)
    long best_px = -1;
    for (int tw = 64; tw >= 8; tw >>= 1) {
        const int th = 64 / tw;
        if (d.kh == 1 && th != 1) continue;
        if ((th + d.kh - 1) * (tw + 8) > 256) continue;
        const long px = (long)fsc::ceil_div(d.h, th) * fsc::ceil_div(d.w, tw) * 64;
        if (best_px < 0 || px < best_px) { best_px = px; g.th = th; g.tw = tw; }
    }
    if (best_px < 0 || (double)d.h * d.w < 0.75 * (double)best_px) return false;
    g.tiles_h = fsc::ceil_div(d.h, g.th); g.tiles_w = fsc::ceil_div(d.w, g.tw);
    g.units = d.n * g.tiles_h * g.tiles_w;
    g.rowp = g.tw + 8;
    g.plane = (g.th + d.kh - 1) * g.rowp;
    while (g.plane % 64 != 4 && g.plane % 64 != 36) ++g.plane;      // channel rows on distinct 16-byte bank slots
    if (g.plane > 256) g.plane = (g.th + d.kh - 1) * g.rowp;
    g.in_instr = fsc::ceil_div((g.th + d.kh - 1) * g.rowp, 64);
    // workgroup shape: ng co groups x nt ci tiles, <= 4 co tiles per wave
    const int tiles_co = fsc::ceil_div(d.c_out, 16), tiles_ci = fsc::ceil_div(d.c_in, 16);
    double best_eff = -1.0;
    for (int nt = 2; nt <= 8; nt *= 2) {
        const int ng = kWgxWaves / nt;
        const int ci_blocks = fsc::ceil_div(tiles_ci, nt);
        const int co_blocks = fsc::ceil_div(tiles_co, ng * 4);
        const int tpb = fsc::ceil_div(tiles_co, co_blocks);
        const int tpg = fsc::ceil_div(tpb, ng);
        const size_t lds = 2 * sizeof(float) * ((size_t)ng * tpg * 16 * kWgxDso + (size_t)nt * 16 * g.plane);
        if (lds > 160 * 1024) continue;
        // MFMA share of a wave's issue slots grows with the co tiles it owns (54 MFMAs per 44 + 201/tpg VALU)
        static const double kTileWeight[5] = {0.0, 0.55, 0.78, 0.92, 1.0};
        // Dead tiles are skipped, so a workgroup is as slow as its busiest SIMD (waves s and s + 4 share SIMD s;
        // wave = (co group wid / nt, ci tile wid % nt)): sum that over all workgroups of the grid
        long busiest = 0;
        for (int cb = 0; cb < co_blocks; ++cb)
            for (int ib = 0; ib < ci_blocks; ++ib) {
                int worst = 0;
                for (int sd = 0; sd < 4; ++sd) {
                    int load = 0;
                    for (int wv = sd; wv < kWgxWaves; wv += 4) {
                        const int cog = wv / nt, cit = wv % nt;
                        const int blk_live = cb * tpb + tpb < tiles_co ? tpb : tiles_co - cb * tpb;
                        int first, live;
                        wgx_group(blk_live > 0 ? blk_live : 0, ng, cog, &first, &live);
                        if (ib * nt + cit >= tiles_ci) live = 0;
                        load += live;
                    }
                    if (load > worst) worst = load;
                }
                busiest += worst;
            }
        const double eff = (double)tiles_co * tiles_ci / (4.0 * (double)busiest) * kTileWeight[tpg];
        if (eff > best_eff) {
            best_eff = eff;
            g.ng = ng; g.nt = nt; g.tpb = tpb; g.tpg = tpg; g.co_blocks = co_blocks; g.ci_blocks = ci_blocks;
            p.lds_bytes = lds;
        }
    }
    if (best_eff < 0.5) return false;
    p.mt = g.tpg;
    p.nprod = nprod;
    g.co_pad = g.co_blocks * g.tpb * 16;
    g.ci_pad = g.ci_blocks * g.nt * 16;
    // split-K: one workgroup per CU; fill 256 slots without a straggler round, >= 4 units per split
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

template <int KH, int KW, int MT, int NPROD>
void launch_wgrad_x3_k(const WgxPlan& p, const float* in, const float* dout, float* part, const float* in_amax,
                       const float* dout_amax, hipStream_t st) {
    auto kern = conv_wgrad_x3_kernel<KH, KW, MT, NPROD>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    dim3 grid(p.g.co_blocks * p.g.ci_blocks, p.g.nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(kWgxWaves * 64), p.lds_bytes, st, p.g, in, dout, part, in_amax, dout_amax);
}

template <int KH, int KW>
int launch_wgrad_x3(const WgxPlan& p, const float* in, const float* dout, float* part, const float* in_amax,
                    const float* dout_amax, hipStream_t st) {
#define FSC_WGX(MT_)                                                                                          \
    if (p.nprod == 3) launch_wgrad_x3_k<KH, KW, MT_, 3>(p, in, dout, part, in_amax, dout_amax, st);          \
    else if (p.nprod == 1) launch_wgrad_x3_k<KH, KW, MT_, 1>(p, in, dout, part, in_amax, dout_amax, st);     \
    else if (p.nprod == 6) launch_wgrad_x3_k<KH, KW, MT_, 6>(p, in, dout, part, in_amax, dout_amax, st);     \
    else launch_wgrad_x3_k<KH, KW, MT_, 9>(p, in, dout, part, in_amax, dout_amax, st);                       \
    break;
    switch (p.mt) {
        case 1: FSC_WGX(1)
        case 2: FSC_WGX(2)
        case 3: FSC_WGX(3)
        default: FSC_WGX(4)
    }
#undef FSC_WGX
    FSC_LAUNCH_CHECK("fsc_conv_wgrad(x3)");
    return 0;
}

template <int KH, int KW, int MT>
int launch_wgrad_mt(const WgPlan& p, const float* in, const float* dout, float* part, hipStream_t st) {
    dim3 grid(p.co_blocks * p.g.ci_blocks, p.g.nsplit);
    if (p.packed && KH * KW > 1) {
        auto kern = conv_wgrad_kernel<KH, KW, MT, (KH * KW > 1)>;
        hipLaunchKernelGGL(kern, grid, dim3(p.threads), p.lds_bytes, st, p.g, in, dout, part);
    } else {
        auto kern = conv_wgrad_kernel<KH, KW, MT, false>;
        if (p.lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
        hipLaunchKernelGGL(kern, grid, dim3(p.threads), p.lds_bytes, st, p.g, in, dout, part);
    }
    FSC_LAUNCH_CHECK("fsc_conv_wgrad");
    return 0;
}

template <int KH, int KW>
int launch_wgrad(const WgPlan& p, const float* in, const float* dout, float* part, hipStream_t st) {
    switch (p.mt) {
        case 1: return launch_wgrad_mt<KH, KW, 1>(p, in, dout, part, st);
        case 2: return launch_wgrad_mt<KH, KW, 2>(p, in, dout, part, st);
        case 3: return launch_wgrad_mt<KH, KW, 3>(p, in, dout, part, st);
        case 4: return launch_wgrad_mt<KH, KW, 4>(p, in, dout, part, st);
        case 5: return launch_wgrad_mt<KH, KW, 5>(p, in, dout, part, st);
        case 6: return launch_wgrad_mt<KH, KW, 6>(p, in, dout, part, st);
        default: return launch_wgrad_mt<KH, KW, 7>(p, in, dout, part, st);
    }
}

// the weight gradient of a small 1-d layer in the bf16 arithmetic (conv_s1d.hip), or false
bool s1d_wgrad_plan(const fsc_conv_desc& d, fsc::s1d::WPlan* out) {
    if (arith_of(d) != 1) return false;
    fsc_conv_desc dd = d;
    dd.arith = 1;
    return fsc::s1d::plan_wgrad(dd, out);
}

}  // namespace

extern "C" {

size_t fsc_conv_packed_floats(const fsc_conv_desc* d, int dgrad) {
    if (!valid_desc(d)) return 0;
    FwdPlan p;
    if (!plan_fwd(*d, dgrad, &p)) return 0;
    if (p.x3) return x3_limb_floats(p) + (p.x3 == 3 ? 4 : 0);
    return (size_t)d->kh * d->kw * p.g.k_pad * p.g.m_pad;
}

int fsc_conv_default_arith(void) { return default_arith(); }


int fsc_conv_pack_weights(const fsc_conv_desc* d, const float* weight, int dgrad, float* packed, fsc_stream_t stream) {
    FSC_CHECK_ARG(valid_desc(d) && weight && packed, "fsc_conv_pack_weights: bad descriptor or null pointer");
    FwdPlan p;
    FSC_CHECK_ARG(plan_fwd(*d, dgrad, &p), "fsc_conv_pack_weights: no tiling for this shape");
    if (p.x3) {
        const long items = (long)p.co_blocks * p.g.x_steps * p.cot * 512;
        long xb = (items + 255) / 256;
        if (xb > 8192) xb = 8192;
        float* w_amax = nullptr;
        if (p.x3 == 3) {
            w_amax = packed + x3_limb_floats(p);
            const int rc = launch_amax(weight, (long)d->c_out * d->c_in * d->kh * d->kw, w_amax, fsc::as_stream(stream), 1);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(pack_x3_kernel, dim3((unsigned)xb), dim3(256), 0, fsc::as_stream(stream), weight,
                           reinterpret_cast<unsigned short*>(packed), d->c_out, d->c_in, d->kh * d->kw, p.cot, p.co_blocks,
                           p.g.x_nfull, p.g.x_tail_oct, p.g.x_steps, 1, dgrad, w_amax, limbs_of(p.x3));
        FSC_LAUNCH_CHECK("fsc_conv_pack_weights(x3)");
        return 0;
    }
    const long total = (long)d->kh * d->kw * p.g.k_pad * p.g.m_pad;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, fsc::as_stream(stream), weight, packed,
                       d->c_out, d->c_in, d->kh, d->kw, p.g.k_pad, p.g.m_pad, dgrad);
    FSC_LAUNCH_CHECK("fsc_conv_pack_weights");
    return 0;
}

int fsc_conv_pack_weights_multi_supported(const fsc_conv_desc* d, int dgrad) {
    FwdPlan p;
    return valid_desc(d) && plan_fwd(*d, dgrad, &p) && p.x3 != 0 && p.x3 != 3 ? 1 : 0;
}

int fsc_conv_pack_weights_multi(int count, const fsc_conv_desc* descs, const float* const* weights, const int* dgrad,
                                float* const* packed, fsc_stream_t stream) {
    FSC_CHECK_ARG(count > 0 && descs && weights && dgrad && packed, "fsc_conv_pack_weights_multi: bad arguments");
    hipStream_t st = fsc::as_stream(stream);
    for (int first = 0; first < count; first += kPackJobs) {
        X3PackJobs jobs{};
        jobs.n = count - first < kPackJobs ? count - first : kPackJobs;
        int blocks = 0;
        for (int i = 0; i < jobs.n; ++i) {
            const fsc_conv_desc* d = descs + first + i;
            FwdPlan p;
            FSC_CHECK_ARG(valid_desc(d) && weights[first + i] && packed[first + i] && plan_fwd(*d, dgrad[first + i], &p) &&
                              p.x3 != 0 && p.x3 != 3,
                          "fsc_conv_pack_weights_multi: job %d is not a bf16-limb tiling (fsc_conv_pack_weights_multi_supported)", first + i);
            jobs.j[i] = X3PackJob{weights[first + i], reinterpret_cast<unsigned short*>(packed[first + i]), d->c_out, d->c_in,
                                  d->kh * d->kw, p.cot, p.co_blocks, p.g.x_nfull, p.g.x_tail_oct, p.g.x_steps, dgrad[first + i] ? 1 : 0,
                                  limbs_of(p.x3)};
            const long items = (long)p.co_blocks * p.g.x_steps * p.cot * 512;
            long nb = (items + 4 * 256 - 1) / (4 * 256);              // four fragments' elements per thread
            if (nb > 256) nb = 256;
            jobs.block_start[i] = blocks;
            blocks += (int)nb;
        }
        jobs.block_start[jobs.n] = blocks;
        hipLaunchKernelGGL(pack_x3_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, jobs);
        FSC_LAUNCH_CHECK("fsc_conv_pack_weights_multi");
    }
    return 0;
}

int fsc_conv_fwd(const fsc_conv_desc* d, const float* in, const float* packed, const float* bias, int dgrad,
                 int accumulate, float* out, const float* in_amax, fsc_stream_t stream) {
    FSC_CHECK_ARG(valid_desc(d) && in && packed && out, "fsc_conv_fwd: bad descriptor or null pointer");
    FSC_CHECK_ARG(!(dgrad && bias), "fsc_conv_fwd: dgrad takes no bias");
    FwdPlan p;
    FSC_CHECK_ARG(plan_fwd(*d, dgrad, &p), "fsc_conv_fwd: no tiling for this shape");
    FSC_CHECK_ARG(p.x3 != 3 || in_amax, "fsc_conv_fwd: the split-fp16 kernels need in_amax (fsc_amax of `in`)");
    hipStream_t st = fsc::as_stream(stream);
    if (p.s1d) return fsc::s1d::launch_fwd(p.sp, in, reinterpret_cast<const unsigned short*>(packed), bias, out, accumulate, st);
    if (stem_shape(*d) && !p.x3) {
        // stem layer: direct kernels organised around the wide tensor (see conv_stem_*_kernel)
        const int qpr = (d->w + 3) / 4;
        if (!dgrad) {
            const int wrow = d->c_in <= 2 ? 20 : 36;
            dim3 grid(fsc::ceil_div((long)d->h * qpr, kStemThreads), d->n);
            const size_t lds = sizeof(float) * (size_t)d->c_out * wrow;
#define FSC_STEM_F(CI_) hipLaunchKernelGGL(conv_stem_fwd_kernel<CI_>, grid, dim3(kStemThreads), lds, st, in, packed, bias, out, \
                                           d->c_out, d->h, d->w, p.g.k_pad, p.g.m_pad, accumulate)
            switch (d->c_in) { case 1: FSC_STEM_F(1); break; case 2: FSC_STEM_F(2); break; case 3: FSC_STEM_F(3); break; default: FSC_STEM_F(4); }
#undef FSC_STEM_F
        } else {
            dim3 grid(fsc::ceil_div(qpr, 32) * fsc::ceil_div(d->h, 8), d->n);
            const size_t lds = sizeof(float) * (size_t)d->c_out * ((9 * d->c_in + 3) & ~3);
#define FSC_STEM_D(CI_) hipLaunchKernelGGL(conv_stem_dgrad_kernel<CI_>, grid, dim3(kStemThreads), lds, st, in, packed, out, \
                                           d->c_out, d->h, d->w, p.g.k_pad, p.g.m_pad, accumulate)
            switch (d->c_in) { case 1: FSC_STEM_D(1); break; case 2: FSC_STEM_D(2); break; case 3: FSC_STEM_D(3); break; default: FSC_STEM_D(4); }
#undef FSC_STEM_D
        }
        FSC_LAUNCH_CHECK("fsc_conv_fwd(stem)");
        return 0;
    }
    if (d->kh == 3) return launch_fwd<3, 3>(p, in, packed, bias, out, accumulate, in_amax, st);
    if (d->kw == 3) return launch_fwd<1, 3>(p, in, packed, bias, out, accumulate, in_amax, st);
    return launch_fwd<1, 1>(p, in, packed, bias, out, accumulate, in_amax, st);
}

/* statistics records of fsc_conv_fwd_stats: out4 = {workers, channel blocks, channels per block, order (2: transposed, [channel][slot])};
 * workers * 8 * channels-per-block float4 {sum (y - pivot), sum (y - pivot)^2, min, max} -- the records of fsc_conv_l16_stats_layout.  0: no such kernel
 * for this layer (anything but plain bf16 arithmetic on 1-d rows with a ring-kernel tiling). */
int fsc_conv_fwd_stats_layout(const fsc_conv_desc* d, int* out4) {
    FwdPlan p;
    if (!valid_desc(d) || !out4 || !plan_fwd(*d, 0, &p) || !x3_stats_ok(p, *d)) return 0;
    out4[0] = (int)p.launch_x; out4[1] = p.co_blocks; out4[2] = p.cot * 16; out4[3] = 2;      // (order 2: [channel][slot])
    return 1;
}

int fsc_conv_fwd_stats(const fsc_conv_desc* d, const float* in, const float* packed, const float* bias, float* out,
                       const float* stat_pivot, void* stat_rec, fsc_stream_t stream) {
    FSC_CHECK_ARG(valid_desc(d) && in && packed && out && stat_rec, "fsc_conv_fwd_stats: bad descriptor or null pointer");
    FwdPlan p;
    FSC_CHECK_ARG(plan_fwd(*d, 0, &p) && x3_stats_ok(p, *d), "fsc_conv_fwd_stats: unsupported layer (see fsc_conv_fwd_stats_layout)");
    const XStat sa{stat_pivot, reinterpret_cast<float4*>(stat_rec), nullptr};
    hipStream_t st = fsc::as_stream(stream);
    if (d->kw == 3) return launch_fwd<1, 3>(p, in, packed, bias, out, 0, nullptr, st, sa);
    return launch_fwd<1, 1>(p, in, packed, bias, out, 0, nullptr, st, sa);
}

/* fsc_conv_fwd_stats of a k3 Conv1d followed by MaxPool1d(2) in one launch: `pooled` (n, c_out, 1, w / 2) and `pool_idx` (same
 * shape, bytes) are what fsc_maxpool_fwd would give on the convolution's output -- which is never written --, the records are the
 * statistics of the POOLED tensor (layout: fsc_conv_fwd_stats_layout of the same descriptor). */
int fsc_conv_fwd_pool_stats_supported(const fsc_conv_desc* d) {
    FwdPlan p;
    return (valid_desc(d) && d->kh == 1 && d->kw == 3 && d->h == 1 && d->w >= 2 && plan_fwd(*d, 0, &p) && x3_stats_ok(p, *d)) ? 1 : 0;
}

int fsc_conv_fwd_pool_stats(const fsc_conv_desc* d, const float* in, const float* packed, const float* bias, float* pooled,
                            uint8_t* pool_idx, const float* stat_pivot, void* stat_rec, fsc_stream_t stream) {
    FSC_CHECK_ARG(valid_desc(d) && in && packed && pooled && pool_idx && stat_rec, "fsc_conv_fwd_pool_stats: bad descriptor or null pointer");
    FSC_CHECK_ARG(fsc_conv_fwd_pool_stats_supported(d), "fsc_conv_fwd_pool_stats: unsupported layer (see fsc_conv_fwd_pool_stats_supported)");
    FwdPlan p;
    plan_fwd(*d, 0, &p);
    const XStat sa{stat_pivot, reinterpret_cast<float4*>(stat_rec), pool_idx};
    return launch_fwd<1, 3>(p, in, packed, bias, pooled, 0, nullptr, fsc::as_stream(stream), sa);
}

int fsc_amax(const float* x, long n, float* out, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && out && n > 0, "fsc_amax: bad arguments");
    return launch_amax(x, n, out, fsc::as_stream(stream), fsc::kAmaxSlots);
}

int fsc_conv_pool_supported(const fsc_conv_desc* d) {
    return valid_desc(d) && stem_shape(*d) && d->c_in <= 2 && d->h >= 2 && d->w >= 8 ? 1 : 0;
}

int fsc_conv_pool_fwd(const fsc_conv_desc* d, const float* in, const float* packed, const float* bias, float* pooled,
                      uint8_t* idx, fsc_stream_t stream) {
    FSC_CHECK_ARG(fsc_conv_pool_supported(d) && in && packed && pooled && idx,
                  "fsc_conv_pool_fwd: only 3x3 stem layers (c_in <= 2) are fused with the 2x2 max-pool");
    FwdPlan p;
    FSC_CHECK_ARG(plan_fwd(*d, 0, &p) && !p.x3, "fsc_conv_pool_fwd: no tiling for this shape");
    const int oh = d->h / 2, ow = d->w / 2, qpr = (ow + 3) / 4;
    dim3 grid(fsc::ceil_div((long)oh * qpr, kStemThreads), d->n);
    const size_t lds = sizeof(float) * (size_t)d->c_out * 20;
    hipStream_t st = fsc::as_stream(stream);
    if (d->c_in == 1)
        hipLaunchKernelGGL(conv_stem_pool_fwd_kernel<1>, grid, dim3(kStemThreads), lds, st, in, packed, bias, pooled, idx,
                           d->c_out, d->h, d->w, p.g.k_pad, p.g.m_pad);
    else
        hipLaunchKernelGGL(conv_stem_pool_fwd_kernel<2>, grid, dim3(kStemThreads), lds, st, in, packed, bias, pooled, idx,
                           d->c_out, d->h, d->w, p.g.k_pad, p.g.m_pad);
    FSC_LAUNCH_CHECK("fsc_conv_pool_fwd");
    return 0;
}

int fsc_conv_plan_describe(const fsc_conv_desc* d, int mode, char* buf, size_t buf_len) {
    FSC_CHECK_ARG(valid_desc(d) && buf && buf_len > 0, "fsc_conv_plan_describe: bad arguments");
    if (mode == 2) {
        fsc::s1d::WPlan ws;
        if (s1d_wgrad_plan(*d, &ws)) {
            snprintf(buf, buf_len, "conv_s1d_wgrad_kernel<%d,%d> units=%d split=%d grid=%dx%dx%d", d->kh, d->kw, ws.ksteps, ws.nsplit,
                     ws.ci_blocks, ws.co_blocks, ws.nsplit);
            return 0;
        }
        WgxPlan px;
        if (arith_of(*d) && plan_wgrad_x3(*d, arith_of(*d), &px)) {
            snprintf(buf, buf_len, "conv_wgrad_x3_kernel<%d,%d,%d,%d> box=%dx%d groups=%dx%d tiles/block=%d units=%d split=%d grid=%dx%d lds=%zu",
                     d->kh, d->kw, px.mt, px.nprod, px.g.th, px.g.tw, px.g.ng, px.g.nt, px.g.tpb, px.g.units, px.g.nsplit,
                     px.g.co_blocks * px.g.ci_blocks, px.g.nsplit, px.lds_bytes);
            return 0;
        }
        WgPlan p;
        FSC_CHECK_ARG(plan_wgrad(*d, &p), "fsc_conv_plan_describe: no tiling for this shape");
        snprintf(buf, buf_len, "conv_wgrad_kernel<%d,%d,%d%s> box=%dx%dx%d units=%d split=%d grid=%dx%d lds=%zu",
                 d->kh, d->kw, p.mt, p.packed ? ",packed" : "", p.g.nb, p.g.th, p.g.tw, p.g.units, p.g.nsplit,
                 p.co_blocks * p.g.ci_blocks, p.g.nsplit, p.lds_bytes);
    } else {
        FwdPlan p;
        FSC_CHECK_ARG(plan_fwd(*d, mode, &p), "fsc_conv_plan_describe: no tiling for this shape");
        if (p.s1d)
            snprintf(buf, buf_len, "conv_s1d_fwd_kernel<%d,%d%s> tile=%dx%d grid=%dx%d steps=%d waves=%d", d->kh, d->kw,
                     p.sp.nprod == 9 ? ",9" : "", p.sp.cot * 16, fsc::s1d::kPt * 16, p.sp.px_groups, p.sp.co_blocks, p.sp.steps,
                     fsc::s1d::kWaves);
        else if (stem_shape(*d) && !p.x3)
            snprintf(buf, buf_len, "%s<%d> quads=%d", mode ? "conv_stem_dgrad_kernel" : "conv_stem_fwd_kernel", d->c_in,
                     d->h * ((d->w + 3) / 4));
        else if (p.x3)
            snprintf(buf, buf_len, "conv_fwd_x3_kernel<%d,%d,%d,%d,%d> box=%dx%dx%d grid=%ldx%dx%d steps=%d lds=%zu", d->kh,
                     d->kw, p.cot, p.pt, p.x3, p.g.nb, p.g.th, p.g.tw, p.grid_x, p.co_blocks, p.g.ksplit, p.g.x_steps,
                     p.lds_bytes);
        else
            snprintf(buf, buf_len, "conv_fwd_kernel<%d,%d,%d,%d> box=%dx%dx%d flat=%d grid=%ldx%dx%d kc=%d lds=%zu", d->kh,
                     d->kw, p.cot, fwd_pt(d->kh * d->kw), p.g.nb, p.g.th, p.g.tw, p.g.flat, p.grid_x, p.co_blocks,
                     p.g.ksplit, p.kc, p.lds_bytes);
    }
    return 0;
}

size_t fsc_conv_wgrad_workspace_bytes(const fsc_conv_desc* d) {
    if (!valid_desc(d)) return 0;
    fsc::s1d::WPlan ws;
    if (s1d_wgrad_plan(*d, &ws)) return (size_t)ws.nsplit * d->kh * d->kw * ws.ci_pad * ws.co_pad * sizeof(float);
    WgxPlan px;
    if (arith_of(*d) && plan_wgrad_x3(*d, arith_of(*d), &px))
        return (size_t)px.g.nsplit * d->kh * d->kw * px.g.ci_pad * px.g.co_pad * sizeof(float);
    WgPlan p;
    if (!plan_wgrad(*d, &p)) return 0;
    return (size_t)p.part_splits * d->kh * d->kw * p.g.ci_pad * p.g.co_pad * sizeof(float);
}

// the split-K slices of a weight gradient into `workspace`; *ci_pad, *co_pad, *nsplit describe them for the reduce
static int wgrad_partial(const fsc_conv_desc* d, const float* in, const float* dout, void* workspace, const float* in_amax,
                         const float* dout_amax, hipStream_t st, int* ci_pad, int* co_pad, int* nsplit) {
    float* part = reinterpret_cast<float*>(workspace);
    fsc::s1d::WPlan ws;
    if (s1d_wgrad_plan(*d, &ws)) {                              // bf16, few positions: conv_s1d.hip
        *ci_pad = ws.ci_pad; *co_pad = ws.co_pad; *nsplit = ws.nsplit;
        if (!in) return 0;
        return fsc::s1d::launch_wgrad(ws, in, dout, part, st);
    }
    WgxPlan px;
    if (arith_of(*d) && plan_wgrad_x3(*d, arith_of(*d), &px)) {
        FSC_CHECK_ARG(px.nprod != 3 || (in_amax && dout_amax),
                      "fsc_conv_wgrad: the split-fp16 kernels need in_amax and dout_amax (fsc_amax of the operands)");
        *ci_pad = px.g.ci_pad; *co_pad = px.g.co_pad; *nsplit = px.g.nsplit;
        if (!in) return 0;                                       // (geometry only)
        if (d->kh == 3) return launch_wgrad_x3<3, 3>(px, in, dout, part, in_amax, dout_amax, st);
        return launch_wgrad_x3<1, 3>(px, in, dout, part, in_amax, dout_amax, st);
    }
    WgPlan p;
    FSC_CHECK_ARG(plan_wgrad(*d, &p), "fsc_conv_wgrad: no tiling for this shape");
    *ci_pad = p.g.ci_pad; *co_pad = p.g.co_pad; *nsplit = p.part_splits;
    if (!in) return 0;
    if (d->kh == 3) return launch_wgrad<3, 3>(p, in, dout, part, st);
    if (d->kw == 3) return launch_wgrad<1, 3>(p, in, dout, part, st);
    return launch_wgrad<1, 1>(p, in, dout, part, st);
}

int fsc_conv_wgrad(const fsc_conv_desc* d, const float* in, const float* dout, float* dweight, void* workspace,
                   const float* in_amax, const float* dout_amax, fsc_stream_t stream) {
    FSC_CHECK_ARG(valid_desc(d) && in && dout && dweight && workspace, "fsc_conv_wgrad: bad descriptor or null pointer");
    hipStream_t st = fsc::as_stream(stream);
    int ci_pad, co_pad, nsplit;
    const int rc = wgrad_partial(d, in, dout, workspace, in_amax, dout_amax, st, &ci_pad, &co_pad, &nsplit);
    if (rc) return rc;
    const int taps = d->kh * d->kw;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(fsc::ceil_div(d->c_out, kRedX), taps * d->c_in), dim3(kRedX, kRedY), 0, st,
                       reinterpret_cast<const float*>(workspace), dweight, d->c_out, d->c_in, taps, ci_pad, co_pad, nsplit);
    FSC_LAUNCH_CHECK("fsc_conv_wgrad(reduce)");
    return 0;
}

int fsc_conv_wgrad_partial(const fsc_conv_desc* d, const float* in, const float* dout, void* workspace, const float* in_amax,
                           const float* dout_amax, fsc_stream_t stream) {
    FSC_CHECK_ARG(valid_desc(d) && in && dout && workspace, "fsc_conv_wgrad_partial: bad descriptor or null pointer");
    int ci_pad, co_pad, nsplit;
    const int rc = wgrad_partial(d, in, dout, workspace, in_amax, dout_amax, fsc::as_stream(stream), &ci_pad, &co_pad, &nsplit);
    if (rc) return rc;
    FSC_LAUNCH_CHECK("fsc_conv_wgrad_partial");
    return 0;
}

int fsc_conv_wgrad_reduce_multi(int count, const fsc_conv_desc* descs, const void* const* workspaces, float* const* dweights,
                                fsc_stream_t stream) {
    FSC_CHECK_ARG(count > 0 && descs && workspaces && dweights, "fsc_conv_wgrad_reduce_multi: bad arguments");
    hipStream_t st = fsc::as_stream(stream);
    const float one = 1.f;                  // (any non-null amax: only the geometry is asked for)
    for (int first = 0; first < count; first += kRedJobs) {
        RedJobs jobs{};
        jobs.n = count - first < kRedJobs ? count - first : kRedJobs;
        int blocks = 0;
        for (int i = 0; i < jobs.n; ++i) {
            const fsc_conv_desc* d = descs + first + i;
            FSC_CHECK_ARG(valid_desc(d) && workspaces[first + i] && dweights[first + i], "fsc_conv_wgrad_reduce_multi: job %d", first + i);
            int ci_pad, co_pad, nsplit;
            const int rc = wgrad_partial(d, nullptr, nullptr, nullptr, &one, &one, st, &ci_pad, &co_pad, &nsplit);
            if (rc) return rc;
            const int taps = d->kh * d->kw, bx = fsc::ceil_div(d->c_out, kRedX);
            jobs.j[i] = RedJob{reinterpret_cast<const float*>(workspaces[first + i]), dweights[first + i], d->c_out, d->c_in, taps,
                               ci_pad, co_pad, nsplit, bx};
            jobs.block_start[i] = blocks;
            blocks += bx * taps * d->c_in;
        }
        jobs.block_start[jobs.n] = blocks;
        hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)blocks), dim3(kRedX, kRedY), 0, st, jobs);
        FSC_LAUNCH_CHECK("fsc_conv_wgrad_reduce_multi");
    }
    return 0;
}

}  // extern "C"
