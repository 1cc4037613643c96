// Fused audio front-end for gfx950: waveform -> reflect-padded Hann frames -> real FFT ->
// magnitude -> (banded mel projection) -> log -> (frequency-encoding channel).
// Replaces ops/utils.py:110-127 + networks/classifiers.py:565-582 of the reference, which
// materialise the complex STFT (226 MB at cfg 2) and its magnitude before the mel conv1d.
//
// One workgroup owns FG consecutive frames of one clip.  A frame's n_fft real samples are
// packed into an n_fft/2-point complex sequence, transformed by an LDS Stockham FFT
// (radix-4 passes, one radix-2 pass when log2 is odd) whose twiddles are staged in LDS once
// per workgroup, unpacked to the one-sided spectrum, and reduced to mel bins straight from
// LDS.  Results are collected in an LDS tile [feature][frame] so the global store writes
// contiguous runs along the frame axis of the (N, F, frames) output.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__global__ void tables_kernel(float* tables, int n_fft) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_fft) return;
    const double x = (double)i / (double)n_fft;
    // periodic Hann: 0.5 - 0.5 cos(2 pi i / n_fft)
    tables[i] = (float)(0.5 - 0.5 * cospi(2.0 * x));
    double s, c;
    sincospi(-2.0 * x, &s, &c);
    tables[n_fft + 2 * i] = (float)c;
    tables[n_fft + 2 * i + 1] = (float)s;
}

struct FrontendArgs {
    const float* wave;
    long wave_stride;
    int t;
    int n_fft, hop, frames;
    const float* tables;
    const int* mel_start;
    const int* mel_len;
    const float* mel_w;
    int n_mel;     // number of output features (n_mel in mel mode, n_fft/2+1 in stft mode)
    float log_eps;
    int apply_log;
    float* out;
    long out_n_stride;
    int freq_channel;
    int fg;        // frames per workgroup
    int tpf;       // threads per frame (power of two, 64..256)
    int block_sync;   // FSC_FE_BLOCK_SYNC: workgroup barriers between the passes of a frame even with one wave per frame (debugging)
};

// MEL = true: banded mel projection + log.  MEL = false: (log) magnitude of every bin.
template <bool MEL>
__global__ __launch_bounds__(kThreads) void frontend_kernel(FrontendArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n_fft = a.n_fft, nc = n_fft >> 1, nbins = nc + 1;
    const int tid = threadIdx.x;
    const int tpf = a.tpf, fpb = kThreads / tpf;
    const int sub = tid / tpf, st = tid - sub * tpf;
    const int clip = blockIdx.y;
    const int f_base = blockIdx.x * a.fg;
    const int tile_ld = a.fg + 1;

    float2* tw = reinterpret_cast<float2*>(smem);                 // n_fft entries
    float2* buf0 = tw + n_fft + (size_t)sub * 2 * nc;             // nc entries
    float2* buf1 = buf0 + nc;
    float* tile = smem + 2 * n_fft + (size_t)fpb * 4 * nc;        // n_out x (fg + 1)

    const float* win = a.tables;
    const float2* tw_g = reinterpret_cast<const float2*>(a.tables + n_fft);
    for (int i = tid; i < n_fft; i += kThreads) tw[i] = tw_g[i];
    __syncthreads();
    // a frame's buffers belong to its `tpf` threads: with one wavefront per frame (n_fft <= 256) the passes of a frame are
    // ordered by the wave's own LDS queue and need no workgroup barrier (seven per frame otherwise)
    const bool wave_frames = tpf <= 64 && !a.block_sync;
    auto frame_sync = [&]() {
        if (wave_frames) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };

    const float* wav = a.wave + (long)clip * a.wave_stride;
    const int t = a.t;
    const int iters = (a.fg + fpb - 1) / fpb;

    for (int it = 0; it < iters; ++it) {
        const int fl = it * fpb + sub;           // frame index inside the group
        const int f = f_base + fl;
        const bool live = (fl < a.fg) && (f < a.frames);
        // ---- load + window + pack (even, odd) -> complex
        if (live) {
            const long s0 = (long)f * a.hop - nc;
            for (int j = st; j < nc; j += tpf) {
                long i0 = s0 + 2 * j, i1 = i0 + 1;
                if (i0 < 0) i0 = -i0; else if (i0 >= t) i0 = 2L * (t - 1) - i0;
                if (i1 < 0) i1 = -i1; else if (i1 >= t) i1 = 2L * (t - 1) - i1;
                buf0[j] = make_float2(wav[i0] * win[2 * j], wav[i1] * win[2 * j + 1]);
            }
        }
        frame_sync();
        // ---- Stockham FFT of nc complex points
        float2* src = buf0;
        float2* dst = buf1;
        int p = 1;
        const int quarter = nc >> 2;
        while (p * 4 <= nc) {
            if (live) {
                const int tstep = n_fft / (4 * p);
                for (int b = st; b < quarter; b += tpf) {
                    const int k = b & (p - 1);
                    const int j = ((b - k) << 2) + k;
                    const int m1 = k * tstep;
                    const float2 u0 = src[b];
                    const float2 u1 = cmul(src[b + quarter], tw[m1]);
                    const float2 u2 = cmul(src[b + 2 * quarter], tw[2 * m1]);
                    const float2 u3 = cmul(src[b + 3 * quarter], tw[3 * m1]);
                    const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y);
                    const float2 v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
                    const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
                    const float2 v3 = make_float2(u1.y - u3.y, u3.x - u1.x);   // -i (u1 - u3)
                    dst[j] = make_float2(v0.x + v2.x, v0.y + v2.y);
                    dst[j + p] = make_float2(v1.x + v3.x, v1.y + v3.y);
                    dst[j + 2 * p] = make_float2(v0.x - v2.x, v0.y - v2.y);
                    dst[j + 3 * p] = make_float2(v1.x - v3.x, v1.y - v3.y);
                }
            }
            frame_sync();
            float2* tmp = src; src = dst; dst = tmp;
            p <<= 2;
        }
        if (p < nc) {   // one radix-2 pass left (p * 2 == nc)
            if (live) {
                const int half = nc >> 1;
                const int tstep = n_fft / (2 * p);
                for (int b = st; b < half; b += tpf) {
                    const int k = b & (p - 1);
                    const int j = ((b - k) << 1) + k;
                    const float2 u0 = src[b];
                    const float2 u1 = cmul(src[b + half], tw[k * tstep]);
                    dst[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
                    dst[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
                }
            }
            frame_sync();
            float2* tmp = src; src = dst; dst = tmp;
        }
        // ---- unpack to the one-sided spectrum of the real frame, magnitude -> dst (as floats)
        float* mag = reinterpret_cast<float*>(dst);
        if (live) {
            for (int k = st; k < nbins; k += tpf) {
                const float2 zk = src[k & (nc - 1)];
                const float2 zr = src[(nc - k) & (nc - 1)];
                // even part E = (zk + conj(zr)) / 2, odd part O = (zk - conj(zr)) / (2i)
                const float2 e = make_float2(0.5f * (zk.x + zr.x), 0.5f * (zk.y - zr.y));
                const float2 o = make_float2(0.5f * (zk.y + zr.y), -0.5f * (zk.x - zr.x));
                const float2 wo = cmul(tw[k], o);
                const float re = e.x + wo.x, im = e.y + wo.y;
                mag[k] = sqrtf(re * re + im * im);
            }
        }
        frame_sync();
        if (live) {
            if (MEL) {
                // the band of a high mel bin is ~90 spectrum bins long: four independent partial sums keep four
                // weight loads (L2) in flight instead of one load-to-use latency per bin
                for (int m = st; m < a.n_mel; m += tpf) {
                    const int s = a.mel_start[m], len = a.mel_len[m];
                    const float* wcol = a.mel_w + m;
                    const float* mg = mag + s;
                    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
                    int j = 0;
                    for (; j + 3 < len; j += 4) {
                        const float w0 = wcol[(long)j * a.n_mel], w1 = wcol[(long)(j + 1) * a.n_mel];
                        const float w2 = wcol[(long)(j + 2) * a.n_mel], w3 = wcol[(long)(j + 3) * a.n_mel];
                        acc0 = fmaf(w0, mg[j], acc0);
                        acc1 = fmaf(w1, mg[j + 1], acc1);
                        acc2 = fmaf(w2, mg[j + 2], acc2);
                        acc3 = fmaf(w3, mg[j + 3], acc3);
                    }
                    for (; j < len; ++j) acc0 = fmaf(wcol[(long)j * a.n_mel], mg[j], acc0);
                    tile[m * tile_ld + fl] = logf((acc0 + acc1) + (acc2 + acc3) + a.log_eps);
                }
            } else {
                for (int k = st; k < nbins; k += tpf)
                    tile[k * tile_ld + fl] = a.apply_log ? logf(mag[k] + a.log_eps) : mag[k];
            }
        }
        frame_sync();
    }
    __syncthreads();
    // ---- coalesced store of the tile: rows = features, contiguous along frames
    const int fvalid = min(a.fg, a.frames - f_base);
    float* dst_g = a.out + (long)clip * a.out_n_stride;
    const int total = a.n_mel * fvalid;
    for (int i = tid; i < total; i += kThreads) {
        const int row = i / fvalid, fl = i - row * fvalid;
        dst_g[(long)row * a.frames + f_base + fl] = tile[row * tile_ld + fl];
    }
    if (a.freq_channel) {
        // torch.linspace(-1, 1, h): symmetric evaluation in fp32
        float* fq = dst_g + (long)a.n_mel * a.frames;
        const int h = a.n_mel;
        const float step = 2.0f / (float)(h - 1);
        for (int i = tid; i < total; i += kThreads) {
            const int row = i / fvalid, fl = i - row * fvalid;
            const float v = (row < h / 2) ? (-1.0f + step * (float)row)
                                          : (1.0f - step * (float)(h - 1 - row));
            fq[(long)row * a.frames + f_base + fl] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------
// n_fft = 2048 (the 44.1 kHz configurations): one WAVE per frame, no workgroup barrier inside a frame.
// The 1024-point complex FFT of the packed frame is 16 x 16 x 4 (Cooley-Tukey, decimation in time):
//   A  lane L holds z[64 n1 + L], n1 = 0..15: a 16-point FFT in registers, twiddle W_1024^(L k1);
//   B  transpose through a wave-private LDS patch: lane (k1 = L >> 2, c = L & 3) takes row k1, columns 4 m + c:
//      a second 16-point FFT in registers, twiddle W_64^(c k);
//   C  the remaining 4-point FFT runs across the four lanes of a quad (two xor shuffles).
// Then the spectrum goes back to LDS in natural order, lane pairs (k, 1024 - k) unpack the real transform, and the
// magnitudes feed the banded mel sums.  The generic kernel above takes 5 radix-4 passes with a workgroup barrier each,
// one butterfly per thread: 0.78 ms at cfg 2 against 0.03 ms of HBM time.
// Measured at cfg 2 (tools/ab/fe.sh, same box): 16 frames / launch bound 3 waves per SIMD 0.335 ms; branch-free reflect indexing +
// v_sqrt_f32 (no spilled lane masks) with the bound relaxed to 2: 0.251 ms; 12 frames per workgroup: 0.243 ms; 8: 0.251; 32: 0.297.
// (With the bound at 3 the same source compiles to a 0.366 ms kernel at the same 166 VGPRs: the scheduler's choice, not occupancy.)
#ifndef FSC_FE_FRAMES
#define FSC_FE_FRAMES 12
#endif
#ifndef FSC_FE_MINWAVES
#define FSC_FE_MINWAVES 2
#endif
constexpr int kF2Frames = FSC_FE_FRAMES;           // frames per workgroup (mel mode); 4 waves x 3 frames
constexpr int kF2Patch = 16 * 68;                  // float2 per wave: 16 rows of 64 (+ 4 pad: conflict-free column reads)

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }      // a * (-i)

// forward 4-point DFT (e^{-2 pi i jk / 4})
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = mul_mi(csub(a1, a3));
    a0 = cadd(s02, s13);
    a1 = cadd(d02, d13);
    a2 = csub(s02, s13);
    a3 = csub(d02, d13);
}

// forward 16-point DFT in registers, natural order in and out: 4 x 4 Cooley-Tukey with constant twiddles W_16^(n2 k1)
__device__ __forceinline__ void fft16(float2 (&x)[16]) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    // columns n2 = 0..3: DFT4 over n1 of x[4 n1 + n2] -> y[k1][n2] kept at x[4 k1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
    // twiddles W_16^(n2 k1), k1 = 1..3, n2 = 1..3
    const float2 w1 = make_float2(c1, -s1), w2 = make_float2(h, -h), w3 = make_float2(s1, -c1);
    const float2 w6 = make_float2(-h, -h), w9 = make_float2(-c1, s1);
    x[4 + 1] = cmul(x[4 + 1], w1); x[4 + 2] = cmul(x[4 + 2], w2); x[4 + 3] = cmul(x[4 + 3], w3);
    x[8 + 1] = cmul(x[8 + 1], w2); x[8 + 2] = mul_mi(x[8 + 2]);   x[8 + 3] = cmul(x[8 + 3], w6);
    x[12 + 1] = cmul(x[12 + 1], w3); x[12 + 2] = cmul(x[12 + 2], w6); x[12 + 3] = cmul(x[12 + 3], w9);
    // rows k1 = 0..3: DFT4 over n2 -> X[k1 + 4 k2] at x[4 k1 + k2]
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
    // to natural order: X[k1 + 4 k2] sits at x[4 k1 + k2] -> transpose the 4 x 4 index
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = k1 + 1; k2 < 4; ++k2) {
            const float2 t = x[4 * k1 + k2];
            x[4 * k1 + k2] = x[4 * k2 + k1];
            x[4 * k2 + k1] = t;
        }
}

// exp(-2 pi i m / 2048) for 0 <= m < 2048 from the 1025-entry LDS table (m >= 1024: negated)
__device__ __forceinline__ float2 tw2048(const float2* tw, int m) {
    const float2 v = tw[m & 1023];
    return (m & 1024) ? make_float2(-v.x, -v.y) : v;
}

// lane ^ 1 / lane ^ 2 inside a quad as DPP moves (quad_perm [1,0,3,2] / [2,3,0,1]); __shfl_xor takes the LDS crossbar
__device__ __forceinline__ float quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}

template <bool MEL>
__global__ __launch_bounds__(kThreads, FSC_FE_MINWAVES) void frontend2048_kernel(FrontendArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = 1024;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int clip = blockIdx.y;
    const int f_base = blockIdx.x * a.fg;
    const int tile_ld = a.fg + 1;
    float2* tw = reinterpret_cast<float2*>(smem);                                   // 1025 entries (+ pad to 1032)
    float2* patch = tw + 1032 + (size_t)wid * kF2Patch;                             // this wave's transpose / spectrum buffer
    float* tile = smem + 2 * 1032 + (size_t)(kThreads / 64) * 2 * kF2Patch;         // n_out x (fg + 1)
    const float* win = a.tables;
    const float2* tw_g = reinterpret_cast<const float2*>(a.tables + 2048);
    for (int i = tid; i < 1025; i += kThreads) tw[i] = i < 1024 ? tw_g[i] : make_float2(-1.f, 0.f);
    __syncthreads();

    const float* wav = a.wave + (long)clip * a.wave_stride;
    const int t = a.t;
    const int k1q = lane >> 2, cq = lane & 3;
    const int k2q = ((cq & 1) << 1) | (cq >> 1);            // quad lane c ends up with output k2 = bit-reversed c

    for (int fl = wid; fl < a.fg; fl += kThreads / 64) {
        const int f = f_base + fl;
        if (f >= a.frames) break;                           // (wave-uniform)
        // ---- A: load + window + pack: z[j] = (x[2j] w[2j], x[2j+1] w[2j+1]), j = 64 n1 + lane
        float2 x[16];
        const long s0 = (long)f * a.hop - NC;
        // interior frames with an 8-byte aligned start read sample pairs (hop and clip stride even: every cfg); frames that
        // touch a clip end take the reflect-padded scalar path (ops/utils.py:110-127: center=True, pad_mode="reflect")
        const bool pairs = s0 >= 0 && s0 + 2 * NC <= t && ((s0 | a.wave_stride) & 1) == 0 &&
                           (reinterpret_cast<uintptr_t>(a.wave) & 7) == 0;
        if (pairs) {                                        // (wave-uniform: one of the two loops runs)
            const float2* src = reinterpret_cast<const float2*>(wav + s0) + lane;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const float2 w = reinterpret_cast<const float2*>(win)[64 * n1 + lane];
                const float2 v = src[64 * n1];
                x[n1] = make_float2(v.x * w.x, v.y * w.y);
            }
        } else {
            // reflect padding without branches: index -> |index|, then mirrored at t - 1 (one reflection suffices: t > n_fft / 2)
            const int base = (int)s0 + 2 * lane, hi = 2 * (t - 1);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const float2 w = reinterpret_cast<const float2*>(win)[64 * n1 + lane];
                int i0 = base + 128 * n1, i1 = i0 + 1;
                i0 = i0 < 0 ? -i0 : i0; i1 = i1 < 0 ? -i1 : i1;
                i0 = i0 >= t ? hi - i0 : i0; i1 = i1 >= t ? hi - i1 : i1;
                x[n1] = make_float2(wav[i0] * w.x, wav[i1] * w.y);
            }
        }
        fft16(x);
        // twiddle W_1024^(lane k1) = exp(-2 pi i 2 lane k1 / 2048), then rows [k1][n2 = lane] into the patch
        patch[lane] = x[0];
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) patch[k1 * 68 + lane] = cmul(x[k1], tw2048(tw, 2 * lane * k1));
        // ---- B: lane (k1q, cq) takes row k1q, columns 4 m + cq
#pragma unroll
        for (int m = 0; m < 16; ++m) x[m] = patch[k1q * 68 + 4 * m + cq];
        fft16(x);
        // twiddle W_64^(cq k) = exp(-2 pi i 32 cq k / 2048)
#pragma unroll
        for (int k = 1; k < 16; ++k) x[k] = cmul(x[k], tw2048(tw, 32 * cq * k));
        // ---- C: 4-point DFT over the quad (lanes cq = 0..3 hold a0..a3): after two xor exchanges lane cq holds X[k2q]
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float2 v = x[k];
            const float2 o2 = make_float2(quad_xor2(v.x), quad_xor2(v.y));
            float2 s = (cq & 2) ? csub(o2, v) : cadd(v, o2);          // lanes 0,1: a_c + a_{c+2}; lanes 2,3: a_{c-2} - a_c
            if (cq == 3) s = mul_mi(s);                                // -i (a1 - a3)
            const float2 o1 = make_float2(quad_xor1(s.x), quad_xor1(s.y));
            x[k] = (cq & 1) ? csub(o1, s) : cadd(s, o1);              // even lane: t + t'; odd lane: t' - t (= t_even - t_odd)
        }
        // ---- spectrum Z[k1q + 16 k + 256 k2q] to the patch in natural order (+ 8 pad per 256: conflict-free)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = k1q + 16 * k + 256 * k2q;
            patch[idx + 8 * k2q] = x[k];
        }
        // ---- unpack the real transform: lane pairs bins (k, 1024 - k), k = lane + 64 i, i = 0..8 (k <= 512)
        float mg_lo[9], mg_hi[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int k = lane + 64 * i;
            mg_lo[i] = 0.f; mg_hi[i] = 0.f;
            if (k <= 512) {
                const int kr = (NC - k) & (NC - 1);
                const float2 zk = patch[k + 8 * (k >> 8)], zr = patch[kr + 8 * (kr >> 8)];
                // even part E = (zk + conj(zr)) / 2, odd part O = (zk - conj(zr)) / (2i); X[k] = E + W^k O,
                // X[1024 - k] = conj(E) - conj(W^k O) ... evaluated directly from the swapped pair below
                const float2 e = make_float2(0.5f * (zk.x + zr.x), 0.5f * (zk.y - zr.y));
                const float2 o = make_float2(0.5f * (zk.y + zr.y), -0.5f * (zk.x - zr.x));
                const float2 wo = cmul(tw[k], o);
                const float re = e.x + wo.x, im = e.y + wo.y;
                // v_sqrt_f32 (1 ulp): the IEEE-exact sqrtf expands to a dozen instructions and three lane masks each -- eighteen of
                // them unrolled were the kernel's 84 spilled SGPRs
                mg_lo[i] = __builtin_amdgcn_sqrtf(re * re + im * im);
                // bin 1024 - k: E' = conj(E), O' = conj(O), W^(1024 - k) = -conj(W^k)  =>  X[1024 - k] = conj(E - W^k O)
                const float re2 = e.x - wo.x, im2 = e.y - wo.y;
                mg_hi[i] = __builtin_amdgcn_sqrtf(re2 * re2 + im2 * im2);
            }
        }
        // every lane has read its spectrum values: the magnitudes overwrite the patch (1025 floats)
        float* mag = reinterpret_cast<float*>(patch);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int k = lane + 64 * i;
            if (k <= 512) {
                mag[k] = mg_lo[i];
                mag[NC - k] = mg_hi[i];
            }
        }
        if (MEL) {
            for (int m = lane; m < a.n_mel; m += 64) {
                const int s = a.mel_start[m], len = a.mel_len[m];
                const float* wcol = a.mel_w + m;
                const float* mg = mag + s;
                float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
                int j = 0;
                for (; j + 3 < len; j += 4) {
                    const float w0 = wcol[(long)j * a.n_mel], w1 = wcol[(long)(j + 1) * a.n_mel];
                    const float w2 = wcol[(long)(j + 2) * a.n_mel], w3 = wcol[(long)(j + 3) * a.n_mel];
                    acc0 = fmaf(w0, mg[j], acc0);
                    acc1 = fmaf(w1, mg[j + 1], acc1);
                    acc2 = fmaf(w2, mg[j + 2], acc2);
                    acc3 = fmaf(w3, mg[j + 3], acc3);
                }
                for (; j < len; ++j) acc0 = fmaf(wcol[(long)j * a.n_mel], mg[j], acc0);
                tile[m * tile_ld + fl] = logf((acc0 + acc1) + (acc2 + acc3) + a.log_eps);
            }
        } else {
            for (int k = lane; k <= NC; k += 64) tile[k * tile_ld + fl] = a.apply_log ? logf(mag[k] + a.log_eps) : mag[k];
        }
    }
    __syncthreads();
    // ---- coalesced store of the tile: rows = features, contiguous along frames
    const int fvalid = min(a.fg, a.frames - f_base);
    float* dst_g = a.out + (long)clip * a.out_n_stride;
    const int total = a.n_mel * fvalid;
    for (int i = tid; i < total; i += kThreads) {
        const int row = i / fvalid, fl = i - row * fvalid;
        dst_g[(long)row * a.frames + f_base + fl] = tile[row * tile_ld + fl];
    }
    if (a.freq_channel) {
        float* fq = dst_g + (long)a.n_mel * a.frames;
        const int h = a.n_mel;
        const float step = 2.0f / (float)(h - 1);
        for (int i = tid; i < total; i += kThreads) {
            const int row = i / fvalid, fl = i - row * fvalid;
            const float v = (row < h / 2) ? (-1.0f + step * (float)row)
                                          : (1.0f - step * (float)(h - 1 - row));
            fq[(long)row * a.frames + f_base + fl] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------
// n_fft = 256, magnitude / log-magnitude output (the 1-d model's stft_256_* features): EIGHT LANES per frame, eight frames per wave.
// The 128-point complex FFT of the packed frame is 16 x 8 (decimation in time):
//   A  lane L of a frame holds z[8 n1 + L], n1 = 0..15: a 16-point FFT in registers, twiddle W_128^(L k1);
//   B  transpose through the wave's LDS patch: lane j takes rows k1 = 2 j, 2 j + 1 (8 columns each): two 8-point FFTs in registers;
//   C  the spectrum returns to the patch in natural order and lane L unpacks bins L + 8 m of the real transform into the
//      workgroup's [bin][frame] tile, which leaves as 128-byte runs along the frame axis.
// Three LDS round trips per EIGHT frames of a wave; the generic kernel above takes seven per frame (one wave per frame, four
// butterfly passes with half of the lanes idle): 0.47 ms for the 453 MB of cfg 3 (tools/frontend256_bench.py).
constexpr int kF8Frames = 32;                     // frames per workgroup: 4 waves x 8
constexpr int kF8Stride = 272;                    // floats per frame in a patch: 128 complex + 16 (frames of a lane group on disjoint banks)
constexpr int kF8TileLd = 36;                     // tile row stride: lanes (frame, bin + 1) four banks apart

// forward 8-point DFT in registers, natural order in and out
__device__ __forceinline__ void fft8(float2 (&v)[8]) {
    constexpr float h = 0.70710678118654752f;
    float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    o1 = cmul(o1, make_float2(h, -h));
    o2 = mul_mi(o2);
    o3 = cmul(o3, make_float2(-h, -h));
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

__global__ __launch_bounds__(kThreads) void frontend256_kernel(FrontendArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = 128;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fw = lane >> 3, L = lane & 7;
    const int clip = blockIdx.y;
    const int f_base = blockIdx.x * kF8Frames;
    float2* tw = reinterpret_cast<float2*>(smem);                          // exp(-2 pi i m / 256), m = 0..255
    float* win = smem + 512;                                               // Hann window, 256
    float* patch = smem + 768 + (wid * 8 + fw) * kF8Stride;                // this frame's 128 complex
    float* tile = smem + 768;                                              // 129 x kF8TileLd, over the patches once they are done
    tw[tid] = reinterpret_cast<const float2*>(a.tables + 256)[tid];
    win[tid] = a.tables[tid];
    __syncthreads();

    const float* wav = a.wave + (long)clip * a.wave_stride;
    const int t = a.t;
    const int fl = wid * 8 + fw;
    const int f = f_base + fl;
    if (f < a.frames) {
        // ---- A: load + window + pack: z[n] = (x[2n] w[2n], x[2n+1] w[2n+1]), n = 8 n1 + L
        float2 x[16];
        const long s0 = (long)f * a.hop - NC;
        const bool pairs = s0 >= 0 && s0 + 2 * NC <= t && ((s0 | a.wave_stride) & 1) == 0 &&
                           (reinterpret_cast<uintptr_t>(a.wave) & 7) == 0;
        if (pairs) {
            const float2* src = reinterpret_cast<const float2*>(wav + s0) + L;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const float2 w = reinterpret_cast<const float2*>(win)[8 * n1 + L];
                const float2 v = src[8 * n1];
                x[n1] = make_float2(v.x * w.x, v.y * w.y);
            }
        } else {
            // reflect padding without branches (ops/utils.py:110-127: center=True, pad_mode="reflect"; t > n_fft / 2)
            const int base = (int)s0 + 2 * L, hi = 2 * (t - 1);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const float2 w = reinterpret_cast<const float2*>(win)[8 * n1 + L];
                int i0 = base + 16 * n1, i1 = i0 + 1;
                i0 = i0 < 0 ? -i0 : i0; i1 = i1 < 0 ? -i1 : i1;
                i0 = i0 >= t ? hi - i0 : i0; i1 = i1 >= t ? hi - i1 : i1;
                x[n1] = make_float2(wav[i0] * w.x, wav[i1] * w.y);
            }
        }
        fft16(x);
        // twiddle W_128^(L k1) = exp(-2 pi i 2 L k1 / 256); rows [k1][L] into the patch
        float2* pc = reinterpret_cast<float2*>(patch);
        pc[L] = x[0];
#pragma unroll
        for (int k1 = 1; k1 < 16; ++k1) pc[k1 * 8 + L] = cmul(x[k1], tw[2 * L * k1]);
    }
    // a frame's patch belongs to its eight lanes: the wave's own LDS queue orders the passes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (f < a.frames) {
        // ---- B: lane j = L takes rows k1 = 2 L, 2 L + 1: 16 consecutive complex of the patch
        float2 r0[8], r1[8];
        const float4* p4 = reinterpret_cast<const float4*>(patch + 32 * L);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 u = p4[q], v = p4[4 + q];
            r0[2 * q] = make_float2(u.x, u.y); r0[2 * q + 1] = make_float2(u.z, u.w);
            r1[2 * q] = make_float2(v.x, v.y); r1[2 * q + 1] = make_float2(v.z, v.w);
        }
        fft8(r0);
        fft8(r1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();                   // every lane of the frame has read its rows
        // ---- C: Z[k1 + 16 k2] in natural order: (k1, k1 + 1) = (2 L, 2 L + 1) are neighbours
        float4* o4 = reinterpret_cast<float4*>(patch + 4 * L);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) o4[8 * k2] = make_float4(r0[k2].x, r0[k2].y, r1[k2].x, r1[k2].y);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float val[17];
    if (f < a.frames) {
        // ---- unpack the real transform: bins k = L + 8 m (m = 0..15) and bin 128 (lane 0 of the frame; the others repeat bin L)
        const float2* pc = reinterpret_cast<const float2*>(patch);
#pragma unroll
        for (int m = 0; m <= 16; ++m) {
            const int k = (m == 16 && L != 0) ? L : L + 8 * m;
            const float2 zk = pc[k & (NC - 1)], zr = pc[(NC - k) & (NC - 1)];
            const float2 e = make_float2(0.5f * (zk.x + zr.x), 0.5f * (zk.y - zr.y));
            const float2 o = make_float2(0.5f * (zk.y + zr.y), -0.5f * (zk.x - zr.x));
            const float2 wo = cmul(tw[k], o);
            const float re = e.x + wo.x, im = e.y + wo.y;
            const float mg = __builtin_amdgcn_sqrtf(re * re + im * im);
            val[m] = a.apply_log ? logf(mg + a.log_eps) : mg;
        }
    }
    __syncthreads();                                        // every wave is through with its patches: the tile takes their place
    if (f < a.frames) {
#pragma unroll
        for (int m = 0; m < 16; ++m) tile[(L + 8 * m) * kF8TileLd + fl] = val[m];
        if (L == 0) tile[NC * kF8TileLd + fl] = val[16];
    }
    __syncthreads();
    // ---- coalesced store of the tile: rows = bins, 32 consecutive frames each
    const int fvalid = min(kF8Frames, a.frames - f_base);
    float* dst_g = a.out + (long)clip * a.out_n_stride;
    for (int i = tid; i < (NC + 1) * kF8Frames; i += kThreads) {
        const int row = i >> 5, c = i & 31;
        if (c < fvalid) dst_g[(long)row * a.frames + f_base + c] = tile[row * kF8TileLd + c];
    }
    if (a.freq_channel) {
        float* fq = dst_g + (long)(NC + 1) * a.frames;
        const float step = 2.0f / (float)NC;
        for (int i = tid; i < (NC + 1) * kF8Frames; i += kThreads) {
            const int row = i >> 5, c = i & 31;
            const float v = (row < (NC + 1) / 2) ? (-1.0f + step * (float)row) : (1.0f - step * (float)(NC - row));
            if (c < fvalid) fq[(long)row * a.frames + f_base + c] = v;
        }
    }
}

int launch256(const FrontendArgs& base, int n, hipStream_t stream) {
    FrontendArgs a = base;
    a.fg = kF8Frames;
    static_assert((kThreads / 8) * kF8Stride >= 129 * kF8TileLd, "the tile reuses the patches");
    const size_t lds = sizeof(float) * ((size_t)768 + (size_t)(kThreads / 8) * kF8Stride);
    dim3 grid(fsc::ceil_div(a.frames, kF8Frames), n);
    hipLaunchKernelGGL(frontend256_kernel, grid, dim3(kThreads), lds, stream, a);
    FSC_LAUNCH_CHECK("fsc_frontend(256)");
    return 0;
}

int launch2048(const FrontendArgs& base, bool mel, int n, hipStream_t stream) {
    FrontendArgs a = base;
    a.fg = mel ? kF2Frames : 8;
    const size_t lds = sizeof(float) * ((size_t)2 * 1032 + (size_t)(kThreads / 64) * 2 * kF2Patch + (size_t)a.n_mel * (a.fg + 1));
    if (lds > 160 * 1024) return -1;
    dim3 grid(fsc::ceil_div(a.frames, a.fg), n);
    if (mel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&frontend2048_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(frontend2048_kernel<true>, grid, dim3(kThreads), lds, stream, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&frontend2048_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(frontend2048_kernel<false>, grid, dim3(kThreads), lds, stream, a);
    }
    FSC_LAUNCH_CHECK("fsc_frontend(2048)");
    return 0;
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

int launch2048(const FrontendArgs& base, bool mel, int n, hipStream_t stream);
int launch256(const FrontendArgs& base, int n, hipStream_t stream);

int launch(const FrontendArgs& base, bool mel, int n, hipStream_t stream) {
    if (base.n_fft == 2048 && !fsc::env().frontend_generic) {
        const int rc = launch2048(base, mel, n, stream);
        if (rc >= 0) return rc;
    }
    if (base.n_fft == 256 && !mel && base.t > 128 && !fsc::env().frontend_generic) return launch256(base, n, stream);
    FrontendArgs a = base;
    const int nc = a.n_fft / 2;
    int tpf = nc / 8;            // two butterflies per thread and pass: two frames share the barriers of a workgroup
#ifndef FSC_FE_TPF_MIN
#define FSC_FE_TPF_MIN 64
#endif
    if (tpf < FSC_FE_TPF_MIN) tpf = FSC_FE_TPF_MIN;
    if (tpf > kThreads) tpf = kThreads;
    a.tpf = tpf;
    a.block_sync = fsc::env().fe_block_sync ? 1 : 0;
    const int fpb = kThreads / tpf;
    int fg = 32;
    auto lds_bytes = [&](int g) {
        return sizeof(float) * ((size_t)2 * a.n_fft + (size_t)fpb * 4 * nc + (size_t)a.n_mel * (g + 1));
    };
    while (fg > fpb && lds_bytes(fg) > 64 * 1024) fg >>= 1;
    if (fg < fpb) fg = fpb;
    const size_t lds = lds_bytes(fg);
    FSC_CHECK_ARG(lds <= 160 * 1024, "frontend: LDS tile of %zu bytes does not fit (n_fft=%d, features=%d)",
                  lds, a.n_fft, a.n_mel);
    a.fg = fg;
    dim3 grid(fsc::ceil_div(a.frames, fg), n);
    if (mel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&frontend_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(frontend_kernel<true>, grid, dim3(kThreads), lds, stream, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&frontend_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(frontend_kernel<false>, grid, dim3(kThreads), lds, stream, a);
    }
    FSC_LAUNCH_CHECK("fsc_frontend");
    return 0;
}

}  // namespace

extern "C" {

size_t fsc_frontend_table_floats(int n_fft) { return (size_t)3 * n_fft; }

int fsc_frontend_tables_init(float* tables, int n_fft, fsc_stream_t stream) {
    FSC_CHECK_ARG(tables && pow2(n_fft) && n_fft >= 64 && n_fft <= 4096,
                  "fsc_frontend_tables_init: n_fft=%d must be a power of two in [64, 4096]", n_fft);
    hipLaunchKernelGGL(tables_kernel, dim3(fsc::ceil_div(n_fft, 256)), dim3(256), 0,
                       fsc::as_stream(stream), tables, n_fft);
    FSC_LAUNCH_CHECK("fsc_frontend_tables_init");
    return 0;
}

int fsc_frontend_logmel_fwd(const float* wave, int n, int t, long wave_stride, int n_fft, int hop,
                            const float* tables, const int* mel_start, const int* mel_len,
                            const float* mel_w, int n_mel, int max_band, float log_eps, float* out,
                            long out_n_stride, int freq_channel, fsc_stream_t stream) {
    FSC_CHECK_ARG(wave && tables && mel_start && mel_len && mel_w && out, "fsc_frontend_logmel_fwd: null pointer");
    FSC_CHECK_ARG(pow2(n_fft) && n_fft >= 64 && n_fft <= 4096, "fsc_frontend_logmel_fwd: bad n_fft %d", n_fft);
    FSC_CHECK_ARG(hop > 0 && n > 0 && n_mel > 0 && max_band >= 0, "fsc_frontend_logmel_fwd: bad sizes");
    FSC_CHECK_ARG(t > n_fft / 2, "fsc_frontend_logmel_fwd: reflect padding needs T (%d) > n_fft/2 (%d)", t, n_fft / 2);
    FrontendArgs a{};
    a.wave = wave; a.wave_stride = wave_stride; a.t = t; a.n_fft = n_fft; a.hop = hop;
    a.frames = 1 + t / hop; a.tables = tables; a.mel_start = mel_start; a.mel_len = mel_len;
    a.mel_w = mel_w; a.n_mel = n_mel; a.log_eps = log_eps; a.apply_log = 1; a.out = out;
    a.out_n_stride = out_n_stride; a.freq_channel = freq_channel;
    return launch(a, true, n, fsc::as_stream(stream));
}

int fsc_frontend_stft_fwd(const float* wave, int n, int t, long wave_stride, int n_fft, int hop,
                          const float* tables, int apply_log, float log_eps, float* out,
                          long out_n_stride, int freq_channel, fsc_stream_t stream) {
    FSC_CHECK_ARG(wave && tables && out, "fsc_frontend_stft_fwd: null pointer");
    FSC_CHECK_ARG(pow2(n_fft) && n_fft >= 64 && n_fft <= 4096, "fsc_frontend_stft_fwd: bad n_fft %d", n_fft);
    FSC_CHECK_ARG(hop > 0 && n > 0, "fsc_frontend_stft_fwd: bad sizes");
    FSC_CHECK_ARG(t > n_fft / 2, "fsc_frontend_stft_fwd: reflect padding needs T (%d) > n_fft/2 (%d)", t, n_fft / 2);
    FrontendArgs a{};
    a.wave = wave; a.wave_stride = wave_stride; a.t = t; a.n_fft = n_fft; a.hop = hop;
    a.frames = 1 + t / hop; a.tables = tables; a.n_mel = n_fft / 2 + 1; a.log_eps = log_eps;
    a.apply_log = apply_log; a.out = out; a.out_n_stride = out_n_stride; a.freq_channel = freq_channel;
    return launch(a, false, n, fsc::as_stream(stream));
}

}  // extern "C"
