// Weight gradient of the stem convolution (3x3, c_in <= 2; reference networks/classifiers.py:526-532) taken directly from
// the gradient at the POOLED resolution and the max-pool's arg-max indices:
//
//     dW[co][ci][ty][tx] = sum over windows (n, oy, ox)  d[n][co][oy][ox] * a[n][ci][2 oy + ry + ty - 1][2 ox + rx + tx - 1]
//
// where (ry, rx) is the window's arg-max position.  The un-pooled gradient dc (2.8 GB at cfg 2) has exactly one non-zero
// per 2x2 window; materialising it cost a 0.98 ms un-pool pass and a 0.95 ms dense weight-gradient pass.  Here each lane
// owns pooled columns, gathers its 18 inputs from an LDS tile of the (tiny) conv input and accumulates the 18 sums of one
// output channel; a wave walks its channels one after the other.  The same pass yields the eight border sums of dc that
// functional._stem_bn_grads needs (first / last row, first / last column, corners).
// Output: per (block, channel) 32 floats [18 weight-gradient sums | 8 border sums | pad], summed over blocks by
// fsc_conv_stem_grads_finish, which also turns them into the parameter gradients of the BatchNorm in front of the stem.
// With (mean, invstd) given the kernel reads the BatchNorm's INPUT x and correlates the gradient with xhat = (x - mean) invstd
// (zero outside the image, like the padded conv input): dW' = sum dc xhat.  Then, with a = gamma xhat + beta the conv input,
//     dW = gamma dW' + beta T          dgamma = sum da xhat = sum_{co,tap} w dW'          dbeta = sum da = sum_{co,tap} w T
// (T[co][tap] = sum of dc[co] over the pixels whose tap neighbour lies inside the image) -- no division by gamma anywhere.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTileRows = 4;            // pooled rows per block

// Sum over the 64 lanes as DPP moves inside the rows of 16 (xor 1, xor 2, half-row mirror, row mirror) and four v_readlane across
// the rows: 26 sums per (wave, channel) through __shfl_xor -- six ds_bpermute round trips each -- cost as much as the FMAs they fold.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);         // row_half_mirror
    v += dpp_mov<0x140>(v);         // row_mirror: every lane holds its row's total
    const int b = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}

template <int CIN>
__global__ __launch_bounds__(kThreads) void stem_wgrad_pooled_kernel(const float* __restrict__ a, const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd, const float* __restrict__ dp,
                                                                      const uint8_t* __restrict__ idx, float* __restrict__ part,
                                                                      int cout, int h, int w, int oh, int ow, int wp) {
    extern __shared__ __attribute__((aligned(16))) float atile[];      // [CIN][2 * kTileRows + 2][wp]
    constexpr int TROWS = 2 * kTileRows + 2;
    const int n = blockIdx.y, r0 = blockIdx.x * kTileRows;            // first pooled row of the tile
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // ---- the conv input rows 2 r0 - 1 .. 2 r0 + 2 kTileRows, columns -1 .. w (zero outside the image)
    // A tile row is stored de-interleaved: even tile columns in [0, wh), odd ones in [wh, 2 wh) with wh = 32 (mod 64) -- lane ox
    // reads tile column 2 ox + rx + tx, i.e. element ox + ((rx + tx) >> 1) of the even or the odd half: consecutive lanes hit
    // consecutive banks and the two halves sit 32 banks apart (interleaved, the stride-2 gather was a 2-way bank conflict).
    const int wh = wp >> 1;
    float sc[CIN], sh[CIN];                 // input affine: identity, or x -> xhat
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
        sc[ci] = mean ? invstd[ci] : 1.f;
        sh[ci] = mean ? -mean[ci] * invstd[ci] : 0.f;
    }
    for (int i = threadIdx.x; i < CIN * TROWS * wp; i += kThreads) {
        const int ci = i / (TROWS * wp), rem = i - ci * (TROWS * wp);
        const int tr = rem / wp, tc = rem - tr * wp;
        const int y = 2 * r0 - 1 + tr, x = tc - 1;
        const float s_ = (CIN == 1 || ci == 0) ? sc[0] : sc[CIN - 1], t_ = (CIN == 1 || ci == 0) ? sh[0] : sh[CIN - 1];
        atile[(ci * TROWS + tr) * wp + (tc & 1) * wh + (tc >> 1)] =
            (y >= 0 && y < h && x >= 0 && x < w) ? fmaf(a[((long)(n * CIN + ci) * h + y) * w + x], s_, t_) : 0.f;
    }
    __syncthreads();
    const int rows = min(kTileRows, oh - r0);
    for (int co = wv; co < cout; co += kThreads / 64) {
        float acc[CIN * 9], bs[8];
#pragma unroll
        for (int k = 0; k < CIN * 9; ++k) acc[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) bs[k] = 0.f;
        const long plane = ((long)n * cout + co) * oh;
        // a pooled row = up to kChunks x 64 columns: all of a row's loads are issued before its arithmetic, and the next
        // row's loads before this row's (the loop is latency-bound otherwise: two dependent 4-byte loads per 18 FMAs)
        constexpr int kChunks = 4;
        float dv[2][kChunks];
        int pv[2][kChunks];
        auto load_row = [&](int pr, int buf) {
            const long rowbase = (plane + r0 + pr) * ow;
#pragma unroll
            for (int cch = 0; cch < kChunks; ++cch) {
                const int ox = cch * 64 + lane;
                const bool live = pr < rows && ox < ow;
                dv[buf][cch] = live ? dp[rowbase + ox] : 0.f;
                pv[buf][cch] = live ? (int)idx[rowbase + ox] : 0;
            }
        };
        auto row_math = [&](int pr, int buf, int cch0) {
#pragma unroll
            for (int cch = 0; cch < kChunks; ++cch) {
                const int ox = (cch0 + cch) * 64 + lane;
                const float d = dv[buf][cch];                               // (0 for dead lanes: contributes nothing)
                const int pos = pv[buf][cch];
                const int ry = pos >> 1, rx = pos & 1;
                const int y = 2 * pr + ry, x = ox < ow ? 2 * ox + rx : 0;   // tile row of tap ty = 0 / column of tap tx = 0
                const int oxs = ox < ow ? ox : 0;
                // tile column x + tx = 2 oxs + rx + tx -> half (rx + tx) & 1, element oxs + ((rx + tx) >> 1)
                const float* t0 = atile + y * wp + oxs + (rx ? wh : 0);            // tx = 0
                const float* t1 = atile + y * wp + oxs + (rx ? 1 : wh);            // tx = 1
                const float* t2 = atile + y * wp + oxs + (rx ? wh + 1 : 1);        // tx = 2
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty) {
                        const int o = (ci * TROWS + ty) * wp;
                        acc[(ci * 3 + ty) * 3 + 0] = fmaf(d, t0[o], acc[(ci * 3 + ty) * 3 + 0]);
                        acc[(ci * 3 + ty) * 3 + 1] = fmaf(d, t1[o], acc[(ci * 3 + ty) * 3 + 1]);
                        acc[(ci * 3 + ty) * 3 + 2] = fmaf(d, t2[o], acc[(ci * 3 + ty) * 3 + 2]);
                    }
                // border sums: only the first / last pooled row and the chunks that hold the first / last pooled column can
                // contribute (wave-uniform test)
                const int gy = 2 * (r0 + pr) + ry;                           // position of the non-zero in the un-pooled gradient
                const int c_abs = cch0 + cch;
                if (r0 + pr == 0 || 2 * (r0 + pr) + 1 >= h - 1 || c_abs == 0 || c_abs == ((ow - 1) >> 6)) {
                    const bool top = gy == 0, bot = gy == h - 1, lef = x == 0, rig = x == w - 1;
                    bs[0] += top ? d : 0.f; bs[1] += bot ? d : 0.f; bs[2] += lef ? d : 0.f; bs[3] += rig ? d : 0.f;
                    bs[4] += (top && lef) ? d : 0.f; bs[5] += (top && rig) ? d : 0.f;
                    bs[6] += (bot && lef) ? d : 0.f; bs[7] += (bot && rig) ? d : 0.f;
                }
            }
        };
        if (ow <= kChunks * 64) {
            load_row(0, 0);
#pragma unroll 1
            for (int pr = 0; pr < rows; pr += 2) {
                load_row(pr + 1, 1);
                row_math(pr, 0, 0);
                load_row(pr + 2, 0);
                if (pr + 1 < rows) row_math(pr + 1, 1, 0);
            }
        } else {                                                        // wide images: chunk groups one after the other
            for (int pr = 0; pr < rows; ++pr)
                for (int c0 = 0; c0 * 64 < ow; c0 += kChunks) {
                    const long rowbase = (plane + r0 + pr) * ow;
#pragma unroll
                    for (int cch = 0; cch < kChunks; ++cch) {
                        const int ox = (c0 + cch) * 64 + lane;
                        dv[0][cch] = ox < ow ? dp[rowbase + ox] : 0.f;
                        pv[0][cch] = ox < ow ? (int)idx[rowbase + ox] : 0;
                    }
                    row_math(pr, 0, c0);
                }
        }
        float* o = part + (((long)blockIdx.y * gridDim.x + blockIdx.x) * cout + co) * 32;
#pragma unroll
        for (int k = 0; k < CIN * 9; ++k) {
            const float v = wave_sum_dpp(acc[k]);
            if (lane == 0) o[k] = v;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = wave_sum_dpp(bs[k]);
            if (lane == 0) o[18 + k] = v;
        }
        if (lane < 32 && (lane >= 26 || (lane >= CIN * 9 && lane < 18))) o[lane] = 0.f;
    }
}

// Sums the partial rows of one output channel over the blocks and finishes what can be finished per channel: the weight gradient
// (xhat mode: dW = gamma dW' + beta T), the border sums, and this channel's terms of the BatchNorm parameter gradients, which it
// leaves as DOUBLES in floats 16 .. 16 + 4 CIN of its own row of block 0 ([sum_tap w T | sum_tap w dW(')] per input channel; the
// row's own values are in shared memory by then).  Round 5: everything behind the per-block partial rows is summed in fp64 --
// dgamma = sum_{co, tap} w dW' is a cancelling sum of 900 terms that are themselves sums over millions of positions, and the fp32
// reductions here (2048 rows, then 9 taps, then 100 channels) left it 2.9e-3 of its scale from an fp64 evaluation against 4.6e-4 for
// the CPU's direct fp32 sum (tests/test_cfg2_gpu.py, stage 0).
template <int CIN>
__global__ __launch_bounds__(256) void stem_grads_reduce_kernel(float* __restrict__ part, int blocks, int cout,
                                                                 const float* __restrict__ weight, const float* __restrict__ chan_sum,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, int xhat,
                                                                 float* __restrict__ dweight, float* __restrict__ borders) {
    __shared__ double red[8][32];
    const int co = blockIdx.x, k = threadIdx.x & 31, g = threadIdx.x >> 5;
    double acc = 0.0;
    for (int b = g; b < blocks; b += 8) acc += (double)part[((long)b * cout + co) * 32 + k];
    red[g][k] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][k];
        red[0][k] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double* tot = red[0];
        const double s = (double)chan_sum[co];
        double T[9];
        for (int ty = 0; ty < 3; ++ty)
            for (int tx = 0; tx < 3; ++tx) {
                double v = s;
                if (ty == 0) v -= tot[18 + 0];
                if (ty == 2) v -= tot[18 + 1];
                if (tx == 0) v -= tot[18 + 2];
                if (tx == 2) v -= tot[18 + 3];
                if (ty == 0 && tx == 0) v += tot[18 + 4];
                if (ty == 0 && tx == 2) v += tot[18 + 5];
                if (ty == 2 && tx == 0) v += tot[18 + 6];
                if (ty == 2 && tx == 2) v += tot[18 + 7];
                T[ty * 3 + tx] = v;
            }
        double* scratch = reinterpret_cast<double*>(part + (long)co * 32 + 16);
        for (int ci = 0; ci < CIN; ++ci) {
            double cb = 0.0, cs = 0.0;
            for (int tap = 0; tap < 9; ++tap) {
                const double wv = (double)weight[((long)co * CIN + ci) * 9 + tap], dwx = tot[ci * 9 + tap];
                cb += wv * T[tap];
                cs += wv * dwx;
                dweight[((long)co * CIN + ci) * 9 + tap] = (float)(xhat ? (double)gamma[ci] * dwx + (double)beta[ci] * T[tap] : dwx);
            }
            scratch[2 * ci] = cb;
            scratch[2 * ci + 1] = cs;
        }
        if (borders)
            for (int i = 0; i < 8; ++i) borders[co * 8 + i] = (float)tot[18 + i];
    }
}

// dbeta[ci] = sum_co cb, dgamma[ci] = sum_co cs (xhat mode) or (sum_co cs - beta dbeta) / gamma (the conv input itself was
// correlated: singular at gamma = 0, callers guard it -- functional._gamma_ok).
template <int CIN>
__global__ __launch_bounds__(256) void stem_bn_finish_kernel(const float* __restrict__ part, int cout, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int xhat, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
    __shared__ double scratch[4];
    double v[2 * CIN];
#pragma unroll
    for (int i = 0; i < 2 * CIN; ++i) v[i] = 0.0;
    for (int co = threadIdx.x; co < cout; co += 256)
#pragma unroll
        for (int i = 0; i < 2 * CIN; ++i) v[i] += reinterpret_cast<const double*>(part + (long)co * 32 + 16)[i];
#pragma unroll
    for (int i = 0; i < 2 * CIN; ++i) v[i] = fsc::block_sum<double, 4>(v[i], scratch);
    if (threadIdx.x == 0)
        for (int ci = 0; ci < CIN; ++ci) {
            const double db = v[2 * ci], x = v[2 * ci + 1];
            dbeta[ci] = (float)db;
            dgamma[ci] = (float)(xhat ? x : (x - (double)beta[ci] * db) / (double)gamma[ci]);
        }
}

// tile row pitch: >= w + 2, even, half of it = 32 (mod 64) floats (the two de-interleaved halves 32 banks apart)
int tile_pitch(int w) {
    int wh = (w + 2 + 1) / 2;
    wh += (32 - (wh & 63) + 64) & 63;
    return 2 * wh;
}

bool supported(const fsc_conv_desc* d) {
    return d && d->kh == 3 && d->kw == 3 && d->c_in >= 1 && d->c_in <= 2 && d->h >= 2 && d->w >= 8 && d->n > 0 && d->c_out > 0 &&
           (size_t)d->c_in * (2 * kTileRows + 2) * tile_pitch(d->w) * sizeof(float) <= 80 * 1024;
}

}  // namespace

extern "C" {

/* blocks (= rows of the partial buffer) of fsc_conv_stem_wgrad_pooled for this shape; 0 = unsupported */
size_t fsc_conv_stem_wgrad_pooled_blocks(const fsc_conv_desc* d) {
    if (!supported(d)) return 0;
    return (size_t)fsc::ceil_div(d->h / 2, kTileRows) * d->n;
}

int fsc_conv_stem_wgrad_pooled(const fsc_conv_desc* d, const float* in, const float* in_mean, const float* in_invstd,
                               const float* dpooled, const uint8_t* pool_idx, float* partial, fsc_stream_t stream) {
    FSC_CHECK_ARG(supported(d) && in && dpooled && pool_idx && partial, "fsc_conv_stem_wgrad_pooled: unsupported shape or null pointer");
    FSC_CHECK_ARG((in_mean == nullptr) == (in_invstd == nullptr), "fsc_conv_stem_wgrad_pooled: in_mean and in_invstd come together");
    const int oh = d->h / 2, ow = d->w / 2;
    const int wp = tile_pitch(d->w);
    dim3 grid(fsc::ceil_div(oh, kTileRows), d->n);
    const size_t lds = sizeof(float) * (size_t)d->c_in * (2 * kTileRows + 2) * wp;
    hipStream_t st = fsc::as_stream(stream);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad_pooled_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad_pooled_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (d->c_in == 1)
        hipLaunchKernelGGL(stem_wgrad_pooled_kernel<1>, grid, dim3(kThreads), lds, st, in, in_mean, in_invstd, dpooled, pool_idx, partial, d->c_out, d->h, d->w, oh, ow, wp);
    else
        hipLaunchKernelGGL(stem_wgrad_pooled_kernel<2>, grid, dim3(kThreads), lds, st, in, in_mean, in_invstd, dpooled, pool_idx, partial, d->c_out, d->h, d->w, oh, ow, wp);
    FSC_LAUNCH_CHECK("fsc_conv_stem_wgrad_pooled");
    return 0;
}

int fsc_conv_stem_grads_finish(const fsc_conv_desc* d, float* partial, const float* weight, const float* chan_sum,
                               const float* gamma, const float* beta, int xhat, float* dweight, float* borders, float* dgamma,
                               float* dbeta, fsc_stream_t stream) {
    FSC_CHECK_ARG(supported(d) && partial && weight && chan_sum && dweight, "fsc_conv_stem_grads_finish: unsupported shape or null pointer");
    FSC_CHECK_ARG(!xhat || (gamma && beta), "fsc_conv_stem_grads_finish: xhat mode needs gamma and beta");
    FSC_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr) && (!dgamma || (gamma && beta)),
                  "fsc_conv_stem_grads_finish: dgamma and dbeta come together and need gamma and beta");
    const int blocks = (int)fsc_conv_stem_wgrad_pooled_blocks(d);
    hipStream_t st = fsc::as_stream(stream);
    if (d->c_in == 1) {
        hipLaunchKernelGGL(stem_grads_reduce_kernel<1>, dim3(d->c_out), dim3(256), 0, st, partial, blocks, d->c_out, weight, chan_sum,
                           gamma, beta, xhat, dweight, borders);
        if (dgamma) hipLaunchKernelGGL(stem_bn_finish_kernel<1>, dim3(1), dim3(256), 0, st, partial, d->c_out, gamma, beta, xhat, dgamma, dbeta);
    } else {
        hipLaunchKernelGGL(stem_grads_reduce_kernel<2>, dim3(d->c_out), dim3(256), 0, st, partial, blocks, d->c_out, weight, chan_sum,
                           gamma, beta, xhat, dweight, borders);
        if (dgamma) hipLaunchKernelGGL(stem_bn_finish_kernel<2>, dim3(1), dim3(256), 0, st, partial, d->c_out, gamma, beta, xhat, dgamma, dbeta);
    }
    FSC_LAUNCH_CHECK("fsc_conv_stem_grads_finish");
    return 0;
}

}  // extern "C"
