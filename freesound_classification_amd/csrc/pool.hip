// Max-pooling kernels (HBM-bound streaming).
//   fsc_maxpool_*        : nn.MaxPool2d(2,2) / nn.MaxPool1d(2,2), floor mode
//                          (reference networks/classifiers.py:532, :155)
//   fsc_global_maxpool_* : nn.AdaptiveMaxPool2d(1) / 1d(1) deep-supervision heads
//                          (reference networks/classifiers.py:540,591 and :163,201)
// Tie-breaking follows ATen: the first maximum in row-major window order wins; NaN wins.
#include "common.h"

namespace {

constexpr int kThreads = 256;

// blockIdx.x = (n, c) plane, blockIdx.y strides over row groups; threads are split into (row lane,
// column lane) with a power-of-two column count: no integer division anywhere, short rows still
// fill the block.
__global__ __launch_bounds__(kThreads) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               uint8_t* __restrict__ idx, int h, int w, int ph,
                                                               int oh, int ow, int colp_log2) {
    const int colp = 1 << colp_log2, rows_per_block = kThreads >> colp_log2;
    const int tr = threadIdx.x >> colp_log2, tc = threadIdx.x & (colp - 1);
    const long pl = blockIdx.x;
    const float* px = x + pl * h * w;
    float* py = y + pl * oh * ow;
    uint8_t* pi = idx + pl * oh * ow;
    for (int oy = blockIdx.y * rows_per_block + tr; oy < oh; oy += gridDim.y * rows_per_block) {
        const float* p0 = px + (long)oy * ph * w;
        for (int ox = tc; ox < ow; ox += colp) {
            const float* p = p0 + 2 * ox;
            float best = p[0];
            int bi = 0;
            float v = p[1];
            if (v > best || v != v) { best = v; bi = 1; }
            if (ph == 2) {
                v = p[w];
                if ((v > best || v != v) && best == best) { best = v; bi = 2; }
                v = p[w + 1];
                if ((v > best || v != v) && best == best) { best = v; bi = 3; }
            }
            py[oy * ow + ox] = best;
            pi[oy * ow + ox] = (uint8_t)bi;
        }
    }
}

// Single-row planes with few windows (the 1-d model's late blocks: 61 k planes of 3 ... 215 windows): a thread per window over ALL
// planes.  (A workgroup per plane ran 3 ... 215 of its 256 threads: 8 - 19 us for 1.5 - 14 MB tensors, the smallest the slowest.)
__global__ __launch_bounds__(kThreads) void maxpool_rows_flat_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                     uint8_t* __restrict__ idx, int w, int ow, long total) {
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total) return;
    const long pl = i / ow;
    const int ox = (int)(i - pl * ow);
    const float* p = x + pl * w + 2 * ox;
    float best = p[0];
    int bi = 0;
    const float v = p[1];
    if (v > best || v != v) { best = v; bi = 1; }
    y[i] = best;
    idx[i] = (uint8_t)bi;
}

// gather form: every input element is written exactly once
__global__ __launch_bounds__(kThreads) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                               const uint8_t* __restrict__ idx,
                                                               float* __restrict__ dx, int h, int w, int ph, int oh,
                                                               int ow, int colp_log2) {
    const int colp = 1 << colp_log2, rows_per_block = kThreads >> colp_log2;
    const int tr = threadIdx.x >> colp_log2, tc = threadIdx.x & (colp - 1);
    const long pl = blockIdx.x;
    const float* pdy = dy + pl * oh * ow;
    const uint8_t* pi = idx + pl * oh * ow;
    float* pdx = dx + pl * h * w;
    for (int yy = blockIdx.y * rows_per_block + tr; yy < h; yy += gridDim.y * rows_per_block) {
        const int oy = ph == 2 ? (yy >> 1) : yy;
        const bool row_live = oy < oh;
        const int orow = oy * ow;
        const int rbit = ph == 2 ? ((yy & 1) << 1) : 0;
        for (int xx = tc; xx < w; xx += colp) {
            const int ox = xx >> 1;
            float g = 0.f;
            if (row_live && ox < ow && pi[orow + ox] == (rbit | (xx & 1))) g = pdy[orow + ox];
            pdx[yy * w + xx] = g;
        }
    }
}

// (value, index) packed into one sortable 64-bit key: larger value wins, NaN beats every number,
// equal values -> smaller index (ATen's first-maximum rule).
__device__ __forceinline__ unsigned long long gmax_key(float v, unsigned idx) {
    unsigned u;
    if (v != v) {
        u = 0xFFFFFFFFu;
    } else {
        if (v == 0.f) v = 0.f;              // -0 and +0 compare equal
        u = __float_as_uint(v);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

// one workgroup per (n, c) plane
__global__ __launch_bounds__(kThreads) void gmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            int* __restrict__ idx, long planes, long hw) {
    __shared__ unsigned long long sk[kThreads / 64];
    for (long pl = blockIdx.x; pl < planes; pl += gridDim.x) {
        const float* p = x + pl * hw;
        unsigned long long best = 0ull;     // below every real key (index field of a real key is > 0 only if idx < 2^32-1)
        for (long i = threadIdx.x; i < hw; i += kThreads) {
            const unsigned long long k = gmax_key(p[i], (unsigned)i);
            best = k > best ? k : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) sk[wid] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 1; k < kThreads / 64; ++k) best = sk[k] > best ? sk[k] : best;
            const unsigned ii = 0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull);
            y[pl] = p[ii];
            idx[pl] = (int)ii;
        }
    }
}

__global__ __launch_bounds__(kThreads) void gmax_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx,
                                                            const float* __restrict__ dx_in, float* __restrict__ dx,
                                                            long planes, long hw) {
    const long total = planes * hw;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const long pl = i / hw;
        float g = dx_in ? dx_in[i] : 0.f;
        if (i - pl * hw == (long)idx[pl]) g += dy[pl];
        dx[i] = g;
    }
}

int col_log2(int cols) {
    int l = 0;
    while ((1 << l) < cols && l < 8) ++l;
    return l;
}

unsigned row_grid(int rows, int colp_log2) {
    // row groups per plane: a few rows per block keep >= 8 iterations of work per thread
    const int per_block = kThreads >> colp_log2;
    int b = (rows + per_block * 8 - 1) / (per_block * 8);
    if (b < 1) b = 1;
    if (b > 64) b = 64;
    return (unsigned)b;
}

unsigned stream_grid(long total) {
    long b = (total + kThreads - 1) / kThreads;
    if (b > 256L * 16) b = 256L * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int fsc_maxpool_fwd(const float* x, float* y, uint8_t* idx, int nc, int h, int w, int ph, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && y && idx, "fsc_maxpool_fwd: null pointer");
    FSC_CHECK_ARG((ph == 1 || ph == 2) && nc > 0 && h >= ph && w >= 2, "fsc_maxpool_fwd: bad shape nc=%d h=%d w=%d ph=%d", nc, h, w, ph);
    const int oh = h / ph, ow = w / 2;
    if (h == 1 && ph == 1 && ow <= 256) {
        const long total = (long)nc * ow;
        hipLaunchKernelGGL(maxpool_rows_flat_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                           fsc::as_stream(stream), x, y, idx, w, ow, total);
        FSC_LAUNCH_CHECK("fsc_maxpool_fwd(rows)");
        return 0;
    }
    const int cl = col_log2(ow);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(nc, row_grid(oh, cl)), dim3(kThreads), 0, fsc::as_stream(stream), x, y,
                       idx, h, w, ph, oh, ow, cl);
    FSC_LAUNCH_CHECK("fsc_maxpool_fwd");
    return 0;
}

int fsc_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int nc, int h, int w, int ph, fsc_stream_t stream) {
    FSC_CHECK_ARG(dy && dx && idx, "fsc_maxpool_bwd: null pointer");
    FSC_CHECK_ARG((ph == 1 || ph == 2) && nc > 0 && h >= ph && w >= 2, "fsc_maxpool_bwd: bad shape");
    const int oh = h / ph, ow = w / 2;
    const int cl = col_log2(w);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nc, row_grid(h, cl)), dim3(kThreads), 0, fsc::as_stream(stream), dy,
                       idx, dx, h, w, ph, oh, ow, cl);
    FSC_LAUNCH_CHECK("fsc_maxpool_bwd");
    return 0;
}

int fsc_global_maxpool_fwd(const float* x, float* y, int* idx, int nc, long hw, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && y && idx && nc > 0 && hw > 0, "fsc_global_maxpool_fwd: bad arguments");
    long blocks = nc < 256L * 16 ? nc : 256L * 16;
    hipLaunchKernelGGL(gmax_fwd_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, fsc::as_stream(stream), x, y, idx,
                       (long)nc, hw);
    FSC_LAUNCH_CHECK("fsc_global_maxpool_fwd");
    return 0;
}

int fsc_global_maxpool_bwd(const float* dy, const int* idx, const float* dx_in, float* dx, int nc, long hw,
                           fsc_stream_t stream) {
    FSC_CHECK_ARG(dy && idx && dx && nc > 0 && hw > 0, "fsc_global_maxpool_bwd: bad arguments");
    hipLaunchKernelGGL(gmax_bwd_kernel, dim3(stream_grid((long)nc * hw)), dim3(kThreads), 0, fsc::as_stream(stream),
                       dy, idx, dx_in, dx, (long)nc, hw);
    FSC_LAUNCH_CHECK("fsc_global_maxpool_bwd");
    return 0;
}

}  // extern "C"
