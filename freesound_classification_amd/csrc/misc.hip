// Library plumbing (version, last-error text), small streaming helpers, and the batched
// MixUp kernel (reference ops/audio.py:32-52 applied to device-resident batches).
#include <stdarg.h>
#include <string.h>

#include "common.h"
#include <stdlib.h>

namespace fsc {
const EnvFlags& env() {
    static const EnvFlags flags = [] {
        auto on = [](const char* name) { return getenv(name) != nullptr; };
        auto num = [](const char* name) { const char* v = getenv(name); return v ? atoi(v) : 0; };
        EnvFlags f{};
        f.no_l16 = on("FSC_NO_L16");
        f.no_l16_pool = on("FSC_NO_L16_POOL");
        f.no_l16_wgrad = on("FSC_NO_L16_WGRAD");
        f.l16_no_xcd = on("FSC_L16_NO_XCD");
        f.l16_vec1 = on("FSC_L16_VEC1");
        f.dbg_noksplit = on("FSC_DBG_NOKSPLIT");
        f.frontend_generic = on("FSC_FRONTEND_GENERIC");
        f.fe_block_sync = on("FSC_FE_BLOCK_SYNC");
        f.l16_cot = num("FSC_L16_COT");
        f.l16_pt = num("FSC_L16_PT");
        f.l16w_tw = num("FSC_L16W_TW");
        return f;
    }();
    return flags;
}
}  // namespace fsc

namespace fsc {

static char g_error[512] = "no error";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

}  // namespace fsc

namespace {

__global__ void fill_kernel(float* __restrict__ x, float v, long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) x[i] = v;
}
// (N, sum widths) row-major <-> its column pieces (N, w_i), each contiguous: the features of the deep-supervision heads
// (classifiers.py:586-595 torch.cat) and the way back for their gradient.  Tables travel in the kernel arguments.
constexpr int kMaxPieces = 16;
struct ColPieces {
    float* ptr[kMaxPieces];
    int start[kMaxPieces + 1];
    int count;
};
__global__ void cat_cols_kernel(ColPieces cp, float* __restrict__ full, int rows, int total, int split) {
    const long n = (long)rows * total;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / total), col = (int)(i - (long)r * total);
        int k = 0;
        while (k + 1 < cp.count && col >= cp.start[k + 1]) ++k;
        float* piece = cp.ptr[k] + (long)r * (cp.start[k + 1] - cp.start[k]) + (col - cp.start[k]);
        if (split) *piece = full[i]; else full[i] = *piece;
    }
}

// num_batches_tracked of every BatchNorm a training forward went through, one launch (torch/nn/modules/batchnorm.py: += 1)
constexpr int kMaxCounters = 64;
struct CounterTable {
    long* ptr[kMaxCounters];
    int count;
};
__global__ void bump_counters_kernel(CounterTable t) {
    // (atomic: a BatchNorm applied twice in one forward appears twice in the table and must advance by two, like torch's add_)
    if ((int)threadIdx.x < t.count) atomicAdd(reinterpret_cast<unsigned long long*>(t.ptr[threadIdx.x]), 1ull);
}


__global__ __launch_bounds__(256) void absmin_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float part[4];
    float m = INFINITY;
    for (long i = threadIdx.x; i < n; i += 256) m = fminf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fminf(fminf(part[0], part[1]), fminf(part[2], part[3]));
}

__global__ void axpy_kernel(const float* __restrict__ x, float a, float* __restrict__ y, long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = fmaf(a, x[i], y[i]);
}

// Row n: equal lengths -> (a + b) / 2.  Otherwise the longer clip is scaled by f32(alpha) and
// the window [start, start + shorter) is REPLACED by shorter * f32(1 - alpha): audio.py:50
// reads `longer[start:end] =+ shorter * (1 - a)`, an assignment.  Products are single fp32
// multiplies, so the result is bit-identical to numpy's.
__global__ void mixup_kernel(const float* __restrict__ a, const float* __restrict__ b, const int* __restrict__ len_a,
                             const int* __restrict__ len_b, const int* __restrict__ start,
                             const float* __restrict__ alpha, const float* __restrict__ oma, float* __restrict__ out,
                             long t_a, long t_b, long t_out) {
    const int n = blockIdx.y;
    const int la = len_a[n], lb = len_b[n];
    const float* pa = a + (long)n * t_a;
    const float* pb = b + (long)n * t_b;
    float* po = out + (long)n * t_out;
    const bool a_longer = la > lb;
    const float* plong = a_longer ? pa : pb;
    const float* pshort = a_longer ? pb : pa;
    const int ll = a_longer ? la : lb, ls = a_longer ? lb : la;
    const int s0 = start[n];
    const float al = alpha[n], om = oma[n];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < t_out; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (la == lb) {
            if (i < la) v = (pa[i] + pb[i]) / 2.f;
        } else if (i < ll) {
            v = (i >= s0 && i < s0 + ls) ? pshort[i - s0] * om : plong[i] * al;
        }
        po[i] = v;
    }
}

// Batched index copy for the waveform transforms that are pure index work (reference ops/transforms.py:292-309
// SampleLongAudio: crop; :256-271 ShuffleAudio -> ops/audio.py:55-67: permute ~0.5 s chunks).  Row n of the output is
// the concatenation of seg_count[n] segments of row src_row[n] of `src`: segment k copies
// src[seg_src[n][k] ... + (seg_dst[n][k+1] - seg_dst[n][k])) to out[seg_dst[n][k] ...]; everything past
// seg_dst[n][count] is the collate padding value (0).  Exact copies: bit-identical to the numpy slicing.
constexpr int kMaxSeg = 256;
__global__ __launch_bounds__(256) void segments_gather_kernel(const float* __restrict__ src, long src_stride,
                                                              const int* __restrict__ src_row,
                                                              const int* __restrict__ seg_count,
                                                              const int* __restrict__ seg_src,
                                                              const int* __restrict__ seg_dst, int max_seg,
                                                              float* __restrict__ out, long t_out) {
    __shared__ int s_src[kMaxSeg], s_dst[kMaxSeg + 1];
    const int n = blockIdx.y;
    const int cnt = seg_count[n];
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) s_src[i] = seg_src[(long)n * max_seg + i];
    for (int i = threadIdx.x; i <= cnt; i += blockDim.x) s_dst[i] = seg_dst[(long)n * (max_seg + 1) + i];
    __syncthreads();
    const float* ps = src + (long)src_row[n] * src_stride;
    float* po = out + (long)n * t_out;
    const int total = cnt > 0 ? s_dst[cnt] : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < t_out; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < total) {
            int lo = 0, hi = cnt - 1;                      // last segment with s_dst[k] <= i
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_dst[mid] <= (int)i) lo = mid; else hi = mid - 1;
            }
            v = ps[s_src[lo] + ((int)i - s_dst[lo])];
        }
        po[i] = v;
    }
}

// mixup_kernel with a partner table: row n is mixed with row partner[n] of `b` (lengths len_a[n], len_b[n]), or copied
// unchanged when partner[n] < 0 (MixUp not drawn for this sample, ops/transforms.py:57).
__global__ void mixup_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, const int* __restrict__ partner,
                                  const int* __restrict__ len_a, const int* __restrict__ len_b,
                                  const int* __restrict__ start, const float* __restrict__ alpha,
                                  const float* __restrict__ oma, float* __restrict__ out, long t_a, long t_b, long t_out) {
    const int n = blockIdx.y;
    const int pr = partner[n];
    const int la = len_a[n], lb = pr < 0 ? 0 : len_b[n];
    const float* pa = a + (long)n * t_a;
    const float* pb = b + (long)(pr < 0 ? 0 : pr) * t_b;
    float* po = out + (long)n * t_out;
    const bool a_longer = la > lb;
    const float* plong = a_longer ? pa : pb;
    const float* pshort = a_longer ? pb : pa;
    const int ll = a_longer ? la : lb, ls = a_longer ? lb : la;
    const int s0 = start[n];
    const float al = alpha[n], om = oma[n];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < t_out; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (pr < 0) {
            if (i < la) v = pa[i];
        } else if (la == lb) {
            if (i < la) v = (pa[i] + pb[i]) / 2.f;
        } else if (i < ll) {
            v = (i >= s0 && i < s0 + ls) ? pshort[i - s0] * om : plong[i] * al;
        }
        po[i] = v;
    }
}

__global__ void or_labels_rows_kernel(const float* __restrict__ la, const float* __restrict__ lb, const int* __restrict__ partner,
                                      float* __restrict__ lo, int n, int c) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)n * c; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / c), col = (int)(i - (long)row * c);
        const int pr = partner[row];
        lo[i] = pr < 0 ? la[i] : fminf(fmaxf(la[i] + lb[(long)pr * c + col], 0.f), 1.f);
    }
}

__global__ void or_labels_kernel(const float* __restrict__ la, const float* __restrict__ lb, float* __restrict__ lo,
                                 long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        lo[i] = fminf(fmaxf(la[i] + lb[i], 0.f), 1.f);
}

unsigned grid_for(long count) {
    long b = (count + 255) / 256;
    if (b > 8192) b = 8192;
    return (unsigned)(b < 1 ? 1 : b);
}

// Border sums of (N, C, H, W) planes, per channel: out[c][8] += [first row, last row, first column, last column, and the four
// corners (0,0), (0,W-1), (H-1,0), (H-1,W-1)] summed over the images (out pre-zeroed).  With them the sums of a gradient over
// the pixels a convolution tap can reach follow from its total: what ConvBlockFn needs to obtain the first block's
// input-BN parameter gradients without running the stem input-gradient convolution (functional._stem_bn_grads).
__global__ __launch_bounds__(256) void border_sums_kernel(const float* __restrict__ x, int c, int h, int w, float* __restrict__ out) {
    __shared__ float red[4][4];
    const int ch = blockIdx.x, img = blockIdx.y;
    const float* p = x + ((long)img * c + ch) * h * w;
    float r0 = 0.f, rl = 0.f, c0 = 0.f, cl = 0.f;
    for (int i = threadIdx.x; i < w; i += 256) { r0 += p[i]; rl += p[(long)(h - 1) * w + i]; }
    for (int i = threadIdx.x; i < h; i += 256) { c0 += p[(long)i * w]; cl += p[(long)i * w + w - 1]; }
    r0 = fsc::wave_sum(r0); rl = fsc::wave_sum(rl); c0 = fsc::wave_sum(c0); cl = fsc::wave_sum(cl);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = r0; red[threadIdx.x >> 6][1] = rl; red[threadIdx.x >> 6][2] = c0; red[threadIdx.x >> 6][3] = cl; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        unsafeAtomicAdd(out + ch * 8 + threadIdx.x, v);
    } else if (threadIdx.x < 8) {
        const int k = threadIdx.x - 4;
        const float v = p[(long)((k >> 1) ? h - 1 : 0) * w + ((k & 1) ? w - 1 : 0)];
        unsafeAtomicAdd(out + ch * 8 + threadIdx.x, v);
    }
}

// First block of the 1-d model, the arithmetic behind functional._first_block_grads_1d in one launch (it was ~15 broadcast /
// reduction launches of torch, 70 - 90 us per step on 25 k values): with dWx = the weight-gradient pass over the BatchNorm's RAW
// input, T[co][t] = the border-corrected channel sums of dc, C = invstd (dWx - mean T):  dW = gamma C + beta T,
// dgamma[ci] = sum_{co,t} w C, dbeta[ci] = sum_{co,t} w T.  One workgroup per input channel; every workgroup rebuilds the
// (c_out, 3) table T from dc's first / last columns (2 * n * c_out values, L2-resident) in LDS.
__global__ __launch_bounds__(256) void first_block_1d_kernel(const float* __restrict__ dwx, const float* __restrict__ dc,
                                                             const float* __restrict__ dc_sum, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ w, int n,
                                                             int c_in, int c_out, int len, float* __restrict__ dw,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
    extern __shared__ float tS[];                 // [c_out][3] then [c_out][2][4] partial border sums
    __shared__ double red[2][4];
    const int ci = blockIdx.x, tid = threadIdx.x;
    float* part = tS + 3 * c_out;
    // border columns of dc summed over the images: thread = (channel, quarter of the batch)
    for (int e = tid; e < 4 * c_out; e += 256) {
        const int co = e >> 2, q = e & 3;
        float f[4] = {0.f, 0.f, 0.f, 0.f}, l[4] = {0.f, 0.f, 0.f, 0.f};      // (eight loads in flight: a chain of n / 4 round trips otherwise)
        int b = q;
        for (; b + 12 < n; b += 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* p = dc + ((long)(b + 4 * u) * c_out + co) * len;
                f[u] += p[0];
                l[u] += p[len - 1];
            }
        }
        for (; b < n; b += 4) {
            const float* p = dc + ((long)b * c_out + co) * len;
            f[0] += p[0];
            l[0] += p[len - 1];
        }
        part[(co * 2) * 4 + q] = (f[0] + f[1]) + (f[2] + f[3]);
        part[(co * 2 + 1) * 4 + q] = (l[0] + l[1]) + (l[2] + l[3]);
    }
    __syncthreads();
    for (int co = tid; co < c_out; co += 256) {
        const float* pf = part + (co * 2) * 4;
        const float* pl = part + (co * 2 + 1) * 4;
        const float tot = dc_sum[co];
        tS[co * 3] = tot - ((pf[0] + pf[1]) + (pf[2] + pf[3]));
        tS[co * 3 + 1] = tot;
        tS[co * 3 + 2] = tot - ((pl[0] + pl[1]) + (pl[2] + pl[3]));
    }
    __syncthreads();
    const float mu = mean[ci], is = invstd[ci], g = gamma[ci], b = beta[ci];
    double dg = 0.0, db = 0.0;
    for (int e = tid; e < 3 * c_out; e += 256) {
        const int co = e / 3, tap = e - co * 3;
        const long idx = ((long)co * c_in + ci) * 3 + tap;
        const float t = tS[e];
        const float core = (dwx[idx] - mu * t) * is;
        dw[idx] = core * g + b * t;
        dg += (double)(w[idx] * core);
        db += (double)(w[idx] * t);
    }
    dg = fsc::wave_sum(dg);
    db = fsc::wave_sum(db);
    if ((tid & 63) == 0) { red[0][tid >> 6] = dg; red[1][tid >> 6] = db; }
    __syncthreads();
    if (tid == 0) {
        dgamma[ci] = (float)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        dbeta[ci] = (float)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

}  // namespace

extern "C" {

int fsc_first_block_1d_finish(const float* dwx, const float* dc, const float* dc_chan_sum, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, const float* weight, int n, int c_in, int c_out, int len,
                              float* dw, float* dgamma, float* dbeta, fsc_stream_t stream) {
    FSC_CHECK_ARG(dwx && dc && dc_chan_sum && mean && invstd && gamma && beta && weight && dw && dgamma && dbeta,
                  "fsc_first_block_1d_finish: null pointer");
    FSC_CHECK_ARG(n > 0 && c_in > 0 && c_out > 0 && c_out <= 2048 && len >= 2, "fsc_first_block_1d_finish: bad shape");
    const size_t lds = sizeof(float) * (size_t)c_out * 11;
    hipLaunchKernelGGL(first_block_1d_kernel, dim3(c_in), dim3(256), lds, fsc::as_stream(stream), dwx, dc, dc_chan_sum, mean, invstd,
                       gamma, beta, weight, n, c_in, c_out, len, dw, dgamma, dbeta);
    FSC_LAUNCH_CHECK("fsc_first_block_1d_finish");
    return 0;
}

int fsc_version(void) { return 100; /* 0.1.0 */ }

const char* fsc_last_error_string(void) { return fsc::g_error; }

int fsc_cat_cols(const float* const* pieces, const int* widths, int count, int rows, float* out, int split, fsc_stream_t stream) {
    FSC_CHECK_ARG(pieces && widths && out && rows > 0 && count > 0 && count <= kMaxPieces, "fsc_cat_cols: bad arguments");
    ColPieces cp{};
    int total = 0;
    for (int i = 0; i < count; ++i) {
        FSC_CHECK_ARG(pieces[i] && widths[i] > 0, "fsc_cat_cols: piece %d is empty", i);
        cp.ptr[i] = const_cast<float*>(pieces[i]);
        cp.start[i] = total;
        total += widths[i];
    }
    cp.start[count] = total;
    cp.count = count;
    const long n = (long)rows * total;
    hipLaunchKernelGGL(cat_cols_kernel, dim3(grid_for(n)), dim3(256), 0, fsc::as_stream(stream), cp, out, rows, total, split);
    FSC_LAUNCH_CHECK("fsc_cat_cols");
    return 0;
}

int fsc_bump_counters(long* const* counters, int count, fsc_stream_t stream) {
    FSC_CHECK_ARG(counters && count > 0, "fsc_bump_counters: bad arguments");
    for (int first = 0; first < count; first += kMaxCounters) {
        CounterTable t{};
        t.count = count - first < kMaxCounters ? count - first : kMaxCounters;
        for (int i = 0; i < t.count; ++i) {
            FSC_CHECK_ARG(counters[first + i], "fsc_bump_counters: null counter %d", first + i);
            t.ptr[i] = counters[first + i];
        }
        hipLaunchKernelGGL(bump_counters_kernel, dim3(1), dim3(kMaxCounters), 0, fsc::as_stream(stream), t);
    }
    FSC_LAUNCH_CHECK("fsc_bump_counters");
    return 0;
}

int fsc_absmin(const float* x, long n, float* out, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && out && n > 0, "fsc_absmin: bad arguments");
    hipLaunchKernelGGL(absmin_kernel, dim3(1), dim3(256), 0, fsc::as_stream(stream), x, n, out);
    FSC_LAUNCH_CHECK("fsc_absmin");
    return 0;
}

int fsc_fill(float* x, float value, long count, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && count > 0, "fsc_fill: bad arguments");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(count)), dim3(256), 0, fsc::as_stream(stream), x, value, count);
    FSC_LAUNCH_CHECK("fsc_fill");
    return 0;
}

int fsc_axpy(const float* x, float a, float* y, long count, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && y && count > 0, "fsc_axpy: bad arguments");
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(count)), dim3(256), 0, fsc::as_stream(stream), x, a, y, count);
    FSC_LAUNCH_CHECK("fsc_axpy");
    return 0;
}

int fsc_mixup_batch(const float* a, const float* b, const int* len_a, const int* len_b, const int* start,
                    const float* alpha, const float* one_minus_alpha, float* out, int n, long t_a, long t_b,
                    long t_out, const float* labels_a, const float* labels_b, float* labels_out, int c,
                    fsc_stream_t stream) {
    FSC_CHECK_ARG(a && b && len_a && len_b && start && alpha && one_minus_alpha && out, "fsc_mixup_batch: null pointer");
    FSC_CHECK_ARG(n > 0 && t_a > 0 && t_b > 0 && t_out >= (t_a > t_b ? t_a : t_b), "fsc_mixup_batch: bad sizes");
    hipStream_t st = fsc::as_stream(stream);
    unsigned gx = grid_for(t_out);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(mixup_kernel, dim3(gx, n), dim3(256), 0, st, a, b, len_a, len_b, start, alpha,
                       one_minus_alpha, out, t_a, t_b, t_out);
    if (labels_out) {
        FSC_CHECK_ARG(labels_a && labels_b && c > 0, "fsc_mixup_batch: label pointers");
        hipLaunchKernelGGL(or_labels_kernel, dim3(grid_for((long)n * c)), dim3(256), 0, st, labels_a, labels_b,
                           labels_out, (long)n * c);
    }
    FSC_LAUNCH_CHECK("fsc_mixup_batch");
    return 0;
}

int fsc_segments_gather(const float* src, long src_stride, const int* src_row, const int* seg_count, const int* seg_src,
                        const int* seg_dst, int max_seg, float* out, int n, long t_out, fsc_stream_t stream) {
    FSC_CHECK_ARG(src && src_row && seg_count && seg_src && seg_dst && out, "fsc_segments_gather: null pointer");
    FSC_CHECK_ARG(n > 0 && t_out > 0 && max_seg > 0 && max_seg <= kMaxSeg, "fsc_segments_gather: bad sizes (max_seg <= %d)", kMaxSeg);
    unsigned gx = grid_for(t_out);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(segments_gather_kernel, dim3(gx, n), dim3(256), 0, fsc::as_stream(stream), src, src_stride, src_row,
                       seg_count, seg_src, seg_dst, max_seg, out, t_out);
    FSC_LAUNCH_CHECK("fsc_segments_gather");
    return 0;
}

int fsc_mixup_rows(const float* a, const float* b, const int* partner, const int* len_a, const int* len_b, const int* start,
                   const float* alpha, const float* one_minus_alpha, float* out, int n, long t_a, long t_b, long t_out,
                   const float* labels_a, const float* labels_b, float* labels_out, int c, fsc_stream_t stream) {
    FSC_CHECK_ARG(a && b && partner && len_a && len_b && start && alpha && one_minus_alpha && out, "fsc_mixup_rows: null pointer");
    FSC_CHECK_ARG(n > 0 && t_a > 0 && t_b > 0 && t_out >= (t_a > t_b ? t_a : t_b), "fsc_mixup_rows: bad sizes");
    hipStream_t st = fsc::as_stream(stream);
    unsigned gx = grid_for(t_out);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(mixup_rows_kernel, dim3(gx, n), dim3(256), 0, st, a, b, partner, len_a, len_b, start, alpha,
                       one_minus_alpha, out, t_a, t_b, t_out);
    if (labels_out) {
        FSC_CHECK_ARG(labels_a && labels_b && c > 0, "fsc_mixup_rows: label pointers");
        hipLaunchKernelGGL(or_labels_rows_kernel, dim3(grid_for((long)n * c)), dim3(256), 0, st, labels_a, labels_b, partner,
                           labels_out, n, c);
    }
    FSC_LAUNCH_CHECK("fsc_mixup_rows");
    return 0;
}

int fsc_plane_border_sums(const float* x, int n, int c, int h, int w, float* out, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && out && n > 0 && c > 0 && h > 0 && w > 0, "fsc_plane_border_sums: bad arguments");
    hipLaunchKernelGGL(border_sums_kernel, dim3(c, n), dim3(256), 0, fsc::as_stream(stream), x, c, h, w, out);
    FSC_LAUNCH_CHECK("fsc_plane_border_sums");
    return 0;
}

}  // extern "C"
