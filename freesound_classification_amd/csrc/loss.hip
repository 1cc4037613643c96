// Loss kernels: LSEP pairwise ranking loss (reference networks/losses.py:47-58), binary
// cross-entropy on logits (networks/losses.py:19-22) and the sigmoid used for lwlrap
// (networks/classifiers.py:687).  One wavefront per sample; the (C x C) pairwise table the
// reference materialises three times is never written -- each lane owns rows i = lane,
// lane+64, ... and sweeps j from LDS.  exp() is deliberately un-stabilised like the reference.
#include "common.h"

namespace {

constexpr int kMaxClasses = 4096;

// loss[n] = log(1 + sum_{i,j : t_j < t_i} exp(s_j - s_i))
__global__ __launch_bounds__(64) void lsep_fwd_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                      float* __restrict__ loss, int c) {
    extern __shared__ float sm[];
    float* ss = sm;
    float* tt = sm + c;
    const int n = blockIdx.x, lane = threadIdx.x;
    for (int k = lane; k < c; k += 64) { ss[k] = s[(long)n * c + k]; tt[k] = t[(long)n * c + k]; }
    __syncthreads();
    float acc = 0.f;
    for (int i = lane; i < c; i += 64) {
        const float si = ss[i], ti = tt[i];
        float row = 0.f;
        for (int j = 0; j < c; ++j)
            if (tt[j] < ti) row += expf(ss[j] - si);
        acc += row;
    }
    acc = fsc::wave_sum(acc);
    if (lane == 0) loss[n] = logf(1.f + acc);
}

// d loss / d s_k = ( sum_{i: t_k < t_i} e^{s_k - s_i}  -  sum_{j: t_j < t_k} e^{s_j - s_k} ) / (1 + S)
__global__ __launch_bounds__(64) void lsep_bwd_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                      const float* __restrict__ dloss, float* __restrict__ ds, int c) {
    extern __shared__ float sm[];
    float* ss = sm;
    float* tt = sm + c;
    const int n = blockIdx.x, lane = threadIdx.x;
    for (int k = lane; k < c; k += 64) { ss[k] = s[(long)n * c + k]; tt[k] = t[(long)n * c + k]; }
    __syncthreads();
    float total = 0.f;
    // per-lane: plus-term and minus-term of its classes; total S from the minus terms
    for (int k = lane; k < c; k += 64) {
        const float sk = ss[k], tk = tt[k];
        float minus = 0.f;
        for (int j = 0; j < c; ++j)
            if (tt[j] < tk) minus += expf(ss[j] - sk);
        total += minus;
    }
    total = fsc::wave_sum(total);
    const float coef = dloss[n] / (1.f + total);
    for (int k = lane; k < c; k += 64) {
        const float sk = ss[k], tk = tt[k];
        float plus = 0.f, minus = 0.f;
        for (int j = 0; j < c; ++j) {
            const float tj = tt[j];
            if (tk < tj) plus += expf(sk - ss[j]);
            if (tj < tk) minus += expf(ss[j] - sk);
        }
        ds[(long)n * c + k] = coef * (plus - minus);
    }
}

// The same sums over the pairs that EXIST.  Only a class i whose target exceeds the sample's smallest target has partners j with
// t_j < t_i; with multi-hot labels that is the sample's positives -- 1 ... 3 of 80 classes -- while the kernels above evaluate all
// 80 x 80 exponentials under a mask (20 / 43 us per launch for 10 k values).  The wave lists those classes (ballot), then
// walks the list with its lanes over j.  Soft targets: the list is (nearly) every class, the cost that of the kernels above.
// Same terms, summed in another order.  c <= 256.
constexpr int kListSlots = 4;
__device__ __forceinline__ int lsep_list(const float* tt, int c, int* list, int lane) {
    float tmin = INFINITY;
    for (int k = lane; k < c; k += 64) tmin = fminf(tmin, tt[k]);
    tmin = -fsc::wave_max(-tmin);
    int count = 0;
    for (int k0 = 0; k0 < c; k0 += 64) {
        const int k = k0 + lane;
        const bool on = k < c && tt[k] > tmin;
        const unsigned long long m = __ballot(on);
        if (on) list[count + __popcll(m & ((1ull << lane) - 1ull))] = k;
        count += __popcll(m);
    }
    return count;
}

__global__ __launch_bounds__(64) void lsep_fwd_list_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                           float* __restrict__ loss, int c) {
    extern __shared__ float sm[];
    float* ss = sm;
    float* tt = sm + c;
    int* list = reinterpret_cast<int*>(sm + 2 * c);
    const int n = blockIdx.x, lane = threadIdx.x;
    for (int k = lane; k < c; k += 64) { ss[k] = s[(long)n * c + k]; tt[k] = t[(long)n * c + k]; }
    __syncthreads();
    const int count = lsep_list(tt, c, list, lane);
    __syncthreads();
    float acc = 0.f;
    for (int p = 0; p < count; ++p) {
        const int i = list[p];
        const float si = ss[i], ti = tt[i];
        for (int j = lane; j < c; j += 64)
            if (tt[j] < ti) acc += expf(ss[j] - si);
    }
    acc = fsc::wave_sum(acc);
    if (lane == 0) loss[n] = logf(1.f + acc);
}

__global__ __launch_bounds__(64) void lsep_bwd_list_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                           const float* __restrict__ dloss, float* __restrict__ ds, int c) {
    extern __shared__ float sm[];
    float* ss = sm;
    float* tt = sm + c;
    float* minus = sm + 2 * c;                      // sum_{j: t_j < t_k} e^{s_j - s_k} per class k (0 off the list)
    int* list = reinterpret_cast<int*>(sm + 3 * c);
    const int n = blockIdx.x, lane = threadIdx.x;
    for (int k = lane; k < c; k += 64) { ss[k] = s[(long)n * c + k]; tt[k] = t[(long)n * c + k]; minus[k] = 0.f; }
    __syncthreads();
    const int count = lsep_list(tt, c, list, lane);
    __syncthreads();
    float plus[kListSlots] = {0.f, 0.f, 0.f, 0.f};   // sum_{i: t_j < t_i} e^{s_j - s_i} of this lane's classes j = lane + 64 slot
    float total = 0.f;
    for (int p = 0; p < count; ++p) {
        const int i = list[p];
        const float si = ss[i], ti = tt[i];
        float row = 0.f;
#pragma unroll
        for (int q = 0; q < kListSlots; ++q) {
            const int j = lane + 64 * q;
            if (j < c && tt[j] < ti) {
                const float e = expf(ss[j] - si);
                plus[q] += e;
                row += e;
            }
        }
        row = fsc::wave_sum(row);
        if (lane == 0) minus[i] = row;
        total += row;
    }
    __syncthreads();
    const float coef = dloss[n] / (1.f + total);
#pragma unroll
    for (int q = 0; q < kListSlots; ++q) {
        const int k = lane + 64 * q;
        if (k < c) ds[(long)n * c + k] = coef * (plus[q] - minus[k]);
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void bce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t, double* acc, long count) {
    __shared__ double scratch[4];
    double local = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float p = sigmoidf_(x[i]);
        const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
        local -= (double)(t[i] * lp + (1.f - t[i]) * lq);
    }
    const double tot = fsc::block_sum<double, 4>(local, scratch);
    if (threadIdx.x == 0) atomicAdd(acc, tot);
}

__global__ void bce_finish_kernel(const double* acc, float* loss, long count) { loss[0] = (float)(acc[0] / (double)count); }

__global__ void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ dl,
                               float* __restrict__ dx, long count) {
    const float g = dl[0] / (float)count;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        dx[i] = (sigmoidf_(x[i]) - t[i]) * g;
}

__global__ void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, long count) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = sigmoidf_(x[i]);
}

__global__ void mean_fwd_kernel(const float* __restrict__ x, float* y, long count, float scale) {
    __shared__ double scratch[4];
    double local = 0.0;
    for (long i = threadIdx.x; i < count; i += blockDim.x) local += (double)x[i];
    const double tot = fsc::block_sum<double, 4>(local, scratch);
    if (threadIdx.x == 0) y[0] = (float)(tot / (double)count) * scale;
}

__global__ void mean_bwd_kernel(const float* dy, float* __restrict__ dx, long count, float scale) {
    const float g = dy[0] * scale / (float)count;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) dx[i] = g;
}

unsigned grid_for(long count) {
    long b = (count + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int fsc_lsep_fwd(const float* logits, const float* targets, float* loss, int n, int c, fsc_stream_t stream) {
    FSC_CHECK_ARG(logits && targets && loss && n > 0 && c > 0 && c <= kMaxClasses, "fsc_lsep_fwd: bad arguments (n=%d, c=%d)", n, c);
    if (c <= 64 * kListSlots)
        hipLaunchKernelGGL(lsep_fwd_list_kernel, dim3(n), dim3(64), 3 * c * sizeof(float), fsc::as_stream(stream), logits, targets,
                           loss, c);
    else
        hipLaunchKernelGGL(lsep_fwd_kernel, dim3(n), dim3(64), 2 * c * sizeof(float), fsc::as_stream(stream), logits, targets, loss, c);
    FSC_LAUNCH_CHECK("fsc_lsep_fwd");
    return 0;
}

int fsc_lsep_bwd(const float* logits, const float* targets, const float* dloss, float* dlogits, int n, int c,
                 fsc_stream_t stream) {
    FSC_CHECK_ARG(logits && targets && dloss && dlogits && n > 0 && c > 0 && c <= kMaxClasses, "fsc_lsep_bwd: bad arguments");
    if (c <= 64 * kListSlots)
        hipLaunchKernelGGL(lsep_bwd_list_kernel, dim3(n), dim3(64), 4 * c * sizeof(float), fsc::as_stream(stream), logits, targets,
                           dloss, dlogits, c);
    else
        hipLaunchKernelGGL(lsep_bwd_kernel, dim3(n), dim3(64), 2 * c * sizeof(float), fsc::as_stream(stream), logits, targets, dloss,
                           dlogits, c);
    FSC_LAUNCH_CHECK("fsc_lsep_bwd");
    return 0;
}

int fsc_bce_fwd(const float* logits, const float* targets, float* loss_scalar, double* workspace, long count,
                fsc_stream_t stream) {
    FSC_CHECK_ARG(logits && targets && loss_scalar && workspace && count > 0, "fsc_bce_fwd: bad arguments");
    hipStream_t st = fsc::as_stream(stream);
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(double), st);
    FSC_CHECK_ARG(e == hipSuccess, "fsc_bce_fwd: memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(grid_for(count) > 256 ? 256 : grid_for(count)), dim3(256), 0, st, logits,
                       targets, workspace, count);
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(1), 0, st, workspace, loss_scalar, count);
    FSC_LAUNCH_CHECK("fsc_bce_fwd");
    return 0;
}

int fsc_bce_bwd(const float* logits, const float* targets, const float* dloss_scalar, float* dlogits, long count,
                fsc_stream_t stream) {
    FSC_CHECK_ARG(logits && targets && dloss_scalar && dlogits && count > 0, "fsc_bce_bwd: bad arguments");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, fsc::as_stream(stream), logits, targets,
                       dloss_scalar, dlogits, count);
    FSC_LAUNCH_CHECK("fsc_bce_bwd");
    return 0;
}

int fsc_sigmoid(const float* x, float* y, long count, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && y && count > 0, "fsc_sigmoid: bad arguments");
    hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_for(count)), dim3(256), 0, fsc::as_stream(stream), x, y, count);
    FSC_LAUNCH_CHECK("fsc_sigmoid");
    return 0;
}

int fsc_mean_fwd(const float* x, float* y, long count, float scale, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && y && count > 0, "fsc_mean_fwd: bad arguments");
    hipLaunchKernelGGL(mean_fwd_kernel, dim3(1), dim3(256), 0, fsc::as_stream(stream), x, y, count, scale);
    FSC_LAUNCH_CHECK("fsc_mean_fwd");
    return 0;
}

int fsc_mean_bwd(const float* dy, float* dx, long count, float scale, fsc_stream_t stream) {
    FSC_CHECK_ARG(dy && dx && count > 0, "fsc_mean_bwd: bad arguments");
    hipLaunchKernelGGL(mean_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, fsc::as_stream(stream), dy, dx, count, scale);
    FSC_LAUNCH_CHECK("fsc_mean_bwd");
    return 0;
}

}  // extern "C"
