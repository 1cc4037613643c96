// Multi-tensor optimizer steps (HBM-bound: Adam-amsgrad streams 5 reads + 4 writes per
// parameter).  Reference ops/training.py:9-12 builds torch.optim.Adam(amsgrad=True) and
// torch.optim.SGD(momentum=0.9, nesterov=True); the arithmetic below follows torch's
// single-tensor update rules (weight decay folded into the gradient, lerp-style first moment).
// Tensor tables travel in kernel arguments, kBatch tensors per launch, so nothing has to stay
// alive on the host and no device-side table is allocated.
#include "common.h"

namespace {

// 64 tensors per launch (3.4 KB of the 4 KB of kernel arguments) in chunks of 4096 elements: the 1-d model's ~230 small tensors
// took ten launches of sixteen-trip workgroups with 24 x 16384 (0.2 ms per step; four launches of four-trip workgroups now)
constexpr int kBatch = 64;
constexpr int kChunk = 4096;     // elements per workgroup
constexpr int kThreads = 256;

struct Table {
    fsc_opt_tensor t[kBatch];
    int chunk_start[kBatch + 1];
    int n;
};

struct AdamHyper {
    float beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt, grad_scale;
    const float* dev;        // non-null: {step_size, inv_bc2_sqrt} are read from device memory (a step recorded in a HIP graph)
};

struct SgdHyper {
    float lr, momentum, weight_decay, grad_scale;
    int first_step;
};

__device__ __forceinline__ int find_tensor(const Table& tb, int chunk) {      // the last i with chunk_start[i] <= chunk
    int lo = 0, hi = tb.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (chunk >= tb.chunk_start[mid]) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ void adam_one(const AdamHyper& h, float& p, float g, float& m, float& v, float& vmax) {
    g *= h.grad_scale;
    if (h.weight_decay != 0.f) g = fmaf(h.weight_decay, p, g);
    m = m + (1.f - h.beta1) * (g - m);
    v = fmaf(v, h.beta2, (1.f - h.beta2) * g * g);
    vmax = fmaxf(vmax, v);
    const float denom = sqrtf(vmax) * h.inv_bc2_sqrt + h.eps;
    p = p - h.step_size * (m / denom);
}

__global__ __launch_bounds__(kThreads) void adam_kernel(Table tb, AdamHyper h) {
    if (h.dev != nullptr) {
        h.step_size = h.dev[0];
        h.inv_bc2_sqrt = h.dev[1];
    }
    const int ti = find_tensor(tb, blockIdx.x);
    const fsc_opt_tensor t = tb.t[ti];
    const long base = (long)(blockIdx.x - tb.chunk_start[ti]) * kChunk;
    const long end = base + kChunk < t.count ? base + kChunk : t.count;
    long i0 = base;
    // nine 16-byte streams when the five arrays allow it (a gradient that aliases a bucket of the reducer may sit at any 4-byte
    // offset); the dword loop below takes the rest
    const size_t bits = (size_t)t.param | (size_t)t.grad | (size_t)t.state0 | (size_t)t.state1 | (size_t)t.state2;
    if ((bits & 15) == 0) {
        const long n4 = (end - base) >> 2;
        float4* p4 = reinterpret_cast<float4*>(t.param + base);
        const float4* g4 = reinterpret_cast<const float4*>(t.grad + base);
        float4* m4 = reinterpret_cast<float4*>(t.state0 + base);
        float4* v4 = reinterpret_cast<float4*>(t.state1 + base);
        float4* x4 = reinterpret_cast<float4*>(t.state2 + base);
        for (long i = threadIdx.x; i < n4; i += kThreads) {
            float4 p = p4[i], m = m4[i], v = v4[i], x = x4[i];
            const float4 g = g4[i];
            adam_one(h, p.x, g.x, m.x, v.x, x.x);
            adam_one(h, p.y, g.y, m.y, v.y, x.y);
            adam_one(h, p.z, g.z, m.z, v.z, x.z);
            adam_one(h, p.w, g.w, m.w, v.w, x.w);
            m4[i] = m; v4[i] = v; x4[i] = x; p4[i] = p;
        }
        i0 = base + 4 * n4;
    }
    for (long i = i0 + threadIdx.x; i < end; i += kThreads) {
        float p = t.param[i], m = t.state0[i], v = t.state1[i], vmax = t.state2[i];
        adam_one(h, p, t.grad[i], m, v, vmax);
        t.state0[i] = m;
        t.state1[i] = v;
        t.state2[i] = vmax;
        t.param[i] = p;
    }
}

__global__ __launch_bounds__(kThreads) void sgd_kernel(Table tb, SgdHyper h) {
    const int ti = find_tensor(tb, blockIdx.x);
    const fsc_opt_tensor t = tb.t[ti];
    const long base = (long)(blockIdx.x - tb.chunk_start[ti]) * kChunk;
    const long end = base + kChunk < t.count ? base + kChunk : t.count;
    for (long i = base + threadIdx.x; i < end; i += kThreads) {
        const float p = t.param[i];
        float g = t.grad[i] * h.grad_scale;
        if (h.weight_decay != 0.f) g = fmaf(h.weight_decay, p, g);
        const float buf = h.first_step ? g : fmaf(t.state0[i], h.momentum, g);
        t.state0[i] = buf;
        g = fmaf(h.momentum, buf, g);
        t.param[i] = p - h.lr * g;
    }
}

template <typename Hyper, typename Kernel>
int run(const fsc_opt_tensor* tensors, int n_tensors, const Hyper& h, Kernel kernel, bool need_all_state,
        hipStream_t st, const char* name) {
    for (int first = 0; first < n_tensors; first += kBatch) {
        Table tb{};
        tb.n = n_tensors - first < kBatch ? n_tensors - first : kBatch;
        int chunks = 0;
        for (int i = 0; i < tb.n; ++i) {
            const fsc_opt_tensor& t = tensors[first + i];
            FSC_CHECK_ARG(t.param && t.grad && t.state0 && t.count > 0, "%s: tensor %d has null pointers or zero size", name, first + i);
            FSC_CHECK_ARG(!need_all_state || (t.state1 && t.state2), "%s: tensor %d misses optimizer state", name, first + i);
            tb.t[i] = t;
            tb.chunk_start[i] = chunks;
            chunks += fsc::ceil_div(t.count, kChunk);
        }
        tb.chunk_start[tb.n] = chunks;
        hipLaunchKernelGGL(kernel, dim3(chunks), dim3(kThreads), 0, st, tb, h);
        FSC_LAUNCH_CHECK(name);
    }
    return 0;
}

}  // namespace

extern "C" {

int fsc_adam_amsgrad_step(const fsc_opt_tensor* tensors_host, int n_tensors, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int step, float grad_scale, fsc_stream_t stream) {
    FSC_CHECK_ARG(tensors_host && n_tensors > 0 && step >= 1, "fsc_adam_amsgrad_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamHyper h{beta1, beta2, eps, weight_decay, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, nullptr};
    return run(tensors_host, n_tensors, h, adam_kernel, true, fsc::as_stream(stream), "fsc_adam_amsgrad_step");
}

void fsc_adam_step_factors(float lr, float beta1, float beta2, int step, float* out2_host) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    out2_host[0] = (float)((double)lr / bc1);
    out2_host[1] = (float)(1.0 / sqrt(bc2));
}

int fsc_adam_amsgrad_step_dev(const fsc_opt_tensor* tensors_host, int n_tensors, const float* factors_dev, float beta1,
                              float beta2, float eps, float weight_decay, float grad_scale, fsc_stream_t stream) {
    FSC_CHECK_ARG(tensors_host && n_tensors > 0 && factors_dev, "fsc_adam_amsgrad_step_dev: bad arguments");
    AdamHyper h{beta1, beta2, eps, weight_decay, 0.f, 0.f, grad_scale, factors_dev};
    return run(tensors_host, n_tensors, h, adam_kernel, true, fsc::as_stream(stream), "fsc_adam_amsgrad_step_dev");
}

int fsc_sgd_nesterov_step(const fsc_opt_tensor* tensors_host, int n_tensors, float lr, float momentum,
                          float weight_decay, int first_step, float grad_scale, fsc_stream_t stream) {
    FSC_CHECK_ARG(tensors_host && n_tensors > 0, "fsc_sgd_nesterov_step: bad arguments");
    SgdHyper h{lr, momentum, weight_decay, grad_scale, first_step};
    return run(tensors_host, n_tensors, h, sgd_kernel, false, fsc::as_stream(stream), "fsc_sgd_nesterov_step");
}

}  // extern "C"
