// Library plumbing (version, last-error text), small streaming helpers, and the batched
// MixUp kernel (reference ops/audio.py:32-52 applied to device-resident batches).
#include <stdarg.h>
#include <string.h>

#include "common.h"

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

}  // namespace

extern "C" {

int fsc_version(void) { return 100; /* 0.1.0 */ }

const char* fsc_last_error_string(void) { return fsc::g_error; }

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

}  // extern "C"
