// Batch normalisation (train + eval) fused with the per-channel PReLU and residual add that
// follow it in the reference's blocks (networks/classifiers.py:524,533-534, 37-104, 543-546).
// All kernels are HBM-bound streaming passes over (N, C, HW) fp32 activations:
//   stats    : 1 read of x                       -> per-channel mean / invstd / scale / shift
//   fwd      : 1 read of x (+ residual), 1 write -> y = prelu(x*scale + shift + residual)
//   bwd      : reduce pass (reads dy, x, residual) + apply pass (reads the same, writes dx,
//              dresidual); the pre-activation is recomputed instead of stored.
// Per-channel reductions: fp32 per-thread partials on pivot-shifted data, fp64 across threads,
// blocks and splits (deterministic two-stage reduce, no atomics on the statistics).
#include "common.h"
#include "l16.h"
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
constexpr int kMaxSplit = 64;

// workspace layout (doubles): partial[c][split][8], then coef[c][4] floats
struct Partials {
    double* part;       // c * kMaxSplit * kPartStride
    float* coef;        // c * 4
    unsigned* tickets;  // c arrival counters of the fused finalisations (zero between calls, see fsc_hip.h FSC_BN_TICKETS)
};

constexpr int kPartStride = 8;
__host__ __device__ inline size_t part_doubles(int c) { return (size_t)c * kMaxSplit * kPartStride; }

inline Partials carve(void* ws, int c) {
    Partials p;
    p.part = reinterpret_cast<double*>(ws);
    p.coef = reinterpret_cast<float*>(p.part + part_doubles(c));
    p.tickets = reinterpret_cast<unsigned*>(p.coef + (size_t)c * 4);
    return p;
}

// Arrival ticket of one channel's partial results (include/fsc_hip.h, FSC_BN_TICKETS): the workgroup that arrives LAST folds the
// partials of all `nsplit` workgroups and finalises the channel, so the reduce pass needs no second launch.  Called by the ONE
// thread that stored its workgroup's partials with device-scope atomic stores (`global_store ... sc1`: written through to the
// memory side, no line kept in this XCD's L2).  The hand-off is the "8-byte agent-scope atomics on both sides + drained flag"
// form of MI355X_MICROARCH.md (inter-workgroup visibility): `s_waitcnt vmcnt(0)` holds the ticket back until the partial stores
// are acknowledged, and the winner reads the partials with device-scope atomic loads (served by neither its L1 nor a stale line of
// its own XCD's L2).  For the language: the inline asm and the signal fence keep the compiler from moving the stores below / the
// loads above the ticket, and the winner alone runs an agent-scope acquire fence (`buffer_inv sc1`: one per channel and launch,
// nothing measurable).  -DFSC_BN_TICKET_RELEASE builds the textbook form instead -- a release fetch_add, i.e. `buffer_wbl2 sc1`
// (write-back of the XCD's whole L2) in EVERY workgroup: measured 2.6 -> 5.3 ms per cfg-2 step on the backward reduce pass with a
// full fence, see DESIGN.md section 4.7 for the release-only figure.
// The winner leaves the counter at zero: a workspace that starts zeroed stays usable call after call.
__device__ __forceinline__ bool ticket_last(unsigned* ticket, int nsplit) {
#ifdef FSC_BN_TICKET_RELEASE
    const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
#endif
    if (prev != (unsigned)nsplit - 1u) return false;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

// ---------------------------------------------------------------- statistics
struct FinalizeArgs {
    const float* x; int c; long hw; double count; int nsplit; const double* part; const float* gamma; const float* beta;
    float eps, momentum; float* running_mean; float* running_var; float* save_mean; float* save_invstd; float* scale; float* shift;
    double* sync; int phase; float* x_minmax; int pivot_rm, minmax_only;
};

// (s1, s2): the channel's sums of (x - pivot), (x - pivot)^2 over `count` elements (unused in phase 2)
// `store` false: only the returned scale / shift (every thread of a one-workgroup unit evaluates them, thread 0 stores)
__device__ __forceinline__ void finalize_channel(const FinalizeArgs& a, int ch, double s1, double s2, double pivot,
                                                 bool store = true, float* sc_out = nullptr, float* sh_out = nullptr) {
    double count = a.count;
    if (a.phase == 2) {
        s1 = a.sync[ch * 4];
        s2 = a.sync[ch * 4 + 1];
        count = a.sync[ch * 4 + 2];
        pivot = 0.0;
    } else if (a.phase == 1) {                               // moments about zero: sum (a + p) and sum (a + p)^2
        a.sync[ch * 4] = s1 + count * pivot;
        a.sync[ch * 4 + 1] = s2 + 2.0 * pivot * s1 + count * pivot * pivot;
        a.sync[ch * 4 + 2] = count;
        a.sync[ch * 4 + 3] = 0.0;
        return;
    }
    const double m1 = s1 / count;
    const double mean = pivot + m1;
    double var = s2 / count - m1 * m1;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)a.eps);
    const float g = a.gamma ? a.gamma[ch] : 1.f, b = a.beta ? a.beta[ch] : 0.f;
    const float sc = g * (float)invstd;
    const float sh = b - (float)mean * sc;
    if (sc_out) { *sc_out = sc; *sh_out = sh; }
    if (!store) return;
    if (a.running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        a.running_mean[ch] = (float)((1.0 - a.momentum) * a.running_mean[ch] + a.momentum * mean);
        a.running_var[ch] = (float)((1.0 - a.momentum) * a.running_var[ch] + a.momentum * unbiased);
    }
    a.save_mean[ch] = (float)mean;
    a.save_invstd[ch] = (float)invstd;
    a.scale[ch] = sc;
    a.shift[ch] = sh;
}

// Small planes in the two reduce passes: a lane owns ONE quad of a plane, 256 / 2^l4 planes side by side, several such groups in
// flight.  Planes whose length is not a multiple of 4 (the 1-d model: 861, 215, 107, 53 ...) start at any 4-byte offset: their
// whole 16-byte quads begin `head` elements in, and two more lanes of the plane take the <= 3 elements in front of them and the
// <= 3 behind (one plane per trip and 53 of 256 lanes busy left the 215-pixel layers of cfg 3 at 0.5 - 1 TB/s).
struct QuadPos { int base, count; };            // element index of .x inside the plane; elements that exist (4: a whole quad)
__host__ __device__ inline int small_plane_lanes(long hw) { return (int)(hw >> 2) + ((hw & 3) ? 2 : 0); }
__host__ __device__ inline bool small_planes(long hw) { return hw <= 4 * kThreads && small_plane_lanes(hw) <= kThreads; }
__device__ __forceinline__ QuadPos quad_pos(long plane_off, int hw, int ti4) {
    const int hr = (int)((4 - (plane_off & 3)) & 3);
    const int head = hr < hw ? hr : hw;
    const int n4 = (hw - head) >> 2;
    if (ti4 < n4) return QuadPos{head + 4 * ti4, 4};
    if (ti4 == n4) return QuadPos{0, head};
    if (ti4 == n4 + 1) return QuadPos{head + 4 * n4, hw - head - 4 * n4};
    return QuadPos{0, 0};
}
__device__ __forceinline__ float4 load_quad(const float* plane, QuadPos q, float fill) {
    if (q.count == 4) return *reinterpret_cast<const float4*>(plane + q.base);
    float4 r = make_float4(fill, fill, fill, fill);
    if (q.count > 0) r.x = plane[q.base];
    if (q.count > 1) r.y = plane[q.base + 1];
    if (q.count > 2) r.z = plane[q.base + 2];
    return r;
}

__device__ __forceinline__ void store_quad(float* plane, QuadPos q, const float4& v) {
    if (q.count == 4) { *reinterpret_cast<float4*>(plane + q.base) = v; return; }
    if (q.count > 0) plane[q.base] = v.x;
    if (q.count > 1) plane[q.base + 1] = v.y;
    if (q.count > 2) plane[q.base + 2] = v.z;
}

// Larger planes: one plane at a time per workgroup, 16-byte loads behind an alignment peel.
// With `tickets` the workgroup that finishes a channel last also finalises it (stats_finalize_kernel's arithmetic in the same
// order; see bwd_partial_kernel): the single-replica statistics pass is one launch.
__global__ __launch_bounds__(kThreads) void stats_partial_kernel(
    const float* __restrict__ x, int n, int c, long hw, int nsplit, double* part, FinalizeArgs fa, unsigned* tickets) {
    __shared__ double scratch[kThreads / 64];
    const int ch = blockIdx.x, sp = blockIdx.y;
    const float pivot = x[(long)ch * hw];
    const int ti = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    float mn = pivot, mx = pivot;          // smallest / largest x of the channel (the L16 producers' operand bound)
    if (small_planes(hw)) {
        // four groups of planes in flight (one image per trip left the 208-pixel layers of cfg 2 at 1 TB/s).  Idle lanes and
        // trips past the batch read the pivot, which adds nothing to the shifted sums and lies inside [min, max].
        const int q4 = small_plane_lanes(hw);
        int l4 = 0;
        while ((1 << l4) < q4) ++l4;
        const int g4 = kThreads >> l4, tn4 = threadIdx.x >> l4, ti4 = threadIdx.x & ((1 << l4) - 1);
        const int stride = nsplit * g4;
        const float4 fill = make_float4(pivot, pivot, pivot, pivot);
        for (int b = sp * g4 + tn4; b < n; b += 4 * stride) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int bb = b + k * stride;
                const long off = ((long)bb * c + ch) * hw;
                v[k] = bb < n ? load_quad(x + off, quad_pos(off, (int)hw, ti4), pivot) : fill;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a0 = v[k].x - pivot, a1 = v[k].y - pivot, a2 = v[k].z - pivot, a3 = v[k].w - pivot;
                s1 += (a0 + a1) + (a2 + a3);
                s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                mn = fminf(fminf(mn, fminf(v[k].x, v[k].y)), fminf(v[k].z, v[k].w));
                mx = fmaxf(fmaxf(mx, fmaxf(v[k].x, v[k].y)), fmaxf(v[k].z, v[k].w));
            }
        }
    } else
    for (int b = sp; b < n; b += nsplit) {
        const float* p = x + ((long)b * c + ch) * hw;
        {
            const int head = (int)((4 - ((((long)b * c + ch) * hw) & 3)) & 3);
            const float4* p4 = reinterpret_cast<const float4*>(p + head);
            const long n4 = (hw - head) >> 2;
            if (ti < 8) {                                   // the <= 6 elements outside the whole quads
                const long e = ti < head ? (long)ti : head + 4 * n4 + (ti - head);
                if (e < hw && (ti < head || e >= head + 4 * n4)) {
                    const float a0 = p[e] - pivot;
                    s1 += a0;
                    s2 += a0 * a0;
                    mn = fminf(mn, p[e]);
                    mx = fmaxf(mx, p[e]);
                }
            }
            long i = ti;
            // four 16-byte loads in flight per thread (one reached 4.1 TB/s on the 704 MB tensors, short of the
            // ~5.5 TB/s of the apply passes)
            float t1[3] = {0.f, 0.f, 0.f}, t2[3] = {0.f, 0.f, 0.f};
            for (; i + 3 * kThreads < n4; i += 4 * kThreads) {
                const float4 v = p4[i];
                float4 u[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) u[k] = p4[i + (k + 1) * kThreads];
                const float a0 = v.x - pivot, a1 = v.y - pivot, a2 = v.z - pivot, a3 = v.w - pivot;
                s1 += (a0 + a1) + (a2 + a3);
                s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                mn = fminf(fminf(mn, fminf(v.x, v.y)), fminf(v.z, v.w));
                mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float b0 = u[k].x - pivot, b1 = u[k].y - pivot, b2 = u[k].z - pivot, b3 = u[k].w - pivot;
                    t1[k] += (b0 + b1) + (b2 + b3);
                    t2[k] += (b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3);
                    mn = fminf(fminf(mn, fminf(u[k].x, u[k].y)), fminf(u[k].z, u[k].w));
                    mx = fmaxf(fmaxf(mx, fmaxf(u[k].x, u[k].y)), fmaxf(u[k].z, u[k].w));
                }
            }
            for (; i + kThreads < n4; i += 2 * kThreads) {
                const float4 v = p4[i], u0 = p4[i + kThreads];
                const float a0 = v.x - pivot, a1 = v.y - pivot, a2 = v.z - pivot, a3 = v.w - pivot;
                const float b0 = u0.x - pivot, b1 = u0.y - pivot, b2 = u0.z - pivot, b3 = u0.w - pivot;
                s1 += (a0 + a1) + (a2 + a3);
                s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                t1[0] += (b0 + b1) + (b2 + b3);
                t2[0] += (b0 * b0 + b1 * b1) + (b2 * b2 + b3 * b3);
                mn = fminf(fminf(fminf(mn, fminf(v.x, v.y)), fminf(v.z, v.w)), fminf(fminf(u0.x, u0.y), fminf(u0.z, u0.w)));
                mx = fmaxf(fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w)), fmaxf(fmaxf(u0.x, u0.y), fmaxf(u0.z, u0.w)));
            }
            for (; i < n4; i += kThreads) {
                const float4 v = p4[i];
                const float a0 = v.x - pivot, a1 = v.y - pivot, a2 = v.z - pivot, a3 = v.w - pivot;
                s1 += (a0 + a1) + (a2 + a3);
                s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                mn = fminf(fminf(mn, fminf(v.x, v.y)), fminf(v.z, v.w));
                mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            }
            s1 += (t1[0] + t1[1]) + t1[2];
            s2 += (t2[0] + t2[1]) + t2[2];
        }
    }
    const double t1 = fsc::block_sum<double, kThreads / 64>((double)s1, scratch);
    const double t2 = fsc::block_sum<double, kThreads / 64>((double)s2, scratch);
    mx = fsc::wave_max(mx);
    mn = -fsc::wave_max(-mn);
    __shared__ float mm[2][kThreads / 64];
    if ((threadIdx.x & 63) == 0) { mm[0][threadIdx.x >> 6] = mn; mm[1][threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double* o = part + ((size_t)ch * kMaxSplit + sp) * kPartStride;
    for (int i = 1; i < kThreads / 64; ++i) { mn = fminf(mn, mm[0][i]); mx = fmaxf(mx, mm[1][i]); }
    const double vals[4] = {t1, t2, (double)mn, (double)mx};
    if (tickets == nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = vals[k];
        return;
    }
    if (nsplit == 1) {
        // the channel's only workgroup has nothing to hand over: no written-through partials, no ticket round trip, no re-load
        // (3 - 4 us of the 9 - 11 us such a launch takes on the 1-d model's last blocks); the same numbers as a fold of one split
        if (fa.x_minmax) {
            fa.x_minmax[2 * ch] = mn;
            fa.x_minmax[2 * ch + 1] = mx;
        }
        if (!fa.minmax_only) finalize_channel(fa, ch, 0.0 + t1, 0.0 + t2, (double)pivot);
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) __hip_atomic_store(o + k, vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!ticket_last(tickets + ch, nsplit)) return;
    double f1 = 0.0, f2 = 0.0;
    float fmn = INFINITY, fmx = -INFINITY;
    for (int k = 0; k < nsplit; ++k) {
        double* q = part + ((size_t)ch * kMaxSplit + k) * kPartStride;
        f1 += __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f2 += __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fmn = fminf(fmn, (float)__hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        fmx = fmaxf(fmx, (float)__hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (fa.x_minmax) {
        fa.x_minmax[2 * ch] = fmn;
        fa.x_minmax[2 * ch + 1] = fmx;
    }
    if (fa.minmax_only) return;
    finalize_channel(fa, ch, f1, f2, (double)pivot);
}

// HW == 1: x is (N, C); one thread per channel, coalesced across channels.
// blockIdx.y = one of gridDim.y row groups (split `sp` of the partials takes rows sp, sp + gridDim.y, ...): one thread walking all
// rows of its channel was a chain of 128 loads on 17 workgroups (34 us for 1 MB)
__host__ __device__ inline int rows_split(int n) { return n >= 64 ? 16 : n >= 16 ? 4 : 1; }

__global__ void stats_rows_kernel(const float* __restrict__ x, int n, int c, double* __restrict__ part) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const float pivot = x[ch];
    double s1 = 0.0, s2 = 0.0;
    float mn = pivot, mx = pivot;
    for (int b = blockIdx.y; b < n; b += gridDim.y) {
        const float v = x[(long)b * c + ch];
        const double a = (double)(v - pivot);
        s1 += a;
        s2 += a * a;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    double* o = part + ((size_t)ch * kMaxSplit + blockIdx.y) * kPartStride;
    o[0] = s1;
    o[1] = s2;
    o[2] = (double)mn;
    o[3] = (double)mx;
}

// Cross-replica statistics (SyncBN): `sync` holds per channel [sum x, sum x^2, count, -] about ZERO in
// fp64.  phase 1 stops after writing the local moments there; the caller sum-all-reduces the buffer; phase 2
// finalises from it.  phase 0 is the single-replica path (local pivot-shifted sums, no detour through `sync`).
__global__ void stats_finalize_kernel(FinalizeArgs a) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= a.c) return;
    const double* part = a.part;
    if (a.x_minmax && a.phase != 2) {      // (phase 2 re-runs on the partials of phase 1: already written)
        float mn = (float)part[(size_t)ch * kMaxSplit * kPartStride + 2], mx = (float)part[(size_t)ch * kMaxSplit * kPartStride + 3];
        for (int s = 1; s < a.nsplit; ++s) {
            mn = fminf(mn, (float)part[((size_t)ch * kMaxSplit + s) * kPartStride + 2]);
            mx = fmaxf(mx, (float)part[((size_t)ch * kMaxSplit + s) * kPartStride + 3]);
        }
        a.x_minmax[2 * ch] = mn;
        a.x_minmax[2 * ch + 1] = mx;
    }
    if (a.minmax_only) return;             // (inference: the BatchNorm runs on its running statistics)
    double s1 = 0.0, s2 = 0.0, pivot = 0.0;
    if (a.phase != 2) {
        for (int s = 0; s < a.nsplit; ++s) {
            s1 += part[((size_t)ch * kMaxSplit + s) * kPartStride];
            s2 += part[((size_t)ch * kMaxSplit + s) * kPartStride + 1];
        }
        // pivot of the shifted sums: the channel's first element, or (sums from a STATS convolution) the running mean as it is
        // BEFORE this call updates it -- 0 without running statistics
        pivot = a.pivot_rm ? (a.running_mean ? (double)a.running_mean[ch] : 0.0) : (double)a.x[(long)ch * a.hw];
    }
    finalize_channel(a, ch, s1, s2, pivot);
}

// The same for sums that a STATS convolution took about the BatchNorm's RUNNING mean (pivot_rm, split 0 only; phases 0 and 1),
// one workgroup per channel.  Those sums are fp32 per lane: when the batch mean lies far from the pivot (the first step after
// loading a checkpoint from another domain, a large conv bias under running_mean = 0) sum (y - p)^2 - (sum (y - p))^2 / n
// cancels and the variance loses (mean - p)^2 / var of its digits.  Gate: |mean - p| > kGateSigmas standard deviations (or a
// non-positive variance estimate) -> the workgroup re-reduces its channel of x about the (accurate) mean estimate, in fp64, and
// finalises from that.  The min / max of the records are exact either way.
constexpr double kGateSigmas = 4.0;
__global__ __launch_bounds__(kThreads) void stats_finalize_gated_kernel(FinalizeArgs a, int n) {
    __shared__ double scratch[kThreads / 64];
    const int ch = blockIdx.x;
    const double* part = a.part + (size_t)ch * kMaxSplit * kPartStride;
    if (a.x_minmax && threadIdx.x == 0) {
        a.x_minmax[2 * ch] = (float)part[2];
        a.x_minmax[2 * ch + 1] = (float)part[3];
    }
    double s1 = part[0], s2 = part[1];
    double pivot = a.running_mean ? (double)a.running_mean[ch] : 0.0;
    const double m1 = s1 / a.count, var = s2 / a.count - m1 * m1;
    if (!(var > 0.0) || m1 * m1 > kGateSigmas * kGateSigmas * var) {      // (workgroup-uniform)
        const double p2 = pivot + m1;
        const float p2f = (float)p2;
        double t1 = 0.0, t2 = 0.0;
        for (int b = 0; b < n; ++b) {
            const float* px = a.x + ((long)b * a.c + ch) * a.hw;
            for (long i = threadIdx.x; i < a.hw; i += kThreads) {
                const double d = (double)(px[i] - p2f);
                t1 += d;
                t2 += d * d;
            }
        }
        __syncthreads();
        s1 = fsc::block_sum<double, kThreads / 64>(t1, scratch);
        s2 = fsc::block_sum<double, kThreads / 64>(t2, scratch);
        pivot = (double)p2f;
    }
    if (threadIdx.x == 0) finalize_channel(a, ch, s1, s2, pivot);
}

__global__ void eval_prepare_kernel(int c, const float* gamma, const float* beta, const float* rm,
                                    const float* rv, float eps, float* scale, float* shift) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const float invstd = 1.0f / sqrtf(rv[ch] + eps);
    const float sc = (gamma ? gamma[ch] : 1.f) * invstd;
    scale[ch] = sc;
    shift[ch] = (beta ? beta[ch] : 0.f) - rm[ch] * sc;
}

// ---------------------------------------------------------------- forward apply
__device__ __forceinline__ float act(float z, float alpha, bool has_alpha) {
    return (has_alpha && !(z > 0.f)) ? alpha * z : z;
}

// one (n, c) plane per blockIdx.x, blockIdx.y strides over the plane
__global__ __launch_bounds__(kThreads) void fwd_plane_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ alpha, float* __restrict__ y, int c, long hw,
    float* y_amax) {
    const long plane = blockIdx.x;
    float mx = 0.f;
    const int ch = (int)(plane % c);
    const float sc = scale[ch], sh = shift[ch];
    const bool has_alpha = alpha != nullptr;
    const float al = has_alpha ? alpha[ch] : 0.f;
    const float* px = x + plane * hw;
    const float* pr = res ? res + plane * hw : nullptr;
    float* py = y + plane * hw;
    // 16-byte accesses for any plane length: `head` elements up to the first 16-byte boundary of the plane (planes of odd length
    // -- the 1-d model's 1723, 861, ... frames -- start at any 4-byte offset), whole quads, then the tail; the <= 6 stragglers are
    // done by the first threads of slice 0
    const int head = (int)((4 - ((plane * hw) & 3)) & 3);
    const long n4 = (hw - head) >> 2;
    auto point = [&](long i) {
        float z = fmaf(px[i], sc, sh);
        if (pr) z += pr[i];
        z = act(z, al, has_alpha);
        py[i] = z;
        mx = fmaxf(mx, fabsf(z));
    };
    for (long i = (long)blockIdx.y * kThreads + threadIdx.x; i < n4; i += (long)gridDim.y * kThreads) {
        float4 v = reinterpret_cast<const float4*>(px + head)[i];
        float4 z = make_float4(fmaf(v.x, sc, sh), fmaf(v.y, sc, sh), fmaf(v.z, sc, sh), fmaf(v.w, sc, sh));
        if (pr) {
            const float4 r = reinterpret_cast<const float4*>(pr + head)[i];
            z.x += r.x; z.y += r.y; z.z += r.z; z.w += r.w;
        }
        z.x = act(z.x, al, has_alpha); z.y = act(z.y, al, has_alpha);
        z.z = act(z.z, al, has_alpha); z.w = act(z.w, al, has_alpha);
        reinterpret_cast<float4*>(py + head)[i] = z;
        mx = fmaxf(fmaxf(mx, fabsf(z.x)), fmaxf(fabsf(z.y), fmaxf(fabsf(z.z), fabsf(z.w))));
    }
    if (blockIdx.y == 0 && threadIdx.x < 8) {
        const long t = threadIdx.x;
        if (t < head) point(t);
        else if (head + 4 * n4 + (t - head) < hw) point(head + 4 * n4 + (t - head));
    }
    if (y_amax) fsc::publish_amax(y_amax, mx);
}

// Planes of 2..511 pixels: a group of 2^glog <= 64 lanes per (n, c) plane, 64 >> glog consecutive planes per wavefront (a
// 2 x 6 plane of the last cfg-2 block on a whole wave left 52 lanes idle: 97 k waves for 4.7 MB, 55 us for the backward apply).
__host__ __device__ inline int group_log2(long hw) {
    int l = 2;
    while (l < 6 && (1L << l) < hw) ++l;
    return l;
}
template <typename T>
__device__ __forceinline__ T group_sum(T v, int glog) {       // sum over the lane's group, valid in every lane of it
    for (int o = (1 << glog) >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(kThreads) void fwd_wave_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ alpha, float* __restrict__ y, int c, long hw,
    long planes, float* y_amax, int glog) {
    const int gsz = 1 << glog;
    long plane = ((long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * (64 >> glog) + ((threadIdx.x & 63) >> glog);
    const long hw_live = plane < planes ? hw : 0;        // (no early exit: publish_amax synchronises the block)
    if (plane >= planes) plane = planes - 1;
    const int lane = threadIdx.x & (gsz - 1);
    const int ch = (int)(plane % c);
    const float sc = scale[ch], sh = shift[ch];
    const bool has_alpha = alpha != nullptr;
    const float al = has_alpha ? alpha[ch] : 0.f;
    const long base = plane * hw;
    float mx = 0.f;
    for (long i = lane; i < hw_live; i += gsz) {
        float z = fmaf(x[base + i], sc, sh);
        if (res) z += res[base + i];
        z = act(z, al, has_alpha);
        y[base + i] = z;
        mx = fmaxf(mx, fabsf(z));
    }
    if (y_amax) fsc::publish_amax(y_amax, mx);
}

// ---------------------------------------------------------------- forward apply that leaves records
// The output of a block's last unit is read twice more right away: by the statistics pass of the NEXT block's input
// BatchNorm (classifiers.py:524) and by the global max-pool of the hierarchical head (classifiers.py:586-590).  These
// variants of the plane / wave kernels reduce what both need while they hold the values: per (plane, slice) the sums of
// (y - pivot), (y - pivot)^2 with the pivot of stats_partial_kernel (y at image 0, position 0 of the channel -- recomputed
// from x by every block with the same instruction sequence as the element that is stored), min / max, and the (value,
// index) key of pool.hip's global max.  fold_records_kernel turns them into the statistics partials and the pooled values.
struct PlaneRec { double s1, s2; float mn, mx; unsigned long long key; };

__device__ __forceinline__ unsigned long long rec_key(float v, unsigned idx) {      // (= pool.hip gmax_key)
    unsigned u;
    if (v != v) {
        u = 0xFFFFFFFFu;
    } else {
        if (v == 0.f) v = 0.f;
        u = __float_as_uint(v);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    }
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

__device__ __forceinline__ float bn_point(float x, float r, bool has_res, float sc, float sh, float al, bool has_alpha) {
    float z = fmaf(x, sc, sh);
    if (has_res) z += r;
    return act(z, al, has_alpha);
}

struct RecAcc {
    float s1 = 0.f, s2 = 0.f, mn, mx;
    unsigned long long key = 0ull;
    float pivot;
    __device__ __forceinline__ void init(float p) { pivot = p; mn = p; mx = p; }
    __device__ __forceinline__ void add(float z, unsigned idx) {
        const float a = z - pivot;
        s1 += a;
        s2 += a * a;
        mn = fminf(mn, z);
        mx = fmaxf(mx, z);
        const unsigned long long k = rec_key(z, idx);
        key = k > key ? k : key;
    }
    // fold over aligned groups of 2 * top lanes (the whole wave by default); every lane of a group holds the result
    __device__ __forceinline__ void wave_fold(double& d1, double& d2, int top = 32) {
        d1 = (double)s1; d2 = (double)s2;
        for (int o = top; o > 0; o >>= 1) {
            d1 += __shfl_xor(d1, o, 64);
            d2 += __shfl_xor(d2, o, 64);
            mn = fminf(mn, __shfl_xor(mn, o, 64));
            mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            const unsigned long long other = __shfl_xor(key, o, 64);
            key = other > key ? other : key;
        }
    }
};

__global__ __launch_bounds__(kThreads) void fwd_plane_rec_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ alpha, float* __restrict__ y, int c, long hw,
    PlaneRec* __restrict__ rec) {
    __shared__ PlaneRec fold[kThreads / 64];
    const long plane = blockIdx.x;
    const int ch = (int)(plane % c);
    const float sc = scale[ch], sh = shift[ch];
    const bool has_alpha = alpha != nullptr, has_res = res != nullptr;
    const float al = has_alpha ? alpha[ch] : 0.f;
    const float* px = x + plane * hw;
    const float* pr = has_res ? res + plane * hw : nullptr;
    float* py = y + plane * hw;
    RecAcc acc;
    acc.init(bn_point(x[(long)ch * hw], has_res ? res[(long)ch * hw] : 0.f, has_res, sc, sh, al, has_alpha));
    if ((hw & 3) == 0) {
        const long n4 = hw >> 2;
        for (long i = (long)blockIdx.y * kThreads + threadIdx.x; i < n4; i += (long)gridDim.y * kThreads) {
            const float4 v = reinterpret_cast<const float4*>(px)[i];
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (has_res) r = reinterpret_cast<const float4*>(pr)[i];
            float4 z;
            z.x = bn_point(v.x, r.x, has_res, sc, sh, al, has_alpha);
            z.y = bn_point(v.y, r.y, has_res, sc, sh, al, has_alpha);
            z.z = bn_point(v.z, r.z, has_res, sc, sh, al, has_alpha);
            z.w = bn_point(v.w, r.w, has_res, sc, sh, al, has_alpha);
            reinterpret_cast<float4*>(py)[i] = z;
            const unsigned i0 = (unsigned)(i * 4);
            acc.add(z.x, i0); acc.add(z.y, i0 + 1); acc.add(z.z, i0 + 2); acc.add(z.w, i0 + 3);
        }
    } else {
        for (long i = (long)blockIdx.y * kThreads + threadIdx.x; i < hw; i += (long)gridDim.y * kThreads) {
            const float z = bn_point(px[i], has_res ? pr[i] : 0.f, has_res, sc, sh, al, has_alpha);
            py[i] = z;
            acc.add(z, (unsigned)i);
        }
    }
    double d1, d2;
    acc.wave_fold(d1, d2);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) fold[wid] = PlaneRec{d1, d2, acc.mn, acc.mx, acc.key};
    __syncthreads();
    if (threadIdx.x == 0) {
        PlaneRec o = fold[0];
#pragma unroll
        for (int k = 1; k < kThreads / 64; ++k) {
            o.s1 += fold[k].s1; o.s2 += fold[k].s2;
            o.mn = fminf(o.mn, fold[k].mn); o.mx = fmaxf(o.mx, fold[k].mx);
            o.key = fold[k].key > o.key ? fold[k].key : o.key;
        }
        rec[plane * gridDim.y + blockIdx.y] = o;
    }
}

// planes of 2..511 pixels: one wavefront per plane, one record per plane
__global__ __launch_bounds__(kThreads) void fwd_wave_rec_kernel(
    const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ alpha, float* __restrict__ y, int c, long hw,
    long planes, PlaneRec* __restrict__ rec, int glog) {
    const int gsz = 1 << glog;
    long plane = ((long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * (64 >> glog) + ((threadIdx.x & 63) >> glog);
    const long hw_live = plane < planes ? hw : 0;        // (a dead group of a live wave still takes part in the fold's shuffles)
    const bool dead = plane >= planes;
    if (dead) plane = planes - 1;
    const int lane = threadIdx.x & (gsz - 1);
    const int ch = (int)(plane % c);
    const float sc = scale[ch], sh = shift[ch];
    const bool has_alpha = alpha != nullptr, has_res = res != nullptr;
    const float al = has_alpha ? alpha[ch] : 0.f;
    const long base = plane * hw;
    RecAcc acc;
    acc.init(bn_point(x[(long)ch * hw], has_res ? res[(long)ch * hw] : 0.f, has_res, sc, sh, al, has_alpha));
    for (long i = lane; i < hw_live; i += gsz) {
        const float z = bn_point(x[base + i], has_res ? res[base + i] : 0.f, has_res, sc, sh, al, has_alpha);
        y[base + i] = z;
        acc.add(z, (unsigned)i);
    }
    double d1, d2;
    acc.wave_fold(d1, d2, gsz >> 1);
    if (lane == 0 && !dead) rec[plane] = PlaneRec{d1, d2, acc.mn, acc.mx, acc.key};
}

// one workgroup per channel: a thread folds the slices of one image's plane (-> the global max of that plane), the block
// folds the images (-> split 0 of the statistics partials, what stats_finalize_kernel reads with nsplit = 1)
__global__ __launch_bounds__(kThreads) void fold_records_kernel(const PlaneRec* __restrict__ rec, int slices, int n, int c, long hw,
                                                                const float* __restrict__ y, double* __restrict__ part,
                                                                float* __restrict__ gmax, int* __restrict__ gmax_idx) {
    __shared__ double scratch[kThreads / 64];
    __shared__ float mm[2][kThreads / 64];
    const int ch = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    float mn = rec[(long)ch * slices].mn, mx = mn;          // (image 0: a value of the channel)
    for (int b = threadIdx.x; b < n; b += kThreads) {
        const long plane = (long)b * c + ch;
        const PlaneRec* r = rec + plane * slices;
        unsigned long long key = 0ull;
        for (int j = 0; j < slices; ++j) {
            s1 += r[j].s1; s2 += r[j].s2;
            mn = fminf(mn, r[j].mn); mx = fmaxf(mx, r[j].mx);
            key = r[j].key > key ? r[j].key : key;
        }
        if (gmax) {
            const unsigned ii = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
            gmax[plane] = y[plane * hw + ii];
            gmax_idx[plane] = (int)ii;
        }
    }
    if (!part) return;
    const double t1 = fsc::block_sum<double, kThreads / 64>(s1, scratch);
    const double t2 = fsc::block_sum<double, kThreads / 64>(s2, scratch);
    mx = fsc::wave_max(mx);
    mn = -fsc::wave_max(-mn);
    if ((threadIdx.x & 63) == 0) { mm[0][threadIdx.x >> 6] = mn; mm[1][threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kThreads / 64; ++i) { mn = fminf(mn, mm[0][i]); mx = fmaxf(mx, mm[1][i]); }
        double* o = part + (size_t)ch * kMaxSplit * kPartStride;
        o[0] = t1; o[1] = t2; o[2] = (double)mn; o[3] = (double)mx;
    }
}

// records of a STATS convolution (conv_l16.hip): float4 {s1, s2, min, max} at [(worker * 8 + wave) * co_blk + channel in block],
// worker w holds channel block w % blocks.  One workgroup per channel -> split 0 of the statistics partials.
// order 0 / 1: record of (slot r = worker * 8 + wave, channel) at [r][channel in block], worker w holds channel block w % blocks
// (order 1: (w / 8) % blocks); order 2 (the fp32-input ring kernels): TRANSPOSED -- [channel][r], w % blocks -- so that the
// channel's workgroup below reads 2048 contiguous records instead of one per cache line (9.7 us per fold for 32 KB: the 64 - 125
// workgroups of a fold touched 16 - 32 MB of lines).
__device__ __forceinline__ bool rec_mine(int r, int blocks, int order, int cb) {
    const int w = r >> 3;
    return (order == 1 ? (w >> 3) % blocks : w % blocks) == cb;
}
__device__ __forceinline__ long rec_index(int r, int ch, int within, int co_blk, int nrec, int order) {
    return order == 2 ? (long)ch * nrec + r : (long)r * co_blk + within;
}

__global__ __launch_bounds__(kThreads) void fold_conv_records_kernel(const float4* __restrict__ rec, int workers, int blocks, int co_blk,
                                                                     int order, double* __restrict__ part) {
    __shared__ double scratch[kThreads / 64];
    __shared__ float mm[2][kThreads / 64];
    const int ch = blockIdx.x, cb = ch / co_blk, within = ch - cb * co_blk;
    const int nrec = workers * 8;
    double s1 = 0.0, s2 = 0.0;
    float mn = INFINITY, mx = -INFINITY;
    for (int r = threadIdx.x; r < nrec; r += kThreads) {
        if (!rec_mine(r, blocks, order, cb)) continue;                        // (another channel block's worker)
        const float4 v = rec[rec_index(r, ch, within, co_blk, nrec, order)];
        s1 += (double)v.x; s2 += (double)v.y;
        mn = fminf(mn, v.z); mx = fmaxf(mx, v.w);
    }
    const double t1 = fsc::block_sum<double, kThreads / 64>(s1, scratch);
    const double t2 = fsc::block_sum<double, kThreads / 64>(s2, scratch);
    mx = fsc::wave_max(mx);
    mn = -fsc::wave_max(-mn);
    if ((threadIdx.x & 63) == 0) { mm[0][threadIdx.x >> 6] = mn; mm[1][threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kThreads / 64; ++i) { mn = fminf(mn, mm[0][i]); mx = fmaxf(mx, mm[1][i]); }
        double* o = part + (size_t)ch * kMaxSplit * kPartStride;
        o[0] = t1; o[1] = t2; o[2] = (double)mn; o[3] = (double)mx;
    }
}

// fold_conv_records_kernel + stats_finalize_gated_kernel in one launch (single replica, training): the workgroup of a channel folds
// the channel's records of the STATS convolution and finalises straight away (fsc_bn_train_stats_conv).
__global__ __launch_bounds__(kThreads) void stats_fold_finalize_kernel(FinalizeArgs a, int n, const float4* __restrict__ rec, int workers,
                                                                        int blocks, int co_blk, int order) {
    __shared__ double scratch[kThreads / 64];
    __shared__ float mm[2][kThreads / 64];
    __shared__ double bc[2];
    const int ch = blockIdx.x, cb = ch / co_blk, within = ch - cb * co_blk;
    const int nrec = workers * 8;
    double s1 = 0.0, s2 = 0.0;
    float mn = INFINITY, mx = -INFINITY;
    // (eight record loads in flight per thread -- 2048 records are eight per thread, each on its own cache line --, then ONE LDS
    // exchange for both sums and the extremes: the launch is a latency chain, 9.5 us for 32 KB)
    const float4 none = make_float4(0.f, 0.f, INFINITY, -INFINITY);
    for (int r0 = threadIdx.x; r0 < nrec; r0 += 8 * kThreads) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u * kThreads;
            const bool mine = r < nrec && rec_mine(r, blocks, order, cb);
            v[u] = mine ? rec[rec_index(r, ch, within, co_blk, nrec, order)] : none;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s1 += (double)v[u].x; s2 += (double)v[u].y;
            mn = fminf(mn, v[u].z); mx = fmaxf(mx, v[u].w);
        }
    }
    s1 = fsc::wave_sum(s1);
    s2 = fsc::wave_sum(s2);
    mx = fsc::wave_max(mx);
    mn = -fsc::wave_max(-mn);
    __shared__ double sd2[2][kThreads / 64];
    if ((threadIdx.x & 63) == 0) {
        sd2[0][threadIdx.x >> 6] = s1; sd2[1][threadIdx.x >> 6] = s2;
        mm[0][threadIdx.x >> 6] = mn; mm[1][threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    double t1 = sd2[0][0], t2 = sd2[1][0];
#pragma unroll
    for (int i = 1; i < kThreads / 64; ++i) { t1 += sd2[0][i]; t2 += sd2[1][i]; }
    if (threadIdx.x == 0) {
        mn = mm[0][0]; mx = mm[1][0];
        for (int i = 1; i < kThreads / 64; ++i) { mn = fminf(mn, mm[0][i]); mx = fmaxf(mx, mm[1][i]); }
        if (a.x_minmax) { a.x_minmax[2 * ch] = mn; a.x_minmax[2 * ch + 1] = mx; }
    }
    if (a.minmax_only) return;
    double pivot = a.running_mean ? (double)a.running_mean[ch] : 0.0;
    const double m1 = t1 / a.count, var = t2 / a.count - m1 * m1;
    if (!(var > 0.0) || m1 * m1 > kGateSigmas * kGateSigmas * var) {      // (workgroup-uniform: see stats_finalize_gated_kernel)
        const double p2 = pivot + m1;
        const float p2f = (float)p2;
        double u1 = 0.0, u2 = 0.0;
        for (int b = 0; b < n; ++b) {
            const float* px = a.x + ((long)b * a.c + ch) * a.hw;
            for (long i = threadIdx.x; i < a.hw; i += kThreads) {
                const double d = (double)(px[i] - p2f);
                u1 += d;
                u2 += d * d;
            }
        }
        __syncthreads();
        t1 = fsc::block_sum<double, kThreads / 64>(u1, scratch);
        t2 = fsc::block_sum<double, kThreads / 64>(u2, scratch);
        pivot = (double)p2f;
    }
    if (threadIdx.x == 0) finalize_channel(a, ch, t1, t2, pivot);
    (void)bc;
}

// hw == 1 (BatchNorm1d on (N, C)): flat indexing
__global__ void fwd_flat_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                const float* __restrict__ scale, const float* __restrict__ shift,
                                const float* __restrict__ alpha, float* __restrict__ y, int c, long hw, long total,
                                float* y_amax) {
    float mx = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)((i / hw) % c);
        float z = fmaf(x[i], scale[ch], shift[ch]);
        if (res) z += res[i];
        z = (alpha && !(z > 0.f)) ? alpha[ch] * z : z;
        y[i] = z;
        mx = fmaxf(mx, fabsf(z));
    }
    if (y_amax) fsc::publish_amax(y_amax, mx);
}


// ---------------------------------------------------------------- L16 producers (include/fsc_hip.h "pre-split activations")
// The conv kernels of conv_l16.hip read activations as two scaled fp16 limbs, 8 channels of a position per 16 bytes.
// The kernels below write that layout directly: a thread owns VEC consecutive positions x the 8 channels of one
// (octet, image), reads the 8 fp32 channel planes (coalesced along positions) and stores VEC x 2 x 16 bytes.  The scale
// needs the tensor's largest magnitude BEFORE the pass: the forward bound comes from the per-channel min / max of x the
// statistics pass reports (affine + PReLU take their extremes at the ends of the range, so the bound is the exact
// maximum), the backward bound from per-channel maxima of the reduce pass (see bwd_finalize_kernel).  Every block folds
// the per-channel bounds itself; block (0, 0) publishes the value as the tensor's FSC_AMAX_FLOATS buffer.
// UNI: planes of >= 1024 positions, one (octet, image) per block (per-channel constants in scalar registers); otherwise
// the 256 threads split into (group, position lane) and keep their constants in vector registers.
__device__ __forceinline__ float block_max256(float m, float* red) {
    m = fsc::wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return m;
}

// Stores of a thread's VEC = 4 consecutive positions (hi[p], lo[p] = the two 16-byte limb vectors of position 4 q + p).
// Written directly, a store instruction puts 16 bytes into every fourth 16-byte slot (lane stride 64 bytes): quarter-filled
// lines, 4x the write requests.  Instead each wave transposes through its own LDS patch (80-byte lane pitch: conflict-free
// 16-byte writes) so that instruction k stores positions 64 k + lane of the wave's 256 -- 1 KB contiguous.  Needs the wave's
// lanes on consecutive quads (position lanes >= 64) and EVERY lane of the wave calling (dead lanes pass anything).
// NL = the L16 format code (l16.h): 2 (scaled fp16 pairs), 3 (exact bf16 triples) or 4 (scaled fp16 triples); lv[p][limb]; limb
// planes hw apart from out0.
#define FSC_NLIMBS(NL) l16::fmt_limbs(NL)
template <int NL>
__device__ __forceinline__ void l16_split(const float (&v8)[8], float s, uint4 (&out)[FSC_NLIMBS(NL)]) {
    if constexpr (NL == 2) l16::split8(v8, s, out[0], out[1]);
    else if constexpr (NL == 4) l16::split8_f3(v8, s, out[0], out[1], out[2]);
    else l16::split8_bf3(v8, out[0], out[1], out[2]);
}
template <int NL>
__device__ __forceinline__ void l16_store_quads(uint4* __restrict__ out0, long q, long hw, const uint4 (&lv)[4][FSC_NLIMBS(NL)], uint4* wave_patch) {
    const int lane = threadIdx.x & 63;
    const long p0 = 4 * (q - lane);                          // first position of the wave's 256
#pragma unroll
    for (int limb = 0; limb < FSC_NLIMBS(NL); ++limb) {
#pragma unroll
        for (int p = 0; p < 4; ++p) wave_patch[lane * 5 + p] = lv[p][limb];
        uint4* out = out0 + limb * hw;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pp = k * 64 + lane;
            const uint4 v = wave_patch[(pp >> 2) * 5 + (pp & 3)];
            if (p0 + pp < hw) out[p0 + pp] = v;
        }
    }
}

template <int VEC, bool UNI, int NL>
__global__ __launch_bounds__(kThreads) void fwd_l16_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ alpha, const float* __restrict__ x_minmax, float* __restrict__ y,
    uint4* __restrict__ y16, float* __restrict__ y_amax, int n, int c, long hw, int hwp_log2) {
    __shared__ float red[kThreads / 64];
    const bool has_alpha = alpha != nullptr;
    float s = 1.f;
    if constexpr (l16::fmt_scaled(NL)) {                                 // (bf16 limbs carry the fp32 exponent: no scale, no declared maximum)
        float m = 0.f;
        for (int ch = threadIdx.x; ch < c; ch += kThreads) {
            const float sc = scale[ch], sh = shift[ch], al = has_alpha ? alpha[ch] : 0.f;
            const float lo = act(fmaf(x_minmax[2 * ch], sc, sh), al, has_alpha), hi = act(fmaf(x_minmax[2 * ch + 1], sc, sh), al, has_alpha);
            m = fmaxf(m, fmaxf(fabsf(lo), fabsf(hi)));
        }
        m = block_max256(m, red);
        if (blockIdx.x == 0 && blockIdx.y == 0) l16::store_amax(y_amax, m);
        s = l16::field_to_float(l16::scale_field(m));
    }
    const int oct = (c + 7) >> 3;
    const int hwp = UNI ? kThreads : 1 << hwp_log2, groups = UNI ? 1 : kThreads >> hwp_log2;
    const int tn = UNI ? 0 : threadIdx.x >> hwp_log2, ti = UNI ? threadIdx.x : threadIdx.x & (hwp - 1);
    const long g = (long)blockIdx.x * groups + tn;            // (octet, image), images fastest
    if (g >= (long)n * oct) return;
    const int o = (int)(g / n), img = (int)(g - (long)o * n);
    float sc[8], sh[8], al[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = o * 8 + e;
        const bool live = ch < c;
        sc[e] = live ? scale[ch] : 0.f;
        sh[e] = live ? shift[ch] : 0.f;
        al[e] = live && has_alpha ? alpha[ch] : 0.f;
    }
    const long nq = hw / VEC;
    const long xbase = ((long)img * c + o * 8) * hw;
    uint4* const out0 = y16 + (((long)img * oct + o) * FSC_NLIMBS(NL)) * hw;
    __shared__ uint4 patch[kThreads / 64][64 * 5];
    const bool transpose = VEC == 4 && hwp >= 64;            // (uniform)
    const int lane = threadIdx.x & 63;
    // with the transposed stores the loop is wave-uniform: a lane past the plane still takes part in the wave's stores
    for (long q = (long)blockIdx.y * hwp + ti; transpose ? q - lane < nq : q < nq; q += (long)gridDim.y * hwp) {
        const bool q_live = q < nq;
        float z[8][VEC];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool live = o * 8 + e < c && q_live;
            if (VEC == 4) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) v = reinterpret_cast<const float4*>(x + xbase + e * hw)[q];
                z[e][0] = act(fmaf(v.x, sc[e], sh[e]), al[e], has_alpha);
                z[e][1 % VEC] = act(fmaf(v.y, sc[e], sh[e]), al[e], has_alpha);
                z[e][2 % VEC] = act(fmaf(v.z, sc[e], sh[e]), al[e], has_alpha);
                z[e][3 % VEC] = act(fmaf(v.w, sc[e], sh[e]), al[e], has_alpha);
                if (y && live) reinterpret_cast<float4*>(y + xbase + e * hw)[q] = make_float4(z[e][0], z[e][1 % VEC], z[e][2 % VEC], z[e][3 % VEC]);
            } else {
                const float v = live ? x[xbase + e * hw + q] : 0.f;
                z[e][0] = act(fmaf(v, sc[e], sh[e]), al[e], has_alpha);
                if (y && live) y[xbase + e * hw + q] = z[e][0];
            }
        }
        uint4 lv[VEC][FSC_NLIMBS(NL)];
#pragma unroll
        for (int p = 0; p < VEC; ++p) {
            const float v8[8] = {z[0][p], z[1][p], z[2][p], z[3][p], z[4][p], z[5][p], z[6][p], z[7][p]};
            l16_split<NL>(v8, s, lv[p]);
        }
        if constexpr (VEC == 4) {
            if (transpose) {
                l16_store_quads<NL>(out0, q, hw, lv, patch[threadIdx.x >> 6]);
                continue;
            }
        }
#pragma unroll
        for (int p = 0; p < VEC; ++p)
#pragma unroll
            for (int l = 0; l < FSC_NLIMBS(NL); ++l) out0[l * hw + q * VEC + p] = lv[p][l];
    }
}

// ---------------------------------------------------------------- backward
struct BwdArgs {
    const float* dy;        // may be null (treated as zero)
    const float* gmax_dy;   // (N*C) gradient of a global-max-pool head on the output, may be null
    const int* gmax_idx;    // (N*C) argmax inside the plane
    const float* x;
    const float* res;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    const float* alpha;
    int n, c;
    long hw;
};

__device__ __forceinline__ float upstream(const BwdArgs& a, const float* pdy, long i, float gval, long gpos) {
    float g = pdy ? pdy[i] : 0.f;
    if (i == gpos) g += gval;
    return g;
}

// The per-channel sum of dx (the bias gradient of the convolution in front of this BatchNorm) in closed form: with
// dx = k (dz - c1 - xhat c2) over `count` elements, sum dx = k (sum dz - count c1 - c2 sum xhat), from the sums the reduce pass has
// anyway (+ sum xhat).  Analytically zero (c1 = mean dz, sum xhat = 0): like the sum of the stored dx it is rounding residue, but
// no apply kernel has to add its planes into C addresses (one device-scope atomic per plane, 128 per address at batch 128, cost the
// apply pass of the 1-d model's planes 40 - 77 us per call: tools/bn_atomics_probe.py).
__host__ __device__ inline float chan_sum_of_dx(float k, double sum_dz, double sum_xhat, double count, float c1, float c2) {
    return (float)((double)k * (sum_dz - count * (double)c1 - (double)c2 * sum_xhat));
}
constexpr int kBwdVals = 6;
// partial sums per (channel, split): [0]=sum dz, [1]=sum dz*xhat, [2]=sum dy*min(z,0), [3]=max|dz|, [4]=max|xhat|, [5]=sum xhat
// With fin.tickets the block that finishes a channel LAST (a ticket counter per channel, left at zero again) also does what
// bwd_finalize_kernel does for that channel, in the same order of additions: the single-replica backward is two launches, not three.
struct BwdFinish {
    unsigned* tickets;      // null: bwd_finalize_kernel follows
    double count;
    float* dgamma; float* dbeta; float* dalpha; float* coef; float* dx_chan_sum; float* dx_amax;
    int want_bound;
};

__global__ __launch_bounds__(kThreads) void bwd_partial_kernel(BwdArgs a, int nsplit, double* part, BwdFinish fin) {
    __shared__ double scratch[kThreads / 64];
    const int ch = blockIdx.x, sp = blockIdx.y;
    const float mean = a.mean[ch], invstd = a.invstd[ch];
    const float g = a.gamma ? a.gamma[ch] : 1.f, b = a.beta ? a.beta[ch] : 0.f;
    const bool has_alpha = a.alpha != nullptr;
    const float al = has_alpha ? a.alpha[ch] : 1.f;
    const int ti = threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    float mdz = 0.f, mxh = 0.f;            // max |dz|, max |xhat|: the bound of |dx| for the L16 apply pass
    auto quad = [&](const float4& xv, const float4& rv, const float4& uv) {
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, rs[4] = {rv.x, rv.y, rv.z, rv.w}, us[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xs[e] - mean) * invstd;
            const float z = fmaf(xh, g, b) + rs[e];
            const bool neg = has_alpha && !(z > 0.f);
            const float dz = neg ? al * us[e] : us[e];
            s0 += dz;
            s1 += dz * xh;
            s2 += us[e] * (neg ? z : 0.f);
            s3 += xh;
            mdz = fmaxf(mdz, fabsf(dz));
            mxh = fmaxf(mxh, fabsf(xh));
        }
    };
    if (small_planes(a.hw)) {
        // small planes: one quad per lane, several images side by side, two groups of them in flight (see stats_partial_kernel);
        // idle lanes and missing elements take x = mean, dy = 0, which add nothing
        const int q4 = small_plane_lanes(a.hw);
        int l4 = 0;
        while ((1 << l4) < q4) ++l4;
        const int g4 = kThreads >> l4, tn4 = threadIdx.x >> l4, ti4 = threadIdx.x & ((1 << l4) - 1);
        const int stride = nsplit * g4;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int nb = sp * g4 + tn4; nb < a.n; nb += 2 * stride) {
            float4 xv[2], rv[2], uv[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int bb = nb + k * stride;
                const bool ok = bb < a.n;
                const long plane = (long)(ok ? bb : nb) * a.c + ch;
                const long off = plane * a.hw;
                QuadPos q = quad_pos(off, (int)a.hw, ti4);
                if (!ok) q.count = 0;
                xv[k] = load_quad(a.x + off, q, mean);
                rv[k] = a.res ? load_quad(a.res + off, q, 0.f) : zero4;
                uv[k] = a.dy ? load_quad(a.dy + off, q, 0.f) : zero4;
                if (q.count > 0 && a.gmax_dy) {
                    const int d = a.gmax_idx[plane] - q.base;
                    const float gval = a.gmax_dy[plane];
                    if (d >= 0 && d < q.count) { if (d == 0) uv[k].x += gval; else if (d == 1) uv[k].y += gval; else if (d == 2) uv[k].z += gval; else uv[k].w += gval; }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) quad(xv[k], rv[k], uv[k]);
        }
    } else
    for (int nb = sp; nb < a.n; nb += nsplit) {
        const long plane = (long)nb * a.c + ch;
        const float* px = a.x + plane * a.hw;
        const float* pr = a.res ? a.res + plane * a.hw : nullptr;
        const float* pdy = a.dy ? a.dy + plane * a.hw : nullptr;
        const float gval = a.gmax_dy ? a.gmax_dy[plane] : 0.f;
        const long gpos = a.gmax_dy ? (long)a.gmax_idx[plane] : -1;
        {
            // 16-byte loads whatever the plane's length: head elements up to the plane's first 16-byte boundary (odd lengths -- the
            // 1-d model -- put planes at any 4-byte offset), whole quads, tail
            const int head = (int)((4 - ((plane * a.hw) & 3)) & 3);
            const long n4 = (a.hw - head) >> 2;
            for (long i4 = ti; i4 < n4; i4 += kThreads) {
                const float4 xv = reinterpret_cast<const float4*>(px + head)[i4];
                float4 rv = make_float4(0.f, 0.f, 0.f, 0.f), uv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pr) rv = reinterpret_cast<const float4*>(pr + head)[i4];
                if (pdy) uv = reinterpret_cast<const float4*>(pdy + head)[i4];
                const long d = gpos - (head + i4 * 4);
                if (d >= 0 && d < 4) { if (d == 0) uv.x += gval; else if (d == 1) uv.y += gval; else if (d == 2) uv.z += gval; else uv.w += gval; }
                quad(xv, rv, uv);
            }
            if (ti < 8) {
                const long i = ti < head ? (long)ti : head + 4 * n4 + (ti - head);
                if (i < a.hw && (ti < head || i >= head + 4 * n4)) {
                    const float xh = (px[i] - mean) * invstd;
                    float z = fmaf(xh, g, b);
                    if (pr) z += pr[i];
                    const float up = upstream(a, pdy, i, gval, gpos);
                    const bool neg = has_alpha && !(z > 0.f);
                    const float dz = neg ? al * up : up;
                    s0 += dz;
                    s1 += dz * xh;
                    s2 += up * (neg ? z : 0.f);
                    s3 += xh;
                    mdz = fmaxf(mdz, fabsf(dz));
                    mxh = fmaxf(mxh, fabsf(xh));
                }
            }
        }
    }
    const double t0 = fsc::block_sum<double, kThreads / 64>((double)s0, scratch);
    const double t1 = fsc::block_sum<double, kThreads / 64>((double)s1, scratch);
    const double t2 = fsc::block_sum<double, kThreads / 64>((double)s2, scratch);
    const double t3 = fsc::block_sum<double, kThreads / 64>((double)s3, scratch);
    __shared__ float mred[kThreads / 64];
    mdz = block_max256(mdz, mred);
    mxh = block_max256(mxh, mred);
    const bool alone = fin.tickets != nullptr && nsplit == 1;       // the channel's only workgroup: finalise from registers
    if (threadIdx.x == 0 && !alone) {
        double* o = part + ((size_t)ch * kMaxSplit + sp) * kPartStride;
        const double vals[kBwdVals] = {t0, t1, t2, (double)mdz, (double)mxh, t3};
        if (fin.tickets) {
            // device-coherent stores and loads for the six numbers other workgroups (other XCDs: other L2s) read below.  A
            // __threadfence() here instead writes back and invalidates the XCD's whole L2 once per workgroup: the pass took twice
            // as long (2.6 -> 5.3 ms per step).
#pragma unroll
            for (int k = 0; k < kBwdVals; ++k) __hip_atomic_store(o + k, vals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
            for (int k = 0; k < kBwdVals; ++k) o[k] = vals[k];
        }
    }
    if (fin.tickets == nullptr) return;
    if (fin.dx_amax && !fin.want_bound && ch == 0 && sp == 0)          // (the L16 apply pass stores the bound itself)
        for (int i = threadIdx.x; i < fsc::kAmaxFloats; i += kThreads) fin.dx_amax[i] = 0.f;
    __shared__ int last_s;
    __shared__ double fold_s[kMaxSplit][kBwdVals];
    double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
    float fdz = 0.f, fxh = 0.f;
    if (alone) {
        // (no written-through partials, no ticket round trip, no re-load; the same numbers as a fold of one split)
        if (threadIdx.x != 0) return;
        f0 += t0; f1 += t1; f2 += t2; f3 += t3;
        fdz = fmaxf(fdz, (float)(double)mdz);
        fxh = fmaxf(fxh, (float)(double)mxh);
    } else {
        if (threadIdx.x == 0) last_s = ticket_last(fin.tickets + ch, nsplit) ? 1 : 0;      // (thread 0 issued the stores above)
        __syncthreads();
        if (!last_s) return;
        if ((int)threadIdx.x < nsplit) {
            double* p = part + ((size_t)ch * kMaxSplit + threadIdx.x) * kPartStride;
#pragma unroll
            for (int k = 0; k < kBwdVals; ++k) fold_s[threadIdx.x][k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (threadIdx.x != 0) return;
        for (int k = 0; k < nsplit; ++k) {
            f0 += fold_s[k][0]; f1 += fold_s[k][1]; f2 += fold_s[k][2]; f3 += fold_s[k][5];
            fdz = fmaxf(fdz, (float)fold_s[k][3]);
            fxh = fmaxf(fxh, (float)fold_s[k][4]);
        }
    }
    if (fin.dbeta) fin.dbeta[ch] = (float)f0;
    if (fin.dgamma) fin.dgamma[ch] = (float)f1;
    if (fin.dalpha) fin.dalpha[ch] = (float)f2;
    const float c1 = (float)(f0 / fin.count), c2 = (float)(f1 / fin.count);
    fin.coef[ch * 2] = c1;
    fin.coef[ch * 2 + 1] = c2;
    if (fin.dx_chan_sum) fin.dx_chan_sum[ch] = chan_sum_of_dx(g * invstd, f0, f3, fin.count, c1, c2);
    if (fin.want_bound) fin.coef[2 * a.c + ch] = fabsf(g * invstd) * (fdz + fabsf(c1) + fxh * fabsf(c2));
}

// ---------------------------------------------------------------- units whose channel is ONE workgroup
// The 1-d model's late blocks (195 ... 476 channels on rows of 53 ... 3 frames at batch 128: tensors of <= 1.3 MB) pay per LAUNCH, not
// per byte: a reduce pass takes 7 - 12 us, its apply pass 4 - 6, each a chain of dependent round trips -- parameters, two rounds of
// loads, six block reductions of two barriers each, one thread's finalisation, the stores.  Running the apply pass inside the
// reduce pass's launch as a second loop over memory bought nothing (measured: the apply phase cost what its launch had cost).
// These kernels hold the channel in registers instead: a thread owns at most TRIPS quads (all loads in flight at once), every
// reduction of the unit goes through ONE LDS exchange, every thread finalises (fp64, redundantly: no hand-over), and the apply
// pass is arithmetic on registers plus the stores.  Same additions in the same order as stats_partial_kernel / bwd_partial_kernel
// with one split and the same element arithmetic as fwd_wave_kernel / bwd_apply_wave_kernel / bwd_apply_unpool_kernel: bit-identical
// results (tests/test_bn_fused_gpu.py).
template <int THREADS>
struct UnitGeom {
    int l4, g4, tn4, ti4;
    __device__ __forceinline__ UnitGeom(long hw) {
        const int q4 = small_plane_lanes(hw);
        l4 = 0;
        while ((1 << l4) < q4) ++l4;
        g4 = THREADS >> l4;
        tn4 = threadIdx.x >> l4;
        ti4 = threadIdx.x & ((1 << l4) - 1);
    }
};
inline int unit_trips(int n, long hw, int threads) {
    const int q4 = small_plane_lanes(hw);
    int l4 = 0;
    while ((1 << l4) < q4) ++l4;
    const int g4 = threads >> l4;
    return (n + g4 - 1) / g4;
}
// workgroup size and quads per thread of a unit: 256 threads while <= 8 quads each hold the channel, else 1024 (the 1-d model's
// blocks 3 and 4 at batch 128: 125 / 156 channels of 27.5 k / 13.7 k values); 0 threads: not a unit
struct UnitShape { int threads, trips; };
inline UnitShape unit_shape(int n, long hw) {
    if (hw < 2 || !small_planes(hw)) return UnitShape{0, 0};
    int t = unit_trips(n, hw, 256);
    if (t <= 8) return UnitShape{256, t};
    t = unit_trips(n, hw, 1024);
    if (t <= 8) return UnitShape{1024, t};
    return UnitShape{0, 0};
}

// gmax / gmax_idx (may be null; planes of <= 64 lanes): the (n, c) global max of y and its position, the rule of
// fsc_global_maxpool_fwd (first maximum; NaN wins) -- what the hierarchical head reads of a block output (classifiers.py:586-590)
struct FwdApply { const float* res; const float* alpha; float* y; float* gmax; int* gmax_idx; };

template <int THREADS, int TRIPS>
__global__ __launch_bounds__(THREADS) void unit_fwd_kernel(const float* __restrict__ x, int n, int c, long hw, FinalizeArgs fa,
                                                            FwdApply ap) {
    constexpr int NW = THREADS / 64;
    __shared__ double sd[2][NW];
    __shared__ float sm[2][NW];
    const int ch = blockIdx.x;
    const UnitGeom<THREADS> u(hw);
    const float pivot = x[(long)ch * hw];
    float4 v[TRIPS];
    // (positions are recomputed where they are needed again: 4 registers per quad that 1024 threads x 8 quads do not have)
    auto where = [&](int t, long& off) {
        const int b = u.tn4 + t * u.g4;
        off = ((long)(b < n ? b : 0) * c + ch) * hw;
        QuadPos q = quad_pos(off, (int)hw, u.ti4);
        if (b >= n) q.count = 0;
        return q;
    };
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        long off;
        const QuadPos q = where(t, off);
        v[t] = load_quad(x + off, q, pivot);            // (missing elements read as the pivot: they add nothing to the shifted sums
    }                                                   //  and lie inside [min, max])
    float s1 = 0.f, s2 = 0.f, mn = pivot, mx = pivot;
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        const float a0 = v[t].x - pivot, a1 = v[t].y - pivot, a2 = v[t].z - pivot, a3 = v[t].w - pivot;
        s1 += (a0 + a1) + (a2 + a3);
        s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        mn = fminf(fminf(mn, fminf(v[t].x, v[t].y)), fminf(v[t].z, v[t].w));
        mx = fmaxf(fmaxf(mx, fmaxf(v[t].x, v[t].y)), fmaxf(v[t].z, v[t].w));
    }
    const double w1 = fsc::wave_sum((double)s1), w2 = fsc::wave_sum((double)s2);
    mx = fsc::wave_max(mx);
    mn = -fsc::wave_max(-mn);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sd[0][wid] = w1; sd[1][wid] = w2; sm[0][wid] = mn; sm[1][wid] = mx; }
    __syncthreads();
    double t1 = sd[0][0], t2 = sd[1][0];
#pragma unroll 4
    for (int i = 1; i < NW; ++i) { t1 += sd[0][i]; t2 += sd[1][i]; }
    if (threadIdx.x == 0 && fa.x_minmax) {
        mn = sm[0][0]; mx = sm[1][0];
        for (int i = 1; i < NW; ++i) { mn = fminf(mn, sm[0][i]); mx = fmaxf(mx, sm[1][i]); }
        fa.x_minmax[2 * ch] = mn;
        fa.x_minmax[2 * ch + 1] = mx;
    }
    float sc, sh;
    finalize_channel(fa, ch, 0.0 + t1, 0.0 + t2, (double)pivot, threadIdx.x == 0, &sc, &sh);
    const bool has_alpha = ap.alpha != nullptr;
    const float al = has_alpha ? ap.alpha[ch] : 0.f;
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        long off;
        const QuadPos q = where(t, off);
        if (q.count == 0 && !ap.gmax) continue;        // (with gmax every lane of a plane's group takes part in the shuffles)
        float4 z = make_float4(fmaf(v[t].x, sc, sh), fmaf(v[t].y, sc, sh), fmaf(v[t].z, sc, sh), fmaf(v[t].w, sc, sh));
        if (ap.res) {
            const float4 r = load_quad(ap.res + off, q, 0.f);
            z.x += r.x; z.y += r.y; z.z += r.z; z.w += r.w;
        }
        z.x = act(z.x, al, has_alpha); z.y = act(z.y, al, has_alpha);
        z.z = act(z.z, al, has_alpha); z.w = act(z.w, al, has_alpha);
        store_quad(ap.y + off, q, z);
        if (ap.gmax) {
            // the plane's 2^l4 <= 64 lanes sit in one wave: (value, first index) keys folded with shuffles; the lane that holds
            // the winner writes it (its own bits: the sign of a zero, the payload of a NaN)
            const float zs[4] = {z.x, z.y, z.z, z.w};
            unsigned long long key = 0ull;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned long long k2 = e < q.count ? rec_key(zs[e], (unsigned)(q.base + e)) : 0ull;
                key = k2 > key ? k2 : key;
            }
            for (int o = (1 << u.l4) >> 1; o > 0; o >>= 1) {
                const unsigned long long other = __shfl_xor(key, o, 64);
                key = other > key ? other : key;
            }
            const int ii = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
            if (ii >= q.base && ii < q.base + q.count) {
                const long plane = off / hw;
                const int d = ii - q.base;
                ap.gmax[plane] = d == 0 ? z.x : d == 1 ? z.y : d == 2 ? z.z : z.w;
                ap.gmax_idx[plane] = ii;
            }
        }
    }
}

// dx / dresidual as bwd_apply_wave_kernel writes them, or -- pool_idx given -- the gradient scattered through a (1, 2) max-pool
// as bwd_apply_unpool_kernel does for single-row planes (dc: rows of w = 2 * hw or 2 * hw + 1 values)
struct BwdApply { float* dx; float* dres; const uint8_t* pool_idx; float* dc; int w; };

// RES false: the unit has no residual (four of a block's five: 8 instead of 12 registers per quad)
template <int THREADS, int TRIPS, bool RES>
__global__ __launch_bounds__(THREADS) void unit_bwd_kernel(BwdArgs a, BwdFinish fin, BwdApply ap) {
    constexpr int NW = THREADS / 64;
    __shared__ double sd[4][NW];
    __shared__ float sm[2][NW];
    const int ch = blockIdx.x;
    const UnitGeom<THREADS> u(a.hw);
    const float mean = a.mean[ch], invstd = a.invstd[ch];
    const float g = a.gamma ? a.gamma[ch] : 1.f, b = a.beta ? a.beta[ch] : 0.f;
    const bool has_alpha = a.alpha != nullptr;
    const float al = has_alpha ? a.alpha[ch] : 1.f;
    float4 xv[TRIPS], rv[RES ? TRIPS : 1], uv[TRIPS];
    unsigned pi[TRIPS];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto where = [&](int t, long& plane) {           // (recomputed in the apply loop, see unit_fwd_kernel)
        const int bb = u.tn4 + t * u.g4;
        const bool ok = bb < a.n;
        plane = (long)(ok ? bb : 0) * a.c + ch;
        QuadPos q = quad_pos(plane * a.hw, (int)a.hw, u.ti4);
        if (!ok) q.count = 0;
        return q;
    };
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        long plane;
        const QuadPos q = where(t, plane);
        const long off = plane * a.hw;
        xv[t] = load_quad(a.x + off, q, mean);
        if constexpr (RES) rv[t] = load_quad(a.res + off, q, 0.f);
        uv[t] = a.dy ? load_quad(a.dy + off, q, 0.f) : zero4;
        if (q.count > 0 && a.gmax_dy) {
            const int d = a.gmax_idx[plane] - q.base;
            const float gval = a.gmax_dy[plane];
            if (d >= 0 && d < q.count) { if (d == 0) uv[t].x += gval; else if (d == 1) uv[t].y += gval; else if (d == 2) uv[t].z += gval; else uv[t].w += gval; }
        }
        pi[t] = 0u;
        if (ap.pool_idx)
            for (int e = 0; e < q.count; ++e) pi[t] |= (unsigned)ap.pool_idx[off + q.base + e] << (8 * e);
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, mdz = 0.f, mxh = 0.f;
    // (xv / uv become xhat / dz in place: what the apply pass needs)
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        float xs[4] = {xv[t].x, xv[t].y, xv[t].z, xv[t].w}, us[4] = {uv[t].x, uv[t].y, uv[t].z, uv[t].w};
        const float4 r4 = RES ? rv[RES ? t : 0] : zero4;
        const float rs[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xs[e] - mean) * invstd;
            const float z = fmaf(xh, g, b) + rs[e];
            const bool neg = has_alpha && !(z > 0.f);
            const float dz = neg ? al * us[e] : us[e];
            s0 += dz;
            s1 += dz * xh;
            s2 += us[e] * (neg ? z : 0.f);
            s3 += xh;
            mdz = fmaxf(mdz, fabsf(dz));
            mxh = fmaxf(mxh, fabsf(xh));
            xs[e] = xh;
            us[e] = dz;
        }
        xv[t] = make_float4(xs[0], xs[1], xs[2], xs[3]);
        uv[t] = make_float4(us[0], us[1], us[2], us[3]);
    }
    const double w0 = fsc::wave_sum((double)s0), w1 = fsc::wave_sum((double)s1), w2 = fsc::wave_sum((double)s2),
                 w3 = fsc::wave_sum((double)s3);
    mdz = fsc::wave_max(mdz);
    mxh = fsc::wave_max(mxh);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sd[0][wid] = w0; sd[1][wid] = w1; sd[2][wid] = w2; sd[3][wid] = w3; sm[0][wid] = mdz; sm[1][wid] = mxh; }
    __syncthreads();
    double t0 = sd[0][0], t1 = sd[1][0], t2 = sd[2][0], t3 = sd[3][0];
#pragma unroll 4
    for (int i = 1; i < NW; ++i) { t0 += sd[0][i]; t1 += sd[1][i]; t2 += sd[2][i]; t3 += sd[3][i]; }
    // (the fold of one split, as bwd_partial_kernel's `alone` path)
    double f0 = 0.0, f1 = 0.0, f2 = 0.0, f3 = 0.0;
    f0 += t0; f1 += t1; f2 += t2; f3 += t3;
    const float c1 = (float)(f0 / fin.count), c2 = (float)(f1 / fin.count);
    if (threadIdx.x == 0) {
        if (fin.dbeta) fin.dbeta[ch] = (float)f0;
        if (fin.dgamma) fin.dgamma[ch] = (float)f1;
        if (fin.dalpha) fin.dalpha[ch] = (float)f2;
        fin.coef[ch * 2] = c1;
        fin.coef[ch * 2 + 1] = c2;
        if (fin.dx_chan_sum) fin.dx_chan_sum[ch] = chan_sum_of_dx(g * invstd, f0, f3, fin.count, c1, c2);
    }
    const float k = g * invstd;
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
        long plane;
        const QuadPos q = where(t, plane);
        if (q.count == 0) continue;
        const float xh[4] = {xv[t].x, xv[t].y, xv[t].z, xv[t].w}, dz[4] = {uv[t].x, uv[t].y, uv[t].z, uv[t].w};
        float dv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) dv[e] = k * (dz[e] - c1 - xh[e] * c2);
        if (ap.dc) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            float* r0 = ap.dc + plane * ap.w;
            for (int e = 0; e < q.count; ++e) {
                const int pos = (int)((pi[t] >> (8 * e)) & 0xFFu);
                *reinterpret_cast<f32x2*>(r0 + 2 * (q.base + e)) = (f32x2){pos == 0 ? dv[e] : 0.f, pos == 1 ? dv[e] : 0.f};
            }
            if ((ap.w & 1) && q.base + q.count == (int)a.hw) r0[ap.w - 1] = 0.f;      // the column the pool never read
        } else {
            store_quad(ap.dx + plane * a.hw, q, make_float4(dv[0], dv[1], dv[2], dv[3]));
            if (ap.dres) store_quad(ap.dres + plane * a.hw, q, uv[t]);
        }
    }
}

// HW == 1 version: thread per channel
__global__ void bwd_rows_kernel(BwdArgs a, double* __restrict__ part) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= a.c) return;
    const float mean = a.mean[ch], invstd = a.invstd[ch];
    const float g = a.gamma ? a.gamma[ch] : 1.f, b = a.beta ? a.beta[ch] : 0.f;
    const bool has_alpha = a.alpha != nullptr;
    const float al = has_alpha ? a.alpha[ch] : 1.f;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int nb = blockIdx.y; nb < a.n; nb += gridDim.y) {
        const long i = (long)nb * a.c + ch;
        const float xh = (a.x[i] - mean) * invstd;
        float z = fmaf(xh, g, b);
        if (a.res) z += a.res[i];
        const float up = a.dy ? a.dy[i] : 0.f;
        const bool neg = has_alpha && !(z > 0.f);
        const float dz = neg ? al * up : up;
        s0 += dz; s1 += (double)dz * xh;
        s2 += (double)up * (neg ? z : 0.f);
        s3 += xh;
    }
    double* o = part + ((size_t)ch * kMaxSplit + blockIdx.y) * kPartStride;
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = 0.0; o[4] = 0.0; o[5] = s3;
}

// SyncBN: phase 1 writes the parameter gradients (LOCAL sums: the gradient all-reduce adds the replicas later) and
// [sum dz, sum dz*xhat, count, -] into `sync`, then stops; phase 2 takes the all-reduced `sync` for the two means
// the input gradient needs.  phase 0: single replica.
__global__ void bwd_finalize_kernel(int c, double count, int nsplit, const double* __restrict__ part,
                                    float* dgamma, float* dbeta, float* dalpha, float* coef, float* dx_chan_sum,
                                    float* dx_amax, double* __restrict__ sync, int phase,
                                    const float* __restrict__ gamma, const float* __restrict__ invstd, int want_bound) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (dx_amax && phase != 1 && !want_bound)             // (the L16 apply pass stores the bound itself)
        for (int i = ch; i < fsc::kAmaxFloats; i += gridDim.x * blockDim.x) dx_amax[i] = 0.f;
    if (ch >= c) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    double l0 = 0.0, l3 = 0.0;                  // this replica's sum dz, sum xhat (phase 2: `part` still holds phase 1's partials)
    const double local_count = count;
    for (int s = 0; s < nsplit; ++s) {
        const double* p = part + ((size_t)ch * kMaxSplit + s) * kPartStride;
        l0 += p[0]; l3 += p[5];
    }
    if (phase == 2) {
        s0 = sync[ch * 4];
        s1 = sync[ch * 4 + 1];
        count = sync[ch * 4 + 2];
    } else {
        for (int s = 0; s < nsplit; ++s) {
            const double* p = part + ((size_t)ch * kMaxSplit + s) * kPartStride;
            s0 += p[0]; s1 += p[1]; s2 += p[2];
        }
        if (dbeta) dbeta[ch] = (float)s0;
        if (dgamma) dgamma[ch] = (float)s1;
        if (dalpha) dalpha[ch] = (float)s2;
        if (phase == 1) {
            sync[ch * 4] = s0;
            sync[ch * 4 + 1] = s1;
            sync[ch * 4 + 2] = count;
            sync[ch * 4 + 3] = 0.0;
            return;
        }
    }
    const float c1 = (float)(s0 / count), c2 = (float)(s1 / count);
    coef[ch * 2] = c1;
    coef[ch * 2 + 1] = c2;
    if (dx_chan_sum) dx_chan_sum[ch] = chan_sum_of_dx((gamma ? gamma[ch] : 1.f) * invstd[ch], l0, l3, local_count, c1, c2);
    if (want_bound) {
        // |dx| = |k (dz - c1 - xhat c2)| <= |k| (max |dz| + |c1| + max |xhat| |c2|): the declared maximum of the L16 tensor
        // (an over-estimate, by less than 2x for gradients whose means are small against their extremes; safe)
        float mdz = 0.f, mxh = 0.f;
        for (int s = 0; s < nsplit; ++s) {
            const double* p = part + ((size_t)ch * kMaxSplit + s) * kPartStride;
            mdz = fmaxf(mdz, (float)p[3]);
            mxh = fmaxf(mxh, (float)p[4]);
        }
        const float k = (gamma ? gamma[ch] : 1.f) * invstd[ch];
        coef[2 * c + ch] = fabsf(k) * (mdz + fabsf(c1) + mxh * fabsf(c2));
    }
}

__global__ __launch_bounds__(kThreads) void bwd_apply_plane_kernel(BwdArgs a, const float* __restrict__ coef,
                                                                    float* __restrict__ dx, float* __restrict__ dres,
                                                                    float* dx_amax) {
    const long plane = blockIdx.x;
    float mx = 0.f;
    const int ch = (int)(plane % a.c);
    const float mean = a.mean[ch], invstd = a.invstd[ch];
    const float g = a.gamma ? a.gamma[ch] : 1.f, b = a.beta ? a.beta[ch] : 0.f;
    const bool has_alpha = a.alpha != nullptr;
    const float al = has_alpha ? a.alpha[ch] : 1.f;
    const float c1 = coef[ch * 2], c2 = coef[ch * 2 + 1];
    const float k = g * invstd;
    const float* px = a.x + plane * a.hw;
    const float* pr = a.res ? a.res + plane * a.hw : nullptr;
    const float* pdy = a.dy ? a.dy + plane * a.hw : nullptr;
    const float gval = a.gmax_dy ? a.gmax_dy[plane] : 0.f;
    const long gpos = a.gmax_dy ? (long)a.gmax_idx[plane] : -1;
    float* pdx = dx + plane * a.hw;
    float* pdr = dres ? dres + plane * a.hw : nullptr;
    const int head = (int)((4 - ((plane * a.hw) & 3)) & 3);       // (see fwd_plane_kernel: 16-byte accesses for any plane length)
    const long n4 = (a.hw - head) >> 2;
    auto point = [&](long i) {
        const float xh = (px[i] - mean) * invstd;
        float z = fmaf(xh, g, b);
        if (pr) z += pr[i];
        const float up = upstream(a, pdy, i, gval, gpos);
        const float dz = (has_alpha && !(z > 0.f)) ? al * up : up;
        const float d = k * (dz - c1 - xh * c2);
        pdx[i] = d;
        if (pdr) pdr[i] = dz;
        mx = fmaxf(mx, fabsf(d));
    };
    for (long i4 = (long)blockIdx.y * kThreads + threadIdx.x; i4 < n4; i4 += (long)gridDim.y * kThreads) {
        const float4 xv = reinterpret_cast<const float4*>(px + head)[i4];
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f), uv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pr) rv = reinterpret_cast<const float4*>(pr + head)[i4];
        if (pdy) uv = reinterpret_cast<const float4*>(pdy + head)[i4];
        const long dd = gpos - (head + i4 * 4);
        if (dd >= 0 && dd < 4) { if (dd == 0) uv.x += gval; else if (dd == 1) uv.y += gval; else if (dd == 2) uv.z += gval; else uv.w += gval; }
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, rs[4] = {rv.x, rv.y, rv.z, rv.w}, us[4] = {uv.x, uv.y, uv.z, uv.w};
        float dv[4], zv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (xs[e] - mean) * invstd;
            const float z = fmaf(xh, g, b) + rs[e];
            zv[e] = (has_alpha && !(z > 0.f)) ? al * us[e] : us[e];
            dv[e] = k * (zv[e] - c1 - xh * c2);
            mx = fmaxf(mx, fabsf(dv[e]));
        }
        reinterpret_cast<float4*>(pdx + head)[i4] = make_float4(dv[0], dv[1], dv[2], dv[3]);
        if (pdr) reinterpret_cast<float4*>(pdr + head)[i4] = make_float4(zv[0], zv[1], zv[2], zv[3]);
    }
    if (blockIdx.y == 0 && threadIdx.x < 8) {
        const long t = threadIdx.x;
        if (t < head) point(t);
        else if (head + 4 * n4 + (t - head) < a.hw) point(head + 4 * n4 + (t - head));
    }
    if (dx_amax) fsc::publish_amax(dx_amax, mx);
}

// planes of 2..511 pixels: one wavefront (or a lane group of it) per plane
__global__ __launch_bounds__(kThreads) void bwd_apply_wave_kernel(BwdArgs a, const float* __restrict__ coef,
                                                                   float* __restrict__ dx, float* __restrict__ dres,
                                                                   long planes, float* dx_amax, int glog) {
    const int gsz = 1 << glog;
    long plane = ((long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) * (64 >> glog) + ((threadIdx.x & 63) >> glog);
    const long hw_live = plane < planes ? a.hw : 0;      // (no early exit: publish_amax synchronises the block)
    if (plane >= planes) plane = planes - 1;
    const int lane = threadIdx.x & (gsz - 1);
    const int ch = (int)(plane % a.c);
    const float mean = a.mean[ch], invstd = a.invstd[ch];
    const float g = a.gamma ? a.gamma[ch] : 1.f, b = a.beta ? a.beta[ch] : 0.f;
    const bool has_alpha = a.alpha != nullptr;
    const float al = has_alpha ? a.alpha[ch] : 1.f;
    const float c1 = coef[ch * 2], c2 = coef[ch * 2 + 1];
    const float k = g * invstd;
    const long base = plane * a.hw;
    const float gval = a.gmax_dy ? a.gmax_dy[plane] : 0.f;
    const long gpos = a.gmax_dy ? (long)a.gmax_idx[plane] : -1;
    float mx = 0.f;
    for (long i = lane; i < hw_live; i += gsz) {
        const float xh = (a.x[base + i] - mean) * invstd;
        float z = fmaf(xh, g, b);
        if (a.res) z += a.res[base + i];
        float up = a.dy ? a.dy[base + i] : 0.f;
        if (i == gpos) up += gval;
        const float dz = (has_alpha && !(z > 0.f)) ? al * up : up;
        const float d = k * (dz - c1 - xh * c2);
        dx[base + i] = d;
        if (dres) dres[base + i] = dz;
        mx = fmaxf(mx, fabsf(d));
    }
    if (dx_amax) fsc::publish_amax(dx_amax, mx);
}

// BN backward apply fused with the backward of the 2x2 (or 1x2) max-pool that produced x: instead
// of dx at the pooled resolution, the gradient is scattered to the arg-max position of each window
// of the full-resolution tensor `dc` (all other positions, and the odd trailing row / column that
// floor-mode pooling never read, are zero).  Saves the pooled dx round trip and the separate
// max-pool backward pass.  Grid: blockIdx.x = (n, c) plane, blockIdx.y strides over pooled rows.
__global__ __launch_bounds__(kThreads) void bwd_apply_unpool_kernel(BwdArgs a, const float* __restrict__ coef,
                                                                     const uint8_t* __restrict__ pool_idx,
                                                                     float* __restrict__ dc,
                                                                     int h, int w, int ph, int oh, int ow,
                                                                     int colp_log2, float* dc_amax, long row_units) {
    // A block owns kThreads >> colp_log2 consecutive ROW UNITS (plane, pooled row); a row unit is walked by 2^colp_log2 lanes.  (One
    // block per plane left 32 of 256 lanes busy on the 1-d model's single-row planes and paid a block's fixed cost -- parameter
    // loads, block reduction, two atomics -- per (image, channel): 54 us for a 0.7 MB tensor.)
    const int colp = 1 << colp_log2, rows_per_block = kThreads >> colp_log2;
    const int tr = threadIdx.x >> colp_log2, tc = threadIdx.x & (colp - 1);
    const long unit = (long)blockIdx.x * rows_per_block + tr;
    const bool live = unit < row_units;
    const long plane = live ? unit / oh : 0;
    const int oy = live ? (int)(unit - plane * oh) : 0;
    const int ch = (int)(plane % a.c);
    const float mean = a.mean[ch], invstd = a.invstd[ch];
    const float g = a.gamma ? a.gamma[ch] : 1.f, b = a.beta ? a.beta[ch] : 0.f;
    const bool has_alpha = a.alpha != nullptr;
    const float al = has_alpha ? a.alpha[ch] : 1.f;
    const float c1 = coef[ch * 2], c2 = coef[ch * 2 + 1];
    const float k = g * invstd;
    const float* px = a.x + plane * a.hw;
    const float* pr = a.res ? a.res + plane * a.hw : nullptr;
    const float* pdy = a.dy + plane * a.hw;
    const uint8_t* pi = pool_idx + plane * a.hw;
    float* pdc = dc + plane * h * w;
    float mx = 0.f;
    if (live) {
        float* r0 = pdc + (long)oy * ph * w;
        for (int ox = tc; ox < ow; ox += colp) {
            const int i = oy * ow + ox;
            const float xh = (px[i] - mean) * invstd;
            float z = fmaf(xh, g, b);
            if (pr) z += pr[i];
            const float up = pdy[i];
            const float dz = (has_alpha && !(z > 0.f)) ? al * up : up;
            const float d = k * (dz - c1 - xh * c2);
            mx = fmaxf(mx, fabsf(d));
            const int pos = pi[i];
            // one 8-byte store per window row (4-byte aligned): half the store instructions of the scalar form
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f32x2*>(r0 + 2 * ox) = (f32x2){pos == 0 ? d : 0.f, pos == 1 ? d : 0.f};
            if (ph == 2) *reinterpret_cast<f32x2*>(r0 + w + 2 * ox) = (f32x2){pos == 2 ? d : 0.f, pos == 3 ? d : 0.f};
        }
        if ((w & 1) && tc == 0) {                      // column floor-mode pooling never read
            r0[w - 1] = 0.f;
            if (ph == 2) r0[2 * w - 1] = 0.f;
        }
        if (ph == 2 && (h & 1) && oy == oh - 1) {      // trailing row
            for (int xx = tc; xx < w; xx += colp) pdc[(long)(h - 1) * w + xx] = 0.f;
        }
    }
    if (dc_amax) fsc::publish_amax(dc_amax, mx);
}

// Backward apply pass writing dx as an L16 tensor (and, optionally, as fp32 planes too).  Same arithmetic as
// bwd_apply_plane_kernel; thread layout of fwd_l16_kernel.  The scale comes from the per-channel bounds coef[2c + ch].
template <int VEC, bool UNI, int NL>
__global__ __launch_bounds__(kThreads) void bwd_apply_l16_kernel(BwdArgs a, const float* __restrict__ coef,
                                                                  float* __restrict__ dx, uint4* __restrict__ dx16,
                                                                  float* __restrict__ dres,
                                                                  float* __restrict__ dx_amax, int hwp_log2) {
    __shared__ float red[kThreads / 64];
    float s = 1.f;
    if (l16::fmt_scaled(NL) || dx_amax != nullptr) {                     // (bf16 limbs need no scale; the bound is still published when asked for)
        float m = 0.f;
        for (int ch = threadIdx.x; ch < a.c; ch += kThreads) m = fmaxf(m, coef[2 * a.c + ch]);
        m = block_max256(m, red);
        if (blockIdx.x == 0 && blockIdx.y == 0) l16::store_amax(dx_amax, m);
        s = l16::field_to_float(l16::scale_field(m));
    }
    const int c = a.c, oct = (c + 7) >> 3;
    const long hw = a.hw;
    const int hwp = UNI ? kThreads : 1 << hwp_log2, groups = UNI ? 1 : kThreads >> hwp_log2;
    const int tn = UNI ? 0 : threadIdx.x >> hwp_log2, ti = UNI ? threadIdx.x : threadIdx.x & (hwp - 1);
    const long g = (long)blockIdx.x * groups + tn;            // (octet, image), images fastest
    const bool g_live = g < (long)a.n * oct;
    const int o = g_live ? (int)(g / a.n) : 0, img = g_live ? (int)(g - (long)o * a.n) : 0;
    const bool has_alpha = a.alpha != nullptr;
    const long nq = g_live ? hw / VEC : 0;
    const long xbase = ((long)img * c + o * 8) * hw;
    uint4* const out0 = dx16 + (((long)img * oct + o) * FSC_NLIMBS(NL)) * hw;
    __shared__ uint4 patch[kThreads / 64][64 * 5];
    const bool transpose = VEC == 4 && hwp >= 64;            // (uniform; see l16_store_quads)
    const int lane = threadIdx.x & 63;
    for (long q = (long)blockIdx.y * hwp + ti; transpose ? q - lane < nq : q < nq; q += (long)gridDim.y * hwp) {
        const bool q_live = q < nq;
        float d[8][VEC];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = o * 8 + e;
            if (ch < c && q_live) {
                const float mean = a.mean[ch], invstd = a.invstd[ch];
                const float gm = a.gamma ? a.gamma[ch] : 1.f, bt = a.beta ? a.beta[ch] : 0.f;
                const float al = has_alpha ? a.alpha[ch] : 1.f;
                const float c1 = coef[ch * 2], c2 = coef[ch * 2 + 1];
                const float k = gm * invstd;
                const long pb = xbase + e * hw;
                const long plane = (long)img * c + ch;
                const float gval = a.gmax_dy ? a.gmax_dy[plane] : 0.f;
                const long gpos = a.gmax_dy ? (long)a.gmax_idx[plane] : -1;
                float xs[VEC], rs[VEC], us[VEC], zv[VEC];
                if (VEC == 4) {
                    const float4 xv = reinterpret_cast<const float4*>(a.x + pb)[q];
                    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f), uv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.res) rv = reinterpret_cast<const float4*>(a.res + pb)[q];
                    if (a.dy) uv = reinterpret_cast<const float4*>(a.dy + pb)[q];
                    xs[0] = xv.x; xs[1 % VEC] = xv.y; xs[2 % VEC] = xv.z; xs[3 % VEC] = xv.w;
                    rs[0] = rv.x; rs[1 % VEC] = rv.y; rs[2 % VEC] = rv.z; rs[3 % VEC] = rv.w;
                    us[0] = uv.x; us[1 % VEC] = uv.y; us[2 % VEC] = uv.z; us[3 % VEC] = uv.w;
                } else {
                    xs[0] = a.x[pb + q];
                    rs[0] = a.res ? a.res[pb + q] : 0.f;
                    us[0] = a.dy ? a.dy[pb + q] : 0.f;
                }
#pragma unroll
                for (int p = 0; p < VEC; ++p) {
                    if (q * VEC + p == gpos) us[p] += gval;
                    const float xh = (xs[p] - mean) * invstd;
                    const float z = fmaf(xh, gm, bt) + rs[p];
                    zv[p] = (has_alpha && !(z > 0.f)) ? al * us[p] : us[p];
                    d[e][p] = k * (zv[p] - c1 - xh * c2);
                }
                if (VEC == 4) {
                    if (dx) reinterpret_cast<float4*>(dx + pb)[q] = make_float4(d[e][0], d[e][1 % VEC], d[e][2 % VEC], d[e][3 % VEC]);
                    if (dres) reinterpret_cast<float4*>(dres + pb)[q] = make_float4(zv[0], zv[1 % VEC], zv[2 % VEC], zv[3 % VEC]);
                } else {
                    if (dx) dx[pb + q] = d[e][0];
                    if (dres) dres[pb + q] = zv[0];
                }
            } else {
#pragma unroll
                for (int p = 0; p < VEC; ++p) d[e][p] = 0.f;
            }
        }
        uint4 lv[VEC][FSC_NLIMBS(NL)];
#pragma unroll
        for (int p = 0; p < VEC; ++p) {
            const float v8[8] = {d[0][p], d[1][p], d[2][p], d[3][p], d[4][p], d[5][p], d[6][p], d[7][p]};
            l16_split<NL>(v8, s, lv[p]);
        }
        if constexpr (VEC == 4) {
            if (transpose) {
                l16_store_quads<NL>(out0, q, hw, lv, patch[threadIdx.x >> 6]);
                continue;
            }
        }
#pragma unroll
        for (int p = 0; p < VEC; ++p)
#pragma unroll
            for (int l = 0; l < FSC_NLIMBS(NL); ++l) out0[l * hw + q * VEC + p] = lv[p][l];
    }
}

// L16 form of bwd_apply_unpool_kernel: a thread owns one pooled position x the 8 channels of an (octet, image) and writes the
// 2 x 2 (or 1 x 2) window of the full-resolution gradient dc -- each channel's gradient at ITS arg-max position, zeros
// elsewhere -- as 16-byte limb vectors; the odd trailing row / column of dc is zero.
template <int NL>
__global__ __launch_bounds__(kThreads) void bwd_apply_unpool_l16_kernel(BwdArgs a, const float* __restrict__ coef,
                                                                         const uint8_t* __restrict__ pool_idx,
                                                                         float* __restrict__ dc, uint4* __restrict__ dc16,
                                                                         int h, int w, int ph, int oh, int ow,
                                                                         float* __restrict__ dc_amax, int hwp_log2) {
    __shared__ float red[kThreads / 64];
    float s = 1.f;
    if (l16::fmt_scaled(NL) || dc_amax != nullptr) {
        float m = 0.f;
        for (int ch = threadIdx.x; ch < a.c; ch += kThreads) m = fmaxf(m, coef[2 * a.c + ch]);
        m = block_max256(m, red);
        if (blockIdx.x == 0 && blockIdx.y == 0) l16::store_amax(dc_amax, m);
        s = l16::field_to_float(l16::scale_field(m));
    }
    const int c = a.c, oct = (c + 7) >> 3;
    const long hw = a.hw, HW = (long)h * w;
    const int hwp = 1 << hwp_log2, groups = kThreads >> hwp_log2;
    const int tn = threadIdx.x >> hwp_log2, ti = threadIdx.x & (hwp - 1);
    const long g = (long)blockIdx.x * groups + tn;
    const bool g_live = g < (long)a.n * oct;
    const int o = g_live ? (int)(g / a.n) : 0, img = g_live ? (int)(g - (long)o * a.n) : 0;
    const bool has_alpha = a.alpha != nullptr;
    const long xbase = ((long)img * c + o * 8) * hw;
    uint4* const out0 = dc16 + (((long)img * oct + o) * FSC_NLIMBS(NL)) * HW;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    const long nq = g_live ? hw : 0;
    for (long q = (long)blockIdx.y * hwp + ti; q < nq; q += (long)gridDim.y * hwp) {
        const int oy = (int)(q / ow), ox = (int)(q - (long)oy * ow);
        float d[8];
        int pos[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = o * 8 + e;
            d[e] = 0.f;
            pos[e] = 0;
            if (ch < c) {
                const float mean = a.mean[ch], invstd = a.invstd[ch];
                const float gm = a.gamma ? a.gamma[ch] : 1.f, bt = a.beta ? a.beta[ch] : 0.f;
                const float al = has_alpha ? a.alpha[ch] : 1.f;
                const float k = gm * invstd;
                const long i = xbase + e * hw + q;
                const float xh = (a.x[i] - mean) * invstd;
                const float z = fmaf(xh, gm, bt);
                const float up = a.dy[i];
                const float dz = (has_alpha && !(z > 0.f)) ? al * up : up;
                d[e] = k * (dz - coef[ch * 2] - xh * coef[ch * 2 + 1]);
                pos[e] = pool_idx[i];
                if (dc) {                                   // fp32 planes as well (bwd_apply_unpool_kernel's stores)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    float* r0 = dc + ((long)img * c + ch) * HW + (long)oy * ph * w;
                    *reinterpret_cast<f32x2*>(r0 + 2 * ox) = (f32x2){pos[e] == 0 ? d[e] : 0.f, pos[e] == 1 ? d[e] : 0.f};
                    if (ph == 2) *reinterpret_cast<f32x2*>(r0 + w + 2 * ox) = (f32x2){pos[e] == 2 ? d[e] : 0.f, pos[e] == 3 ? d[e] : 0.f};
                    if ((w & 1) && ox == ow - 1) {
                        r0[w - 1] = 0.f;
                        if (ph == 2) r0[2 * w - 1] = 0.f;
                    }
                }
            }
        }
        for (int wpos = 0; wpos < 2 * ph; ++wpos) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = pos[e] == wpos ? d[e] : 0.f;
            uint4 lv[FSC_NLIMBS(NL)];
            l16_split<NL>(v8, s, lv);
            const long dst = (long)(oy * ph + (wpos >> 1)) * w + 2 * ox + (wpos & 1);
#pragma unroll
            for (int l = 0; l < FSC_NLIMBS(NL); ++l) out0[l * HW + dst] = lv[l];
        }
        if ((w & 1) && ox == ow - 1) {                      // column floor-mode pooling never read
            for (int r = 0; r < ph; ++r)
#pragma unroll
                for (int l = 0; l < FSC_NLIMBS(NL); ++l) out0[l * HW + (long)(oy * ph + r) * w + w - 1] = zero4;
        }
    }
    if (ph == 2 && (h & 1) && blockIdx.y == 0 && g_live) {  // trailing row
        for (int xx = ti; xx < w; xx += hwp) {
#pragma unroll
            for (int l = 0; l < FSC_NLIMBS(NL); ++l) out0[l * HW + (long)(h - 1) * w + xx] = zero4;
            if (dc)
                for (int e = 0; e < 8; ++e)
                    if (o * 8 + e < c) dc[((long)img * c + o * 8 + e) * HW + (long)(h - 1) * w + xx] = 0.f;
        }
    }
}

__global__ void bwd_apply_flat_kernel(BwdArgs a, const float* __restrict__ coef, float* __restrict__ dx,
                                      float* __restrict__ dres, long total, float* dx_amax) {
    float mx = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long plane = i / a.hw;
        const int ch = (int)(plane % a.c);
        const float invstd = a.invstd[ch];
        const float g = a.gamma ? a.gamma[ch] : 1.f, b = a.beta ? a.beta[ch] : 0.f;
        const float xh = (a.x[i] - a.mean[ch]) * invstd;
        float z = fmaf(xh, g, b);
        if (a.res) z += a.res[i];
        float up = a.dy ? a.dy[i] : 0.f;
        if (a.gmax_dy && (i - plane * a.hw) == (long)a.gmax_idx[plane]) up += a.gmax_dy[plane];
        const float dz = (a.alpha && !(z > 0.f)) ? a.alpha[ch] * up : up;
        const float d = g * invstd * (dz - coef[ch * 2] - xh * coef[ch * 2 + 1]);
        dx[i] = d;
        if (dres) dres[i] = dz;
        mx = fmaxf(mx, fabsf(d));
    }
    if (dx_amax) fsc::publish_amax(dx_amax, mx);
}

int pick_split(int n, int c, long hw) {
    // enough blocks to fill 256 CUs a few times over, bounded by the batch and the workspace
    long want = (256L * 8 + c - 1) / c;
    static const long big = [] { const char* e = getenv("FSC_BN_PER_BLOCK"); return e ? atol(e) : 16384L; }();
    const long per_block = small_planes(hw) ? 4096 : big;          // (small planes are latency-bound: more, shorter blocks)
    long by_work = ((long)n * hw + per_block - 1) / per_block;
    long s = want < by_work ? want : by_work;
    if (s > n) s = n;
    if (s > kMaxSplit) s = kMaxSplit;
    if ((long)n * hw <= 8192 && c >= 128) s = 1;       // (a channel's only workgroup finalises from registers: see bwd_partial_kernel)
    if (s < 1) s = 1;
    return (int)s;
}

// launch geometry of the L16 producer kernels: positions per group lane-split, blocks, UNI
struct L16Grid { int hwp_log2; bool uni; dim3 grid; int vec; };
L16Grid l16_grid(int n, int c, long hw, bool allow_vec) {
    L16Grid r;
    r.vec = (allow_vec && (hw & 3) == 0 && !fsc::env().l16_vec1) ? 4 : 1;
    const long nq = hw / r.vec;
    const long groups_total = (long)n * ((c + 7) / 8);
    r.uni = nq >= kThreads;
    r.hwp_log2 = 8;
    if (!r.uni) {
        r.hwp_log2 = 0;
        while ((1L << r.hwp_log2) < nq && r.hwp_log2 < 8) ++r.hwp_log2;
    }
    const int per_block = r.uni ? 1 : kThreads >> r.hwp_log2;
    long gy = r.uni ? (nq + kThreads * 4 - 1) / (kThreads * 4) : 1;
    if (gy > 64) gy = 64;
    r.grid = dim3((unsigned)((groups_total + per_block - 1) / per_block), (unsigned)gy);
    return r;
}

int plane_grid_y(long hw) {
    long per = (hw & 3) == 0 ? hw / 4 : hw;
    long gy = (per + kThreads * 4 - 1) / (kThreads * 4);
    if (gy < 1) gy = 1;
    if (gy > 64) gy = 64;
    return (int)gy;
}

}  // namespace

extern "C" {

size_t fsc_bn_workspace_bytes(int c) {
    return part_doubles(c) * sizeof(double) + (size_t)c * 4 * sizeof(float) + (((size_t)c * sizeof(unsigned) + 7) & ~(size_t)7);
}

size_t fsc_bn_workspace_ticket_offset(int c) { return part_doubles(c) * sizeof(double) + (size_t)c * 4 * sizeof(float); }

int fsc_bn_workspace_reset(void* workspace, int c, fsc_stream_t stream) {
    FSC_CHECK_ARG(workspace && c > 0, "fsc_bn_workspace_reset: bad arguments");
    hipError_t e = hipMemsetAsync(carve(workspace, c).tickets, 0, (size_t)c * sizeof(unsigned), fsc::as_stream(stream));
    FSC_CHECK_ARG(e == hipSuccess, "fsc_bn_workspace_reset: memset failed: %s", hipGetErrorString(e));
    return 0;
}

int fsc_bn_train_stats(const float* x, int n, int c, long hw, const float* gamma, const float* beta, float eps,
                       float momentum, float* running_mean, float* running_var, float* save_mean,
                       float* save_invstd, float* scale, float* shift, void* workspace, double* sync, int phase,
                       float* x_minmax, fsc_stream_t stream) {
    const int minmax_only = (phase & FSC_BN_STATS_MINMAX_ONLY) ? 1 : 0;
    FSC_CHECK_ARG(x && workspace && (minmax_only ? (x_minmax != nullptr && (phase & 3) == 0)
                                                 : (save_mean && save_invstd && scale && shift)),
                  "fsc_bn_train_stats: null pointer");
    FSC_CHECK_ARG((phase & 3) == 0 || (((phase & 3) == 1 || (phase & 3) == 2) && sync),
                  "fsc_bn_train_stats: phase 1 / 2 need `sync`");
    FSC_CHECK_ARG(n > 0 && c > 0 && hw > 0, "fsc_bn_train_stats: bad shape (%d, %d, %ld)", n, c, hw);
    FSC_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "fsc_bn_train_stats: running stats must come in pairs");
    hipStream_t st = fsc::as_stream(stream);
    Partials p = carve(workspace, c);
    int nsplit = 1;
    const bool folded = (phase & FSC_BN_STATS_FOLDED) != 0;      // split 0 of the partials is there already (fsc_bn_records_fold)
    const int pivot_rm = (phase & FSC_BN_STATS_PIVOT_RM) ? 1 : 0;
    const bool zero_tickets = (phase & FSC_BN_TICKETS) != 0;   // the caller vouches for the workspace's ticket words
    phase &= ~(FSC_BN_STATS_FOLDED | FSC_BN_STATS_PIVOT_RM | FSC_BN_STATS_MINMAX_ONLY | FSC_BN_TICKETS);
    if (phase != 2 && !folded) {
        if (hw == 1) {
            nsplit = rows_split(n);
            hipLaunchKernelGGL(stats_rows_kernel, dim3(fsc::ceil_div(c, 128), nsplit), dim3(128), 0, st, x, n, c, p.part);
        } else {
            nsplit = pick_split(n, c, hw);
            FinalizeArgs fa{x, c, hw, (double)n * (double)hw, nsplit, p.part, gamma, beta, eps, momentum, running_mean, running_var,
                            save_mean, save_invstd, scale, shift, sync, phase, x_minmax, pivot_rm, minmax_only};
            unsigned* tickets = nullptr;
            if (phase == 0 && !pivot_rm && zero_tickets) tickets = p.tickets;
            hipLaunchKernelGGL(stats_partial_kernel, dim3(c, nsplit), dim3(kThreads), 0, st, x, n, c, hw, nsplit, p.part, fa, tickets);
            if (tickets) {
                FSC_LAUNCH_CHECK("fsc_bn_train_stats");
                return 0;
            }
        }
    }
    FinalizeArgs fa{x, c, hw, (double)n * (double)hw, nsplit, p.part, gamma, beta, eps, momentum, running_mean, running_var,
                    save_mean, save_invstd, scale, shift, sync, phase, x_minmax, pivot_rm, minmax_only};
    if (folded && pivot_rm && phase != 2 && !minmax_only && hw > 1)
        hipLaunchKernelGGL(stats_finalize_gated_kernel, dim3(c), dim3(kThreads), 0, st, fa, n);
    else
        hipLaunchKernelGGL(stats_finalize_kernel, dim3(fsc::ceil_div(c, 128)), dim3(128), 0, st, fa);
    FSC_LAUNCH_CHECK("fsc_bn_train_stats");
    return 0;
}

// a unit whose channel is one workgroup that holds it in registers (unit_fwd_kernel / unit_bwd_kernel).  The backward keeps 8 - 12
// registers per quad: 8 quads x 1024 threads do not fit (`bwd`: at most 4 there)
static UnitShape fused_unit(int n, int c, long hw, bool bwd = false) {
    static const int mode = [] { const char* e = getenv("FSC_BN_FUSED"); return e ? atoi(e) : 1; }();   // 0: off; 2: 256 threads only (A/B)
    UnitShape u = unit_shape(n, hw);
    if (mode == 0 || n <= 0 || c <= 0 || (mode == 2 && u.threads != 256) || (bwd && u.threads == 1024 && u.trips > 4)) u.threads = 0;
    return u;
}
#define FSC_UNIT_TRIPS(KERNEL_, THREADS_, TRIPS_, ...)                                                                           \
    do {                                                                                                                         \
        if ((TRIPS_) <= 1) hipLaunchKernelGGL((KERNEL_<THREADS_, 1 FSC_UNIT_EXTRA>), dim3(c), dim3(THREADS_), 0, st, __VA_ARGS__);      \
        else if ((TRIPS_) <= 2) hipLaunchKernelGGL((KERNEL_<THREADS_, 2 FSC_UNIT_EXTRA>), dim3(c), dim3(THREADS_), 0, st, __VA_ARGS__); \
        else if ((TRIPS_) <= 4) hipLaunchKernelGGL((KERNEL_<THREADS_, 4 FSC_UNIT_EXTRA>), dim3(c), dim3(THREADS_), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL_<THREADS_, 8 FSC_UNIT_EXTRA>), dim3(c), dim3(THREADS_), 0, st, __VA_ARGS__);                    \
    } while (0)
#define FSC_UNIT_LAUNCH(KERNEL_, U_, ...)                                                \
    do {                                                                                 \
        if ((U_).threads == 256) FSC_UNIT_TRIPS(KERNEL_, 256, (U_).trips, __VA_ARGS__);  \
        else FSC_UNIT_TRIPS(KERNEL_, 1024, (U_).trips, __VA_ARGS__);                     \
    } while (0)

int fsc_bn_train_act_fwd_supported(int n, int c, long hw) {
    if (!fused_unit(n, c, hw).threads) return 0;
    return small_plane_lanes(hw) <= 64 ? 3 : 1;          // bit 1: gmax / gmax_idx can be asked for
}

int fsc_bn_train_act_fwd(const float* x, const float* residual, int n, int c, long hw, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* scale,
                         float* shift, float* x_minmax, const float* alpha, float* y, float* gmax, int* gmax_idx, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && save_mean && save_invstd && scale && shift && y, "fsc_bn_train_act_fwd: null pointer");
    FSC_CHECK_ARG((gmax == nullptr) == (gmax_idx == nullptr), "fsc_bn_train_act_fwd: gmax / gmax_idx must come in pairs");
    FSC_CHECK_ARG(!gmax || small_plane_lanes(hw) <= 64, "fsc_bn_train_act_fwd: the global max needs planes of <= 64 lanes (hw <= ~250)");
    FSC_CHECK_ARG(n > 0 && c > 0 && hw > 0, "fsc_bn_train_act_fwd: bad shape (%d, %d, %ld)", n, c, hw);
    FSC_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "fsc_bn_train_act_fwd: running stats must come in pairs");
    const UnitShape us = fused_unit(n, c, hw);
    FSC_CHECK_ARG(us.threads != 0, "fsc_bn_train_act_fwd: (%d, %d, %ld) is not a one-workgroup-per-channel shape "
                  "(fsc_bn_train_act_fwd_supported)", n, c, hw);
    hipStream_t st = fsc::as_stream(stream);
    FinalizeArgs fa{x, c, hw, (double)n * (double)hw, 1, nullptr, gamma, beta, eps, momentum, running_mean, running_var,
                    save_mean, save_invstd, scale, shift, nullptr, 0, x_minmax, 0, 0};
#define FSC_UNIT_EXTRA
    FSC_UNIT_LAUNCH(unit_fwd_kernel, us, x, n, c, hw, fa, FwdApply{residual, alpha, y, gmax, gmax_idx});
#undef FSC_UNIT_EXTRA
    FSC_LAUNCH_CHECK("fsc_bn_train_act_fwd");
    return 0;
}

int fsc_bn_eval_prepare(int c, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* scale, float* shift, fsc_stream_t stream) {
    FSC_CHECK_ARG(running_mean && running_var && scale && shift && c > 0, "fsc_bn_eval_prepare: bad arguments");
    hipLaunchKernelGGL(eval_prepare_kernel, dim3(fsc::ceil_div(c, 128)), dim3(128), 0, fsc::as_stream(stream), c,
                       gamma, beta, running_mean, running_var, eps, scale, shift);
    FSC_LAUNCH_CHECK("fsc_bn_eval_prepare");
    return 0;
}

static int rec_slices(long hw) { return hw >= 512 ? plane_grid_y(hw) : 1; }

size_t fsc_bn_records_bytes(int n, int c, long hw) {
    if (n <= 0 || c <= 0 || hw < 2) return 0;
    return (size_t)n * c * rec_slices(hw) * sizeof(PlaneRec);
}

int fsc_bn_act_fwd_rec(const float* x, const float* residual, const float* scale, const float* shift,
                       const float* alpha, float* y, int n, int c, long hw, void* records, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && scale && shift && y && records, "fsc_bn_act_fwd_rec: null pointer");
    FSC_CHECK_ARG(n > 0 && c > 0 && hw > 1, "fsc_bn_act_fwd_rec: bad shape (%d, %d, %ld)", n, c, hw);
    hipStream_t st = fsc::as_stream(stream);
    PlaneRec* rec = reinterpret_cast<PlaneRec*>(records);
    const long planes = (long)n * c;
    if (hw >= 512) {
        hipLaunchKernelGGL(fwd_plane_rec_kernel, dim3((unsigned)planes, plane_grid_y(hw)), dim3(kThreads), 0, st, x, residual,
                           scale, shift, alpha, y, c, hw, rec);
    } else {
        const int glog = group_log2(hw), per = 4 * (64 >> glog);
        hipLaunchKernelGGL(fwd_wave_rec_kernel, dim3((unsigned)((planes + per - 1) / per)), dim3(kThreads), 0, st, x, residual, scale,
                           shift, alpha, y, c, hw, planes, rec, glog);
    }
    FSC_LAUNCH_CHECK("fsc_bn_act_fwd_rec");
    return 0;
}

int fsc_bn_records_fold(const void* records, const float* y, int n, int c, long hw, void* stats_workspace, float* gmax,
                        int* gmax_idx, fsc_stream_t stream) {
    FSC_CHECK_ARG(records && y && (stats_workspace || gmax), "fsc_bn_records_fold: null pointer");
    FSC_CHECK_ARG((gmax == nullptr) == (gmax_idx == nullptr), "fsc_bn_records_fold: gmax / gmax_idx must come in pairs");
    FSC_CHECK_ARG(n > 0 && c > 0 && hw > 1, "fsc_bn_records_fold: bad shape (%d, %d, %ld)", n, c, hw);
    double* part = stats_workspace ? carve(stats_workspace, c).part : nullptr;
    hipLaunchKernelGGL(fold_records_kernel, dim3(c), dim3(kThreads), 0, fsc::as_stream(stream),
                       reinterpret_cast<const PlaneRec*>(records), rec_slices(hw), n, c, hw, y, part, gmax, gmax_idx);
    FSC_LAUNCH_CHECK("fsc_bn_records_fold");
    return 0;
}

int fsc_bn_records_fold_conv(const void* records, int workers, int blocks, int co_blk, int order, int c, void* stats_workspace,
                             fsc_stream_t stream) {
    FSC_CHECK_ARG(records && stats_workspace && workers > 0 && blocks > 0 && co_blk > 0 && c > 0 && workers % blocks == 0 &&
                      c <= blocks * co_blk, "fsc_bn_records_fold_conv: bad arguments");
    hipLaunchKernelGGL(fold_conv_records_kernel, dim3(c), dim3(kThreads), 0, fsc::as_stream(stream),
                       reinterpret_cast<const float4*>(records), workers, blocks, co_blk, order, carve(stats_workspace, c).part);
    FSC_LAUNCH_CHECK("fsc_bn_records_fold_conv");
    return 0;
}

int fsc_bn_train_stats_conv(const void* records, int workers, int blocks, int co_blk, int order, const float* x, int n, int c, long hw,
                            const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, float* save_mean, float* save_invstd, float* scale, float* shift, float* x_minmax,
                            fsc_stream_t stream) {
    // save_mean == NULL: only the per-channel min / max of the records (inference: the BatchNorm runs on its running statistics,
    // the L16 producer still needs the range of x)
    const int minmax_only = save_mean == nullptr ? 1 : 0;
    FSC_CHECK_ARG(records && x && (minmax_only ? x_minmax != nullptr : (save_invstd && scale && shift)),
                  "fsc_bn_train_stats_conv: null pointer");
    FSC_CHECK_ARG(workers > 0 && blocks > 0 && co_blk > 0 && workers % blocks == 0 && c > 0 && c <= blocks * co_blk && n > 0 && hw > 1,
                  "fsc_bn_train_stats_conv: bad arguments");
    FSC_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "fsc_bn_train_stats_conv: running stats must come in pairs");
    FinalizeArgs fa{x, c, hw, (double)n * (double)hw, 1, nullptr, gamma, beta, eps, momentum, running_mean, running_var,
                    save_mean, save_invstd, scale, shift, nullptr, 0, x_minmax, 1, minmax_only};
    hipLaunchKernelGGL(stats_fold_finalize_kernel, dim3(c), dim3(kThreads), 0, fsc::as_stream(stream), fa, n,
                       reinterpret_cast<const float4*>(records), workers, blocks, co_blk, order);
    FSC_LAUNCH_CHECK("fsc_bn_train_stats_conv");
    return 0;
}

int fsc_bn_act_fwd(const float* x, const float* residual, const float* scale, const float* shift,
                   const float* alpha, float* y, int n, int c, long hw, float* y_amax, const float* x_minmax,
                   void* y_l16, fsc_stream_t stream) {
    return fsc_bn_act_fwd_limbs(x, residual, scale, shift, alpha, y, n, c, hw, y_amax, x_minmax, y_l16, 2, stream);
}

int fsc_bn_act_fwd_limbs(const float* x, const float* residual, const float* scale, const float* shift,
                         const float* alpha, float* y, int n, int c, long hw, float* y_amax, const float* x_minmax,
                         void* y_l16, int limbs, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && scale && shift && (y || y_l16), "fsc_bn_act_fwd: null pointer");
    FSC_CHECK_ARG(n > 0 && c > 0 && hw > 0, "fsc_bn_act_fwd: bad shape (%d, %d, %ld)", n, c, hw);
    FSC_CHECK_ARG(limbs >= 2 && limbs <= 4, "fsc_bn_act_fwd_limbs: limbs must be 2, 3 or 4 (three scaled fp16 limbs)");
    hipStream_t st = fsc::as_stream(stream);
    if (y_l16) {
        FSC_CHECK_ARG(!residual && hw > 1, "fsc_bn_act_fwd: the L16 output takes no residual and needs hw > 1");
        FSC_CHECK_ARG(limbs == 3 || (x_minmax && y_amax),
                      "fsc_bn_act_fwd: the scaled fp16 L16 outputs need x_minmax (fsc_bn_train_stats) and y_amax");
        const L16Grid g = l16_grid(n, c, hw, true);
        uint4* y16 = reinterpret_cast<uint4*>(y_l16);
#define FSC_FWD_L16(V_, U_)                                                                                                          \
    do {                                                                                                                             \
        if (limbs == 2)                                                                                                              \
            hipLaunchKernelGGL((fwd_l16_kernel<V_, U_, 2>), g.grid, dim3(kThreads), 0, st, x, scale, shift, alpha, x_minmax, y, y16, \
                               y_amax, n, c, hw, g.hwp_log2);                                                                        \
        else if (limbs == 4)                                                                                                         \
            hipLaunchKernelGGL((fwd_l16_kernel<V_, U_, 4>), g.grid, dim3(kThreads), 0, st, x, scale, shift, alpha, x_minmax, y, y16, \
                               y_amax, n, c, hw, g.hwp_log2);                                                                        \
        else                                                                                                                         \
            hipLaunchKernelGGL((fwd_l16_kernel<V_, U_, 3>), g.grid, dim3(kThreads), 0, st, x, scale, shift, alpha, x_minmax, y, y16, \
                               y_amax, n, c, hw, g.hwp_log2);                                                                        \
    } while (0)
        if (g.vec == 4) { if (g.uni) FSC_FWD_L16(4, true); else FSC_FWD_L16(4, false); }
        else { if (g.uni) FSC_FWD_L16(1, true); else FSC_FWD_L16(1, false); }
#undef FSC_FWD_L16
        FSC_LAUNCH_CHECK("fsc_bn_act_fwd(l16)");
        return 0;
    }
    if (y_amax) {
        hipError_t e = hipMemsetAsync(y_amax, 0, fsc::kAmaxFloats * sizeof(float), st);
        FSC_CHECK_ARG(e == hipSuccess, "fsc_bn_act_fwd: memset failed: %s", hipGetErrorString(e));
    }
    const long total = (long)n * c * hw;
    if (hw >= 512) {
        hipLaunchKernelGGL(fwd_plane_kernel, dim3((unsigned)((long)n * c), plane_grid_y(hw)), dim3(kThreads), 0, st, x,
                           residual, scale, shift, alpha, y, c, hw, y_amax);
    } else if (hw > 1) {
        const long planes = (long)n * c;
        const int glog = group_log2(hw), per = 4 * (64 >> glog);
        hipLaunchKernelGGL(fwd_wave_kernel, dim3((unsigned)((planes + per - 1) / per)), dim3(kThreads), 0, st, x, residual,
                           scale, shift, alpha, y, c, hw, planes, y_amax, glog);
    } else {
        long blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(fwd_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, residual, scale, shift,
                           alpha, y, c, hw, total, y_amax);
    }
    FSC_LAUNCH_CHECK("fsc_bn_act_fwd");
    return 0;
}

int fsc_bn_act_bwd(const float* dy, const float* gmax_dy, const int* gmax_idx, const float* x,
                   const float* residual, const float* save_mean, const float* save_invstd,
                   const float* gamma, const float* beta, const float* alpha, float* dx, float* dresidual,
                   float* dgamma, float* dbeta, float* dalpha, float* dx_chan_sum, int n, int c, long hw,
                   void* workspace, float* dx_amax, double* sync, int phase, void* dx_l16, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && save_mean && save_invstd && (dx || dx_l16 || (phase & ~(FSC_BN_TICKETS | FSC_BN_L16_LIMBS3 | FSC_BN_L16_F16X3)) == 1) && workspace,
                  "fsc_bn_act_bwd: null pointer");
    const bool limbs3 = (phase & FSC_BN_L16_LIMBS3) != 0, f16x3 = (phase & FSC_BN_L16_F16X3) != 0;
    FSC_CHECK_ARG(!(limbs3 && f16x3), "fsc_bn_act_bwd: FSC_BN_L16_LIMBS3 and FSC_BN_L16_F16X3 exclude each other");
    FSC_CHECK_ARG(!dx_l16 || ((dx_amax || limbs3) && hw > 1), "fsc_bn_act_bwd: the L16 output needs dx_amax (scaled fp16 limbs) and hw > 1");
    const bool zero_tickets = (phase & FSC_BN_TICKETS) != 0;
    phase &= ~(FSC_BN_TICKETS | FSC_BN_L16_LIMBS3 | FSC_BN_L16_F16X3);
    FSC_CHECK_ARG(phase == 0 || ((phase == 1 || phase == 2) && sync), "fsc_bn_act_bwd: phase 1 / 2 need `sync`");
    FSC_CHECK_ARG(dy || gmax_dy, "fsc_bn_act_bwd: no upstream gradient");
    FSC_CHECK_ARG((gmax_dy == nullptr) == (gmax_idx == nullptr), "fsc_bn_act_bwd: gmax_dy / gmax_idx must come in pairs");
    FSC_CHECK_ARG(n > 0 && c > 0 && hw > 0, "fsc_bn_act_bwd: bad shape (%d, %d, %ld)", n, c, hw);
    FSC_CHECK_ARG((alpha == nullptr) == (dalpha == nullptr) || dalpha == nullptr, "fsc_bn_act_bwd: dalpha without alpha");
    hipStream_t st = fsc::as_stream(stream);
    Partials p = carve(workspace, c);
    BwdArgs a{dy, gmax_dy, gmax_idx, x, residual, save_mean, save_invstd, gamma, beta, alpha, n, c, hw};
    const int nsplit = hw == 1 ? rows_split(n) : pick_split(n, c, hw);
    BwdFinish fin{};
    if (phase == 0 && hw > 1 && zero_tickets) {
        fin = BwdFinish{p.tickets, (double)n * (double)hw, dgamma, dbeta, dalpha, p.coef, dx_chan_sum, dx_amax, dx_l16 ? 1 : 0};
    }
    const UnitShape us = fused_unit(n, c, hw, true);
    if (fin.tickets != nullptr && dx && !dx_l16 && !dx_amax && us.threads) {
        // a channel that is one workgroup: reduce, finalise and apply from registers in one launch
        const BwdApply ap{dx, dresidual, nullptr, nullptr, 0};
        if (residual) {
#define FSC_UNIT_EXTRA , true
            FSC_UNIT_LAUNCH(unit_bwd_kernel, us, a, fin, ap);
#undef FSC_UNIT_EXTRA
        } else {
#define FSC_UNIT_EXTRA , false
            FSC_UNIT_LAUNCH(unit_bwd_kernel, us, a, fin, ap);
#undef FSC_UNIT_EXTRA
        }
        FSC_LAUNCH_CHECK("fsc_bn_act_bwd(unit)");
        return 0;
    }
    if (phase != 2) {
        if (hw == 1) {
            FSC_CHECK_ARG(gmax_dy == nullptr, "fsc_bn_act_bwd: global-max gradient needs hw > 1");
            hipLaunchKernelGGL(bwd_rows_kernel, dim3(fsc::ceil_div(c, 128), nsplit), dim3(128), 0, st, a, p.part);
        } else {
            hipLaunchKernelGGL(bwd_partial_kernel, dim3(c, nsplit), dim3(kThreads), 0, st, a, nsplit, p.part, fin);
        }
    }
    if (!fin.tickets)
        hipLaunchKernelGGL(bwd_finalize_kernel, dim3(fsc::ceil_div(c, 128)), dim3(128), 0, st, c, (double)n * (double)hw,
                           nsplit, p.part, dgamma, dbeta, dalpha, p.coef, dx_chan_sum, dx_amax, sync, phase, gamma, save_invstd,
                           dx_l16 ? 1 : 0);
    if (phase == 1) {
        FSC_LAUNCH_CHECK("fsc_bn_act_bwd");
        return 0;
    }
    if (dx_l16) {
        const L16Grid g = l16_grid(n, c, hw, true);
        uint4* dx16 = reinterpret_cast<uint4*>(dx_l16);
#define FSC_BWD_L16(V_, U_)                                                                                                    \
    do {                                                                                                                       \
        if (f16x3)                                                                                                             \
            hipLaunchKernelGGL((bwd_apply_l16_kernel<V_, U_, 4>), g.grid, dim3(kThreads), 0, st, a, p.coef, dx, dx16, dresidual, \
                               dx_amax, g.hwp_log2);                                                                           \
        else if (!limbs3)                                                                                                      \
            hipLaunchKernelGGL((bwd_apply_l16_kernel<V_, U_, 2>), g.grid, dim3(kThreads), 0, st, a, p.coef, dx, dx16, dresidual, \
                               dx_amax, g.hwp_log2);                                                                           \
        else                                                                                                                   \
            hipLaunchKernelGGL((bwd_apply_l16_kernel<V_, U_, 3>), g.grid, dim3(kThreads), 0, st, a, p.coef, dx, dx16, dresidual, \
                               dx_amax, g.hwp_log2);                                                                           \
    } while (0)
        if (g.vec == 4) { if (g.uni) FSC_BWD_L16(4, true); else FSC_BWD_L16(4, false); }
        else { if (g.uni) FSC_BWD_L16(1, true); else FSC_BWD_L16(1, false); }
#undef FSC_BWD_L16
        FSC_LAUNCH_CHECK("fsc_bn_act_bwd(l16)");
        return 0;
    }
    const long total = (long)n * c * hw;
    if (hw >= 512) {
        hipLaunchKernelGGL(bwd_apply_plane_kernel, dim3((unsigned)((long)n * c), plane_grid_y(hw)), dim3(kThreads), 0,
                           st, a, p.coef, dx, dresidual, dx_amax);
    } else if (hw > 1) {
        const long planes = (long)n * c;
        const int glog = group_log2(hw), per = 4 * (64 >> glog);
        hipLaunchKernelGGL(bwd_apply_wave_kernel, dim3((unsigned)((planes + per - 1) / per)), dim3(kThreads), 0, st, a, p.coef,
                           dx, dresidual, planes, dx_amax, glog);
    } else {
        long blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(bwd_apply_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, p.coef, dx, dresidual,
                           total, dx_amax);
    }
    FSC_LAUNCH_CHECK("fsc_bn_act_bwd");
    return 0;
}

int fsc_bn_act_bwd_unpool(const float* dy, const float* x, const float* save_mean, const float* save_invstd,
                          const float* gamma, const float* beta, const float* alpha, const uint8_t* pool_idx,
                          float* dc, float* dgamma, float* dbeta, float* dalpha, float* dx_chan_sum, int n, int c,
                          int h, int w, int ph, void* workspace, float* dc_amax, double* sync, int phase,
                          void* dc_l16, fsc_stream_t stream) {
    FSC_CHECK_ARG(dy && x && save_mean && save_invstd && pool_idx && (dc || dc_l16) && workspace, "fsc_bn_act_bwd_unpool: null pointer");
    const bool limbs3 = (phase & FSC_BN_L16_LIMBS3) != 0, f16x3 = (phase & FSC_BN_L16_F16X3) != 0;
    FSC_CHECK_ARG(!(limbs3 && f16x3), "fsc_bn_act_bwd_unpool: FSC_BN_L16_LIMBS3 and FSC_BN_L16_F16X3 exclude each other");
    FSC_CHECK_ARG(!dc_l16 || dc_amax || limbs3, "fsc_bn_act_bwd_unpool: the scaled fp16 L16 outputs need dc_amax");
    const bool zero_tickets = (phase & FSC_BN_TICKETS) != 0;
    phase &= ~(FSC_BN_TICKETS | FSC_BN_L16_LIMBS3 | FSC_BN_L16_F16X3);
    FSC_CHECK_ARG(phase == 0 || ((phase == 1 || phase == 2) && sync), "fsc_bn_act_bwd_unpool: phase 1 / 2 need `sync`");
    FSC_CHECK_ARG((ph == 1 || ph == 2) && n > 0 && c > 0 && h >= ph && w >= 2, "fsc_bn_act_bwd_unpool: bad shape");
    const int oh = h / ph, ow = w / 2;
    const long hw = (long)oh * ow;
    hipStream_t st = fsc::as_stream(stream);
    Partials p = carve(workspace, c);
    BwdArgs a{dy, nullptr, nullptr, x, nullptr, save_mean, save_invstd, gamma, beta, alpha, n, c, hw};
    const int nsplit = pick_split(n, c, hw);
    BwdFinish fin{};
    if (phase == 0 && zero_tickets) {
        fin = BwdFinish{p.tickets, (double)n * (double)hw, dgamma, dbeta, dalpha, p.coef, dx_chan_sum, dc_amax, dc_l16 ? 1 : 0};
    }
    const UnitShape us = fused_unit(n, c, hw, true);
    if (fin.tickets != nullptr && dc && !dc_l16 && !dc_amax && h == 1 && ph == 1 && us.threads) {
        const BwdApply ap{nullptr, nullptr, pool_idx, dc, w};
#define FSC_UNIT_EXTRA , false
        FSC_UNIT_LAUNCH(unit_bwd_kernel, us, a, fin, ap);
#undef FSC_UNIT_EXTRA
        FSC_LAUNCH_CHECK("fsc_bn_act_bwd_unpool(unit)");
        return 0;
    }
    if (phase != 2)
        hipLaunchKernelGGL(bwd_partial_kernel, dim3(c, nsplit), dim3(kThreads), 0, st, a, nsplit, p.part, fin);
    if (!fin.tickets)
        hipLaunchKernelGGL(bwd_finalize_kernel, dim3(fsc::ceil_div(c, 128)), dim3(128), 0, st, c, (double)n * (double)hw,
                           nsplit, p.part, dgamma, dbeta, dalpha, p.coef, dx_chan_sum, dc_amax, sync, phase, gamma, save_invstd,
                           dc_l16 ? 1 : 0);
    if (phase == 1) {
        FSC_LAUNCH_CHECK("fsc_bn_act_bwd_unpool");
        return 0;
    }
    if (dc_l16) {
        L16Grid g = l16_grid(n, c, hw, false);
        if (g.uni) g.hwp_log2 = 8;
        if (f16x3)
            hipLaunchKernelGGL(bwd_apply_unpool_l16_kernel<4>, g.grid, dim3(kThreads), 0, st, a, p.coef, pool_idx, dc,
                               reinterpret_cast<uint4*>(dc_l16), h, w, ph, oh, ow, dc_amax, g.hwp_log2);
        else if (!limbs3)
            hipLaunchKernelGGL(bwd_apply_unpool_l16_kernel<2>, g.grid, dim3(kThreads), 0, st, a, p.coef, pool_idx, dc,
                               reinterpret_cast<uint4*>(dc_l16), h, w, ph, oh, ow, dc_amax, g.hwp_log2);
        else
            hipLaunchKernelGGL(bwd_apply_unpool_l16_kernel<3>, g.grid, dim3(kThreads), 0, st, a, p.coef, pool_idx, dc,
                               reinterpret_cast<uint4*>(dc_l16), h, w, ph, oh, ow, dc_amax, g.hwp_log2);
        FSC_LAUNCH_CHECK("fsc_bn_act_bwd_unpool(l16)");
        return 0;
    }
    int cl = 0;
    while ((1 << cl) < ow && cl < 8) ++cl;
    const int per_block = kThreads >> cl;                          // row units (plane, pooled row) per block
    const long row_units = (long)n * c * oh;
    hipLaunchKernelGGL(bwd_apply_unpool_kernel, dim3((unsigned)((row_units + per_block - 1) / per_block)), dim3(kThreads), 0, st, a,
                       p.coef, pool_idx, dc, h, w, ph, oh, ow, cl, dc_amax, row_units);
    FSC_LAUNCH_CHECK("fsc_bn_act_bwd_unpool");
    return 0;
}

}  // extern "C"
