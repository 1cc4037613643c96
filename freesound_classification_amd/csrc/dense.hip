// Classifier head: nn.Linear forward/backward on v_mfma_f32_16x16x4_f32 and nn.Dropout
// (reference networks/classifiers.py:542-549).  The three GEMMs (y = x W^T, dx = dy W,
// dW = dy^T x) share one strided tile kernel: a 64x64 output tile per workgroup, four waves
// in a 2x2 grid owning 32x32 each, K staged through LDS 16 deep in [k][m] order so the MFMA
// operand reads (lane = m, k = lane>>4) are bank-conflict free.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int LDT = BM + 16;   // row stride == 16 (mod 32): the 4 k-planes of one MFMA read hit disjoint banks

struct GemmArgs {
    const float* a; long sam, sak;   // A(m, k) = a[m*sam + k*sak]
    const float* b; long sbk, sbn;   // B(k, n) = b[k*sbk + n*sbn]
    float* c; long ldc;              // C(m, n) = c[m*ldc + n]
    const float* bias;               // per-n, may be null
    int m, n, k;
    int ksplit;       // > 1: blockIdx.z owns a K slice and adds into a zeroed C (few output tiles, long K)
};

__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ float as[BK * LDT];
    __shared__ float bs[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lm = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wm = (wid >> 1) * 32, wn = (wid & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool a_kfast = g.sak == 1, b_kfast = g.sbk == 1;
    const int kchunks = (g.k + BK - 1) / BK;
    const int k_lo = (int)((long)kchunks * blockIdx.z / g.ksplit) * BK;
    const int k_hi = (int)((long)kchunks * (blockIdx.z + 1) / g.ksplit) * BK;
    // the next chunk's global loads are in flight while the MFMAs of the current one run (one wave per SIMD: nothing else
    // hides the ~2 us of a dependent load -> LDS -> MFMA round; 110 -> 40 us on the 128 x 1977 x 1977 products of the head)
    constexpr int EPT = BM * BK / 256;
    float ra[EPT], rb[EPT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * 256;
            int mm, kk;
            if (a_kfast) { kk = e % BK; mm = e / BK; } else { mm = e % BM; kk = e / BM; }
            const int gm = m0 + mm, gk = k0 + kk;
            ra[i] = (gm < g.m && gk < g.k) ? g.a[gm * g.sam + gk * g.sak] : 0.f;
            int nn;
            if (b_kfast) { kk = e % BK; nn = e / BK; } else { nn = e % BN; kk = e / BN; }
            const int gn = n0 + nn, gk2 = k0 + kk;
            rb[i] = (gn < g.n && gk2 < g.k) ? g.b[gk2 * g.sbk + gn * g.sbn] : 0.f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            const int e = tid + i * 256;
            int mm, kk;
            if (a_kfast) { kk = e % BK; mm = e / BK; } else { mm = e % BM; kk = e / BM; }
            as[kk * LDT + mm] = ra[i];
            int nn;
            if (b_kfast) { kk = e % BK; nn = e / BK; } else { nn = e % BN; kk = e / BN; }
            bs[kk * LDT + nn] = rb[i];
        }
    };
    if (k_lo < k_hi) fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += BK) {
        stash();
        __syncthreads();
        if (k0 + BK < k_hi) fetch(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = as[(ks * 4 + kq) * LDT + wm + i * 16 + lm];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = bs[(ks * 4 + kq) * LDT + wn + j * 16 + lm];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: column = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn + j * 16 + lm;
            if (gn >= g.n) continue;
            const float bv = (g.bias && blockIdx.z == 0) ? g.bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm + i * 16 + kq * 4 + r;
                if (gm < g.m) {
                    if (g.ksplit > 1) atomicAdd(g.c + (long)gm * g.ldc + gn, acc[i][j][r] + bv);
                    else g.c[(long)gm * g.ldc + gn] = acc[i][j][r] + bv;
                }
            }
        }
}

// column sums (the bias gradient): 16 columns x 16 row groups per workgroup, the groups folded through LDS.  (A thread per column
// walking all rows was a chain of `rows` dependent loads: 21 us for 128 x 1977.)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int cols) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + tx;
    float s0 = 0.f, s1 = 0.f;
    if (c < cols) {
        int r = ty;
        for (; r + 16 < rows; r += 32) {
            s0 += x[(long)r * cols + c];
            s1 += x[(long)(r + 16) * cols + c];
        }
        if (r < rows) s0 += x[(long)r * cols + c];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && c < cols) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += red[g][tx];
        out[c] = s;
    }
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask,
                                   long count, float p, float inv_keep, uint64_t seed, uint64_t offset) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const uint64_t r = mix64(mix64(seed) ^ (offset + (uint64_t)i));
        const float u = (float)(r >> 40) * (1.0f / 16777216.0f);   // 24-bit uniform in [0, 1)
        const uint8_t keep = u >= p ? 1 : 0;
        mask[i] = keep;
        y[i] = keep ? x[i] * inv_keep : 0.f;
    }
}

__global__ void dropout_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                   float* __restrict__ dx, long count, float inv_keep) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        dx[i] = mask[i] ? dy[i] * inv_keep : 0.f;
}

int run_gemm(const GemmArgs& g_in, hipStream_t st, const char* name) {
    GemmArgs g = g_in;
    dim3 grid(fsc::ceil_div(g.n, BN), fsc::ceil_div(g.m, BM));
    // the head's 128 x 1977 x 1977 products have 62 output tiles: split K so that the chip is filled
    g.ksplit = 1;
    const long tiles = (long)grid.x * grid.y;
    const int kchunks = (g.k + BK - 1) / BK;
    if (tiles < 128 && kchunks >= 16 && g.ldc == g.n) {
        // (512 workgroups: 6.672 -> 6.649 ms per cfg-3 step against 256, the same at 1024 -- beyond that the K slices' atomics cost
        // what the shorter loops save)
        static const long target = [] { const char* e = getenv("FSC_GEMM_WORKGROUPS"); return e ? atol(e) : 512L; }();
        long ks = target / tiles;
        if (ks > kchunks / 4) ks = kchunks / 4;
        if (ks > 16) ks = 16;
        if (ks > 1) {
            g.ksplit = (int)ks;
            hipError_t e = hipMemsetAsync(g.c, 0, sizeof(float) * (size_t)g.m * g.n, st);
            FSC_CHECK_ARG(e == hipSuccess, "%s: memset failed: %s", name, hipGetErrorString(e));
        }
    }
    grid.z = g.ksplit;
    hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, st, g);
    FSC_LAUNCH_CHECK(name);
    return 0;
}

unsigned grid_for(long count) {
    long b = (count + 255) / 256;
    if (b > 4096) b = 4096;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int fsc_linear_fwd(const float* x, const float* w, const float* bias, float* y, int m, int k, int n_out,
                   fsc_stream_t stream) {
    FSC_CHECK_ARG(x && w && y && m > 0 && k > 0 && n_out > 0, "fsc_linear_fwd: bad arguments");
    GemmArgs g{x, k, 1, w, 1, k, y, n_out, bias, m, n_out, k};
    return run_gemm(g, fsc::as_stream(stream), "fsc_linear_fwd");
}

int fsc_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias, int m, int k,
                   int n_out, fsc_stream_t stream) {
    FSC_CHECK_ARG(dy && m > 0 && k > 0 && n_out > 0, "fsc_linear_bwd: bad arguments");
    hipStream_t st = fsc::as_stream(stream);
    if (dx) {
        FSC_CHECK_ARG(w, "fsc_linear_bwd: dx needs w");
        // dx (m, k) = dy (m, n_out) . w (n_out, k)
        GemmArgs g{dy, n_out, 1, w, k, 1, dx, k, nullptr, m, k, n_out};
        int rc = run_gemm(g, st, "fsc_linear_bwd(dx)");
        if (rc) return rc;
    }
    if (dw) {
        FSC_CHECK_ARG(x, "fsc_linear_bwd: dw needs x");
        // dw (n_out, k) = dy^T (n_out, m) . x (m, k)
        GemmArgs g{dy, 1, n_out, x, k, 1, dw, k, nullptr, n_out, k, m};
        int rc = run_gemm(g, st, "fsc_linear_bwd(dw)");
        if (rc) return rc;
    }
    if (dbias) {
        hipLaunchKernelGGL(colsum_kernel, dim3(fsc::ceil_div(n_out, 16)), dim3(256), 0, st, dy, dbias, m, n_out);
        FSC_LAUNCH_CHECK("fsc_linear_bwd(dbias)");
    }
    return 0;
}

int fsc_dropout_fwd(const float* x, float* y, uint8_t* mask, long count, float p, uint64_t seed, uint64_t offset,
                    fsc_stream_t stream) {
    FSC_CHECK_ARG(x && y && mask && count > 0 && p >= 0.f && p < 1.f, "fsc_dropout_fwd: bad arguments (p=%f)", p);
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for(count)), dim3(256), 0, fsc::as_stream(stream), x, y, mask,
                       count, p, 1.0f / (1.0f - p), seed, offset);
    FSC_LAUNCH_CHECK("fsc_dropout_fwd");
    return 0;
}

int fsc_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, long count, float p, fsc_stream_t stream) {
    FSC_CHECK_ARG(dy && dx && mask && count > 0 && p >= 0.f && p < 1.f, "fsc_dropout_bwd: bad arguments");
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, fsc::as_stream(stream), dy, mask, dx,
                       count, 1.0f / (1.0f - p));
    FSC_LAUNCH_CHECK("fsc_dropout_bwd");
    return 0;
}

}  // extern "C"
