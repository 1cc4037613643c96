// RNN aggregation head of the 2-d model (reference networks/classifiers.py:514-522, 592-597):
//   rnn_input = mean(h, dim=2).permute(0, 2, 1)          (N, C, H, W) -> (N, W, C)
//   LayerNorm((C,)) -> bidirectional GRU(C, 128, batch_first); the two final states are the block's features.
// Kernels here: the frequency mean + transpose, LayerNorm, and ONE GRU time step (forward / backward); the
// input projections X W_ih^T for all time steps and the weight gradients are GEMMs (fsc_linear_fwd / fsc_linear_bwd).
// All fp32; gate order r, z, n and the update h' = (1 - z) n + z h follow torch.nn.GRU.
#include "common.h"

namespace {

// ---------------------------------------------------------------- mean over H, (N, C, H, W) -> (N, W, C)
// block = 64 x 4 threads over (w, c-quad) of one image; the H loop reads coalesced rows of x, the result is written
// through an LDS transpose so that both sides stay coalesced.
__global__ __launch_bounds__(256) void freq_mean_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int c, int h,
                                                            int w) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, w0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const float inv = 1.0f / (float)h;
    for (int cc = ty; cc < 32; cc += 8) {
        float s = 0.f;
        if (c0 + cc < c && w0 + tx < w) {
            const float* p = x + (((long)n * c + c0 + cc) * h) * w + w0 + tx;
            for (int r = 0; r < h; ++r) s += p[(long)r * w];
        }
        tile[cc][tx] = s * inv;
    }
    __syncthreads();
    for (int ww = ty; ww < 32; ww += 8)
        if (w0 + ww < w && c0 + tx < c) y[((long)n * w + w0 + ww) * c + c0 + tx] = tile[tx][ww];
}

// dx[n, c, r, w] = dy[n, w, c] / H for every row r
__global__ __launch_bounds__(256) void freq_mean_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int c, int h,
                                                            int w) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, c0 = blockIdx.y * 32, w0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float inv = 1.0f / (float)h;
    for (int ww = ty; ww < 32; ww += 8)
        tile[ww][tx] = (w0 + ww < w && c0 + tx < c) ? dy[((long)n * w + w0 + ww) * c + c0 + tx] * inv : 0.f;
    __syncthreads();
    for (int cc = ty; cc < 32; cc += 8) {
        if (c0 + cc < c && w0 + tx < w) {
            const float v = tile[tx][cc];
            float* p = dx + (((long)n * c + c0 + cc) * h) * w + w0 + tx;
            for (int r = 0; r < h; ++r) p[(long)r * w] = v;
        }
    }
}

// ---------------------------------------------------------------- LayerNorm over the last dim of (rows, C)
// one wave per row; biased variance, eps inside the square root (torch.nn.LayerNorm)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, long rows, int c) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* px = x + row * c;
    float s = 0.f;
    for (int i = lane; i < c; i += 64) s += px[i];
    const float m = fsc::wave_sum(s) / (float)c;
    float v = 0.f;
    for (int i = lane; i < c; i += 64) {
        const float d = px[i] - m;
        v += d * d;
    }
    const float rs = 1.0f / sqrtf(fsc::wave_sum(v) / (float)c + eps);
    float* py = y + row * c;
    for (int i = lane; i < c; i += 64) py[i] = (px[i] - m) * rs * gamma[i] + beta[i];
    if (lane == 0) {
        mean[row] = m;
        rstd[row] = rs;
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; per-row partial dgamma / dbeta go through atomics
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            long rows, int c) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float m = mean[row], rs = rstd[row];
    const float* px = x + row * c;
    const float* pd = dy + row * c;
    float s0 = 0.f, s1 = 0.f;
    for (int i = lane; i < c; i += 64) {
        const float g = pd[i] * gamma[i], xh = (px[i] - m) * rs;
        s0 += g;
        s1 += g * xh;
    }
    s0 = fsc::wave_sum(s0) / (float)c;
    s1 = fsc::wave_sum(s1) / (float)c;
    float* po = dx + row * c;
    for (int i = lane; i < c; i += 64) {
        const float g = pd[i] * gamma[i], xh = (px[i] - m) * rs;
        po[i] = rs * (g - s0 - xh * s1);
    }
}

// dgamma[c] = sum_rows dy * xhat, dbeta[c] = sum_rows dy: one thread per channel, rows strided over blockIdx.y,
// fp32 partials added atomically (dgamma / dbeta cleared by the caller's launch wrapper)
__global__ __launch_bounds__(256) void layernorm_param_grad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta, long rows,
                                                                   int c) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float g = 0.f, b = 0.f;
    for (long r = blockIdx.y; r < rows; r += gridDim.y) {
        const float d = dy[r * c + ch];
        g += d * (x[r * c + ch] - mean[r]) * rstd[r];
        b += d;
    }
    atomicAdd(dgamma + ch, g);
    atomicAdd(dbeta + ch, b);
}

// ---------------------------------------------------------------- one GRU time step
__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + __expf(-v)); }

// thread = (sample n, hidden unit j).  gx: the input projections of this time step, row stride gx_stride, columns
// [r | z | n] x hidden (b_ih already added).  Saves r, z, n~ and the hidden part of the candidate (W_hn h + b_hn).
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const float* __restrict__ gx, long gx_stride,
                                                           const float* __restrict__ h_prev, const float* __restrict__ w_hh,
                                                           const float* __restrict__ b_hh, float* __restrict__ h_out,
                                                           float* __restrict__ r_s, float* __restrict__ z_s, float* __restrict__ n_s,
                                                           float* __restrict__ ghn_s, int batch, int hid) {
    extern __shared__ float hs[];                        // h_prev rows of the samples of this block: [spb][hid]
    const int spb = blockDim.x / hid;                    // samples per block (host: hid divides 256 or blockDim == hid)
    const int ls = threadIdx.x / hid, j = threadIdx.x - ls * hid;
    const int n = blockIdx.x * spb + ls;
    if (n < batch) hs[ls * hid + j] = h_prev[(long)n * hid + j];
    __syncthreads();
    if (n >= batch) return;
    const float* hp = hs + ls * hid;
    const float* wr = w_hh + (long)j * hid;
    const float* wz = w_hh + (long)(hid + j) * hid;
    const float* wn = w_hh + (long)(2 * hid + j) * hid;
    float ar = b_hh[j], az = b_hh[hid + j], an = b_hh[2 * hid + j];
    for (int k = 0; k < hid; ++k) {
        const float hv = hp[k];
        ar = fmaf(wr[k], hv, ar);
        az = fmaf(wz[k], hv, az);
        an = fmaf(wn[k], hv, an);
    }
    const float* g = gx + (long)n * gx_stride;
    const float r = sigmoidf(g[j] + ar), z = sigmoidf(g[hid + j] + az);
    const float nt = tanhf(g[2 * hid + j] + r * an);
    const long o = (long)n * hid + j;
    h_out[o] = (1.0f - z) * nt + z * hp[j];
    if (r_s) {
        r_s[o] = r;
        z_s[o] = z;
        n_s[o] = nt;
        ghn_s[o] = an;
    }
}

// gate gradients of one step: dgx (row stride dgx_stride) and dgh (batch x 3 hid, dense), and the direct part of dh_prev
__global__ __launch_bounds__(256) void gru_step_bwd_gates_kernel(const float* __restrict__ dh, const float* __restrict__ r_s,
                                                                 const float* __restrict__ z_s, const float* __restrict__ n_s,
                                                                 const float* __restrict__ ghn_s, const float* __restrict__ h_prev,
                                                                 float* __restrict__ dgx, long dgx_stride, float* __restrict__ dgh,
                                                                 float* __restrict__ dh_prev, int batch, int hid) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)batch * hid) return;
    const int n = (int)(idx / hid), j = (int)(idx - (long)n * hid);
    const float d = dh[idx], r = r_s[idx], z = z_s[idx], nt = n_s[idx];
    const float dn = d * (1.0f - z), dz = d * (h_prev[idx] - nt);
    const float dan = dn * (1.0f - nt * nt);
    const float dar = dan * ghn_s[idx] * r * (1.0f - r);
    const float daz = dz * z * (1.0f - z);
    float* gxo = dgx + (long)n * dgx_stride;
    float* gho = dgh + (long)n * 3 * hid;
    gxo[j] = dar;
    gxo[hid + j] = daz;
    gxo[2 * hid + j] = dan;
    gho[j] = dar;
    gho[hid + j] = daz;
    gho[2 * hid + j] = dan * r;
    dh_prev[idx] = d * z;
}

// dh_prev[n, k] += sum_g dgh[n, g] * w_hh[g, k]   (g over 3 hid)
__global__ __launch_bounds__(256) void gru_step_bwd_hidden_kernel(const float* __restrict__ dgh, const float* __restrict__ w_hh,
                                                                  float* __restrict__ dh_prev, int batch, int hid) {
    extern __shared__ float gs[];                        // dgh rows of this block's samples: [spb][3 hid]
    const int spb = blockDim.x / hid;
    const int ls = threadIdx.x / hid, k = threadIdx.x - ls * hid;
    const int n = blockIdx.x * spb + ls;
    for (int i = k; i < 3 * hid; i += hid) gs[ls * 3 * hid + i] = n < batch ? dgh[(long)n * 3 * hid + i] : 0.f;
    __syncthreads();
    if (n >= batch) return;
    const float* g = gs + ls * 3 * hid;
    float acc = 0.f;
    for (int q = 0; q < 3 * hid; ++q) acc = fmaf(g[q], w_hh[(long)q * hid + k], acc);       // coalesced over k
    dh_prev[(long)n * hid + k] += acc;
}

}  // namespace

extern "C" {

int fsc_freq_mean_fwd(const float* x, float* y, int n, int c, int h, int w, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && y && n > 0 && c > 0 && h > 0 && w > 0, "fsc_freq_mean_fwd: bad arguments");
    hipLaunchKernelGGL(freq_mean_fwd_kernel, dim3(fsc::ceil_div(w, 32), fsc::ceil_div(c, 32), n), dim3(256), 0,
                       fsc::as_stream(stream), x, y, c, h, w);
    FSC_LAUNCH_CHECK("fsc_freq_mean_fwd");
    return 0;
}

int fsc_freq_mean_bwd(const float* dy, float* dx, int n, int c, int h, int w, fsc_stream_t stream) {
    FSC_CHECK_ARG(dy && dx && n > 0 && c > 0 && h > 0 && w > 0, "fsc_freq_mean_bwd: bad arguments");
    hipLaunchKernelGGL(freq_mean_bwd_kernel, dim3(fsc::ceil_div(w, 32), fsc::ceil_div(c, 32), n), dim3(256), 0,
                       fsc::as_stream(stream), dy, dx, c, h, w);
    FSC_LAUNCH_CHECK("fsc_freq_mean_bwd");
    return 0;
}

int fsc_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, float* mean, float* rstd,
                      long rows, int c, fsc_stream_t stream) {
    FSC_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0 && c > 0, "fsc_layernorm_fwd: bad arguments");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, fsc::as_stream(stream), x, gamma,
                       beta, eps, y, mean, rstd, rows, c);
    FSC_LAUNCH_CHECK("fsc_layernorm_fwd");
    return 0;
}

int fsc_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                      float* dgamma, float* dbeta, long rows, int c, fsc_stream_t stream) {
    FSC_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && rows > 0 && c > 0, "fsc_layernorm_bwd: bad arguments");
    hipStream_t st = fsc::as_stream(stream);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, dy, x, mean, rstd, gamma, dx, rows, c);
    hipError_t e = hipMemsetAsync(dgamma, 0, sizeof(float) * c, st);
    if (e == hipSuccess) e = hipMemsetAsync(dbeta, 0, sizeof(float) * c, st);
    FSC_CHECK_ARG(e == hipSuccess, "fsc_layernorm_bwd: memset failed: %s", hipGetErrorString(e));
    long gy = (rows + 63) / 64;
    if (gy > 256) gy = 256;
    hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3(fsc::ceil_div(c, 256), (unsigned)gy), dim3(256), 0, st, dy, x, mean, rstd,
                       dgamma, dbeta, rows, c);
    FSC_LAUNCH_CHECK("fsc_layernorm_bwd");
    return 0;
}

static int gru_block(int hid) { return hid >= 256 ? hid : (256 / hid) * hid; }

int fsc_gru_step_fwd(const float* gx, long gx_stride, const float* h_prev, const float* w_hh, const float* b_hh, float* h_out,
                     float* r_save, float* z_save, float* n_save, float* ghn_save, int batch, int hidden, fsc_stream_t stream) {
    FSC_CHECK_ARG(gx && h_prev && w_hh && b_hh && h_out && batch > 0, "fsc_gru_step_fwd: bad arguments");
    FSC_CHECK_ARG(hidden > 0 && hidden <= 1024 && (hidden >= 256 || 256 % hidden == 0), "fsc_gru_step_fwd: hidden size %d", hidden);
    FSC_CHECK_ARG((r_save == nullptr) == (z_save == nullptr) && (r_save == nullptr) == (n_save == nullptr) &&
                  (r_save == nullptr) == (ghn_save == nullptr), "fsc_gru_step_fwd: the four gate buffers come together");
    const int threads = gru_block(hidden), spb = threads / hidden;
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(fsc::ceil_div(batch, spb)), dim3(threads), sizeof(float) * threads,
                       fsc::as_stream(stream), gx, gx_stride, h_prev, w_hh, b_hh, h_out, r_save, z_save, n_save, ghn_save, batch, hidden);
    FSC_LAUNCH_CHECK("fsc_gru_step_fwd");
    return 0;
}

int fsc_gru_step_bwd(const float* dh, const float* r_save, const float* z_save, const float* n_save, const float* ghn_save,
                     const float* h_prev, const float* w_hh, float* dgx, long dgx_stride, float* dgh, float* dh_prev, int batch,
                     int hidden, fsc_stream_t stream) {
    FSC_CHECK_ARG(dh && r_save && z_save && n_save && ghn_save && h_prev && w_hh && dgx && dgh && dh_prev && batch > 0,
                  "fsc_gru_step_bwd: bad arguments");
    FSC_CHECK_ARG(hidden > 0 && hidden <= 1024 && (hidden >= 256 || 256 % hidden == 0), "fsc_gru_step_bwd: hidden size %d", hidden);
    hipStream_t st = fsc::as_stream(stream);
    const long total = (long)batch * hidden;
    hipLaunchKernelGGL(gru_step_bwd_gates_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dh, r_save, z_save,
                       n_save, ghn_save, h_prev, dgx, dgx_stride, dgh, dh_prev, batch, hidden);
    const int threads = gru_block(hidden), spb = threads / hidden;
    hipLaunchKernelGGL(gru_step_bwd_hidden_kernel, dim3(fsc::ceil_div(batch, spb)), dim3(threads), sizeof(float) * 3 * threads, st,
                       dgh, w_hh, dh_prev, batch, hidden);
    FSC_LAUNCH_CHECK("fsc_gru_step_bwd");
    return 0;
}

}  // extern "C"
