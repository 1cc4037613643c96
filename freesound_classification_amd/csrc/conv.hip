// Implicit-GEMM convolutions for gfx950 on v_mfma_f32_16x16x4_f32 (exact fp32, 157 TF peak).
// Replaces nn.Conv2d 3x3/1x1 and nn.Conv1d k3/k1 forward, input-gradient and weight-gradient
// (reference networks/classifiers.py:526-531, 77-81, 149-154, 42-46), NCHW, stride 1, "same"
// padding.  These kernels carry ~99 % of the step's FLOPs and are MFMA-bound.
//
// Forward / dgrad (one kernel; dgrad = forward with the flipped, transposed packed weights):
//   D[m = out channel][n = pixel] += A[m][k] * B[k][n],  k = (tap, in channel)
//   * workgroup = 4 waves; output tile = COT*16 channels x PT*64 pixels; the pixel tile is a
//     (nb images) x (th rows) x (tw cols) box chosen per layer on the host so odd widths
//     (431, 215, 107, 53, ...) waste few MFMA columns;
//   * per K-chunk of KC input channels the halo'd input box [KC][rows][cols] and the weight
//     slab [tap][KC][COT*16] are staged in LDS; plane strides are == 16 (mod 32) so the
//     four k-planes an MFMA operand read touches sit on disjoint banks;
//   * global loads of chunk c+1 are issued into registers before the MFMAs of chunk c
//     (issue-early / write-late), one LDS buffer.
// Wgrad:
//   D[m = out channel][n = (in channel, tap)] += dOut[m][pixel] * In[n][pixel + tap]
//   * split-K over pixel tiles; partials [split][tap][ci][co] reduced by a second kernel that
//     also transposes into the (c_out, c_in, kh, kw) layout of the state dict.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;

// -------------------------------------------------------------------------------------------
// geometry shared by host and device
struct Geom {
    int n, cin, cout, h, w;   // cin: channels of the tensor read, cout: channels of the tensor written
    long hw;
    int nb, th, tw;           // pixel box: images x rows x cols
    int tiles_n, tiles_h, tiles_w;
    int rows, cols;           // staged box incl. halo
    int plane;                // LDS stride between staged channels (== 16 mod 32 for fwd, == 2 mod 32 for wgrad)
    int npos;                 // nb * rows * cols
    int npix;                 // nb * th * tw   (<= tile capacity)
    int k_pad, m_pad;         // packed weight extents
    int flat;                 // 1x1: pixels are a flat (n, hw) range, box = [p0, p0 + npix)
    long flat_total;          // n * hw
};

template <int KH, int KW>
struct FwdCfg {
    static constexpr int TAPS = KH * KW;
    static constexpr int KC = TAPS == 1 ? 32 : 8;                 // input channels per chunk
    static constexpr int MAXE = TAPS == 1 ? 16 : 8;               // staged input elements per thread
};

// -------------------------------------------------------------------------------------------
template <int KH, int KW, int COT, int PT>
__global__ __launch_bounds__(kThreads) void conv_fwd_kernel(Geom g, const float* __restrict__ in,
                                                            const float* __restrict__ packed,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out, int accumulate) {
    using C = FwdCfg<KH, KW>;
    constexpr int TAPS = C::TAPS, KC = C::KC, MAXE = C::MAXE;
    constexpr int CO_BLK = COT * 16;
    constexpr int COS = CO_BLK + ((CO_BLK % 32 == 16) ? 0 : 16);   // == 16 (mod 32)
    constexpr int W4_PER_ROW = CO_BLK / 4;
    constexpr int W4_TOTAL = TAPS * KC * W4_PER_ROW;
    constexpr int NW4 = (W4_TOTAL + kThreads - 1) / kThreads;
    constexpr int PADH = KH / 2, PADW = KW / 2;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                       // [TAPS*KC][COS]
    float* il = smem + TAPS * KC * COS;     // [KC][plane]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lm = lane & 15, kq = lane >> 4;

    // ---- which box
    int t = blockIdx.x;
    const int twi = t % g.tiles_w; t /= g.tiles_w;
    const int thi = t % g.tiles_h; t /= g.tiles_h;
    const int n0 = t * g.nb;
    const int h0 = thi * g.th, w0 = twi * g.tw;
    const long p0 = (long)blockIdx.x * g.npix;          // flat mode
    const int co0 = blockIdx.y * CO_BLK;

    // ---- per-thread staging plan for the input box (chunk-invariant)
    int e_goff[MAXE];     // offset inside the input tensor relative to channel ci0 of image 0, or -1
    int e_lk[MAXE];       // (k << 20) | lds offset
    const int n_elems = KC * g.npos;
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        const int e = tid + q * kThreads;
        e_goff[q] = -1;
        e_lk[q] = -1;
        if (e < n_elems) {
            const int k = e / g.npos, pos = e - k * g.npos;
            long goff = -1;
            int loff;
            if (g.flat) {
                const long pg = p0 + pos;
                loff = pos;
                if (pos < g.npix && pg < g.flat_total) {
                    const long img = pg / g.hw, i = pg - img * g.hw;
                    goff = (img * g.cin + k) * g.hw + i;
                }
            } else {
                const int per = g.rows * g.cols;
                const int b = pos / per, rem = pos - b * per;
                const int rr = rem / g.cols, cc = rem - rr * g.cols;
                const int gh = h0 + rr - PADH, gw = w0 + cc - PADW;
                loff = pos;
                if (n0 + b < g.n && gh >= 0 && gh < g.h && gw >= 0 && gw < g.w)
                    goff = ((long)(n0 + b) * g.cin + k) * g.hw + (long)gh * g.w + gw;
            }
            e_goff[q] = (int)goff;      // tensors on this path stay below 2^31 elements (checked on host)
            e_lk[q] = (k << 20) | (k * g.plane + loff);
        }
    }

    // ---- this lane's output pixels
    int pix_l[PT];        // LDS offset of the pixel inside one staged channel (tap (0,0))
    long pix_g[PT];       // offset inside the output tensor for channel 0, or -1
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wid * PT + pt) * 16 + lm;
        pix_l[pt] = 0;
        pix_g[pt] = -1;
        if (p < g.npix) {
            if (g.flat) {
                const long pg = p0 + p;
                pix_l[pt] = p;
                if (pg < g.flat_total) {
                    const long img = pg / g.hw, i = pg - img * g.hw;
                    pix_g[pt] = img * g.cout * g.hw + i;
                }
            } else {
                const int per = g.th * g.tw;
                const int b = p / per, rem = p - b * per;
                const int r = rem / g.tw, c = rem - r * g.tw;
                pix_l[pt] = (b * g.rows + r) * g.cols + c;
                if (n0 + b < g.n && h0 + r < g.h && w0 + c < g.w)
                    pix_g[pt] = (long)(n0 + b) * g.cout * g.hw + (long)(h0 + r) * g.w + (w0 + c);
            }
        }
    }

    f32x4 acc[COT][PT];
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 sw[NW4];
    float sx[MAXE];

    auto load_chunk = [&](int ci0) {
#pragma unroll
        for (int q = 0; q < NW4; ++q) {
            const int idx = tid + q * kThreads;
            sw[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < W4_TOTAL) {
                const int row = idx / W4_PER_ROW, c4 = idx - row * W4_PER_ROW;
                const int tap = row / KC, k = row - tap * KC;
                sw[q] = *reinterpret_cast<const float4*>(packed + ((long)tap * g.k_pad + ci0 + k) * g.m_pad + co0 + c4 * 4);
            }
        }
        const long cbase = (long)ci0 * g.hw;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            float v = 0.f;
            if (e_goff[q] >= 0 && ci0 + (e_lk[q] >> 20) < g.cin) v = in[cbase + e_goff[q]];
            sx[q] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int q = 0; q < NW4; ++q) {
            const int idx = tid + q * kThreads;
            if (idx < W4_TOTAL) {
                const int row = idx / W4_PER_ROW, c4 = idx - row * W4_PER_ROW;
                *reinterpret_cast<float4*>(wl + row * COS + c4 * 4) = sw[q];
            }
        }
#pragma unroll
        for (int q = 0; q < MAXE; ++q)
            if (e_lk[q] >= 0) il[e_lk[q] & 0xFFFFF] = sx[q];
    };

    const int nchunks = g.k_pad / KC;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        if (more) load_chunk((c + 1) * KC);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int tapoff = (tap / KW) * g.cols + (tap % KW);
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                float a[COT], b[PT];
                const float* wrow = wl + (tap * KC + ks * 4 + kq) * COS + lm;
#pragma unroll
                for (int i = 0; i < COT; ++i) a[i] = wrow[i * 16];
                const float* irow = il + (ks * 4 + kq) * g.plane + tapoff;
#pragma unroll
                for (int j = 0; j < PT; ++j) b[j] = irow[pix_l[j]];
#pragma unroll
                for (int i = 0; i < COT; ++i)
#pragma unroll
                    for (int j = 0; j < PT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue: D row = channel (kq*4 + r), column = pixel (lm)
#pragma unroll
    for (int i = 0; i < COT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + i * 16 + kq * 4 + r;
            if (co >= g.cout) continue;
            const float bv = bias ? bias[co] : 0.f;
#pragma unroll
            for (int j = 0; j < PT; ++j)
                if (pix_g[j] >= 0) {
                    float* o = out + pix_g[j] + (long)co * g.hw;
                    const float v = acc[i][j][r] + bv;
                    *o = accumulate ? *o + v : v;
                }
        }
    }
}

// -------------------------------------------------------------------------------------------
// weight (c_out, c_in, kh, kw) -> packed[tap][k][m]  (fwd: k = c_in, m = c_out;
// dgrad: k = c_out, m = c_in, taps mirrored)
__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int c_out, int c_in, int kh,
                            int kw, int k_pad, int m_pad, int dgrad) {
    const int taps = kh * kw;
    const long total = (long)taps * k_pad * m_pad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i % m_pad);
        const long r = i / m_pad;
        const int k = (int)(r % k_pad);
        const int tap = (int)(r / k_pad);
        const int co = dgrad ? k : m, ci = dgrad ? m : k;
        float v = 0.f;
        if (co < c_out && ci < c_in) {
            const int src_tap = dgrad ? (taps - 1 - tap) : tap;
            v = w[((long)co * c_in + ci) * taps + src_tap];
        }
        packed[i] = v;
    }
}

// -------------------------------------------------------------------------------------------
// weight gradient
template <int KH, int KW>
struct WgCfg {
    static constexpr int TAPS = KH * KW;
    static constexpr int WAVES = TAPS == 1 ? 4 : 6;
    static constexpr int NPW = TAPS == 9 ? 3 : 2;                 // (ci-tile, tap) pairs per wave
    static constexpr int CIT = WAVES * NPW / TAPS;                // ci tiles (16 channels) per workgroup
    static constexpr int PIXC = 64;                               // pixels per chunk (K of the GEMM)
    static constexpr int MAXPOS = TAPS == 1 ? 1 : (TAPS == 3 ? 2 : 3);   // staged positions per lane (x 64)
};

struct WgGeom {
    int n, cin, cout, h, w;
    long hw;
    int nb, th, tw, tiles_n, tiles_h, tiles_w;
    int rows, cols, plane, npos, npix;
    int ci_pad, co_pad;
    int units;            // pixel tiles in total
    int nsplit;
    int ci_blocks;
};

template <int KH, int KW>
constexpr int wg_threads() { return WgCfg<KH, KW>::WAVES * 64; }

template <int KH, int KW, int MT>
__global__ __launch_bounds__((wg_threads<KH, KW>())) void conv_wgrad_kernel(WgGeom g, const float* __restrict__ in,
                                                                               const float* __restrict__ dout,
                                                                               float* __restrict__ part) {
    using C = WgCfg<KH, KW>;
    constexpr int TAPS = C::TAPS, WAVES = C::WAVES, NPW = C::NPW, CIT = C::CIT, PIXC = C::PIXC, MAXPOS = C::MAXPOS;
    constexpr int CO_BLK = MT * 16, CI_BLK = CIT * 16;
    constexpr int DS = PIXC + 2;                                   // == 2 (mod 32)
    constexpr int ND = (CO_BLK + WAVES - 1) / WAVES;               // dout rows staged per wave
    constexpr int NI = (CI_BLK + WAVES - 1) / WAVES;               // input channels staged per wave
    constexpr int PADH = KH / 2, PADW = KW / 2;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dl = smem;                              // [CO_BLK][DS]
    float* il = smem + CO_BLK * DS;                // [CI_BLK][plane]
    int* ptab = reinterpret_cast<int*>(il + CI_BLK * g.plane);   // [PIXC] LDS offset of pixel p inside a staged channel

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lm = lane & 15, kq = lane >> 4;
    const int co0 = (blockIdx.x / g.ci_blocks) * CO_BLK;
    const int ci0 = (blockIdx.x % g.ci_blocks) * CI_BLK;
    const int split = blockIdx.y;

    // chunk-invariant decode of this lane's pixel (for dout) and staged positions (for in)
    const int per_pix = g.th * g.tw;
    int pb = 0, pr = 0, pc = 0;
    const bool pix_live = lane < g.npix;
    if (pix_live) {
        pb = lane / per_pix;
        const int rem = lane - pb * per_pix;
        pr = rem / g.tw;
        pc = rem - pr * g.tw;
    }
    if (tid < PIXC) ptab[tid] = pix_live ? (pb * g.rows + pr) * g.cols + pc : 0;
    int qb[MAXPOS], qr[MAXPOS], qc[MAXPOS];
    const int per_pos = g.rows * g.cols;
#pragma unroll
    for (int j = 0; j < MAXPOS; ++j) {
        const int pos = lane + 64 * j;
        qb[j] = -1; qr[j] = 0; qc[j] = 0;
        if (pos < g.npos) {
            qb[j] = pos / per_pos;
            const int rem = pos - qb[j] * per_pos;
            qr[j] = rem / g.cols;
            qc[j] = rem - qr[j] * g.cols;
        }
    }

    f32x4 acc[NPW][MT];
#pragma unroll
    for (int s = 0; s < NPW; ++s)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[s][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float sd[ND];
    float sx[NI][MAXPOS];

    auto load_unit = [&](int u) {
        int t = u;
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h; t /= g.tiles_h;
        const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
        long dgo = -1;
        if (pix_live && n0 + pb < g.n && h0 + pr < g.h && w0 + pc < g.w)
            dgo = (long)(n0 + pb) * g.cout * g.hw + (long)(h0 + pr) * g.w + (w0 + pc);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int col = wid + i * WAVES;
            float v = 0.f;
            if (col < CO_BLK && co0 + col < g.cout && dgo >= 0) v = dout[dgo + (long)(co0 + col) * g.hw];
            sd[i] = v;
        }
        long xgo[MAXPOS];
#pragma unroll
        for (int j = 0; j < MAXPOS; ++j) {
            xgo[j] = -1;
            if (qb[j] >= 0) {
                const int gh = h0 + qr[j] - PADH, gw = w0 + qc[j] - PADW;
                if (n0 + qb[j] < g.n && gh >= 0 && gh < g.h && gw >= 0 && gw < g.w)
                    xgo[j] = (long)(n0 + qb[j]) * g.cin * g.hw + (long)gh * g.w + gw;
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int cl = wid + i * WAVES;
            const bool chan_ok = cl < CI_BLK && ci0 + cl < g.cin;
#pragma unroll
            for (int j = 0; j < MAXPOS; ++j) {
                float v = 0.f;
                if (chan_ok && xgo[j] >= 0) v = in[xgo[j] + (long)(ci0 + cl) * g.hw];
                sx[i][j] = v;
            }
        }
    };
    auto store_unit = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int col = wid + i * WAVES;
            if (col < CO_BLK) dl[col * DS + lane] = sd[i];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int cl = wid + i * WAVES;
            if (cl < CI_BLK) {
#pragma unroll
                for (int j = 0; j < MAXPOS; ++j)
                    if (qb[j] >= 0) il[cl * g.plane + lane + 64 * j] = sx[i][j];
            }
        }
    };

    int u = split;
    if (u < g.units) {
        load_unit(u);
        store_unit();
    }
    __syncthreads();
    for (; u < g.units; u += g.nsplit) {
        const int un = u + g.nsplit;
        const bool more = un < g.units;
        if (more) load_unit(un);
#pragma unroll 4
        for (int ks = 0; ks < PIXC / 4; ++ks) {
            const int p = ks * 4 + kq;
            float a[MT], b[NPW];
            const int poff = ptab[p];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = dl[(i * 16 + lm) * DS + p];
#pragma unroll
            for (int s = 0; s < NPW; ++s) {
                const int nt = wid * NPW + s;
                const int cit = nt / TAPS, tap = nt - cit * TAPS;
                b[s] = il[(cit * 16 + lm) * g.plane + poff + (tap / KW) * g.cols + (tap % KW)];
            }
#pragma unroll
            for (int s = 0; s < NPW; ++s)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    acc[s][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[s], acc[s][i], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_unit();
            __syncthreads();
        }
    }

    // partial[split][tap][ci][co]; D row = co (kq*4 + r), column = ci (lm)
#pragma unroll
    for (int s = 0; s < NPW; ++s) {
        const int nt = wid * NPW + s;
        const int cit = nt / TAPS, tap = nt - cit * TAPS;
        if (cit >= CIT) continue;
        const long row = ((long)split * TAPS + tap) * g.ci_pad + ci0 + cit * 16 + lm;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float4 v = make_float4(acc[s][i][0], acc[s][i][1], acc[s][i][2], acc[s][i][3]);
            *reinterpret_cast<float4*>(part + row * g.co_pad + co0 + i * 16 + kq * 4) = v;
        }
    }
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int c_out, int c_in,
                                    int taps, int ci_pad, int co_pad, int nsplit) {
    const long total = (long)c_out * c_in * taps;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i enumerates (tap, ci, co) with co fastest so the partial reads coalesce
        const int co = (int)(i % c_out);
        const long r = i / c_out;
        const int ci = (int)(r % c_in);
        const int tap = (int)(r / c_in);
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += part[(((long)sp * taps + tap) * ci_pad + ci) * co_pad + co];
        dw[((long)co * c_in + ci) * taps + tap] = s;
    }
}

// -------------------------------------------------------------------------------------------
// host-side planning
struct FwdPlan {
    Geom g;
    int cot;          // channel tiles per workgroup
    int co_blocks;
    int kc;
    size_t lds_bytes;
    long grid_x;
};

int pad_plane(int floats, int want) {   // smallest p >= floats with p % 32 == want
    int p = floats;
    while (p % 32 != want) ++p;
    return p;
}

// Narrow boxes cost global-memory efficiency (row segments shorter than a 64/128-byte line), so
// the box search trades a few percent of MFMA columns for wider rows.  Returned in 1/100.
long box_penalty(int tw, int w) {
    if (tw >= 32 || tw == w) return 100;
    if (tw >= 16) return 102;
    if (tw >= 8) return 108;
    return 125;
}

// Few channel tiles per workgroup mean little operand reuse per LDS read; weigh padded tiles
// by this factor (1/100) when splitting the output channels into blocks.
long tile_penalty(int tiles_per_block) {
    switch (tiles_per_block) {
        case 1: return 140;
        case 2: return 120;
        case 3: return 110;
        case 4: return 105;
        case 5: return 102;
        default: return 100;
    }
}

constexpr int kPT = 2;                  // pixel tiles (16) per wave -> 128 pixels per workgroup
constexpr int kPixCap = 4 * kPT * 16;

bool plan_fwd(const fsc_conv_desc& d, int dgrad, FwdPlan* out) {
    FwdPlan p{};
    Geom& g = p.g;
    const int taps = d.kh * d.kw;
    g.n = d.n; g.h = d.h; g.w = d.w; g.hw = (long)d.h * d.w;
    g.cin = dgrad ? d.c_out : d.c_in;
    g.cout = dgrad ? d.c_in : d.c_out;
    const int kc = taps == 1 ? 32 : 8;
    const int maxe = taps == 1 ? 16 : 8;
    p.kc = kc;
    // channel tiling: minimise padded tiles, prefer fewer blocks
    const int tiles = fsc::ceil_div(g.cout, 16);
    int best_cot = 1, best_blocks = tiles;
    long best_tile_cost = (long)tiles * tile_penalty(1);
    for (int cot = 2; cot <= 8; ++cot) {
        const int blocks = fsc::ceil_div(tiles, cot);
        const long cost = (long)blocks * cot * tile_penalty(cot);
        if (cost < best_tile_cost || (cost == best_tile_cost && blocks < best_blocks)) {
            best_cot = cot; best_blocks = blocks; best_tile_cost = cost;
        }
    }
    p.cot = best_cot;
    p.co_blocks = best_blocks;
    g.m_pad = best_blocks * best_cot * 16;
    g.k_pad = (int)fsc::round_up(g.cin, kc);
    if (taps == 1) {
        g.flat = 1;
        g.flat_total = (long)d.n * g.hw;
        g.nb = 1; g.th = 1; g.tw = kPixCap; g.rows = 1; g.cols = kPixCap;
        g.npix = kPixCap; g.npos = kPixCap;
        g.tiles_n = 1; g.tiles_h = 1; g.tiles_w = 1;
        p.grid_x = (g.flat_total + kPixCap - 1) / kPixCap;
        g.tiles_w = (int)p.grid_x;      // so the box decode in the kernel stays in range
    } else {
        g.flat = 0;
        const int cap_pos = 256 * maxe / kc;
        long best_cost = -1;
        int bnb = 1, bth = 1, btw = 1;
        for (int tw = 1; tw <= d.w && tw <= kPixCap; ++tw) {
            int th = kPixCap / tw;
            if (th > d.h) th = d.h;
            int nb = 1;
            if (th == d.h && tw == d.w) {            // whole image fits: pack several images per box
                nb = kPixCap / (th * tw);
                if (nb > d.n) nb = d.n;
                if (nb < 1) nb = 1;
            }
            while (nb > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) --nb;
            while (th > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) --th;
            if (nb * (th + d.kh - 1) * (tw + d.kw - 1) > cap_pos) continue;
            const long tiles = (long)fsc::ceil_div(d.n, nb) * fsc::ceil_div(d.h, th) * fsc::ceil_div(d.w, tw);
            const long cost = tiles * box_penalty(tw, d.w);
            if (best_cost < 0 || cost < best_cost || (cost == best_cost && tw > btw)) {
                best_cost = cost; bnb = nb; bth = th; btw = tw;
            }
        }
        if (best_cost < 0) return false;
        g.nb = bnb; g.th = bth; g.tw = btw;
        g.rows = bth + d.kh - 1; g.cols = btw + d.kw - 1;
        g.npix = bnb * bth * btw; g.npos = bnb * g.rows * g.cols;
        g.tiles_n = fsc::ceil_div(d.n, bnb); g.tiles_h = fsc::ceil_div(d.h, bth); g.tiles_w = fsc::ceil_div(d.w, btw);
        p.grid_x = (long)g.tiles_n * g.tiles_h * g.tiles_w;
    }
    g.plane = pad_plane(g.npos, 16);
    const int co_blk = p.cot * 16;
    const int cos = co_blk + ((co_blk % 32 == 16) ? 0 : 16);
    p.lds_bytes = sizeof(float) * ((size_t)taps * kc * cos + (size_t)kc * g.plane);
    *out = p;
    return true;
}

template <int KH, int KW, int COT>
int launch_fwd_cot(const FwdPlan& p, const float* in, const float* packed, const float* bias, float* out,
                   int accumulate, hipStream_t st) {
    auto kern = conv_fwd_kernel<KH, KW, COT, kPT>;
    if (p.lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)p.grid_x, p.co_blocks), dim3(kThreads), p.lds_bytes, st, p.g, in, packed, bias, out, accumulate);
    FSC_LAUNCH_CHECK("fsc_conv_fwd");
    return 0;
}

template <int KH, int KW>
int launch_fwd(const FwdPlan& p, const float* in, const float* packed, const float* bias, float* out, int accumulate,
               hipStream_t st) {
    switch (p.cot) {
        case 1: return launch_fwd_cot<KH, KW, 1>(p, in, packed, bias, out, accumulate, st);
        case 2: return launch_fwd_cot<KH, KW, 2>(p, in, packed, bias, out, accumulate, st);
        case 3: return launch_fwd_cot<KH, KW, 3>(p, in, packed, bias, out, accumulate, st);
        case 4: return launch_fwd_cot<KH, KW, 4>(p, in, packed, bias, out, accumulate, st);
        case 5: return launch_fwd_cot<KH, KW, 5>(p, in, packed, bias, out, accumulate, st);
        case 6: return launch_fwd_cot<KH, KW, 6>(p, in, packed, bias, out, accumulate, st);
        case 7: return launch_fwd_cot<KH, KW, 7>(p, in, packed, bias, out, accumulate, st);
        default: return launch_fwd_cot<KH, KW, 8>(p, in, packed, bias, out, accumulate, st);
    }
}

bool valid_desc(const fsc_conv_desc* d) {
    if (!d || d->n <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->h <= 0 || d->w <= 0) return false;
    const bool k33 = d->kh == 3 && d->kw == 3, k11 = d->kh == 1 && d->kw == 1, k13 = d->kh == 1 && d->kw == 3;
    if (!(k33 || k11 || k13)) return false;
    const long big = 1L << 31;
    return (long)d->n * d->c_in * d->h * d->w < big && (long)d->n * d->c_out * d->h * d->w < big;
}

// ---- wgrad planning
struct WgPlan {
    WgGeom g;
    int mt, co_blocks;
    size_t lds_bytes;
    int threads;
};

bool plan_wgrad(const fsc_conv_desc& d_in, WgPlan* out) {
    fsc_conv_desc d = d_in;
    if (d.kh == 1 && d.kw == 1) {       // no halo: treat each (n, c) plane as one row of h*w pixels
        d.w = d.h * d.w;
        d.h = 1;
    }
    WgPlan p{};
    WgGeom& g = p.g;
    const int taps = d.kh * d.kw;
    const int waves = taps == 1 ? 4 : 6;
    const int npw = taps == 9 ? 3 : 2;
    const int cit = waves * npw / taps;
    const int pixc = 64, maxpos = taps == 1 ? 1 : (taps == 3 ? 2 : 3);
    g.n = d.n; g.cin = d.c_in; g.cout = d.c_out; g.h = d.h; g.w = d.w; g.hw = (long)d.h * d.w;
    const int tiles = fsc::ceil_div(d.c_out, 16);
    int best_mt = 1, best_blocks = tiles;
    long best_tile_cost = (long)tiles * tile_penalty(1);
    for (int mt = 2; mt <= 7; ++mt) {
        const int blocks = fsc::ceil_div(tiles, mt);
        const long cost = (long)blocks * mt * tile_penalty(mt);
        if (cost < best_tile_cost || (cost == best_tile_cost && blocks < best_blocks)) {
            best_mt = mt; best_blocks = blocks; best_tile_cost = cost;
        }
    }
    p.mt = best_mt; p.co_blocks = best_blocks;
    g.co_pad = best_blocks * best_mt * 16;
    g.ci_blocks = fsc::ceil_div(d.c_in, cit * 16);
    g.ci_pad = g.ci_blocks * cit * 16;
    // pixel box of <= 64 pixels whose halo'd footprint fits maxpos*64 positions
    long best_cost = -1;
    int bnb = 1, bth = 1, btw = 1;
    for (int tw = 1; tw <= d.w && tw <= pixc; ++tw) {
        int th = pixc / tw;
        if (th > d.h) th = d.h;
        int nb = 1;
        if (th == d.h && tw == d.w) {
            nb = pixc / (th * tw);
            if (nb > d.n) nb = d.n;
            if (nb < 1) nb = 1;
        }
        while (nb > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > maxpos * 64) --nb;
        while (th > 1 && nb * (th + d.kh - 1) * (tw + d.kw - 1) > maxpos * 64) --th;
        if (nb * (th + d.kh - 1) * (tw + d.kw - 1) > maxpos * 64) continue;
        const long tiles = (long)fsc::ceil_div(d.n, nb) * fsc::ceil_div(d.h, th) * fsc::ceil_div(d.w, tw);
        const long cost = tiles * box_penalty(tw, d.w);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && tw > btw)) {
            best_cost = cost; bnb = nb; bth = th; btw = tw;
        }
    }
    if (best_cost < 0) return false;
    g.nb = bnb; g.th = bth; g.tw = btw;
    g.rows = bth + d.kh - 1; g.cols = btw + d.kw - 1;
    g.npix = bnb * bth * btw; g.npos = bnb * g.rows * g.cols;
    g.tiles_n = fsc::ceil_div(d.n, bnb); g.tiles_h = fsc::ceil_div(d.h, bth); g.tiles_w = fsc::ceil_div(d.w, btw);
    g.units = g.tiles_n * g.tiles_h * g.tiles_w;
    g.plane = pad_plane(g.npos, 2);
    // split-K: aim for ~4 workgroups per CU, at least 4 units per split, partials <= 256 MB
    const long base = (long)p.co_blocks * g.ci_blocks;
    long ns = (256L * 4 + base - 1) / base;
    if (ns > g.units / 4) ns = g.units / 4;
    const long part_bytes_per_split = (long)taps * g.ci_pad * g.co_pad * 4;
    while (ns > 1 && ns * part_bytes_per_split > (256L << 20)) --ns;
    if (ns < 1) ns = 1;
    g.nsplit = (int)ns;
    p.threads = waves * 64;
    p.lds_bytes = sizeof(float) * ((size_t)p.mt * 16 * (pixc + 2) + (size_t)cit * 16 * g.plane + pixc);
    *out = p;
    return true;
}

template <int KH, int KW, int MT>
int launch_wgrad_mt(const WgPlan& p, const float* in, const float* dout, float* part, hipStream_t st) {
    auto kern = conv_wgrad_kernel<KH, KW, MT>;
    if (p.lds_bytes > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL(kern, dim3(p.co_blocks * p.g.ci_blocks, p.g.nsplit), dim3(p.threads), p.lds_bytes, st, p.g, in, dout, part);
    FSC_LAUNCH_CHECK("fsc_conv_wgrad");
    return 0;
}

template <int KH, int KW>
int launch_wgrad(const WgPlan& p, const float* in, const float* dout, float* part, hipStream_t st) {
    switch (p.mt) {
        case 1: return launch_wgrad_mt<KH, KW, 1>(p, in, dout, part, st);
        case 2: return launch_wgrad_mt<KH, KW, 2>(p, in, dout, part, st);
        case 3: return launch_wgrad_mt<KH, KW, 3>(p, in, dout, part, st);
        case 4: return launch_wgrad_mt<KH, KW, 4>(p, in, dout, part, st);
        case 5: return launch_wgrad_mt<KH, KW, 5>(p, in, dout, part, st);
        case 6: return launch_wgrad_mt<KH, KW, 6>(p, in, dout, part, st);
        default: return launch_wgrad_mt<KH, KW, 7>(p, in, dout, part, st);
    }
}

}  // namespace

extern "C" {

size_t fsc_conv_packed_floats(const fsc_conv_desc* d, int dgrad) {
    if (!valid_desc(d)) return 0;
    FwdPlan p;
    if (!plan_fwd(*d, dgrad, &p)) return 0;
    return (size_t)d->kh * d->kw * p.g.k_pad * p.g.m_pad;
}

int fsc_conv_pack_weights(const fsc_conv_desc* d, const float* weight, int dgrad, float* packed, fsc_stream_t stream) {
    FSC_CHECK_ARG(valid_desc(d) && weight && packed, "fsc_conv_pack_weights: bad descriptor or null pointer");
    FwdPlan p;
    FSC_CHECK_ARG(plan_fwd(*d, dgrad, &p), "fsc_conv_pack_weights: no tiling for this shape");
    const long total = (long)d->kh * d->kw * p.g.k_pad * p.g.m_pad;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, fsc::as_stream(stream), weight, packed,
                       d->c_out, d->c_in, d->kh, d->kw, p.g.k_pad, p.g.m_pad, dgrad);
    FSC_LAUNCH_CHECK("fsc_conv_pack_weights");
    return 0;
}

int fsc_conv_fwd(const fsc_conv_desc* d, const float* in, const float* packed, const float* bias, int dgrad,
                 int accumulate, float* out, fsc_stream_t stream) {
    FSC_CHECK_ARG(valid_desc(d) && in && packed && out, "fsc_conv_fwd: bad descriptor or null pointer");
    FSC_CHECK_ARG(!(dgrad && bias), "fsc_conv_fwd: dgrad takes no bias");
    FwdPlan p;
    FSC_CHECK_ARG(plan_fwd(*d, dgrad, &p), "fsc_conv_fwd: no tiling for this shape");
    hipStream_t st = fsc::as_stream(stream);
    if (d->kh == 3) return launch_fwd<3, 3>(p, in, packed, bias, out, accumulate, st);
    if (d->kw == 3) return launch_fwd<1, 3>(p, in, packed, bias, out, accumulate, st);
    return launch_fwd<1, 1>(p, in, packed, bias, out, accumulate, st);
}

int fsc_conv_plan_describe(const fsc_conv_desc* d, int mode, char* buf, size_t buf_len) {
    FSC_CHECK_ARG(valid_desc(d) && buf && buf_len > 0, "fsc_conv_plan_describe: bad arguments");
    if (mode == 2) {
        WgPlan p;
        FSC_CHECK_ARG(plan_wgrad(*d, &p), "fsc_conv_plan_describe: no tiling for this shape");
        snprintf(buf, buf_len, "conv_wgrad_kernel<%d,%d,%d> box=%dx%dx%d units=%d split=%d grid=%dx%d lds=%zu",
                 d->kh, d->kw, p.mt, p.g.nb, p.g.th, p.g.tw, p.g.units, p.g.nsplit, p.co_blocks * p.g.ci_blocks,
                 p.g.nsplit, p.lds_bytes);
    } else {
        FwdPlan p;
        FSC_CHECK_ARG(plan_fwd(*d, mode, &p), "fsc_conv_plan_describe: no tiling for this shape");
        snprintf(buf, buf_len, "conv_fwd_kernel<%d,%d,%d,%d> box=%dx%dx%d flat=%d grid=%ldx%d lds=%zu", d->kh, d->kw,
                 p.cot, kPT, p.g.nb, p.g.th, p.g.tw, p.g.flat, p.grid_x, p.co_blocks, p.lds_bytes);
    }
    return 0;
}

size_t fsc_conv_wgrad_workspace_bytes(const fsc_conv_desc* d) {
    if (!valid_desc(d)) return 0;
    WgPlan p;
    if (!plan_wgrad(*d, &p)) return 0;
    return (size_t)p.g.nsplit * d->kh * d->kw * p.g.ci_pad * p.g.co_pad * sizeof(float);
}

int fsc_conv_wgrad(const fsc_conv_desc* d, const float* in, const float* dout, float* dweight, void* workspace,
                   fsc_stream_t stream) {
    FSC_CHECK_ARG(valid_desc(d) && in && dout && dweight && workspace, "fsc_conv_wgrad: bad descriptor or null pointer");
    WgPlan p;
    FSC_CHECK_ARG(plan_wgrad(*d, &p), "fsc_conv_wgrad: no tiling for this shape");
    hipStream_t st = fsc::as_stream(stream);
    float* part = reinterpret_cast<float*>(workspace);
    int rc;
    if (d->kh == 3) rc = launch_wgrad<3, 3>(p, in, dout, part, st);
    else if (d->kw == 3) rc = launch_wgrad<1, 3>(p, in, dout, part, st);
    else rc = launch_wgrad<1, 1>(p, in, dout, part, st);
    if (rc) return rc;
    const int taps = d->kh * d->kw;
    const long total = (long)d->c_out * d->c_in * taps;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, dweight, d->c_out, d->c_in,
                       taps, p.g.ci_pad, p.g.co_pad, p.g.nsplit);
    FSC_LAUNCH_CHECK("fsc_conv_wgrad(reduce)");
    return 0;
}

}  // extern "C"
