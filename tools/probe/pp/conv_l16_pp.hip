// 3x3 L16 forward / input-gradient convolution, ping-pong form (gfx950).
//
// conv_l16_fwd_kernel (conv_l16_fwd.h) lets all eight waves of a workgroup interleave LDS reads, LDS-DMA issue, step bookkeeping and
// MFMAs and meet at a barrier every two taps: behind every barrier both waves of a SIMD stand still together (fresh A fragments, DMA
// issue), and each wave's MFMA groups stall on its own reads -- the matrix pipe is ~55 % busy (profiles/r03d_pmc_conv_l16_fwd.txt).
// Here the two waves of a SIMD (wave w and wave w + 4) work in ANTI-PHASE, one barrier per half step:
//
//     segment 2S     waves 0-3: LOAD(S)      | waves 4-7: COMPUTE(S - 1)
//     segment 2S + 1 waves 0-3: COMPUTE(S)   | waves 4-7: LOAD(S)
//
// COMPUTE(S) is nothing but the step's COT * 2 * 3 MFMAs on operands that are already in registers; LOAD(S) is everything else:
// the step's A (weight) and B (activation) fragments LDS -> registers, the LDS-DMA copies (waves 0-3: the weight slot three steps
// ahead; waves 4-7: the next chunk's input box, a few pieces per step), the step bookkeeping, and a slice of the PREVIOUS item's
// epilogue.  So the matrix pipe of a SIMD always has exactly one wave feeding it from registers while its partner does the memory
// work (MI355X_MICROARCH.md, "Two waves per SIMD": matrix beside memory is the pairing that nets), and a wave that arrives at a
// barrier finds its partner already waiting.
//
// Epilogue without a hole: two accumulator sets.  Item k accumulates into set k & 1; while item k + 1 runs, each LOAD drains one
// (channel tile, pixel tile) of item k's set -- scale, bias, LDS transpose, 16-byte store, statistics / max-pool exactly as the
// eight-wave kernel's epilogue -- and clears it.  The stores trickle out beside the MFMAs instead of 114 KB per workgroup at once.
//
// Products and their order per accumulator are those of conv_l16_fwd_kernel (l*h, h*l, h*h per tap and 32-channel chunk): results
// are bit-identical to it and to the fp32-input f16x3 kernels (tests/test_l16_gpu.py, tests/test_pp_gpu.py).
// Same plans (LPlan), same packed weights, same records as the eight-wave kernel; FSC_L16_V1=1 selects that one (A/B, tests).
//
// Replaces nn.Conv2d 3x3 forward and input gradient (reference networks/classifiers.py:526-531, 77-81) and the MaxPool2d(2) behind
// the entry convolution of a block (:532).
#include "conv_l16_fwd.h"

#include <type_traits>
#include <utility>

namespace {

__device__ unsigned long long g_pp_clock[2];

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): every index a compile-time constant (register arrays)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// Development (-DFSC_L16_PROFILE): shader-clock stamps per wave of workgroup 0, summed over the launch: [0] LOAD work, [1] LOAD wait +
// barrier, [2] COMPUTE (MFMAs + side work), [3] COMPUTE wait + barrier, [4] steps, [5] whole kernel; fsc_debug_pp_prof reads and clears.
#ifdef FSC_L16_PROFILE
__device__ unsigned long long g_pp_prof[8][8];
#define PP_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pp_acc[i] += n_ - pp_t; pp_t = n_; } while (0)
#else
#define PP_STAMP(i)
#endif

template <int COT, bool POOL, bool STATS>
__global__ __launch_bounds__(kWaves * 64) void conv_l16_pp_kernel(LGeom g, const uint4* __restrict__ in,
                                                                   const float* __restrict__ packed,
                                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                                   int accumulate, const float* __restrict__ in_amax,
                                                                   const float* __restrict__ w_amax,
                                                                   uint8_t* __restrict__ pool_idx,
                                                                   const float* __restrict__ stat_pivot,
                                                                   float4* __restrict__ stat_rec) {
    constexpr int KW = 3, TAPS = 9, PT = 2;
    constexpr int CO_BLK = COT * 16;
    constexpr int NSTG = 2;
    constexpr int WUNITS = COT * 2;                 // 1 KB fragment images per step
    constexpr int WSLOT_F = WUNITS * 256;           // floats per ring slot
    constexpr int RING = 4;
    constexpr int NWP = (WUNITS + 3) / 4;           // weight pieces EVERY copying wave issues per step (a constant count keeps the
                                                    // counted waits simple: a unit past the end repeats the wave's previous one)
    constexpr int NT = COT * PT;                    // accumulator tiles per wave

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const wring = smem;
    uint4* const ibase = reinterpret_cast<uint4*>(smem + RING * WSLOT_F);
    const int istage = 8 * g.plane;                 // uint4 per stage
    float* const scratch0 = reinterpret_cast<float*>(ibase + NSTG * istage);
    float* const scratch = scratch0 + (threadIdx.x >> 6) * (16 * kScr);
    // The bias and the statistics pivot of this worker's channel block live in the PAD columns of the epilogue scratch (rows of 16
    // floats at a pitch of kScr = 20: 4 spare floats x 16 rows x 8 waves = 512 floats): entry e at wave e / 64, row (e % 64) / 4,
    // column 16 + e % 4 -- a quad of four consecutive entries is one aligned 16-byte read.  Bias: entries [0, CO_BLK), pivot: [128, 128 + CO_BLK).
    auto pad_ptr = [&](int e) -> float* { return scratch0 + (e >> 6) * (16 * kScr) + ((e & 63) >> 2) * kScr + 16 + (e & 3); };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wid >> 2, wq = wid & 3;        // half 0: waves 0-3 (copy the weights), half 1: waves 4-7 (copy the input)
    const int lm = lane & 15, kq = lane >> 4;
    const unsigned long long ck0 = __builtin_readcyclecounter(), cr0 = __builtin_amdgcn_s_memrealtime();

    const float ax = block_amax512(in_amax, smem);
    const float aw = *w_amax;
    const int fx = scale_field(ax), fw = scale_field(aw);
    const float inv_x = inv_scale(fx, ax), inv_w = inv_scale(fw, aw);

    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };   // (a scalar register)
    const float inv_per = uni(1.0f / (float)(g.rows * g.cols)), inv_cols = uni(1.0f / (float)g.cols);
    const float inv_tw = uni(1.0f / (float)g.tw), inv_thw = uni(1.0f / (float)(g.th * g.tw));
    int pix_b[PT];               // byte offset of this lane's pixel inside a staged plane
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int p = (wid * PT + pt) * 16 + lm;
        int pl = 0;
        if (p < g.npix) {
            if (POOL) {                                  // tile = a 2 x 8 block: (image, row pair, column octet)
                const int tpr = g.tw >> 3, tpi = (g.th >> 1) * tpr;
                const int t = wid * PT + pt;
                const int b = t / tpi, rem = t - b * tpi;
                const int tr = rem / tpr, tc = rem - tr * tpr;
                pl = (b * g.rows + 2 * tr + (lm >> 3)) * g.cols + 8 * tc + (lm & 7);
            } else {
                const int per = g.th * g.tw;
                const int b = fdiv(p, inv_thw), rem = p - b * per;
                const int r = fdiv(rem, inv_tw), c = rem - r * g.tw;
                pl = (b * g.rows + r) * g.cols + c;
            }
        }
        pix_b[pt] = pl * 16;
    }
    const int ntiles = g.tiles_n * g.tiles_h * g.tiles_w;
    // item order: conv_l16_fwd_kernel's (a worker keeps one channel block; with g.xcd the blocks of a tile share an XCD)
    int cb_w, t0;
    const int ts = (int)gridDim.x / g.coblk;
    if (g.xcd) {
        const int l = (int)blockIdx.x >> 3;
        cb_w = l % g.coblk;
        t0 = ((int)blockIdx.x & 7) + 8 * (l / g.coblk);
    } else {
        cb_w = (int)blockIdx.x % g.coblk;
        t0 = (int)blockIdx.x / g.coblk;
    }
    const int co0 = cb_w * CO_BLK;
    const int nchunks = g.nfull + (g.tail_oct ? 1 : 0);
    for (int i = tid; i < CO_BLK; i += kWaves * 64) {
        const int co = co0 + i;
        *pad_ptr(i) = (bias != nullptr && co < g.cout) ? bias[co] : 0.f;
        *pad_ptr(128 + i) = (STATS && stat_pivot != nullptr && co < g.cout) ? stat_pivot[co] : 0.f;
    }

    // The lane number behind an opaque move: whatever is derived from it is recomputed where it is used instead of being hoisted
    // out of the item loop and kept in registers (the hoisted store offsets and copy addresses were what spilled).
    auto fresh_lane = [&]() -> int {
        int l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };

    // ---- accumulators: two sets (item k uses set k & 1, the other one is being drained)
    f32x4 acc[2][COT][PT];       // (an item's first step writes its set: compute_step(first))
    constexpr int NST = STATS ? COT : 1;
    float st_s1[NST], st_s2[NST], st_mn[NST], st_mx[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) { st_s1[i] = 0.f; st_s2[i] = 0.f; st_mn[i] = INFINITY; st_mx[i] = -INFINITY; }
    auto stat_add = [&](int i, float y, float pv) {
        const float a = y - pv;
        st_s1[i] += a;
        st_s2[i] = fmaf(a, a, st_s2[i]);
        st_mn[i] = fminf(st_mn[i], y);
        st_mx[i] = fmaxf(st_mx[i], y);
    };

    // ---- weight copies (waves 0-3): W(S + 3) during LOAD(S), across items
    const uint4* const zero = reinterpret_cast<const uint4*>(g_zero16_l);
    int wp_item = t0, wp_left = g.steps, wp_slot = 0, wp_issued = 0;
    const float* const wp_base = packed + (long)cb_w * g.steps * WSLOT_F;      // (uniform; the lane's 16 bytes are added at the copy)
    const float* wp_src = wp_base;
    auto w_piece = [&](int q) {                              // piece q of NWP of the slot being filled (q static)
        if (wp_item >= ntiles) return;
        float* dst = wring + wp_slot * WSLOT_F;
        int u = q * 4 + wq;
        if ((q + 1) * 4 > WUNITS && u >= WUNITS) u -= 4;
        glds16(wp_src + u * 256 + fresh_lane() * 4, dst + u * 256);
    };
    auto w_advance = [&]() {
        if (wp_item >= ntiles) return;
        wp_src += WSLOT_F;
        wp_slot = wp_slot == RING - 1 ? 0 : wp_slot + 1;
        ++wp_issued;
        if (--wp_left == 0) {
            wp_left = g.steps;
            wp_item += ts;
            wp_src = wp_base;
        }
    };
    auto produce_w = [&]() {
#pragma unroll
        for (int q = 0; q < NWP; ++q) w_piece(q);
        w_advance();
    };

    // ---- input copies (waves 4-7): the box of the chunk AFTER the one being consumed, piece by piece.  Wave 4 + q copies units
    //      q and q + 4 of the chunk's eight (octet, limb) planes: piece p = (p / npt: which of the two, p % npt: 64 positions).
    int ip_item = t0, ip_c = 0, ip_stg = 0, ip_p = 0;
    int ip_n0 = 0, ip_h0 = 0, ip_w0 = 0;
    auto ip_set_tile = [&](int tile) {
        int t = tile;
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h; t /= g.tiles_h;
        ip_n0 = t * g.nb; ip_h0 = thi * g.th; ip_w0 = twi * g.tw;
    };
    const int npieces = 2 * g.npt;
    auto issue_piece = [&](int p) {                          // (p uniform)
        const int uu = p >= g.npt ? 1 : 0, q = p - uu * g.npt;
        const int unit = wq + 4 * uu;
        const int oct = ip_c * 4 + (unit >> 1);
        const int pos = q * 64 + fresh_lane();
        if (pos < g.plane) {                                 // (lanes past the plane would land in the next unit)
            int off = -1;
            if (pos < g.npos && oct < g.oct_in) {
                const int per = g.rows * g.cols;
                const int b = fdiv(pos, inv_per), rem = pos - b * per;
                const int rr = fdiv(rem, inv_cols), cc = rem - rr * g.cols;
                const int gh = ip_h0 + rr - 1, gw = ip_w0 + cc - 1;
                if (ip_n0 + b < g.n && gh >= 0 && gh < g.h && gw >= 0 && gw < g.w)
                    off = (int)((long)(ip_n0 + b) * g.img_stride + (long)gh * g.w + gw);
            }
            const uint4* src = in + (long)(oct * 2 + (unit & 1)) * g.hw;
            glds16(off >= 0 ? src + off : zero, ibase + ip_stg * istage + unit * g.plane + q * 64);
        }
    };
    auto produce_i = [&](int quota) {
        for (int k = 0; k < quota && ip_item < ntiles && ip_p < npieces; ++k) issue_piece(ip_p++);
    };
    auto ip_next_chunk = [&]() {
        ip_p = 0;
        ip_stg ^= 1;
        if (++ip_c == nchunks) {
            ip_c = 0;
            ip_item += ts;
            if (ip_item < ntiles) ip_set_tile(ip_item);
        }
    };

    // ---- B operand address (conv_l16_fwd_kernel's)
    const int limb_b = g.plane * 16;                        // bytes between the limb planes of an octet
    const int stage_b = 8 * limb_b;
    const int kq_b = kq * 2 * limb_b;
    auto b_off = [&](int stage, int c, int s) -> int {      // (stage, c, s uniform)
        if (c >= g.nfull) {                                  // remainder chunk: (tap, octet) flattened over the lane groups
            const int noct = g.tail_oct;
            int gi = 4 * s + kq;
            if (gi >= TAPS * noct) gi = 0;                   // its weights are zero
            const int tap = fdiv(gi, 1.0f / (float)noct);
            const int oct = gi - tap * noct;
            const int ty = fdiv(tap, 1.0f / (float)KW), tx = tap - ty * KW;
            return stage * stage_b + oct * 2 * limb_b + (ty * g.cols + tx) * 16;
        }
        const int ty = (s * 11) >> 5, tx = s - ty * KW;       // s / 3 for s < 10
        return stage * stage_b + (ty * g.cols + tx) * 16 + kq_b;
    };

    // ---- prologue: weights of steps 0 .. 2, the first chunk's input box
    if (t0 < ntiles) {
        if (half == 0) {
            produce_w();
            produce_w();
            produce_w();
        } else {
            ip_set_tile(t0);
            produce_i(npieces);
            ip_next_chunk();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                         // (also publishes the bias / pivot tables)
    if (half == 1) raw_barrier();                            // waves 4-7 sit out segment 0

    // consumer position (uniform): chunk c, step sc of nst inside it, input stage stg, weight slot
    int c = 0, sc = 0, nst = g.nfull > 0 ? TAPS : g.tail_steps, stg = 0, slot = 0, s_global = 0;
    // draining: the previous item's tile, its next accumulator tile, the store positions of this lane's quads / windows
    int pv_tile = -1, dr_k = NT;
    int quad_g[PT];              // (element offsets: valid_l16_desc keeps tensors below 2^31 elements)
    int quad_ok[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) { quad_g[pt] = 0; quad_ok[pt] = 0; }
    // B-waves skip the slice in the last LOAD of a chunk (it ends with vmcnt(0)): g.steps - nchunks LOADs per item drain
    const int drain_quota = (NT + (g.steps - nchunks) - 1) / (g.steps - nchunks);

    u32x4 fa[COT][2], fb[2][PT];
    const char* const ib = reinterpret_cast<const char*>(ibase);
    long hw_t = g.hw;
    asm volatile("" : "+s"(hw_t));

    auto plan_drain = [&](int tile) {                        // store offsets of tile `tile` for this lane (once per item)
        const int lane = fresh_lane();
        int t = tile;
        const int twi = t % g.tiles_w; t /= g.tiles_w;
        const int thi = t % g.tiles_h; t /= g.tiles_h;
        const int n0 = t * g.nb, h0 = thi * g.th, w0 = twi * g.tw;
        if constexpr (POOL) {
            const int oh = g.h >> 1, ow = g.w >> 1;
            const int tpr = g.tw >> 3, tpi = (g.th >> 1) * tpr;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int tt = wid * PT + pt;
                quad_g[pt] = -1;
                if (tt * 16 < g.npix) {
                    const int b = tt / tpi, rem = tt - b * tpi;
                    const int tr = rem / tpr, tc = rem - tr * tpr;
                    const int pr = (h0 >> 1) + tr, pc = (w0 >> 1) + 4 * tc + (lane & 3);
                    if (n0 + b < g.n && pr < oh && pc < ow) quad_g[pt] = (((n0 + b) * g.cout * oh + pr) * ow + pc);
                }
            }
        } else {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int p = (wid * PT + pt) * 16 + (lane & 3) * 4;
                quad_g[pt] = 0;
                quad_ok[pt] = 0;
                if (p < g.npix) {
                    const int per = g.th * g.tw;
                    const int b = fdiv(p, inv_thw), rem = p - b * per;
                    const int r = fdiv(rem, inv_tw), cq = rem - r * g.tw;
                    if (n0 + b < g.n && h0 + r < g.h) {
                        quad_g[pt] = (n0 + b) * g.cout * (int)g.hw + (h0 + r) * g.w + (w0 + cq);
                        const int left = g.w - (w0 + cq), inbox = g.npix - p;
                        const int nv = left < inbox ? left : inbox;
                        quad_ok[pt] = nv >= 4 ? 15 : nv <= 0 ? 0 : (1 << nv) - 1;
                    }
                }
            }
        }
    };

    // one accumulator tile (set Q, channel tile I, pixel tile J) of the previous item -> memory.  The set is only READ here (a
    // write inside the jump below would make every tile of the set a phi of sixteen paths: measured +130 registers); the first
    // step of an item starts its accumulators from a zero operand instead of clearing them.
    auto drain_tile = [&](auto Q_, auto I_, auto J_) __attribute__((always_inline)) {
        constexpr int Q = decltype(Q_)::value, i = decltype(I_)::value, j = decltype(J_)::value;
        const int lane = fresh_lane();
        const int lm = lane & 15, kq = lane >> 4, ch = lane >> 2;
        const f32x4 bq = *reinterpret_cast<const f32x4*>(pad_ptr(i * 16 + kq * 4));
        const int co = co0 + i * 16 + ch;
        float pv = 0.f;
        if (STATS) pv = *pad_ptr(128 + i * 16 + ch);
        if constexpr (POOL) {
            // after the DPP moves the four even lanes lm = 0, 2, 4, 6 of a column group hold the pooled value and window index of
            // 4 channels x one window; through the scratch tile lane L = (channel L >> 2, window L & 3) stores one pooled pixel
            const long ohw = (long)(g.h >> 1) * (g.w >> 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v0 = fmaf(acc[Q][i][j][r] * inv_x, inv_w, bq[r]);
                const float v1 = dpp_xor1(v0);
                const float v2 = dpp_xor8(v0);
                const float v3 = dpp_xor8(v1);
                float best = v0;                     // first maximum in window order, NaN wins (fsc_maxpool_fwd)
                int bi = 0;
                if (v1 > best || v1 != v1) { best = v1; bi = 1; }
                if ((v2 > best || v2 != v2) && best == best) { best = v2; bi = 2; }
                if ((v3 > best || v3 != v3) && best == best) { best = v3; bi = 3; }
                if ((lm & 9) == 0) {                 // lanes lm = 0, 2, 4, 6: the window's first pixel
                    scratch[(kq * 4 + r) * kScr + (lm >> 1)] = best;
                    scratch[(kq * 4 + r) * kScr + 8 + (lm >> 1)] = __int_as_float(bi);
                }
            }
            const float val = scratch[ch * kScr + (lane & 3)];
            const int bidx = __float_as_int(scratch[ch * kScr + 8 + (lane & 3)]);
            if (co < g.cout && quad_g[j] >= 0) {
                out[(long)quad_g[j] + (long)co * ohw] = val;
                pool_idx[(long)quad_g[j] + (long)co * ohw] = (uint8_t)bidx;
                if (STATS) stat_add(i, val, pv);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) scratch[(kq * 4 + r) * kScr + lm] = fmaf(acc[Q][i][j][r] * inv_x, inv_w, bq[r]);
            const f32x4 v = *reinterpret_cast<const f32x4*>(scratch + ch * kScr + (lane & 3) * 4);
            if (STATS && co < g.cout && quad_ok[j]) {
                if (quad_ok[j] == 15) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) stat_add(i, v[k], pv);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (quad_ok[j] & (1 << k)) stat_add(i, v[k], pv);
                }
            }
            if (co < g.cout && quad_ok[j]) {
                float* o = out + (long)quad_g[j] + (long)co * hw_t;
                if (!accumulate && quad_ok[j] == 15) {
                    *reinterpret_cast<f32x4*>(o) = v;
                } else if (quad_ok[j] == 15) {
                    const f32x4 old = *reinterpret_cast<const f32x4*>(o);
                    *reinterpret_cast<f32x4*>(o) = old + v;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (quad_ok[j] & (1 << k)) o[k] = accumulate ? o[k] + v[k] : v[k];
                }
            }
        }
    };
    // tile number k (uniform, run time) of set Q: a jump over the NT statically indexed bodies
    auto drain_k = [&](auto Q_, int k) __attribute__((always_inline)) {
        auto body = [&](auto K_) __attribute__((always_inline)) {
            constexpr int K = decltype(K_)::value;
            drain_tile(Q_, std::integral_constant<int, K / PT>{}, std::integral_constant<int, K % PT>{});
        };
        switch (k) {
#define FSC_PP_CASE(K_)                                                  \
    case K_:                                                             \
        if constexpr (K_ < NT) body(std::integral_constant<int, K_>{});  \
        break;
            FSC_PP_CASE(0) FSC_PP_CASE(1) FSC_PP_CASE(2) FSC_PP_CASE(3) FSC_PP_CASE(4) FSC_PP_CASE(5) FSC_PP_CASE(6) FSC_PP_CASE(7)
            FSC_PP_CASE(8) FSC_PP_CASE(9) FSC_PP_CASE(10) FSC_PP_CASE(11) FSC_PP_CASE(12) FSC_PP_CASE(13) FSC_PP_CASE(14) FSC_PP_CASE(15)
#undef FSC_PP_CASE
            default: break;
        }
    };

    // ---- LOAD(S): the step's fragments LDS -> registers and a slice of the previous item's epilogue -- what needs LDS latency or
    //      memory round trips.  P = the accumulator set of the CURRENT item (the other one drains).  `noff` / `slot` were prepared by
    //      the previous COMPUTE.  Keep this segment SHORT: while it runs, the partner wave's MFMAs are all the SIMD has; copy issue and
    //      the step bookkeeping ride in the issue slots BETWEEN the MFMAs of COMPUTE instead (a first version with everything here
    //      measured 1900 cycles per LOAD against 700 per COMPUTE).
    int noff = b_off(0, 0, 0);
#ifdef FSC_L16_PROFILE
    unsigned long long pp_acc[6] = {0, 0, 0, 0, 0, 0}, pp_t = __builtin_readcyclecounter();
    const unsigned long long pp_k0 = pp_t;
#endif
    auto load_step = [&](auto P_) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value;
        const u32x4* wl = reinterpret_cast<const u32x4*>(wring + slot * WSLOT_F) + lane;
#pragma unroll
        for (int i = 0; i < COT; ++i)
#pragma unroll
            for (int l = 0; l < 2; ++l) fa[i][l] = wl[(i * 2 + l) * 64];
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            fb[0][j] = *reinterpret_cast<const u32x4*>(ib + noff + pix_b[j]);
            fb[1][j] = *reinterpret_cast<const u32x4*>(ib + noff + pix_b[j] + limb_b);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool chunk_last = sc + 1 == nst;
#ifndef FSC_PP_NODRAIN
        if (!(half == 1 && chunk_last)) {                    // (that LOAD of waves 4-7 ends with vmcnt(0): no fresh store in front of it)
            for (int q = 0; q < drain_quota && dr_k < NT; ++q) {
                drain_k(std::integral_constant<int, 1 - P>{}, dr_k);
                ++dr_k;
            }
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (half == 1 && chunk_last) {
            // the next chunk's box must be in LDS before waves 0-3 read it in the next segment
            produce_i(npieces);                              // (pieces a short chunk's COMPUTE slots did not get to)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ip_next_chunk();
        }
        PP_STAMP(0);
        raw_barrier();
        PP_STAMP(1);
    };

    // ---- COMPUTE(S): the step's MFMAs, l*h, h*l, h*h per accumulator -- smallest first, as in conv_l16_fwd_kernel -- and, in the
    //      issue slots between them, this wave's copies and the bookkeeping of the next step
    auto compute_step = [&](auto P_, bool first) __attribute__((always_inline)) {
        constexpr int P = decltype(P_)::value;
        constexpr int kLa[3] = {1, 0, 0}, kLb[3] = {0, 1, 0};
        constexpr int NM = 3 * NT;                           // MFMAs of the step
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const bool chunk_last = sc + 1 == nst;
        const int quota = half == 1 && !chunk_last ? (npieces + (nst - 2)) / (nst - 1) : 0;
        // side work rides between the MFMAs, one piece at a time (a burst of four copies between two MFMA groups cost the wave
        // ~1000 cycles): slot k of kSlots sits behind MFMA (k + 1) * NM / (kSlots + 1)
        constexpr int kSlots = 6;
        auto side = [&](int k) __attribute__((always_inline)) {
#ifndef FSC_PP_NOCOPY
            if (half == 0) {
                if (k < NWP) w_piece(k);
                if (k == NWP - 1) w_advance();
            } else if (k < quota && ip_item < ntiles && ip_p < npieces) {
                issue_piece(ip_p++);
            }
#endif
            if (k == kSlots - 1) {                           // the next step: position, weight slot, B operand offset
                if (chunk_last) {
                    sc = 0;
                    stg ^= 1;
                    c = c + 1 < nchunks ? c + 1 : 0;
                    nst = c < g.nfull ? TAPS : g.tail_steps;
                } else {
                    ++sc;
                }
                slot = slot == RING - 1 ? 0 : slot + 1;
                noff = b_off(stg, c, sc);
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        if (first) {                                         // (uniform) the item's first products start from zero
#pragma unroll
            for (int i = 0; i < COT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) acc[P][i][j] = mfma16(fa[i][kLa[0]], fb[kLb[0]][j], zero4);
        } else {
#pragma unroll
            for (int i = 0; i < COT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) acc[P][i][j] = mfma16(fa[i][kLa[0]], fb[kLb[0]][j], acc[P][i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        side(0);
        __builtin_amdgcn_sched_barrier(0);
        static_for<NM - NT>([&](auto M_) __attribute__((always_inline)) {
            constexpr int m = NT + decltype(M_)::value;
            constexpr int gq = m / NT, i = (m % NT) / PT, j = m % PT;
            acc[P][i][j] = mfma16(fa[i][kLa[gq]], fb[kLb[gq]][j], acc[P][i][j]);
            static_for<kSlots - 1>([&](auto K_) __attribute__((always_inline)) {
                constexpr int k = 1 + decltype(K_)::value;
                if constexpr (m == NT + (k * (NM - NT)) / kSlots - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    side(k);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
        __builtin_amdgcn_sched_barrier(0);
        if (half == 0) {
            // W(S + 1) -- read by LOAD(S + 1) in the next segment -- has landed: it was issued in COMPUTE(S - 2); the copies of the
            // two COMPUTEs since may stay in flight (stores in the queue only make the wait stricter)
            const int ahead = wp_issued - (s_global + 2);
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NWP) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ++s_global;
        PP_STAMP(2);
        raw_barrier();
        PP_STAMP(3);
#ifdef FSC_L16_PROFILE
        pp_acc[4] += 1;
#endif
    };

    auto run_item = [&](auto P_) __attribute__((always_inline)) {
#pragma unroll 1
        for (int s = 0; s < g.steps; ++s) {
            load_step(P_);
            compute_step(P_, s == 0);
        }
    };

    // The item loop is unrolled by two -- set 0, set 1, set 0, ... in straight line -- so that no control-flow merge has to carry the
    // two accumulator sets through phi copies (an `if (parity) ... else ...` around the two bodies cost a third set of registers).
    auto next_item = [&](int item) {
        pv_tile = item;
        dr_k = 0;
        plan_drain(item);
    };
    auto finish = [&](auto Q_) __attribute__((always_inline)) {     // Q: the set of the last item
        if (half == 0) raw_barrier();                        // (waves 4-7 ran one segment behind)
        if (pv_tile >= 0)
            for (; dr_k < NT; ++dr_k) drain_k(Q_, dr_k);
    };
    for (int item = t0;;) {
        if (item >= ntiles) { finish(std::integral_constant<int, 1>{}); break; }
        run_item(std::integral_constant<int, 0>{});
        next_item(item);
        item += ts;
        if (item >= ntiles) { finish(std::integral_constant<int, 0>{}); break; }
        run_item(std::integral_constant<int, 1>{});
        next_item(item);
        item += ts;
    }
#ifdef FSC_L16_PROFILE
    if (blockIdx.x == 0 && lane == 0) {
        pp_acc[5] = __builtin_readcyclecounter() - pp_k0;
        for (int i = 0; i < 6; ++i) atomicAdd(&g_pp_prof[wid][i], pp_acc[i]);
    }
#endif
    if (blockIdx.x == 0 && tid == 0) {
        g_pp_clock[0] = __builtin_readcyclecounter() - ck0;
        g_pp_clock[1] = __builtin_amdgcn_s_memrealtime() - cr0;
    }
    if constexpr (STATS) {
        // the four lanes of a quad hold the same channel: fold them, lane (lane & 3) == 0 writes the record
#pragma unroll
        for (int i = 0; i < COT; ++i) {
            float a = st_s1[i], b = st_s2[i], mn = st_mn[i], mx = st_mx[i];
            a += dpp_xor1(a); a += dpp_xor2(a);
            b += dpp_xor1(b); b += dpp_xor2(b);
            mn = fminf(mn, dpp_xor1(mn)); mn = fminf(mn, dpp_xor2(mn));
            mx = fmaxf(mx, dpp_xor1(mx)); mx = fmaxf(mx, dpp_xor2(mx));
            if ((lane & 3) == 0)
                stat_rec[((long)blockIdx.x * kWaves + wid) * CO_BLK + i * 16 + (lane >> 2)] = make_float4(a, b, mn, mx);
        }
    }
}

template <int COT, bool POOL, bool STATS>
int launch_one(const LPlan& p, const fsc::L16LaunchArgs& a, hipStream_t st) {
    auto kern = conv_l16_pp_kernel<COT, POOL, STATS>;
    const size_t lds = p.lds_bytes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)p.workers), dim3(kWaves * 64), lds, st, p.g, a.in, a.packed, a.bias, a.out,
                       a.accumulate, a.in_amax, a.w_amax, a.pool_idx, a.stat_pivot, a.stat_rec);
    FSC_LAUNCH_CHECK("fsc_conv_l16_fwd(pp)");
    return 0;
}

template <int COT>
int launch_kind(const LPlan& p, const fsc::L16LaunchArgs& a, hipStream_t st) {
    const bool stats = a.stat_rec != nullptr, pool = a.pool_idx != nullptr;
    if constexpr (COT >= 4) {
        if (pool) return stats ? launch_one<COT, true, true>(p, a, st) : launch_one<COT, true, false>(p, a, st);
    }
    if (pool) {
        fsc::set_error("fsc_conv_l16_pool_fwd(pp): internal: no instantiation for %d channel tiles", COT);
        return 22;
    }
    return stats ? launch_one<COT, false, true>(p, a, st) : launch_one<COT, false, false>(p, a, st);
}

}  // namespace

namespace fsc {

bool l16_pp_has(int cot) {
#ifndef FSC_L16_DEV
    return cot >= 3 && cot <= 8;
#else
    return cot == 5 || cot == 7 || cot == 8;
#endif
}

int l16_launch_pp(const void* plan, const L16LaunchArgs& a, hipStream_t st) {
    const LPlan& p = *reinterpret_cast<const LPlan*>(plan);      // (conv_l16.hip's LPlan: the same header, the same layout)
#ifdef FSC_PP_SINGLE                                             // (development: one instantiation, for resource bisection)
    return launch_one<FSC_PP_SINGLE, false, false>(p, a, st);
#else
    switch (p.cot) {
#ifndef FSC_L16_DEV
        case 3: return launch_kind<3>(p, a, st);
        case 4: return launch_kind<4>(p, a, st);
        case 6: return launch_kind<6>(p, a, st);
#endif
        case 5: return launch_kind<5>(p, a, st);
        case 7: return launch_kind<7>(p, a, st);
        case 8: return launch_kind<8>(p, a, st);
        default: break;
    }
    set_error("fsc_conv_l16_fwd(pp): internal: no instantiation for %d channel tiles", p.cot);
    return 22;
#endif
}

int l16_pp_clock(unsigned long long* v2) {
    hipError_t e = hipMemcpyFromSymbol(v2, HIP_SYMBOL(g_pp_clock), 2 * sizeof(unsigned long long));
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace fsc

#ifdef FSC_L16_PROFILE
/* development: copies the 8 x 8 phase counters of the ping-pong kernel to `out64` (host) and clears them */
extern "C" int fsc_debug_pp_prof(unsigned long long* out64) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_pp_prof), sizeof(unsigned long long) * 64);
    unsigned long long z[64] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_pp_prof), z, sizeof(z));
    return 0;
}
#endif
