// Small 1-d convolutions of the bf16 arithmetic (arith 1; BASELINE.json configs[2], reference networks/classifiers.py:147-163,
// 37-69: the late blocks of the hierarchical model -- 156 ... 476 channels on rows of 107 ... 3 frames).  See conv_s1d.hip.
#pragma once
#include "common.h"

namespace fsc {
namespace s1d {

struct Plan {
    int n, cin, cout, h, w, kh, kw, taps;   // cin / cout as the KERNEL sees them (dgrad: swapped); 1-d rows: h = 1
    int nprod, cot;                    // 1 (bf16 operands) / 9 (three exact bf16 limbs, nine products); channel tiles per workgroup
    int nfull, tail_oct, steps;        // K steps of 32 channels x one tap (conv.hip pack_x3_items: the A-fragment order)
    int co_blocks;                     // blocks of kCot output-channel tiles
    long npix;                         // n * len
    int px_groups;                     // workgroups along the pixels (kPt pixel tiles each)
};
constexpr int kCot = 4;                // output-channel tiles (16 channels) per workgroup
constexpr int kPt = 2;                 // pixel tiles (16 positions) per workgroup
constexpr int kWaves = 8;              // waves of a workgroup: each takes every 8th K step of the whole tile

// forward (dgrad = 0) / input gradient (dgrad = 1) of a stride-1 same-pad Conv1d in arith 1 on few pixels
bool plan_fwd(const fsc_conv_desc& d, int dgrad, Plan* out);
int launch_fwd(const Plan& p, const float* in, const unsigned short* packed, const float* bias, float* out, int accumulate,
               hipStream_t st);

struct WPlan {
    int n, cin, cout, len, taps;
    long npix;
    int seg, spr;                      // seg = 1: K steps of four 8-position segments of single rows (spr segments per row)
    int ksteps;                        // K steps (32 positions, or four segments)
    int nsplit;                        // split-K slices written to the workspace (grid.z)
    int ci_blocks, co_blocks;          // 32 x 32 (ci, co) blocks
    int ci_pad, co_pad;
};
bool plan_wgrad(const fsc_conv_desc& d, WPlan* out);
int launch_wgrad(const WPlan& p, const float* in, const float* dout, float* part, hipStream_t st);

}  // namespace s1d
}  // namespace fsc
