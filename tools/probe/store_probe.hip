// Store pattern of a conv epilogue (development microbenchmark): a tile = 256 pixels x 112 channels of an NCHW fp32 tensor
// (N 128, C 100 -> 112 rows used, HW 13760), 8 waves per workgroup, persistent over tiles like the kernels.
//   A: as conv_l16_fwd_kernel: wave w owns pixels [32 w, 32 w + 32); a store instruction = 16 channel rows x 64 bytes
//   B: wave w owns channel rows {w, w + 8, ...}; a store instruction = one row x 1 KB
//   C: the same bytes as one contiguous stream (what a BatchNorm pass writes)
//   hipcc --offload-arch=gfx950 -O3 -o store_probe tools/probe/store_probe.hip && ./store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kN = 128, kC = 112, kTile = 256;
constexpr long kHW = 13760 + 64;      // (row pitch: a multiple of 256 pixels is not needed, of 32 is)

template <int MODE>
__global__ __launch_bounds__(512) void stores(float* out, int tiles_per_img) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ntiles = kN * tiles_per_img;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int n = t / tiles_per_img, px0 = (t - n * tiles_per_img) * kTile;
        float* base = out + (long)n * kC * kHW + px0;
        if (MODE == 0) {
            for (int i = 0; i < kC / 16; ++i)
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<f32x4*>(base + (long)(i * 16 + (lane >> 2)) * kHW + w * 32 + j * 16 + (lane & 3) * 4) = v;
        } else if (MODE == 1) {
            for (int k = 0; k < kC / 8; ++k)
                *reinterpret_cast<f32x4*>(base + (long)(k * 8 + w) * kHW + lane * 4) = v;
        } else {
            float* lin = out + ((long)t * kC * kTile);
            for (int k = 0; k < kC / 8; ++k)
                *reinterpret_cast<f32x4*>(lin + (long)(k * 8 + w) * kTile + lane * 4) = v;
        }
    }
}

template <int MODE>
void run(const char* name, float* d) {
    const int tpi = 13760 / kTile;        // 53 whole tiles per image
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    stores<MODE><<<256, 512>>>(d, tpi);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        stores<MODE><<<256, 512>>>(d, tpi);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)kN * tpi * kTile * kC * 4;
    printf("%-44s %7.3f ms  %6.2f TB/s (%.0f MB)\n", name, best, bytes / best / 1e9, bytes / 1e6);
}

int main() {
    float* d;
    hipMalloc(&d, sizeof(float) * (size_t)kN * kC * kHW);
    run<0>("A  16 rows x 64 B per instruction", d);
    run<1>("B  one row x 1 KB per instruction", d);
    run<2>("C  contiguous stream", d);
    run<0>("A  again", d);
    hipFree(d);
    return 0;
}
