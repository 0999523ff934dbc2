// Calibration: back-to-back v_mfma_f32_32x32x2_f32 from registers, W waves per SIMD.  Developer tool.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu) {
    int grid = 256 * blocks_per_cu, iters = 20000 / NACC;
    float* out; hipMalloc(&out, grid * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(out, 100, 0.5f, 0.25f);
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() { run<1>(1); run<4>(1); run<4>(2); run<4>(3); run<2>(2); return 0; }
