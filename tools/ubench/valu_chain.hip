// Micro-benchmark: v_fmac_f32 throughput per SIMD as a function of the number of independent accumulator chains per wave and
// of the waves per SIMD (what a kernel like the conv3x3 scoring tail - 3 chains, 2 waves per SIMD - can expect).
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_chain valu_chain.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NCH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    extern __shared__ float dummy[];
    float a[NCH];
    for (int i = 0; i < NCH; ++i) a[i] = threadIdx.x * 1e-3f + i;
    float w[4] = {s, s * 0.5f, s * 0.25f, s * 0.125f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 48 / NCH; ++r)
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(w[r & 3]), "v"(w[(r + 1) & 3]));
    }
    float r = 0;
    for (int i = 0; i < NCH; ++i) r += a[i];
    if (r == 123.456f) dummy[threadIdx.x] = r;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NCH>
static void run(float* d, int waves_per_simd) {
    // 160 KB LDS per CU: a block that asks for 160/(waves_per_simd) KB leaves room for exactly that many blocks of 4 waves
    const int lds = waves_per_simd >= 8 ? 0 : (160 * 1024 / waves_per_simd) - 2048;
    hipFuncSetAttribute((const void*)k<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = 256 * 16, iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NCH><<<blocks, 256, lds>>>(d, 16, 1e-6f);
    (void)hipEventRecord(e0);
    k<NCH><<<blocks, 256, lds>>>(d, iters, 1e-6f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters * 48;
    printf("chains %2d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s  %.2f cycles/instr/SIMD @2.4GHz\n", NCH, waves_per_simd, ms,
           2 * instr * 64 / ms * 1e-9, 2.4e9 * (ms * 1e-3) / (instr / 1024));
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 16 * 256 * 4);
    for (int w : {1, 2, 4, 8}) {
        run<1>(d, w); run<2>(d, w); run<3>(d, w); run<4>(d, w); run<6>(d, w); run<8>(d, w); run<12>(d, w); run<16>(d, w);
    }
    return 0;
}
