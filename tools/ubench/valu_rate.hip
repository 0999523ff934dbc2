// Micro-benchmark: issue rate of v_fmac_f32 vs v_pk_fma_f32 on gfx950 (is packed fp32 twice the rate of scalar fp32 per
// lane?).  Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a[16];
    f2 p[8];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    float w0 = s, w1 = s * 0.5f;
    f2 w = f2{w0, w1};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(w0), "v"(w1));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(w), "v"(w));
        } else {   // broadcast of one scalar input against a weight pair, as a convolution would use it
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(w), "v"(w));
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i];
    for (int i = 0; i < 8; ++i) r += p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
static void run(const char* name, float* d, int macs_per_instr) {
    const int blocks = 256 * 8, iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 16, 1e-6f);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1e-6f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 /*waves*/ * iters * (MODE == 0 ? 64 : 32);
    const double macs = instr * 64 * macs_per_instr;
    printf("%-28s %.3f ms  %.1f G wave-instr/s  %.1f TFLOP/s  (%.2f cycles/instr/SIMD at 2.4 GHz)\n", name, ms, instr / ms * 1e-6,
           2 * macs / ms * 1e-9, 2.4e9 * (ms * 1e-3) / (instr / 1024));
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_fmac_f32", d, 1);
    run<1>("v_pk_fma_f32", d, 2);
    run<2>("v_pk_fma_f32 op_sel bcast", d, 2);
    return 0;
}
