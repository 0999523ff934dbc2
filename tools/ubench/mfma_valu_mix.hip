// Micro-benchmark: what a wave of plain VALU work costs next to a wave of exact-fp32 MFMAs on the same SIMD (the fused dec4
// kernel's two work-groups per CU rely on one group's matrix phase hiding the other's combine phase).
// One work-group of 512 threads per CU = 2 waves per SIMD.  Waves 0-3 run NM v_mfma_f32_32x32x2_f32 (two independent
// accumulator chains); waves 4-7 run NV v_fmac_f32 (CH independent chains).  Each role is timed alone and together.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_valu_mix mfma_valu_mix.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int nm, int nv, float s) {
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    if (wave < 4) {
        f32x16 a0 = {0}, a1 = {0};
        const float x = threadIdx.x * 1e-3f, y = s;
        for (int i = 0; i < nm; i += 2) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) r += a0[i] + a1[i];
    } else {
        float a[CH];
        for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 1e-3f + i;
        float w[4] = {s, s * 0.5f, s * 0.25f, s * 0.125f};
        for (int it = 0; it < nv; it += 48) {
#pragma unroll
            for (int q = 0; q < 48 / CH; ++q)
#pragma unroll
                for (int i = 0; i < CH; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(w[q & 3]), "v"(w[(q + 1) & 3]));
        }
        for (int i = 0; i < CH; ++i) r += a[i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// ONE wave per SIMD: each MFMA followed by K independent v_fmac in the same instruction stream.  F16 = 1: v_mfma_f32_32x32x8_f16
// (16 passes as well) instead of the fp32 instruction.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int K, int F16>
__global__ __launch_bounds__(256) void ks(float* out, unsigned long long* cyc, int nm, float s) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    f32x16 a0 = {0}, a1 = {0};
    const float x = threadIdx.x * 1e-3f, y = s;
    const f16x4 hx = {(_Float16)x, (_Float16)y, (_Float16)x, (_Float16)y};
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < nm; i += 2) {
        if (F16) a0 = __builtin_amdgcn_mfma_f32_32x32x8f16(hx, hx, a0, 0, 0, 0); else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < K; ++q) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[q & 7]) : "v"(x), "v"(y));
        if (F16) a1 = __builtin_amdgcn_mfma_f32_32x32x8f16(hx, hx, a1, 0, 0, 0); else a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < K; ++q) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[q & 7]) : "v"(y), "v"(x));
    }
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += a0[i] + a1[i];
    for (int i = 0; i < 8; ++i) r += a[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int K, int F16>
static void run_same(float* d, unsigned long long* c) {
    const int blocks = 256, nm = 4096;
    static unsigned long long h[256 * 4];
    ks<K, F16><<<blocks, 256>>>(d, c, nm, 1e-6f);
    ks<K, F16><<<blocks, 256>>>(d, c, nm, 1e-6f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks * 4; ++i) m += (double)h[i];
    m /= blocks * 4;
    printf("same wave, %s MFMA + %2d fmac each: %.1f ticks per MFMA\n", F16 ? "f16 32x32x8" : "f32 32x32x2", K, m / nm);
}

template <int CH>
static void run(float* d, unsigned long long* c, int nm, int nv) {
    const int blocks = 256;
    static unsigned long long h[256 * 8];
    k<CH><<<blocks, 512>>>(d, c, nm, nv, 1e-6f);
    k<CH><<<blocks, 512>>>(d, c, nm, nv, 1e-6f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
    m /= blocks * 4; v /= blocks * 4;
    printf("chains %2d  mfma %6d  valu %7d : mfma waves %9.0f ticks (%.1f / mfma)   valu waves %9.0f ticks (%.2f / instr)\n", CH, nm, nv,
           m, nm ? m / nm : 0.0, v, nv ? v / nv : 0.0);
}

int main() {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&c, 256 * 8 * 8);
    printf("(ticks are s_memtime counts; a 32x32x2 fp32 MFMA is 64 shader clocks)\n");
    run<1>(d, c, 4096, 0); run<1>(d, c, 0, 48 * 1024);
    run<1>(d, c, 4096, 48 * 1024); run<2>(d, c, 4096, 48 * 1024); run<4>(d, c, 4096, 48 * 1024); run<8>(d, c, 4096, 48 * 1024);
    run<4>(d, c, 0, 48 * 1024); run<8>(d, c, 0, 48 * 1024);
    run<4>(d, c, 4096, 48 * 256); run<4>(d, c, 4096, 48 * 4096);
    run_same<0, 0>(d, c); run_same<4, 0>(d, c); run_same<8, 0>(d, c); run_same<12, 0>(d, c); run_same<16, 0>(d, c);
    run_same<0, 1>(d, c); run_same<4, 1>(d, c); run_same<8, 1>(d, c); run_same<12, 1>(d, c); run_same<16, 1>(d, c);
    return 0;
}
