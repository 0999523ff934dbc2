// Micro-benchmark: issue rate of DEPENDENT exact-fp32 MFMAs (every instruction accumulates into the previous one's result)
// against the number of independent accumulator chains per wave, one wave per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_chain mfma_chain.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCH, int BIG>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int nm, float s) {
    const float x = threadIdx.x * 1e-3f, y = s;
    f32x4 a4[NCH];
    f32x16 a16[NCH];
    for (int i = 0; i < NCH; ++i) { a4[i] = f32x4{0, 0, 0, 0}; for (int r = 0; r < 16; ++r) a16[i][r] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < nm; i += NCH) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (BIG) a16[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a16[c], 0, 0, 0);
            else a4[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4[c], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int c = 0; c < NCH; ++c) { for (int i = 0; i < 4; ++i) r += a4[c][i]; for (int i = 0; i < 16; ++i) r += a16[c][i]; }
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// bf16 32x32x16 (the training kernels' instruction): NCH independent chains taken round-robin
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NCH>
__global__ __launch_bounds__(256) void kb(float* out, unsigned long long* cyc, int nm, float s) {
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 1e-3f + i); y[i] = (__bf16)(s * i); }
    f32x16 a[NCH];
    for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) a[c][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < nm; i += NCH) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) a[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int c = 0; c < NCH; ++c) for (int i = 0; i < 16; ++i) r += a[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NCH>
static void runb(float* d, unsigned long long* c) {
    static unsigned long long h[256 * 4];
    kb<NCH><<<256, 256>>>(d, c, 4800, 1e-6f); kb<NCH><<<256, 256>>>(d, c, 4800, 1e-6f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 1024; ++i) m += (double)h[i];
    printf("32x32x16 bf16, %d chain(s): %.1f ticks per MFMA\n", NCH, m / 1024 / 4800);
}

// four accumulators of the 16x16x4 instruction, two MFMAs each per trip, in the order 0 1 0 1 2 3 2 3 instead of 0 1 2 3 0 1 2 3
__global__ __launch_bounds__(256) void kpair(float* out, unsigned long long* cyc, int nm, float s) {
    const float x = threadIdx.x * 1e-3f, y = s;
    f32x4 a[4];
    for (int i = 0; i < 4; ++i) a[i] = f32x4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < nm; i += 8) {
        a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[0], 0, 0, 0);
        a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[1], 0, 0, 0);
        a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a[0], 0, 0, 0);
        a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a[1], 0, 0, 0);
        a[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[2], 0, 0, 0);
        a[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[3], 0, 0, 0);
        a[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a[2], 0, 0, 0);
        a[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a[3], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 4; ++i) r += a[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NCH, int BIG>
static void run(float* d, unsigned long long* c) {
    const int blocks = 256, nm = 4800;
    static unsigned long long h[256 * 4];
    k<NCH, BIG><<<blocks, 256>>>(d, c, nm, 1e-6f);
    k<NCH, BIG><<<blocks, 256>>>(d, c, nm, 1e-6f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks * 4; ++i) m += (double)h[i];
    printf("%s, %d chain(s): %.1f ticks per MFMA\n", BIG ? "32x32x2 f32" : "16x16x4 f32", NCH, m / (blocks * 4) / nm);
}
int main() {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&c, 256 * 4 * 8);
    run<1, 0>(d, c); run<2, 0>(d, c); run<3, 0>(d, c); run<4, 0>(d, c);
    run<1, 1>(d, c); run<2, 1>(d, c); run<4, 1>(d, c);
    {
        static unsigned long long h[256 * 4];
        kpair<<<256, 256>>>(d, c, 4800, 1e-6f); kpair<<<256, 256>>>(d, c, 4800, 1e-6f);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 1024; ++i) m += (double)h[i];
        printf("16x16x4 f32, 4 chains in the order 0 1 0 1 2 3 2 3: %.1f ticks per MFMA\n", m / 1024 / 4800);
    }
    runb<1>(d, c); runb<2>(d, c); runb<3>(d, c); runb<4>(d, c); runb<6>(d, c); runb<8>(d, c);
    return 0;
}
