// Shared declarations for libvad_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vad_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

int vad_fail(int code, const char* fmt, ...);

#define VAD_HIP_TRY(expr)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return vad_fail(VAD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                            __FILE__, __LINE__);                                           \
    } while (0)

#define VAD_REQUIRE(cond, ...)                               \
    do {                                                     \
        if (!(cond)) return vad_fail(VAD_ERR_ARG, __VA_ARGS__); \
    } while (0)

// Launch check: kernel launches report configuration errors through hipGetLastError.
#define VAD_LAUNCH_CHECK() VAD_HIP_TRY(hipGetLastError())

// XCD-aware block remap (8 XCDs, blocks dealt round-robin): logical ids that are consecutive
// land on the same XCD, so work-groups that share an input tile or a weight panel share an L2.
// Bijective for any block count.  Speed only, never correctness.
__device__ __forceinline__ unsigned vad_xcd_remap(unsigned b, unsigned nb) {
    const unsigned q = nb >> 3, r = nb & 7u, x = b & 7u;
    const unsigned base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (b >> 3);
}

__device__ __forceinline__ float vad_act(float v, int act) {
    if (act == VAD_ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    if (act == VAD_ACT_RELU) return fmaxf(v, 0.f);
    return v;
}

// Gate non-linearities of the ConvLSTM epilogue on the hardware transcendental unit (v_exp_f32 / v_rcp_f32,
// ~1 ulp each): 4-5 instructions instead of the ~40 of libm's expf/tanhf, absolute error ~1e-7.
__device__ __forceinline__ float vad_sigmoid(float v) { return __frcp_rn(1.0f + __expf(-v)); }
__device__ __forceinline__ float vad_tanh(float v) { return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * v) + 1.0f); }

// per-layer profiling hooks (vad_api.hip)
struct VadProfScope {
    int slot;
    hipStream_t stream;
    VadProfScope(int slot, hipStream_t stream);
    ~VadProfScope();
};
