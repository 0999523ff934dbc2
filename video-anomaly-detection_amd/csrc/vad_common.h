// Shared declarations for libvad_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vad_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

// A positive, finite power of two: multiplying by it is exact (up to overflow / underflow), which is what the gradient
// rescaling of the split-fp16 training mode relies on.
static inline bool vad_is_pow2f(float v) {
    unsigned u;
    __builtin_memcpy(&u, &v, 4);
    return v > 0.f && (u & 0x007fffffu) == 0u && (u >> 23) != 0u && (u >> 23) != 0xffu;
}

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

// LeakyReLU(0.2) as max(v, 0.2 v): the same value as the select for every input (0.2 v < v exactly when v > 0), in two
// VALU instructions instead of v_mul + v_cmp + v_cndmask plus the VCC wait states - these epilogues run on the pipe the
// exact-fp32 MFMAs use.  One plain v_max_f32 each (fmaxf adds a canonicalising v_max(v, v) per operand under IEEE mode).
__device__ __forceinline__ float vad_vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vad_act(float v, int act) {
    if (act == VAD_ACT_LEAKY) return vad_vmax(v, 0.2f * v);
    if (act == VAD_ACT_RELU) return vad_vmax(v, 0.f);
    return v;
}

// Gate non-linearities of the ConvLSTM epilogue on the hardware transcendental unit (v_exp_f32 / v_rcp_f32,
// ~1 ulp each): 4-5 instructions instead of the ~40 of libm's expf/tanhf, absolute error ~1e-7.
// Reciprocal on the transcendental unit (v_rcp_f32, 1 ulp).  `__frcp_rn` / `1.0f / x` is the correctly rounded division: ten VALU
// instructions (v_div_scale x2, v_rcp, four FMAs, v_div_fmas, v_div_fixup) - five of them per ConvLSTM cell made 800 of the
// ~900 VALU instructions of that kernel's epilogue, on the pipe its MFMAs use.  The sigmoid / tanh built on it stay within
// ~2e-7 of libm's (tests hold the scores to 1e-5 of the oracle).
__device__ __forceinline__ float vad_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float vad_sigmoid(float v) { return vad_rcp(1.0f + __expf(-v)); }
__device__ __forceinline__ float vad_tanh(float v) { return __builtin_fmaf(-2.0f, vad_rcp(__expf(2.0f * v) + 1.0f), 1.0f); }
// ConvLSTMCell state update (reference models/video_autoencoder.py:76-83) from the four gate pre-activations.  The
// contractions are spelled out so that every kernel that fuses it (32x32x2 persistent / one-tile forms, 16x16x4 small-grid
// form) rounds identically: c' = fma(sigmoid(f), c, sigmoid(i)*tanh(g)), h' = sigmoid(o) * tanh(c').
__device__ __forceinline__ void vad_lstm_cell(float zi, float zf, float zg, float zo, float c_prev, float& c_next, float& h_next) {
    const float ig = vad_sigmoid(zi) * vad_tanh(zg);
    c_next = __builtin_fmaf(vad_sigmoid(zf), c_prev, ig);
    h_next = vad_sigmoid(zo) * vad_tanh(c_next);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Buffer descriptor for [p, p + bytes): loads past the end return 0 and stores past the end are dropped, which
// is how zero padding and partial tiles are handled without branches (invalid elements get offset 0x80000000).
// The pointer halves go through readfirstlane so hipcc can prove the descriptor wave-uniform (no waterfall loops).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vad_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 vad_bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x2 vad_bload2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float vad_bload1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void vad_bstore1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void vad_bstore_h(unsigned short v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b16((short)v, r, (int)voff, (int)soff, 0);
}
constexpr unsigned VAD_OOB = 0x80000000u;   // byte offset no frame reaches (host checks frames < 2^31 bytes)

// ---- bf16 storage (training mode VAD_PREC_BF16S: activations and activation gradients live in HBM as bf16, every
// arithmetic step is fp32).  Storage type = the 16-bit pattern; conversion to bf16 rounds to nearest even.
typedef unsigned short vad_bf16;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vad_bf16_f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned short vad_f_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned vad_pack_bf16(float lo, float hi) {
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, bf16x2_{(__bf16)lo, (__bf16)hi});
}
// four consecutive elements <-> f32x4 (fp32: one 16-byte access; bf16: one 8-byte access)
template <typename T> struct vad_io4;
template <> struct vad_io4<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *(f32x4*)p = v; }
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
    static __device__ __forceinline__ float round(float v) { return v; }
};
template <> struct vad_io4<vad_bf16> {
    static __device__ __forceinline__ f32x4 ld(const vad_bf16* p) {
        const u32x2 r = *(const u32x2*)p;
        return f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
    }
    static __device__ __forceinline__ void st(vad_bf16* p, f32x4 v) { *(u32x2*)p = u32x2{vad_pack_bf16(v[0], v[1]), vad_pack_bf16(v[2], v[3])}; }
    static __device__ __forceinline__ float ld1(const vad_bf16* p) { return vad_bf16_f(*p); }
    static __device__ __forceinline__ float round(float v) { return vad_bf16_f(vad_f_bf16(v)); }
};

// one element through a buffer descriptor, offsets in BYTES (fp32: dword load; bf16: 16-bit load, widened)
template <typename T> __device__ __forceinline__ float vad_bload_e(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ float vad_bload_e<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return vad_bload1(r, voff, soff); }
template <> __device__ __forceinline__ float vad_bload_e<vad_bf16>(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return vad_bf16_f((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, 0));
}

// Input formats of the ORIGINAL frames handed to the model-level entry points.
//   VAD_X_F32_NCHW : float32 [N,3,H,W], already normalised to [-1,1] (what the reference's datasets produce)
//   VAD_X_U8_NHWC  : uint8 [N,H,W,3] as decoded; the kernels apply ToTensor + Normalize(0.5,0.5) themselves,
//                    (u8/255 - 0.5)/0.5 in fp32 step by step (reference utils/dataset.py:65-70), bit-identical
__device__ __forceinline__ float vad_norm_u8(unsigned v) {
    float f = (float)v;
    f = f / 255.0f;
    f = f - 0.5f;
    return f / 0.5f;
}

// format-aware internals behind the float-only layer entry points of include/vad_hip.h
int vad_conv3x3_c3_fmt(const void* x, int fmt, const float* w, const float* bias, float* out, int n, int h, int wd,
                       int cout, int act, int pool, void* stream);
int vad_conv3x3_c3_stats(const void* x, int fmt, const float* w, const float* bias, float* out, int n, int h, int wd, int cout,
                         int act, int pool, float* stats, int* stats_blocks, void* stream);   // + BatchNorm partial sums (training forward)
int vad_conv3x3_stats(const float* in, long long in_fs, const float* w, const float* bias, float* out, long long out_fs, int n, int h,
                      int wd, int cin, int cout, int act, int pool, int precision, float* stats, int* stats_rows, void* stream);
int vad_conv3x3_kpart(const float* in, long long in_fs, const float* w, const float* bias, float* out, long long out_fs, int n, int h,
                      int wd, int cin, int cin_w, int cout, int act, int pool, int precision, float* stats, int* stats_rows, void* stream);
bool vad_convlstm_small_wins(int n, int h, int wd, int hid);
bool vad_convlstm_gate_wins(int n, int h, int wd, int hid);   // would a ConvLSTM step of n frames run the gate-split kernel?
bool vad_convlstm_hoist_ok(void);
bool vad_conv_stats_available(void);
int vad_convlstm_step_zx(const float* x, long long x_fs, const float* zx, long long zx_fs, const float* h_prev, long long h_prev_fs,
                         const float* c_prev, const float* w, const float* bias, float* h_out, long long h_out_fs,
                         float* c_out, int n, int h, int wd, int cin_x, int hid, int precision, void* stream);
size_t vad_conv3x3_stats_floats(int cout);
int vad_convt2x2_stats(const float* in, long long in_fs, const float* w, const float* bias, float* out, long long out_fs, int n, int h,
                       int wd, int cin, int cout, int act, int precision, float* stats, int* stats_rows, void* stream);
size_t vad_convt2x2_stats_floats(int cout);
int vad_bn_stats_from_partials(const float* partials, int nblocks, long long npix, int c, float eps, float momentum, float* stats,
                               float* running_mean, float* running_var, const float* pivot, void* stream);
int vad_conv3x3_c3_fused_fmt(const void* x, int fmt, const float* w0, const float* b0, const float* w1, const float* b1,
                             float* out, int n, int h, int wd, int precision, void* stream);
int vad_nhwc_to_nchw_ld(const float* in, int in_c, float* out, int n, int h, int w, int c, void* stream);   // first c of in_c channels
// first layer with a bf16 output tensor (out16 != 0; VAD_PREC_BF16S); the other bf16-tensor forms are declared in vad_hip.h
int vad_conv3x3_c3_stats_t(const void* x, int fmt, const float* w, const float* bias, void* out, int out16, int n, int h, int wd, int cout,
                           int act, int pool, float* stats, int* stats_blocks, void* stream);
// vad_score_finalize + the device-side blob check: when hdr != NULL and hdr[1] != want_tag every score becomes NaN
// csrc/wide_io.hip (models with in_channels > 3)
int vad_nchw_to_nhwc_pad(const float* x, float* out, long long n, int h, int w, int c, int cpad, void* stream);
int vad_wide_score_partials(int h, int w);
int vad_tanh_score_nhwc(const float* pre, int cpad, const float* x, int c, float* partials, float* recon, float* errmap,
                        int n, int h, int w, int t, int clip_stride, void* stream);
// (channels: the planes a frame's squared error is averaged over - 3 on the 3-plane path)
int vad_score_finalize_tagged(const float* partials, int nparts, int n, int h2, int w2, int channels, float* frame_scores,
                              float* seq_scores, int t, const unsigned* hdr, unsigned want_tag, void* stream);
int vad_conv3x3_to3_score_fmt(const float* in, const float* w_packed, const float* bias3, const void* x, int fmt,
                              float* partials, float* recon, float* errmap, int n, int h2, int w2, int cin, void* stream);
int vad_convt2x2_to3_score_fmt(const float* in, const float* w_iohw, const float* bias3, const void* x, int fmt,
                               float* partials, float* recon, float* errmap, int n, int h, int w, int cin, int t,
                               int clip_stride, void* stream);

// dec4.0 + dec4.3 + score fused (dec4_fused.hip); w2_gemm = the second form written by vad_pack_conv3x3_to3 (cin 32)
int vad_dec4_score_fmt(const float* in, const float* wt_packed, const float* bt, const float* w2_gemm, const float* bias3,
                       const void* x, int fmt, float* partials, float* recon, float* errmap, int n, int h, int w, void* stream);

// per-layer profiling hooks (vad_api.hip)
struct VadProfScope {
    int slot;
    hipStream_t stream;
    hipEvent_t end;
    VadProfScope(int slot, hipStream_t stream);
    ~VadProfScope();
};
