// Training-step kernels (reference train_video.py:44-65), gfx950, exact fp32.
//
// The reference's step is stock autograd over models/video_autoencoder.py in train() mode.  Here the same arithmetic
// is stated as explicit kernels on NHWC activations:
//   * train-mode BatchNorm2d (batch statistics over N*H*W, eps 1e-5, momentum 0.1, unbiased running variance):
//     per-channel sums with wave/LDS reductions and a fixed-order fp64 finalize, then ONE fused
//     normalise + activation (+ MaxPool2d) pass; backward in two passes (route the pooled gradient, apply act',
//     accumulate sum(dz), sum(dz*xhat); then dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)))
//   * conv / convT weight gradients as one MFMA GEMM (M = input channels, N = output channels, K = pixels) with a
//     deterministic split-K (partials + fixed-order reduce straight into the torch OIHW / IOHW layouts)
//   * conv data gradients reuse the forward implicit-GEMM kernels on re-packed (rotated / transposed) weights;
//     convT data gradients are a 1x1 GEMM over the space-to-depth view that the BatchNorm backward writes directly
//   * ConvLSTM gate non-linearities + state update, forward and backward (BPTT), as pointwise kernels
//   * ConvTranspose2d(32->3) + Tanh + MSELoss forward AND backward in one pass over the last activation
//   * Adam with L2 weight decay folded into the gradient (torch.optim.Adam semantics) over ONE flat parameter buffer.
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "vad_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {

__device__ __forceinline__ long long view_frame(int n, int t, int b) { return t > 0 ? (long long)(n % t) * b + n / t : n; }

// Reduce two per-thread float4 channel accumulators over the pixel rows of a 256-thread block; thread = (row, c4).
__device__ __forceinline__ void block_chan_reduce(f32x4 s0, f32x4 s1, int c, float* dst /*[2][c]*/) {
    __shared__ f32x4 red[2][256];
    const int tid = threadIdx.x, cg = c >> 2, rows = 256 / cg;
    red[0][tid] = s0;
    red[1][tid] = s1;
    __syncthreads();
    if (tid < cg) {
        f32x4 a = red[0][tid], b = red[1][tid];
        for (int r = 1; r < rows; ++r) { a += red[0][r * cg + tid]; b += red[1][r * cg + tid]; }
        *(f32x4*)&dst[4 * tid] = a;
        *(f32x4*)&dst[c + 4 * tid] = b;
    }
}

// ------------------------------------------------------------------------------------------------ channel sums
// ws[block][0][c] = sum over the block's pixels of (v - pivot[c]), ws[block][1][c] = sum of (v - pivot[c])^2.
// pivot (nullable = 0) is a per-channel shift applied BEFORE squaring: BatchNorm statistics pass the channel's first
// sample, so the variance is not formed as E[x^2] - mean^2 of the raw values (that cancels catastrophically in fp32 when
// |mean| >> std: measured 5e-5 relative error in 1/std, enough to flip ReLU decisions in the layers that follow).
template <typename T>
__global__ __launch_bounds__(256) void chan_sums_kernel(const T* y, long long npix, int c, long long chunk, const float* pivot,
                                                        float* ws) {
    const int tid = threadIdx.x, cg = c >> 2, rows = 256 / cg, row = tid / cg, c4 = tid - row * cg;
    const long long p0 = (long long)blockIdx.x * chunk, p1 = (p0 + chunk < npix) ? p0 + chunk : npix;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (row < rows) {
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (pivot) pv = *(const f32x4*)&pivot[4 * c4];
        // four loads in flight per thread (same accumulation order): one at a time, a CU's 2048 threads keep 32 KB in flight and
        // the pass ran at 3.8 TB/s
        long long p = p0 + row;
        for (; p + 3 * rows < p1; p += 4 * rows) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = vad_io4<T>::ld(&y[(p + (long long)u * rows) * c + 4 * c4]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 d = v[u] - pv;
                s0 += d;
                s1 += d * d;
            }
        }
        for (; p < p1; p += rows) {
            const f32x4 v = vad_io4<T>::ld(&y[p * c + 4 * c4]) - pv;
            s0 += v;
            s1 += v * v;
        }
    }
    block_chan_reduce(s0, s1, c, ws + (size_t)blockIdx.x * 2 * c);
}

// mode 0: BatchNorm statistics -> stats[0][c] = mean, stats[1][c] = 1/sqrt(var_biased + eps); running stats updated
//         like torch (momentum; unbiased variance) when given
// mode 1: BatchNorm backward sums -> dgamma = sum(dz*xhat), dbeta = sum(dz), stats = {sum(dz)/M, sum(dz*xhat)/M}
// mode 2: plain sums -> dbeta[c] = sum (bias gradients)
__global__ __launch_bounds__(256) void chan_finalize_kernel(const float* ws, int nblocks, int c, double count, int mode,
                                                            float eps, float momentum, float* stats, float* running_mean,
                                                            float* running_var, float* dgamma, float* dbeta, const float* pivot) {
    // one work-group per channel: thread t adds partial blocks t, t+256, ... in fp64, then a fixed-shape tree over the
    // 256 threads (the result depends only on the data and the block count, never on scheduling)
    __shared__ double rs[256], rq[256];
    const int ch = blockIdx.x, tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = tid; b < nblocks; b += 256) {
        s += (double)ws[(size_t)b * 2 * c + ch];
        q += (double)ws[(size_t)b * 2 * c + c + ch];
    }
    rs[tid] = s;
    rq[tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
        __syncthreads();
    }
    if (tid != 0) return;
    s = rs[0];
    q = rq[0];
    if (mode == 0) {
        const double sm = s / count;                     // mean of the shifted values
        const double mean = (pivot ? (double)pivot[ch] : 0.0) + sm;
        double var = q / count - sm * sm;
        if (var < 0.0) var = 0.0;
        stats[ch] = (float)mean;
        stats[c + ch] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + momentum * mean);
            running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + momentum * unb);
        }
    } else if (mode == 1) {
        dbeta[ch] = (float)s;
        dgamma[ch] = (float)q;
        stats[ch] = (float)(s / count);
        stats[c + ch] = (float)(q / count);
    } else {
        dbeta[ch] = (float)s;
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm forward
struct BnFwdP {     // y / out: fp32 or bf16 (the kernel's storage type)
    const void* y; const float* stats; const float* gamma; const float* beta;
    void* out; long long out_fs; int out_ps, t, b;
    int n, h, w, c, act, pool;
    long long total;      // n * oh * ow * c/4
};

// xhat and the BatchNorm output with every operation rounded on its own (no fma contraction): forward, backward pass A and
// backward pass B must take bit-identical branch decisions whatever code the compiler generates around them.
__device__ __forceinline__ float bn_xhat(float y, float mean, float invstd) { return __fmul_rn(__fsub_rn(y, mean), invstd); }
__device__ __forceinline__ float bn_value(float xhat, float gamma, float beta) { return __fadd_rn(__fmul_rn(xhat, gamma), beta); }

__device__ __forceinline__ f32x4 bn_apply(f32x4 y, f32x4 mean, f32x4 invstd, f32x4 gamma, f32x4 beta, int act) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = vad_act(bn_value(bn_xhat(y[e], mean[e], invstd[e]), gamma[e], beta[e]), act);
    return v;
}

// POOL / ACT are compile-time (host dispatch): with run-time flags every element paid the selects of both activations and the
// window loop's guards - VALU work these passes can no longer hide once the tensors are bf16.
template <typename T, int POOL, int ACT>
__global__ __launch_bounds__(256) void bn_act_pool_fwd_kernel(BnFwdP p) {
    typedef vad_io4<T> io;
    const int cg = p.c >> 2, oh = POOL ? p.h / 2 : p.h, ow = POOL ? p.w / 2 : p.w;
    // (32-bit index arithmetic, host-checked: three 64-bit divisions per item cost more than the item's memory traffic once
    // the tensors are bf16)
    const unsigned total = (unsigned)p.total, ucg = (unsigned)cg, uow = (unsigned)ow, uoh = (unsigned)oh;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned pix = idx / ucg, t_ = pix / uow, un = t_ / uoh;
        const int c4 = (int)(idx - pix * ucg);
        const int x = (int)(pix - t_ * uow);
        const int y = (int)(t_ - un * uoh);
        const int n = (int)un;
        const f32x4 mean = *(const f32x4*)&p.stats[4 * c4], invstd = *(const f32x4*)&p.stats[p.c + 4 * c4];
        const f32x4 gamma = *(const f32x4*)&p.gamma[4 * c4], beta = *(const f32x4*)&p.beta[4 * c4];
        const T* src = (const T*)p.y + (size_t)n * p.h * p.w * p.c + 4 * c4;
        f32x4 v;
        if constexpr (POOL) {
            const size_t o = ((size_t)(2 * y) * p.w + 2 * x) * p.c;
            v = bn_apply(io::ld(&src[o]), mean, invstd, gamma, beta, ACT);
            const f32x4 v1 = bn_apply(io::ld(&src[o + p.c]), mean, invstd, gamma, beta, ACT);
            const f32x4 v2 = bn_apply(io::ld(&src[o + (size_t)p.w * p.c]), mean, invstd, gamma, beta, ACT);
            const f32x4 v3 = bn_apply(io::ld(&src[o + (size_t)p.w * p.c + p.c]), mean, invstd, gamma, beta, ACT);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaxf(v[e], v1[e]), fmaxf(v2[e], v3[e]));
        } else {
            v = bn_apply(io::ld(&src[((size_t)y * p.w + x) * p.c]), mean, invstd, gamma, beta, ACT);
        }
        io::st((T*)p.out + view_frame(n, p.t, p.b) * p.out_fs + ((size_t)y * ow + x) * p.out_ps + 4 * c4, v);
    }
}

// bf16 tensors, EIGHT channels per thread (16-byte accesses: twice the bytes in flight per wave, half the index arithmetic);
// every element goes through the same bn_apply as above: identical results.  c % 8 == 0, strides % 8 == 0 (host-checked).
__device__ __forceinline__ void bf16_ld8(const vad_bf16* p, f32x4& a, f32x4& b) {
    const u32x4 r = *(const u32x4*)p;
    a = f32x4{__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16), __uint_as_float(r[1] & 0xffff0000u)};
    b = f32x4{__uint_as_float(r[2] << 16), __uint_as_float(r[2] & 0xffff0000u), __uint_as_float(r[3] << 16), __uint_as_float(r[3] & 0xffff0000u)};
}
__device__ __forceinline__ void bf16_st8(vad_bf16* p, f32x4 a, f32x4 b) {
    *(u32x4*)p = u32x4{vad_pack_bf16(a[0], a[1]), vad_pack_bf16(a[2], a[3]), vad_pack_bf16(b[0], b[1]), vad_pack_bf16(b[2], b[3])};
}
template <int POOL, int ACT>
__global__ __launch_bounds__(256) void bn_act_pool_fwd8_kernel(BnFwdP p) {
    const int cg = p.c >> 3, oh = POOL ? p.h / 2 : p.h, ow = POOL ? p.w / 2 : p.w;
    const unsigned total = (unsigned)(p.total >> 1), ucg = (unsigned)cg, uow = (unsigned)ow, uoh = (unsigned)oh;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned pix = idx / ucg, t_ = pix / uow, un = t_ / uoh;
        const int c8 = (int)(idx - pix * ucg);
        const int x = (int)(pix - t_ * uow);
        const int y = (int)(t_ - un * uoh);
        const int n = (int)un;
        f32x4 mean[2], invstd[2], gamma[2], beta[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            mean[hf] = *(const f32x4*)&p.stats[8 * c8 + 4 * hf]; invstd[hf] = *(const f32x4*)&p.stats[p.c + 8 * c8 + 4 * hf];
            gamma[hf] = *(const f32x4*)&p.gamma[8 * c8 + 4 * hf]; beta[hf] = *(const f32x4*)&p.beta[8 * c8 + 4 * hf];
        }
        const vad_bf16* src = (const vad_bf16*)p.y + (size_t)n * p.h * p.w * p.c + 8 * c8;
        f32x4 v[2];
        if constexpr (POOL) {
            const size_t o = ((size_t)(2 * y) * p.w + 2 * x) * p.c;
            f32x4 w0[2], w1[2], w2[2], w3[2];
            bf16_ld8(&src[o], w0[0], w0[1]);
            bf16_ld8(&src[o + p.c], w1[0], w1[1]);
            bf16_ld8(&src[o + (size_t)p.w * p.c], w2[0], w2[1]);
            bf16_ld8(&src[o + (size_t)p.w * p.c + p.c], w3[0], w3[1]);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                v[hf] = bn_apply(w0[hf], mean[hf], invstd[hf], gamma[hf], beta[hf], ACT);
                const f32x4 v1 = bn_apply(w1[hf], mean[hf], invstd[hf], gamma[hf], beta[hf], ACT);
                const f32x4 v2 = bn_apply(w2[hf], mean[hf], invstd[hf], gamma[hf], beta[hf], ACT);
                const f32x4 v3 = bn_apply(w3[hf], mean[hf], invstd[hf], gamma[hf], beta[hf], ACT);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[hf][e] = fmaxf(fmaxf(v[hf][e], v1[e]), fmaxf(v2[e], v3[e]));
            }
        } else {
            f32x4 w0[2];
            bf16_ld8(&src[((size_t)y * p.w + x) * p.c], w0[0], w0[1]);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) v[hf] = bn_apply(w0[hf], mean[hf], invstd[hf], gamma[hf], beta[hf], ACT);
        }
        bf16_st8((vad_bf16*)p.out + view_frame(n, p.t, p.b) * p.out_fs + ((size_t)y * ow + x) * p.out_ps + 8 * c8, v[0], v[1]);
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm backward
struct BnBwdP {     // y / dout / dy: fp32 or bf16 (the kernels' storage type)
    const void* y; const float* stats; const float* gamma; const float* beta;
    const void* dout; long long dout_fs; int dout_ps, t, b;
    float* ws;            // pass A: partial sums
    const float* k;       // pass B: {sum(dz)/M, sum(dz*xhat)/M}
    void* dy;             // pass B: gradient of the conv output, dense NHWC or space-to-depth
    int s2d;
    int n, h, w, c, act, pool;
    long long opix, chunk;   // pooled-resolution pixels (n*oh*ow), per block
    unsigned char* dec;   // debug (nullable): [opix][c] bytes, bits 0-1 = pooling argmax (window scan order), bit 2 = value > 0
    unsigned char* codes; // nullable: the same bytes as an OUTPUT of pass A (the routed first-layer weight gradient reads them)
};

__device__ __forceinline__ float act_grad(float v, int act) {
    if (act == VAD_ACT_LEAKY) return v > 0.f ? 1.f : 0.2f;
    if (act == VAD_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    return 1.f;
}

// The routed gradient of one (pooled) output element: which window element receives it (first maximum in window scan
// order, like torch), its xhat, and dz = d(out) * act'(value).  Shared by both passes.
struct Routed { int am; float xh, gz, v; };
__device__ __forceinline__ Routed bn_route(const float y4[4], int nwin, float mean, float invstd, float gamma, float beta, float g, int act) {
    Routed r;
    float xh[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < nwin) { xh[k] = bn_xhat(y4[k], mean, invstd); v[k] = vad_act(bn_value(xh[k], gamma, beta), act); }
    r.am = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < nwin && v[k] > v[r.am]) r.am = k;
    r.xh = xh[r.am];
    r.v = v[r.am];
    r.gz = g * act_grad(r.v, act);
    return r;
}

// pass A: per-channel partial sums of dz and dz*xhat (dz itself is never stored; pass B re-derives it)
template <typename T, int POOL, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(BnBwdP p) {
    typedef vad_io4<T> io;
    const T* py = (const T*)p.y;
    const T* pdout = (const T*)p.dout;
    const int tid = threadIdx.x, cg = p.c >> 2, rows = 256 / cg, row = tid / cg, c4 = tid - row * cg;
    const int oh = POOL ? p.h / 2 : p.h, ow = POOL ? p.w / 2 : p.w; constexpr int nwin = POOL ? 4 : 1;
    const long long p0 = (long long)blockIdx.x * p.chunk, p1 = (p0 + p.chunk < p.opix) ? p0 + p.chunk : p.opix;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (row < rows) {
        const f32x4 mean = *(const f32x4*)&p.stats[4 * c4], invstd = *(const f32x4*)&p.stats[p.c + 4 * c4];
        const f32x4 gamma = *(const f32x4*)&p.gamma[4 * c4], beta = *(const f32x4*)&p.beta[4 * c4];
        auto fetch = [&](long long q, f32x4& g, f32x4 (&yv)[4]) {
            const unsigned uq = (unsigned)q, t_ = uq / (unsigned)ow, un = t_ / (unsigned)oh;       // (opix < 2^31: host-checked)
            const int x = (int)(uq - t_ * (unsigned)ow), y = (int)(t_ - un * (unsigned)oh), n = (int)un;
            g = io::ld(&pdout[view_frame(n, p.t, p.b) * p.dout_fs + ((size_t)y * ow + x) * p.dout_ps + 4 * c4]);
            const size_t o0 = (size_t)n * p.h * p.w * p.c + 4 * c4 + (POOL ? ((size_t)(2 * y) * p.w + 2 * x) : ((size_t)y * p.w + x)) * p.c;
            yv[0] = io::ld(&py[o0]);
            if constexpr (POOL) { yv[1] = io::ld(&py[o0 + p.c]); yv[2] = io::ld(&py[o0 + (size_t)p.w * p.c]); yv[3] = io::ld(&py[o0 + (size_t)p.w * p.c + p.c]); }
        };
        auto add = [&](long long q, const f32x4& g, const f32x4 (&yv)[4]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y4[4] = {yv[0][e], yv[1][e], yv[2][e], yv[3][e]};
                const Routed r = bn_route(y4, nwin, mean[e], invstd[e], gamma[e], beta[e], g[e], ACT);
                if (p.dec) p.dec[q * p.c + 4 * c4 + e] = (unsigned char)(r.am | (r.v > 0.f ? 4 : 0));
                if (p.codes) p.codes[q * p.c + 4 * c4 + e] = (unsigned char)(r.am | (r.v > 0.f ? 4 : 0));
                s0[e] += r.gz;
                s1[e] += r.gz * r.xh;
            }
        };
        // (two pixels per trip with all ten loads first measured SLOWER - 1.61 -> 2.0 ms per bf16 training step: the second
        // register set costs occupancy, and the pass is bound by its ~230 VALU instructions per pixel quad as much as by memory)
        for (long long q = p0 + row; q < p1; q += rows) {
            f32x4 ga, ya[4];
            fetch(q, ga, ya);
            add(q, ga, ya);
        }
    }
    block_chan_reduce(s0, s1, p.c, p.ws + (size_t)blockIdx.x * 2 * p.c);
}

// pass B: dy = gamma * invstd * (dz - k1 - xhat * k2) for every element of the window (dz = 0 off the routed element);
// s2d writes the space-to-depth view [n][h/2][w/2][4][c] (the operand layout of the transposed convolution's gradients)
template <typename T, int POOL, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdP p) {
    typedef vad_io4<T> io;
    const T* py = (const T*)p.y;
    const T* pdout = (const T*)p.dout;
    T* pdy = (T*)p.dy;
    const int cg = p.c >> 2, oh = POOL ? p.h / 2 : p.h, ow = POOL ? p.w / 2 : p.w; constexpr int nwin = POOL ? 4 : 1;
    const unsigned total = (unsigned)(p.opix * cg), ucg = (unsigned)cg;               // (< 2^31: host-checked)
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned q = idx / ucg, t_ = q / (unsigned)ow, un = t_ / (unsigned)oh;
        const int c4 = (int)(idx - q * ucg);
        const int x = (int)(q - t_ * (unsigned)ow), y = (int)(t_ - un * (unsigned)oh), n = (int)un;
        const f32x4 mean = *(const f32x4*)&p.stats[4 * c4], invstd = *(const f32x4*)&p.stats[p.c + 4 * c4];
        const f32x4 gamma = *(const f32x4*)&p.gamma[4 * c4], beta = *(const f32x4*)&p.beta[4 * c4];
        const f32x4 k1 = *(const f32x4*)&p.k[4 * c4], k2 = *(const f32x4*)&p.k[p.c + 4 * c4];
        const f32x4 g = io::ld(&pdout[view_frame(n, p.t, p.b) * p.dout_fs + ((size_t)y * ow + x) * p.dout_ps + 4 * c4]);
        const size_t fb = (size_t)n * p.h * p.w * p.c + 4 * c4;
        const size_t o0 = fb + (POOL ? ((size_t)(2 * y) * p.w + 2 * x) : ((size_t)y * p.w + x)) * p.c;
        const size_t off[4] = {o0, o0 + p.c, o0 + (size_t)p.w * p.c, o0 + (size_t)p.w * p.c + p.c};
        f32x4 yv[4], out[4];
        yv[0] = io::ld(&py[off[0]]);
        if constexpr (POOL) { yv[1] = io::ld(&py[off[1]]); yv[2] = io::ld(&py[off[2]]); yv[3] = io::ld(&py[off[3]]); }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y4[4] = {yv[0][e], yv[1][e], yv[2][e], yv[3][e]};
            const Routed r = bn_route(y4, nwin, mean[e], invstd[e], gamma[e], beta[e], g[e], ACT);
            const float sc = gamma[e] * invstd[e];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nwin) {
                    const float xh = bn_xhat(y4[k], mean[e], invstd[e]);
                    out[k][e] = sc * ((k == r.am ? r.gz : 0.f) - k1[e] - xh * k2[e]);
                }
        }
        if (p.s2d) {          // no-pool layers only (checked on the host): pixel (y, x) of an h x w map
            const size_t o = (((size_t)n * (p.h / 2) + y / 2) * (p.w / 2) + x / 2) * 4 * p.c + ((y & 1) * 2 + (x & 1)) * p.c + 4 * c4;
            io::st(&pdy[o], out[0]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nwin) io::st(&pdy[off[k]], out[k]);
        }
    }
}

// pass B on bf16 tensors, eight channels per thread (see bn_act_pool_fwd8_kernel): the same bn_route / expressions per element
template <int POOL, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_apply8_kernel(BnBwdP p) {
    const vad_bf16* py = (const vad_bf16*)p.y;
    const vad_bf16* pdout = (const vad_bf16*)p.dout;
    vad_bf16* pdy = (vad_bf16*)p.dy;
    const int cg = p.c >> 3, oh = POOL ? p.h / 2 : p.h, ow = POOL ? p.w / 2 : p.w; constexpr int nwin = POOL ? 4 : 1;
    const unsigned total = (unsigned)(p.opix * cg), ucg = (unsigned)cg;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned q = idx / ucg, t_ = q / (unsigned)ow, un = t_ / (unsigned)oh;
        const int c8 = (int)(idx - q * ucg);
        const int x = (int)(q - t_ * (unsigned)ow), y = (int)(t_ - un * (unsigned)oh), n = (int)un;
        f32x4 g[2];
        bf16_ld8(&pdout[view_frame(n, p.t, p.b) * p.dout_fs + ((size_t)y * ow + x) * p.dout_ps + 8 * c8], g[0], g[1]);
        const size_t fb = (size_t)n * p.h * p.w * p.c + 8 * c8;
        const size_t o0 = fb + (POOL ? ((size_t)(2 * y) * p.w + 2 * x) : ((size_t)y * p.w + x)) * p.c;
        const size_t off[4] = {o0, o0 + p.c, o0 + (size_t)p.w * p.c, o0 + (size_t)p.w * p.c + p.c};
        f32x4 yv[4][2], out[4][2];
        bf16_ld8(&py[off[0]], yv[0][0], yv[0][1]);
        if constexpr (POOL) { bf16_ld8(&py[off[1]], yv[1][0], yv[1][1]); bf16_ld8(&py[off[2]], yv[2][0], yv[2][1]); bf16_ld8(&py[off[3]], yv[3][0], yv[3][1]); }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int c4 = 2 * c8 + hf;
            const f32x4 mean = *(const f32x4*)&p.stats[4 * c4], invstd = *(const f32x4*)&p.stats[p.c + 4 * c4];
            const f32x4 gamma = *(const f32x4*)&p.gamma[4 * c4], beta = *(const f32x4*)&p.beta[4 * c4];
            const f32x4 k1 = *(const f32x4*)&p.k[4 * c4], k2 = *(const f32x4*)&p.k[p.c + 4 * c4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y4[4] = {yv[0][hf][e], yv[1][hf][e], yv[2][hf][e], yv[3][hf][e]};
                const Routed r = bn_route(y4, nwin, mean[e], invstd[e], gamma[e], beta[e], g[hf][e], ACT);
                const float sc = gamma[e] * invstd[e];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < nwin) {
                        const float xh = bn_xhat(y4[k], mean[e], invstd[e]);
                        out[k][hf][e] = sc * ((k == r.am ? r.gz : 0.f) - k1[e] - xh * k2[e]);
                    }
            }
        }
        if (p.s2d) {          // no-pool layers only (checked on the host): pixel (y, x) of an h x w map
            const size_t o = (((size_t)n * (p.h / 2) + y / 2) * (p.w / 2) + x / 2) * 4 * p.c + ((y & 1) * 2 + (x & 1)) * p.c + 8 * c8;
            bf16_st8(&pdy[o], out[0][0], out[0][1]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nwin) bf16_st8(&pdy[off[k]], out[k][0], out[k][1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ ConvLSTM pointwise
struct LstmFwdP {   // z / h1 / h2: fp32 or bf16 (the kernel's storage type); the cell state is always fp32
    void* z;                  // [npix][4*hid]: pre-activations in, activated gates (i, f, g, o) out
    const float* c_prev;      // [npix][hid] or null (zeros)
    float* c_out;             // [npix][hid]
    void* h1; long long h1_fs; int h1_ps;      // destination 1 of h (frame b, pixel, channel) or null
    void* h2; long long h2_fs; int h2_ps;      // destination 2 or null
    int hw, hid;
    long long total;          // nb * hw * hid/4
};

template <typename T>
__global__ __launch_bounds__(256) void lstm_gates_fwd_kernel(LstmFwdP p) {
    typedef vad_io4<T> io;
    const int hg = p.hid >> 2;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < p.total; idx += (long long)gridDim.x * 256) {
        const int j4 = (int)(idx % hg);
        const long long pix = idx / hg;
        T* zz = (T*)p.z + pix * 4 * p.hid + 4 * j4;
        f32x4 gi = io::ld(&zz[0]), gf = io::ld(&zz[p.hid]), gg = io::ld(&zz[2 * p.hid]), go = io::ld(&zz[3 * p.hid]);
        f32x4 cp = {0.f, 0.f, 0.f, 0.f};
        if (p.c_prev) cp = *(const f32x4*)&p.c_prev[pix * p.hid + 4 * j4];
        f32x4 cn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // (bf16 storage: the state update uses the gate values AS STORED, the ones the backward will read)
            gi[e] = io::round(vad_sigmoid(gi[e])); gf[e] = io::round(vad_sigmoid(gf[e])); gg[e] = io::round(vad_tanh(gg[e])); go[e] = io::round(vad_sigmoid(go[e]));
            cn[e] = gf[e] * cp[e] + gi[e] * gg[e];
            hn[e] = go[e] * vad_tanh(cn[e]);
        }
        io::st(&zz[0], gi); io::st(&zz[p.hid], gf); io::st(&zz[2 * p.hid], gg); io::st(&zz[3 * p.hid], go);
        *(f32x4*)&p.c_out[pix * p.hid + 4 * j4] = cn;
        const long long b = pix / p.hw, q = pix - b * p.hw;
        if (p.h1) io::st((T*)p.h1 + b * p.h1_fs + q * p.h1_ps + 4 * j4, hn);
        if (p.h2) io::st((T*)p.h2 + b * p.h2_fs + q * p.h2_ps + 4 * j4, hn);
    }
}

struct LstmBwdP {   // gates / dh1 / dh2 / dz: fp32 or bf16 (the kernel's storage type); cell states and their gradients fp32
    const void* gates;        // [npix][4*hid] activated
    const float* c_prev;      // null = zeros
    const float* c;           // [npix][hid]
    const void* dh1; long long dh1_fs; int dh1_ps;    // gradient sources for h (either may be null)
    const void* dh2; long long dh2_fs; int dh2_ps;
    const float* dc_next;     // null = zeros
    void* dz;                 // [npix][4*hid]
    float* dc_prev;           // [npix][hid] (may alias dc_next)
    int hw, hid;
    long long total;
};

template <typename T>
__global__ __launch_bounds__(256) void lstm_gates_bwd_kernel(LstmBwdP p) {
    typedef vad_io4<T> io;
    const int hg = p.hid >> 2;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < p.total; idx += (long long)gridDim.x * 256) {
        const int j4 = (int)(idx % hg);
        const long long pix = idx / hg, b = pix / p.hw, q = pix - b * p.hw;
        const T* gz = (const T*)p.gates + pix * 4 * p.hid + 4 * j4;
        const f32x4 gi = io::ld(&gz[0]), gf = io::ld(&gz[p.hid]), gg = io::ld(&gz[2 * p.hid]), go = io::ld(&gz[3 * p.hid]);
        const f32x4 c = *(const f32x4*)&p.c[pix * p.hid + 4 * j4];
        f32x4 cp = {0.f, 0.f, 0.f, 0.f}, dh = cp, dcn = cp;
        if (p.c_prev) cp = *(const f32x4*)&p.c_prev[pix * p.hid + 4 * j4];
        if (p.dh1) dh += io::ld((const T*)p.dh1 + b * p.dh1_fs + q * p.dh1_ps + 4 * j4);
        if (p.dh2) dh += io::ld((const T*)p.dh2 + b * p.dh2_fs + q * p.dh2_ps + 4 * j4);
        if (p.dc_next) dcn = *(const f32x4*)&p.dc_next[pix * p.hid + 4 * j4];
        f32x4 di, df, dg, dgo, dcp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float tc = vad_tanh(c[e]);
            const float dc = dcn[e] + dh[e] * go[e] * (1.f - tc * tc);
            dgo[e] = dh[e] * tc * go[e] * (1.f - go[e]);
            di[e] = dc * gg[e] * gi[e] * (1.f - gi[e]);
            df[e] = dc * cp[e] * gf[e] * (1.f - gf[e]);
            dg[e] = dc * gi[e] * (1.f - gg[e] * gg[e]);
            dcp[e] = dc * gf[e];
        }
        T* dzp = (T*)p.dz + pix * 4 * p.hid + 4 * j4;
        io::st(&dzp[0], di); io::st(&dzp[p.hid], df); io::st(&dzp[2 * p.hid], dg); io::st(&dzp[3 * p.hid], dgo);
        *(f32x4*)&p.dc_prev[pix * p.hid + 4 * j4] = dcp;
    }
}

// ------------------------------------------------------------------------------------------------ weight gradients
// dW[tap][ci][col] = sum over (n, y, x) of A[n, y+dy-1, x+dx-1, ci] * G[n, y, x, col]   (TAPS == 9: 3x3, pad 1)
// dW[ci][col]      = sum over pixels of A[pix][ci] * G[pix][col]                        (TAPS == 1)
// One wave owns a 32 (ci) x 32*NT (col) tile of every tap and a slice of the image rows (split-K); each
// v_mfma_f32_32x32x2_f32 consumes two horizontally adjacent pixels: lane (li, lh) feeds A[pixel lh][ci li] and
// G[pixel lh][col li], both 128-byte coalesced rows of the NHWC tensors.  Partials go to ws[split][tap][ci][col].
// WGRAD_XCD: consecutive items share operands - the (ci tile, column group) pairs of ONE slice of image rows read the same rows
// of `a` and `g`, each tile pair re-reading them (a 32 x 32 tile per wave: ~144 FLOP per byte requested) - so consecutive
// LOGICAL blocks are placed on the same XCD (vad_xcd_remap): the re-reads then hit that XCD's L2 instead of going to the
// Infinity Cache / HBM once per XCD.
struct WgradP {       // a / g: fp32, or bf16 for the IO16 form of the bf16 kernel
    const void* a; const void* g; float* ws;
    int n, h, w, cin, ncols;
    int ci_tiles, col_groups, splits, rows_per_split;
    unsigned nitems;
};

template <int TAPS, int NT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradP p) {
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    unsigned item = __builtin_amdgcn_readfirstlane(vad_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));   // see WGRAD_XCD
    if (item >= p.nitems) return;
    const int ct = item % p.ci_tiles; item /= p.ci_tiles;
    const int cgp = item % p.col_groups;
    const int split = item / p.col_groups;
    const int H = p.h, W = p.w, total_rows = p.n * H;
    const int r0 = split * p.rows_per_split, r1 = (r0 + p.rows_per_split < total_rows) ? r0 + p.rows_per_split : total_rows;
    const unsigned a_bytes = (unsigned)(H * W) * (unsigned)p.cin * 4u, g_bytes = (unsigned)(H * W) * (unsigned)p.ncols * 4u;
    f32x16 acc[TAPS][NT];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nt][r] = 0.f;

    // software pipeline over (row, pixel pair): the operands of the next pair are in flight while the current pair's
    // TAPS*NT MFMAs run.  Branch-free body: (lrow, lx) is the next pair to LOAD; past the end of the slice every offset is
    // out of range, the loads return 0 and the surplus MFMAs add nothing.
    // Address work is split by rate (PMC: the first version issued 1.4 scalar + 1.0 vector instructions per MFMA and kept
    // the matrix pipe 48 % busy): per ROW three descriptors (zero-sized for rows outside the image) and TAPS scalar tap
    // bases; per PAIR two scalar increments, three lane masks (left / centre / right column validity) and one scalar add
    // per load.  Lane parts of the offsets are loop invariants.
    // Every scalar offset is >= 0 and every lane that passes the range check addresses a pixel inside the frame whether or
    // not the hardware includes the scalar offset in that check: the left-column tap (dx = -1) takes its scalar base at
    // pixel max(x-2, 0) with the lane part one pixel further (at x == 0 only the upper lane half exists: pixel 0).
    constexpr int NR = TAPS == 9 ? 3 : 1;
    float av[2][TAPS], bv[2][NT];
    int lrow = r0, lx = 0;
    const __amdgpu_buffer_rsrc_t rzero = vad_rsrc(p.a, 0);
    __amdgpu_buffer_rsrc_t rrow[NR], rg = vad_rsrc(p.g, 0);
    unsigned rbase[NR], sx = 0, sxm = 0, sgx = 0;     // scalar: row bases of the load row, x offsets (sxm: max(x-2,0))
#pragma unroll
    for (int i = 0; i < NR; ++i) { rrow[i] = rzero; rbase[i] = 0; }
    const unsigned pix_a = (unsigned)(p.cin * 4);
    const unsigned lane_a = (unsigned)((lh * p.cin + ct * 32 + li) * 4), lane_b = (unsigned)((lh * p.ncols + cgp * NT * 32 + li) * 4);
    const unsigned lane_a_p1 = lane_a + pix_a;                              // one pixel to the right of the lane's own
    const unsigned lane_a_x0 = lh ? lane_a - pix_a : VAD_OOB;                // dx = -1 at x == 0: lane half 1 reads pixel 0
    const unsigned step_a = 2 * pix_a, step_b = (unsigned)(2 * p.ncols * 4);
#define WG_LOAD(buf)                                                                                                   \
    {                                                                                                                  \
        const bool valid = lrow < r1;                                                                                  \
        if (valid && lx == 0) {                                                                                        \
            const int n_ = lrow / H, ly = lrow - n_ * H;                                                               \
            const float* fa = (const float*)p.a + (size_t)n_ * H * W * p.cin;                                          \
            rg = vad_rsrc((const float*)p.g + (size_t)n_ * H * W * p.ncols, g_bytes);                                  \
            _Pragma("unroll") for (int i = 0; i < NR; ++i) {                                                           \
                const int yy = ly + (NR == 3 ? i - 1 : 0);                                                             \
                const bool rok = yy >= 0 && yy < H;                                                                    \
                rrow[i] = rok ? vad_rsrc(fa, a_bytes) : rzero;                                                         \
                rbase[i] = rok ? (unsigned)(yy * W) * pix_a : 0u;                                                      \
            }                                                                                                          \
            sx = 0;                                                                                                    \
            sxm = 0;                                                                                                   \
            sgx = (unsigned)(ly * W * p.ncols * 4);                                                                    \
        }                                                                                                              \
        const int px = lx + lh;                                                                                        \
        const bool pok = valid && px < W;                                                                              \
        const unsigned vb = pok ? lane_b : VAD_OOB;                                                                    \
        unsigned va[NR];                                                                                               \
        if (NR == 3) {                                                                                                 \
            va[0] = pok ? (lx >= 2 ? lane_a_p1 : lane_a_x0) : VAD_OOB;                                                 \
            va[1] = pok ? lane_a : VAD_OOB;                                                                            \
            va[2] = (pok && px + 1 < W) ? lane_a_p1 : VAD_OOB;                                                         \
        } else {                                                                                                       \
            va[0] = pok ? lane_a : VAD_OOB;                                                                            \
        }                                                                                                              \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) bv[buf][nt] = vad_bload1(rg, vb, sgx + (unsigned)(nt * 128)); \
        _Pragma("unroll") for (int t = 0; t < TAPS; ++t)                                                               \
            av[buf][t] = vad_bload1(rrow[t / NR], va[t % NR], rbase[t / NR] + ((NR == 3 && t % NR == 0) ? sxm : sx));   \
        sxm = lx >= 2 ? sxm + step_a : (lx == 0 ? 0u : sxm);                                                           \
        sx += step_a;                                                                                                  \
        sgx += step_b;                                                                                                 \
        lx += 2;                                                                                                       \
        if (lx >= W) { lx = 0; ++lrow; }                                                                               \
    }
#define WG_MFMA(buf)                                                                         \
    {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        _Pragma("unroll") for (int t = 0; t < TAPS; ++t)                                     \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) acc[t][nt] = MFMA32(av[buf][t], bv[buf][nt], acc[t][nt]); \
    }
    const int npairs = (r1 - r0) * ((W + 1) / 2);
    WG_LOAD(0);
    for (int i = 0; i < npairs; i += 2) {
        WG_LOAD(1);
        WG_MFMA(0);
        WG_LOAD(0);
        WG_MFMA(1);
    }
#undef WG_MFMA
#undef WG_LOAD
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                p.ws[(((size_t)split * TAPS + t) * p.cin + ci) * p.ncols + (cgp * NT + nt) * 32 + li] = acc[t][nt][r];
            }
}

// bf16 form of the same GEMM (VAD_PREC_BF16, BASELINE configs[4]'s dtype): v_mfma_f32_32x32x16_bf16 consumes 16 pixels per
// instruction - lane (li, kb) supplies the 8 consecutive pixels x0 + 8 kb + (0..7) of ITS channel / column, read as 8 (3x3:
// 10, one pixel of halo either side) dword loads whose 32 lanes cover one 128-byte NHWC row each, rounded to bf16 (nearest
// even) and packed on the fly; the three dx taps of a row are the element windows [0,8) [1,9) [2,10) of those 10 values
// (even- and odd-aligned pair packings).  fp32 accumulation, same split-K partials and reduction as the exact kernel.  Per 16
// pixels a wave issues 38 loads and 9 MFMAs of 32 cycles (the exact kernel: 9 x 8 MFMAs of 64 cycles): it is bound by the
// vector L1, ~8x the exact kernel's rate.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned wg_pk(float a, float b) { return __builtin_bit_cast(unsigned, wg_bf16x2{(__bf16)a, (__bf16)b}); }
__device__ __forceinline__ wg_bf16x8 wg_frag(unsigned a, unsigned b, unsigned c, unsigned d) { return __builtin_bit_cast(wg_bf16x8, u32x4{a, b, c, d}); }

// IO16: both operands are ALREADY bf16 in memory (VAD_PREC_BF16S): the same access pattern with 16-bit loads (half the bytes
// through L1) and a pair of values is packed with one v_perm / v_lshl_or instead of a conversion.
// What bounds it (round 3, measured on the five 3x3 layers of the bf16 training step, all at ~450 TFLOP/s = 0.18 of the bf16
// peak whatever their shape): the NUMBER of load instructions - 38 per 9 MFMAs, each a 64-lane 2- or 4-byte gather through
// the texture addresser.  Not their bytes (bf16 tensors: the same time as fp32 tensors), not the VALU work around them (lane
// offsets as loop invariants with the pixel group in the scalar offset cut it from ~150 to ~50 instructions per group: no
// change), not latency (the two-deep pipeline below: -8 %), not L2 misses (XCD-aware item order: no change).  The next
// step is a workgroup-shared LDS tile written transposed ([channel][pixel]) from 16-byte loads, so that a fragment is one
// ds_read_b128: not built.
template <int TAPS, int NT, int IO16>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(WgradP p) {
    constexpr unsigned ES = IO16 ? 2u : 4u;
    auto LD = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) -> unsigned {     // raw element: fp32 bits or a zero-extended bf16
        if constexpr (IO16) return (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, 0);
        else return __builtin_bit_cast(unsigned, vad_bload1(r, voff, soff));
    };
    auto PK = [](unsigned a, unsigned b) -> unsigned {
        if constexpr (IO16) return a | (b << 16);
        else return wg_pk(__uint_as_float(a), __uint_as_float(b));
    };
    const int lane = threadIdx.x & 63, li = lane & 31, kb = lane >> 5;
    unsigned item = __builtin_amdgcn_readfirstlane(vad_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));   // see WGRAD_XCD
    if (item >= p.nitems) return;
    const int ct = item % p.ci_tiles; item /= p.ci_tiles;
    const int cgp = item % p.col_groups;
    const int split = item / p.col_groups;
    const int H = p.h, W = p.w, total_rows = p.n * H;
    const int r0 = split * p.rows_per_split, r1 = (r0 + p.rows_per_split < total_rows) ? r0 + p.rows_per_split : total_rows;
    const unsigned a_bytes = (unsigned)(H * W) * (unsigned)p.cin * ES, g_bytes = (unsigned)(H * W) * (unsigned)p.ncols * ES;
    f32x16 acc[TAPS][NT];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][nt][r] = 0.f;
    constexpr int NR = TAPS == 9 ? 3 : 1, HALO = TAPS == 9 ? 1 : 0, NE = 8 + 2 * HALO;
    const unsigned pix_a = (unsigned)p.cin * ES, pix_g = (unsigned)p.ncols * ES;
    const unsigned lane_a = (unsigned)(ct * 32 + li) * ES, lane_g = (unsigned)(cgp * NT * 32 + li) * ES;
    const __amdgpu_buffer_rsrc_t rzero = vad_rsrc(p.a, 0);
    // Software pipeline over (row, 16-pixel group): the 38 loads of the NEXT group are in flight while the current group is
    // packed and multiplied (two register sets).  Without it every group paid a full memory round trip in front of its 9
    // MFMAs - with two waves per SIMD the matrix pipe was ~13 % busy whatever the operand width (fp32 or bf16 tensors: 2.19 /
    // 2.18 ms per training step).  The load stage keeps its own position (lrow, lx) and row descriptors; groups past the end
    // of the slice load through out-of-range offsets (zeros) and the surplus MFMAs add nothing.
    const int groups_per_row = (W + 15) / 16;
    int lrow = r0, lx = 0;
    __amdgpu_buffer_rsrc_t rrow[NR], rg = rzero;
    unsigned rbase[NR], gbase = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) { rrow[i] = rzero; rbase[i] = 0; }
    auto LOAD = [&](unsigned (&gv)[NT][8], unsigned (&av)[NR][NE]) {
        // (no load below sits under a condition that involves a uniform value: hipcc turns those into branches around the
        // loads and joins the paths with vmcnt(0) - past the end of the slice the DESCRIPTORS become zero-sized instead)
        if (lx == 0 && lrow >= r1) {
            rg = rzero;
#pragma unroll
            for (int i = 0; i < NR; ++i) rrow[i] = rzero;
        } else if (lx == 0) {
            const int n_ = lrow / H, ly = lrow - n_ * H;
            const char* fa = (const char*)p.a + (size_t)n_ * H * W * p.cin * ES;
            rg = vad_rsrc((const char*)p.g + (size_t)n_ * H * W * p.ncols * ES, g_bytes);
            gbase = (unsigned)(ly * W) * pix_g;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int yy = ly + (NR == 3 ? i - 1 : 0);
                const bool rok = yy >= 0 && yy < H;
                rrow[i] = rok ? vad_rsrc(fa, a_bytes) : rzero;       // rows above / below the image: zero-sized descriptor -> zeros
                rbase[i] = rok ? (unsigned)(yy * W) * pix_a : 0u;
            }
        }
        {
            const int px0 = lx + 8 * kb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int x = px0 + e;
                const unsigned off = x < W ? lane_g + (unsigned)x * pix_g : VAD_OOB;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) gv[nt][e] = LD(rg, off, gbase + (unsigned)nt * 32u * ES);
            }
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int x = px0 + e - HALO;
                    av[i][e] = LD(rrow[i], (unsigned)x < (unsigned)W ? lane_a + (unsigned)x * pix_a : VAD_OOB, rbase[i]);
                }
        }
        lx += 16;
        if (lx >= W) { lx = 0; ++lrow; }
    };
    auto COMPUTE = [&](const unsigned (&gv)[NT][8], const unsigned (&av)[NR][NE]) {
        wg_bf16x8 gb[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            gb[nt] = wg_frag(PK(gv[nt][0], gv[nt][1]), PK(gv[nt][2], gv[nt][3]), PK(gv[nt][4], gv[nt][5]), PK(gv[nt][6], gv[nt][7]));
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if constexpr (TAPS == 9) {
                unsigned pe[5], po[4];
#pragma unroll
                for (int k = 0; k < 5; ++k) pe[k] = PK(av[i][2 * k], av[i][2 * k + 1]);
#pragma unroll
                for (int k = 0; k < 4; ++k) po[k] = PK(av[i][2 * k + 1], av[i][2 * k + 2]);
                const wg_bf16x8 f0 = wg_frag(pe[0], pe[1], pe[2], pe[3]);      // dx = 0: pixels x-1 .. x+6
                const wg_bf16x8 f1 = wg_frag(po[0], po[1], po[2], po[3]);      // dx = 1: pixels x   .. x+7
                const wg_bf16x8 f2 = wg_frag(pe[1], pe[2], pe[3], pe[4]);      // dx = 2: pixels x+1 .. x+8
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[i * 3 + 0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, gb[nt], acc[i * 3 + 0][nt], 0, 0, 0);
                    acc[i * 3 + 1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, gb[nt], acc[i * 3 + 1][nt], 0, 0, 0);
                    acc[i * 3 + 2][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, gb[nt], acc[i * 3 + 2][nt], 0, 0, 0);
                }
            } else {
                const wg_bf16x8 f = wg_frag(PK(av[0][0], av[0][1]), PK(av[0][2], av[0][3]), PK(av[0][4], av[0][5]), PK(av[0][6], av[0][7]));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, gb[nt], acc[0][nt], 0, 0, 0);
            }
        }
    };
    const int ngroups = (r1 - r0) * groups_per_row;
    {
    unsigned gv0[NT][8], av0[NR][NE], gv1[NT][8], av1[NR][NE];
    LOAD(gv0, av0);
    for (int it = 0; it < ngroups; it += 2) {
        LOAD(gv1, av1);
        __builtin_amdgcn_sched_barrier(0);
        COMPUTE(gv0, av0);
        __builtin_amdgcn_sched_barrier(0);
        LOAD(gv0, av0);
        __builtin_amdgcn_sched_barrier(0);
        COMPUTE(gv1, av1);
        __builtin_amdgcn_sched_barrier(0);
    }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
                p.ws[(((size_t)split * TAPS + t) * p.cin + ci) * p.ncols + (cgp * NT + nt) * 32 + li] = acc[t][nt][r];
            }
}

// Split-fp16 form (VAD_PREC_SPLIT, round 4): the same lane / pixel mapping as the bf16 kernel above on fp32 tensors, each operand
// value split into hi = fp16(v) and lo = fp16((v - hi) 2^11) (vad_split, conv_pkernel.h) as it is packed:
//   dw += ah gh + (ah gl + al gh) 2^-11      22-bit products, fp32 accumulation, three v_mfma_f32_32x32x16_f16 per 16 pixels
// with the correction terms in accumulators of their own (they carry the 2^11) that the epilogue folds in.  Twice the
// accumulators: a 3x3 item is ONE kernel row of a 32 x 32 tile (3 taps: 6 accumulator tiles; 18 loads per 9 MFMAs - the
// gradient row is re-read per kernel row), a 1x1 item is the bf16 kernel's.  The gradient operand arrives scaled into the fp16
// range by the step (train_step.hip: grad_mul), the activations are O(1).
typedef _Float16 wg_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void wg_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const wg_f16x2 h = {(_Float16)a, (_Float16)b};
    const wg_f16x2 l = {(_Float16)((a - (float)h[0]) * 2048.0f), (_Float16)((b - (float)h[1]) * 2048.0f)};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// The same split in five instructions per PAIR: one packed conversion, the two residuals a - hi straight from the packed halves
// (v_fma_mix_f32: an f16 source widened inside the FMA), and scale + conversion + packing of the lo halves in
// v_fma_mixlo_f16 / v_fma_mixhi_f16.  Every step is exact or the single rounding of the C form above (a - hi and r * 2048 are
// exact in fp32): bit-identical to wg_split2, which the per-lane kernel keeps - tests/test_hip_train_ops.py compares the two.
__device__ __forceinline__ void wg_split2_fast(float a, float b, unsigned& hi, unsigned& lo) {
    unsigned h, l;
    float ra, rb;
    const float k2048 = 2048.0f;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(h), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(h), "v"(b));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(l) : "v"(ra), "s"(k2048));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(l) : "v"(rb), "s"(k2048));
    hi = h;
    lo = l;
}
__device__ __forceinline__ wg_f16x8 wg_hfrag(unsigned a, unsigned b, unsigned c, unsigned d) { return __builtin_bit_cast(wg_f16x8, u32x4{a, b, c, d}); }

template <int TAPS, int NT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_split_kernel(WgradP p) {
    constexpr int ND = TAPS == 9 ? 3 : 1, HALO = TAPS == 9 ? 1 : 0, NE = 8 + 2 * HALO, NPASS = TAPS == 9 ? 3 : 1;
    const int lane = threadIdx.x & 63, li = lane & 31, kb = lane >> 5;
    unsigned item = __builtin_amdgcn_readfirstlane(vad_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));   // see WGRAD_XCD
    if (item >= p.nitems) return;
    const int ct = item % p.ci_tiles; item /= p.ci_tiles;
    const int cgp = item % p.col_groups; item /= p.col_groups;
    const int pass = item % NPASS;
    const int split = item / NPASS;
    const int H = p.h, W = p.w, total_rows = p.n * H;
    const int r0 = split * p.rows_per_split, r1 = (r0 + p.rows_per_split < total_rows) ? r0 + p.rows_per_split : total_rows;
    const unsigned a_bytes = (unsigned)(H * W) * (unsigned)p.cin * 4u, g_bytes = (unsigned)(H * W) * (unsigned)p.ncols * 4u;
    f32x16 acc[ND][NT], cor[ND][NT];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[d][nt][r] = 0.f; cor[d][nt][r] = 0.f; }
    const unsigned pix_a = (unsigned)p.cin * 4u, pix_g = (unsigned)p.ncols * 4u;
    const unsigned lane_a = (unsigned)(ct * 32 + li) * 4u, lane_g = (unsigned)(cgp * NT * 32 + li) * 4u;
    const __amdgpu_buffer_rsrc_t rzero = vad_rsrc(p.a, 0);
    const int dyk = TAPS == 9 ? pass - 1 : 0;
    // two-deep software pipeline over (row, 16-pixel group) as in the bf16 kernel: the next group's loads are in flight while
    // this one is split and multiplied; past the end of the slice - and on rows whose kernel row falls outside the image -
    // the descriptors are zero-sized (zeros, no branch around a load)
    const int groups_per_row = (W + 15) / 16;
    int lrow = r0, lx = 0;
    __amdgpu_buffer_rsrc_t ra = rzero, rg = rzero;
    unsigned abase = 0, gbase = 0;
    auto LOAD = [&](float (&gv)[NT][8], float (&av)[NE]) {
        if (lx == 0 && lrow >= r1) {
            rg = rzero; ra = rzero;
        } else if (lx == 0) {
            const int n_ = lrow / H, ly = lrow - n_ * H, yy = ly + dyk;
            const bool rok = yy >= 0 && yy < H;
            ra = rok ? vad_rsrc((const char*)p.a + (size_t)n_ * H * W * p.cin * 4u, a_bytes) : rzero;
            rg = rok ? vad_rsrc((const char*)p.g + (size_t)n_ * H * W * p.ncols * 4u, g_bytes) : rzero;
            abase = rok ? (unsigned)(yy * W) * pix_a : 0u;
            gbase = rok ? (unsigned)(ly * W) * pix_g : 0u;
        }
        const int px0 = lx + 8 * kb;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int x = px0 + e;
            const unsigned off = x < W ? lane_g + (unsigned)x * pix_g : VAD_OOB;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) gv[nt][e] = vad_bload1(rg, off, gbase + (unsigned)nt * 128u);
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int x = px0 + e - HALO;
            av[e] = vad_bload1(ra, (unsigned)x < (unsigned)W ? lane_a + (unsigned)x * pix_a : VAD_OOB, abase);
        }
        lx += 16;
        if (lx >= W) { lx = 0; ++lrow; }
    };
    auto COMPUTE = [&](const float (&gv)[NT][8], const float (&av)[NE]) {
        wg_f16x8 gh[NT], gl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            unsigned h[4], l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wg_split2(gv[nt][2 * k], gv[nt][2 * k + 1], h[k], l[k]);
            gh[nt] = wg_hfrag(h[0], h[1], h[2], h[3]);
            gl[nt] = wg_hfrag(l[0], l[1], l[2], l[3]);
        }
        unsigned eh[NE / 2], el[NE / 2];          // even-aligned pairs (elements 2k, 2k+1) of the 8 (10) pixels, hi and lo
#pragma unroll
        for (int k = 0; k < NE / 2; ++k) wg_split2(av[2 * k], av[2 * k + 1], eh[k], el[k]);
        auto mma = [&](int d, wg_f16x8 fh, wg_f16x8 fl) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[d][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, gh[nt], acc[d][nt], 0, 0, 0);
                cor[d][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, gl[nt], cor[d][nt], 0, 0, 0);
                cor[d][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, gh[nt], cor[d][nt], 0, 0, 0);
            }
        };
        if constexpr (TAPS == 9) {
            unsigned oh[4], ol[4];                // odd-aligned pairs (2k+1, 2k+2): the halves of two neighbouring even pairs
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                oh[k] = __builtin_amdgcn_alignbit(eh[k + 1], eh[k], 16);
                ol[k] = __builtin_amdgcn_alignbit(el[k + 1], el[k], 16);
            }
            mma(0, wg_hfrag(eh[0], eh[1], eh[2], eh[3]), wg_hfrag(el[0], el[1], el[2], el[3]));      // dx = 0: pixels x-1 .. x+6
            mma(1, wg_hfrag(oh[0], oh[1], oh[2], oh[3]), wg_hfrag(ol[0], ol[1], ol[2], ol[3]));      // dx = 1: pixels x   .. x+7
            mma(2, wg_hfrag(eh[1], eh[2], eh[3], eh[4]), wg_hfrag(el[1], el[2], el[3], el[4]));      // dx = 2: pixels x+1 .. x+8
        } else {
            mma(0, wg_hfrag(eh[0], eh[1], eh[2], eh[3]), wg_hfrag(el[0], el[1], el[2], el[3]));
        }
    };
    const int ngroups = (r1 - r0) * groups_per_row;
    {
    float gv0[NT][8], av0[NE], gv1[NT][8], av1[NE];
    LOAD(gv0, av0);
    for (int it = 0; it < ngroups; it += 2) {
        LOAD(gv1, av1);
        __builtin_amdgcn_sched_barrier(0);
        COMPUTE(gv0, av0);
        __builtin_amdgcn_sched_barrier(0);
        LOAD(gv0, av0);
        __builtin_amdgcn_sched_barrier(0);
        COMPUTE(gv1, av1);
        __builtin_amdgcn_sched_barrier(0);
    }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int tap = (TAPS == 9 ? 3 * pass : 0) + d;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
                p.ws[(((size_t)split * TAPS + tap) * p.cin + ci) * p.ncols + (cgp * NT + nt) * 32 + li] = fmaf(cor[d][nt][r], 1.0f / 2048.0f, acc[d][nt][r]);
            }
    }
}

// The split-fp16 GEMM with its operands staged ONCE per work-group, TRANSPOSED, through LDS.  The kernel above re-reads every
// operand row per 32 x 32 wave tile from L2 (18 64-lane gathers per 9 MFMAs: 1.4 ms on the 193-GFLOP layers of the 32-clip step,
// ~6.5 TB/s of L2 -> L1 traffic, 140 TFLOP/s) and splits every value once per wave.  Here (32 WM) x (32 WN) x PS waves share a
// tile of ONE kernel row: per group of GP pixels a thread fetches two horizontally adjacent pixels x 4 channels (two 16-byte
// loads), splits them once, and writes the (pixel, pixel + 1) fp16 pairs of each channel as one dword into [channel][pixel]
// planes (hi and lo; rows of PITCH bytes).  A lane's MFMA fragment - 8 consecutive pixels of ITS channel / column - is then
// one ds_read_b128 (3x3: + one dword for the 10-pixel window; the dx = 1 window is four v_alignbit of neighbours).  The next
// group's loads are in flight during the MFMAs (two LDS buffers, one barrier per group).  PS > 1: wave groups take alternate
// 16-pixel sub-groups of a staged group (a split-K factor inside the work-group, partial slot split * PS + ph) so that a
// 32-channel layer still has four waves per staged tile.  Same products and the same hi / lo arithmetic as the kernel above.
template <int TAPS, int WM, int WN, int PS, int GP>
__global__ __launch_bounds__(64 * WM * WN * PS, 3) void conv_wgrad_split_lds_kernel(WgradP p) {
    static_assert(GP == 16 || GP == 32, "groups of 16 or 32 pixels");
    static_assert(GP / 16 >= PS, "pixel split needs a 16-pixel sub-group per wave group");
    constexpr int ND = TAPS == 9 ? 3 : 1, HALO = TAPS == 9 ? 1 : 0, NPASS = TAPS == 9 ? 3 : 1;
    constexpr int NTH = 64 * WM * WN * PS;
    constexpr int CA = 32 * WM, CG = 32 * WN;                        // channels / columns of the work-group tile
    constexpr int NEA = GP + 2 * HALO;                               // A elements (pixels with halo) per group: even
    constexpr int PAIRS_A = NEA / 2, PAIRS_G = GP / 2;
    constexpr int PITCH = GP == 32 ? 80 : 48;                        // bytes per [channel] row: >= 2 NEA, 16-byte multiple, b128 reads of 16 lanes hit 64 distinct banks
    constexpr int NCA = CA / 4, NCG = CG / 4;                        // 4-channel chunks (one 16-byte load per pixel)
    constexpr int UA = PAIRS_A * NCA, UG = PAIRS_G * NCG;            // staging units per group
    constexpr int JA = (UA + NTH - 1) / NTH, JG = (UG + NTH - 1) / NTH;
    constexpr int PLANE_A = CA * PITCH, PLANE_G = CG * PITCH;
    constexpr int BUF = 2 * PLANE_A + 2 * PLANE_G;                   // [A hi][A lo][G hi][G lo]
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kb = lane >> 5;
    const int ph = wave / (WM * WN), wt = wave % (WM * WN), wm = wt / WN, wn = wt % WN;
    unsigned item = vad_xcd_remap(blockIdx.x, gridDim.x);             // see WGRAD_XCD
    const int ct = item % p.ci_tiles; item /= p.ci_tiles;
    const int cgp = item % p.col_groups; item /= p.col_groups;
    const int pass = item % NPASS;
    const int split = item / NPASS;
    const int H = p.h, W = p.w, total_rows = p.n * H;
    const int r0 = split * p.rows_per_split, r1 = (r0 + p.rows_per_split < total_rows) ? r0 + p.rows_per_split : total_rows;
    const int dy = TAPS == 9 ? pass - 1 : 0;
    const unsigned a_bytes = (unsigned)(H * W) * (unsigned)p.cin * 4u, g_bytes = (unsigned)(H * W) * (unsigned)p.ncols * 4u;
    const unsigned pix_a = (unsigned)p.cin * 4u, pix_g = (unsigned)p.ncols * 4u;
    f32x16 acc[ND], cor[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[d][r] = 0.f; cor[d][r] = 0.f; }

    auto row_ok = [&](int row) { const int ly = row % H + dy; return ly >= 0 && ly < H; };
    int row = r0, lx = 0;
    while (row < r1 && !row_ok(row)) ++row;
    // Per-thread invariants of the staging units (unit u = tid + NTH j: 4-channel chunk c = u % NC, pixel pair pr = u / NC): the
    // lane part of the load offsets (element index times the pixel pitch; the group's position goes into the scalar offset,
    // with the descriptor's base one halo pixel BEFORE the frame so that it is never negative), the first element's index for
    // the range check, and the LDS byte offset of the pair.
    unsigned voa[JA], vog[JG];
    int ea[JA], eg[JG], wa[JA], wgo[JG];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int u = tid + NTH * j, c = u % NCA, pr = u / NCA;
        voa[j] = (unsigned)(2 * pr) * pix_a + (unsigned)c * 16u;
        ea[j] = u < UA ? 2 * pr - HALO : (1 << 30);                  // (no such unit: never in range)
        wa[j] = (4 * c) * PITCH + 4 * pr;
    }
#pragma unroll
    for (int j = 0; j < JG; ++j) {
        const int u = tid + NTH * j, c = u % NCG, pr = u / NCG;
        vog[j] = (unsigned)(2 * pr) * pix_g + (unsigned)c * 16u;
        eg[j] = u < UG ? 2 * pr : (1 << 30);
        wgo[j] = 2 * PLANE_A + (4 * c) * PITCH + 4 * pr;
    }
    f32x4 sa[JA][2], sg[JG][2];                      // staging: two adjacent pixels x 4 channels per unit
    auto fetch = [&](int frow, int flx) {
        const int n_ = frow / H, ly = frow - n_ * H;
        const __amdgpu_buffer_rsrc_t ra = vad_rsrc((const char*)p.a + (size_t)n_ * H * W * p.cin * 4u - (size_t)HALO * pix_a, a_bytes + HALO * pix_a);
        const __amdgpu_buffer_rsrc_t rg = vad_rsrc((const char*)p.g + (size_t)n_ * H * W * p.ncols * 4u, g_bytes);
        const unsigned abase = (unsigned)((ly + dy) * W + flx) * pix_a + (unsigned)(ct * CA) * 4u;
        const unsigned gbase = (unsigned)(ly * W + flx) * pix_g + (unsigned)(cgp * CG) * 4u;
#pragma unroll
        for (int j = 0; j < JA; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                sa[j][q] = vad_bload4(ra, (unsigned)(ea[j] + q + flx) < (unsigned)W ? voa[j] + (unsigned)q * pix_a : VAD_OOB, abase);
#pragma unroll
        for (int j = 0; j < JG; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                sg[j][q] = vad_bload4(rg, (unsigned)(eg[j] + q + flx) < (unsigned)W ? vog[j] + (unsigned)q * pix_g : VAD_OOB, gbase);
    };
    auto stash = [&](int b) {
        unsigned char* base = lds + b * BUF;
#pragma unroll
        for (int j = 0; j < JA; ++j)
            if (ea[j] < (1 << 30)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned hi, lo;
                    wg_split2_fast(sa[j][0][e], sa[j][1][e], hi, lo);
                    *(unsigned*)(base + wa[j] + e * PITCH) = hi;
                    *(unsigned*)(base + wa[j] + e * PITCH + PLANE_A) = lo;
                }
            }
#pragma unroll
        for (int j = 0; j < JG; ++j)
            if (eg[j] < (1 << 30)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned hi, lo;
                    wg_split2_fast(sg[j][0][e], sg[j][1][e], hi, lo);
                    *(unsigned*)(base + wgo[j] + e * PITCH) = hi;
                    *(unsigned*)(base + wgo[j] + e * PITCH + PLANE_G) = lo;
                }
            }
    };
    auto compute = [&](int b) {
        const unsigned char* Ah = lds + b * BUF + (wm * 32 + li) * PITCH;
        const unsigned char* Gh = lds + b * BUF + 2 * PLANE_A + (wn * 32 + li) * PITCH;
#pragma unroll
        for (int sub0 = 0; sub0 < GP / 16 / PS; ++sub0) {
            const int sub = PS > 1 ? sub0 * PS + ph : sub0;
            const int off = 2 * (16 * sub + 8 * kb);                       // byte offset of element px0 (= pixel lx + px0 - HALO)
            const u32x4 gh4 = *(const u32x4*)(Gh + off), gl4 = *(const u32x4*)(Gh + PLANE_G + off);
            const wg_f16x8 gh = __builtin_bit_cast(wg_f16x8, gh4), gl = __builtin_bit_cast(wg_f16x8, gl4);
            const u32x4 ah4 = *(const u32x4*)(Ah + off), al4 = *(const u32x4*)(Ah + PLANE_A + off);
            wg_f16x8 fh[ND], fl[ND];
            fh[0] = __builtin_bit_cast(wg_f16x8, ah4); fl[0] = __builtin_bit_cast(wg_f16x8, al4);                   // elements 0..7
            if constexpr (TAPS == 9) {
                const unsigned ah5 = *(const unsigned*)(Ah + off + 16), al5 = *(const unsigned*)(Ah + PLANE_A + off + 16);
                fh[1] = wg_hfrag(__builtin_amdgcn_alignbit(ah4[1], ah4[0], 16), __builtin_amdgcn_alignbit(ah4[2], ah4[1], 16),
                                 __builtin_amdgcn_alignbit(ah4[3], ah4[2], 16), __builtin_amdgcn_alignbit(ah5, ah4[3], 16));      // 1..8
                fl[1] = wg_hfrag(__builtin_amdgcn_alignbit(al4[1], al4[0], 16), __builtin_amdgcn_alignbit(al4[2], al4[1], 16),
                                 __builtin_amdgcn_alignbit(al4[3], al4[2], 16), __builtin_amdgcn_alignbit(al5, al4[3], 16));
                fh[2] = wg_hfrag(ah4[1], ah4[2], ah4[3], ah5); fl[2] = wg_hfrag(al4[1], al4[2], al4[3], al5);       // 2..9
            }
            // (the two correction products of a tap write the same accumulator: issued a tap apart, never back to back)
#pragma unroll
            for (int d = 0; d < ND; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[d], gh, acc[d], 0, 0, 0);
#pragma unroll
            for (int d = 0; d < ND; ++d) cor[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[d], gl, cor[d], 0, 0, 0);
#pragma unroll
            for (int d = 0; d < ND; ++d) cor[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[d], gh, cor[d], 0, 0, 0);
        }
    };
    if (row < r1) {                                   // (uniform over the work-group: every barrier below is reached by all waves)
        fetch(row, lx);
        stash(0);
        __syncthreads();
        int b = 0;
        while (true) {
            int nrow = row, nlx = lx + GP;
            if (nlx >= W) { nlx = 0; ++nrow; while (nrow < r1 && !row_ok(nrow)) ++nrow; }
            const bool more = nrow < r1;
            if (more) fetch(nrow, nlx);               // in flight during this group's MFMAs
            compute(b);
            if (!more) break;
            stash(b ^ 1);                             // the other buffer: its readers passed the last barrier
            __syncthreads();
            b ^= 1; row = nrow; lx = nlx;
        }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int tap = (TAPS == 9 ? 3 * pass : 0) + d;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ct * CA + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
            p.ws[(((size_t)(split * PS + ph) * TAPS + tap) * p.cin + ci) * p.ncols + cgp * CG + wn * 32 + li] = fmaf(cor[d][r], 1.0f / 2048.0f, acc[d][r]);
        }
    }
}

// ROW-RING form of the 3x3 weight gradient (round 4): every operand row is staged ONCE per work-group.
// The LDS-staged kernels above and below take one kernel row per work item, so a tile's three kernel rows stage the gradient
// row three times and the activation rows three times: on bf16 tensors that is 1.5 GB of L2 requests for the 0.5 GB of a
// 64 -> 128 @ 64x64 layer, 4.2 TB/s for 360 us - the fabric, not the matrix pipe (0.21 of its peak).  Here a work-group owns a
// (32 WM) x (32 WN) tile of ALL nine taps and walks DOWN a column strip of GP pixels, frame after frame: per output row it
// stages one new activation row (GP + 2 pixels) into a ring of four and one gradient row into a double buffer, transposed
// ([channel][pixel], a lane's 8-pixel fragment is one ds_read_b128 as in conv_wgrad_split_lds_kernel), and multiplies the three
// activation rows in the ring against the gradient row.  Between frames the stream of activation rows carries one zero row (the
// padding below one frame and above the next), which costs one idle step per frame.
// FMT 0: bf16 tensors, v_mfma_f32_32x32x16_bf16, a wave holds the nine taps of its 32 x 32 tile (144 accumulator registers).
// FMT 1: fp32 tensors in split-fp16 arithmetic (hi / lo planes, three MFMAs per product): two accumulator sets per tap, so the
// three kernel rows of a tile go to three wave groups (96 registers each) that share the staged rows.
// FMT 2: fp32 tensors in exact fp32 (v_mfma_f32_32x32x2_f32): the bf16 form's tiling with fp32 planes and 16-pixel strips.
struct WgradRingP {
    const void* a; const void* g; float* ws;
    int n, h, w, cin, ncols;
    int ci_tiles, col_groups, strips, frames_per_split;
};

template <int FMT, int WM, int WN, int GP>
__global__ __launch_bounds__(64 * WM * WN * (FMT == 1 ? 3 : 1), FMT == 1 ? 3 : 2) void conv_wgrad_ring_kernel(WgradRingP p) {
    static_assert(GP == 16 || GP == 32, "strips of 16 or 32 pixels");
    constexpr bool SPLIT = FMT == 1, F32 = FMT == 2;
    static_assert(!F32 || GP == 16, "exact fp32: 16-pixel strips (the planes of 32 would not fit 64 KB of LDS)");
    constexpr int ES = FMT ? 4 : 2, CPL = 16 / ES;                   // element bytes in memory, channels per 16-byte load
    constexpr int EB = F32 ? 4 : 2;                                  // element bytes in LDS
    constexpr int NKW = SPLIT ? 3 : 1, KRW = 3 / NKW;                // wave groups over kernel rows, kernel rows per wave
    constexpr int NTH = 64 * WM * WN * NKW;
    constexpr int CA = 32 * WM, CG = 32 * WN;
    constexpr int NEA = GP + 2, PAIRS_A = NEA / 2, PAIRS_G = GP / 2;
    constexpr int PITCH = (GP == 32 || F32) ? 80 : 48;               // bytes per [channel] row >= EB (GP + 2) (16-byte multiple; b128 reads of 16 lanes cover all banks)
    constexpr int NCA = CA / CPL, NCG = CG / CPL;
    constexpr int UA = PAIRS_A * NCA, UG = PAIRS_G * NCG;
    constexpr int JA = (UA + NTH - 1) / NTH, JG = (UG + NTH - 1) / NTH;
    constexpr int NPL = SPLIT ? 2 : 1;                               // planes: hi, lo
    constexpr int PLANE_A = CA * PITCH, PLANE_G = CG * PITCH;
    constexpr int SLOT_A = NPL * PLANE_A, BUF_G = NPL * PLANE_G;
    constexpr int DUMP = 4 * SLOT_A + 2 * BUF_G;                     // where threads without a staging unit store (no branch in the step)
    __shared__ __attribute__((aligned(16))) unsigned char lds[DUMP + (SLOT_A > BUF_G ? SLOT_A : BUF_G)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kb = lane >> 5;
    const int kw = wave / (WM * WN), wt = wave % (WM * WN), wm = wt / WN, wn = wt % WN;
    unsigned item = vad_xcd_remap(blockIdx.x, gridDim.x);
    const int ct = item % p.ci_tiles; item /= p.ci_tiles;
    const int cgp = item % p.col_groups; item /= p.col_groups;
    const int strip = item % p.strips;
    const int fs = item / p.strips;
    const int H = p.h, W = p.w, lx = strip * GP;
    const int f0 = fs * p.frames_per_split, f1 = (f0 + p.frames_per_split < p.n) ? f0 + p.frames_per_split : p.n;
    const int nf = f1 - f0;
    const unsigned pix_a = (unsigned)p.cin * ES, pix_g = (unsigned)p.ncols * ES;
    const unsigned a_bytes = (unsigned)(H * W) * pix_a, g_bytes = (unsigned)(H * W) * pix_g;
    constexpr int NACC = 3 * KRW;
    f32x16 acc[NACC], cor[SPLIT ? NACC : 1];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; if constexpr (SPLIT) cor[t][r] = 0.f; }

    // Staging units (unit u = tid + NTH j), invariants as in the kernel above - but PIXEL PAIR fastest (pr = u % PAIRS, channel chunk
    // c = u / PAIRS): a unit's ds_write_b32 go CPL rows apart, and rows CPL apart share their banks whatever the (16-byte
    // aligned) pitch - with the chunk fastest the stores of a wave were 8- to 16-way bank conflicts (SQ_LDS_BANK_CONFLICT: more
    // than half the kernel's cycles), with the pair fastest consecutive lanes hit consecutive banks.  Four chunks stay together
    // (lanes 4i .. 4i+3 load 64 contiguous bytes of one pixel: full use of the lines they touch) at the price of a 2-way (fp32
    // tensors: free) or 4-way (bf16 tensors: twice the store cycles) conflict among them.
    static_assert(NCA % 4 == 0 && NCG % 4 == 0, "chunk quads");
    unsigned voa[JA], vog[JG];
    int ea[JA], eg[JG], wa[JA], wgo[JG];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int u = tid + NTH * j, pr = (u / 4) % PAIRS_A, c = (u / (4 * PAIRS_A)) * 4 + (u & 3);
        voa[j] = (unsigned)(2 * pr) * pix_a + (unsigned)c * 16u;
        ea[j] = u < UA ? 2 * pr - 1 + lx : (1 << 30);                // pixel of the pair's first element (no such unit: never in range)
        wa[j] = (CPL * c) * PITCH + 2 * EB * pr;
    }
#pragma unroll
    for (int j = 0; j < JG; ++j) {
        const int u = tid + NTH * j, pr = (u / 4) % PAIRS_G, c = (u / (4 * PAIRS_G)) * 4 + (u & 3);
        vog[j] = (unsigned)(2 * pr) * pix_g + (unsigned)c * 16u;
        eg[j] = u < UG ? 2 * pr + lx : (1 << 30);
        wgo[j] = 4 * SLOT_A + (CPL * c) * PITCH + 2 * EB * pr;
    }
    // The stream of activation rows: position q = f (H + 1) + r is the zero row for r = 0 and row r - 1 of frame f0 + f otherwise
    // (position nf (H + 1) is the zero row that closes the last frame).  Output rows sit at the positions with r >= 1.
    int af = 0, ar = 0;            // next activation position to fetch
    int gf = 0, gr = 1;            // next gradient position to fetch (the centre of a step)
    u32x4 sa[2][JA][2], sg[2][JG][2];          // two staging sets: a row is fetched two steps before it is written to LDS
    auto fetchA = [&](auto SET) {
        constexpr int X = decltype(SET)::value;
        const bool real = ar >= 1 && af < nf;
        // (a zero row is a zero-sized descriptor: its base is never dereferenced)
        const __amdgpu_buffer_rsrc_t ra = vad_rsrc((const char*)p.a + (size_t)(f0 + af) * H * W * pix_a - pix_a, real ? a_bytes + pix_a : 0u);
        const unsigned abase = (unsigned)((ar - 1) * W + lx) * pix_a + (unsigned)(ct * CA) * ES;
#pragma unroll
        for (int j = 0; j < JA; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                sa[X][j][q] = __builtin_bit_cast(u32x4, vad_bload4(ra, (unsigned)(ea[j] + q) < (unsigned)W ? voa[j] + (unsigned)q * pix_a : VAD_OOB, abase));
        if (++ar > H) { ar = 0; ++af; }
    };
    auto fetchG = [&](auto SET) {
        constexpr int X = decltype(SET)::value;
        const bool real = gr >= 1 && gf < nf;
        const __amdgpu_buffer_rsrc_t rg = vad_rsrc((const char*)p.g + (size_t)(f0 + gf) * H * W * pix_g, real ? g_bytes : 0u);
        const unsigned gbase = (unsigned)((gr - 1) * W + lx) * pix_g + (unsigned)(cgp * CG) * ES;
#pragma unroll
        for (int j = 0; j < JG; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                sg[X][j][q] = __builtin_bit_cast(u32x4, vad_bload4(rg, (unsigned)(eg[j] + q) < (unsigned)W ? vog[j] + (unsigned)q * pix_g : VAD_OOB, gbase));
        if (++gr > H) { gr = 0; ++gf; }
    };
    // two pixels x CPL channels -> CPL dwords (pixel, pixel + 1) of one channel each, written down a column of the [channel][pixel] plane
    auto put = [&](unsigned char* dst, int plane_bytes, const u32x4& p0, const u32x4& p1) {
        if constexpr (FMT == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *(unsigned*)(dst + (2 * k) * PITCH) = __builtin_amdgcn_perm(p1[k], p0[k], 0x05040100u);       // low halves: channel 2k
                *(unsigned*)(dst + (2 * k + 1) * PITCH) = __builtin_amdgcn_perm(p1[k], p0[k], 0x07060302u);   // high halves: channel 2k + 1
            }
        } else if constexpr (F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) *(u32x2*)(dst + e * PITCH) = u32x2{p0[e], p1[e]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned hi, lo;
                wg_split2_fast(__uint_as_float(p0[e]), __uint_as_float(p1[e]), hi, lo);
                *(unsigned*)(dst + e * PITCH) = hi;
                *(unsigned*)(dst + e * PITCH + plane_bytes) = lo;
            }
        }
    };
    auto stashA = [&](auto SET, int slot) {
        constexpr int X = decltype(SET)::value;
#pragma unroll
        for (int j = 0; j < JA; ++j)
            put(lds + (ea[j] < (1 << 30) ? slot * SLOT_A + wa[j] : DUMP), PLANE_A, sa[X][j][0], sa[X][j][1]);
    };
    auto stashG = [&](auto SET, int buf) {
        constexpr int X = decltype(SET)::value;
#pragma unroll
        for (int j = 0; j < JG; ++j)
            put(lds + (eg[j] < (1 << 30) ? buf * BUF_G + wgo[j] : DUMP), PLANE_G, sg[X][j][0], sg[X][j][1]);
    };
    auto compute = [&](int s) {
        const unsigned char* G = lds + 4 * SLOT_A + (s & 1) * BUF_G + (wn * 32 + li) * PITCH;
        if constexpr (F32) {
            // exact fp32 (v_mfma_f32_32x32x2_f32, K = 2 pixels): k-step j of an 8-pixel group pairs pixel j (lanes 0-31) with pixel
            // 4 + j (lanes 32-63), so each half reads ITS four pixels (+ 2 of halo) as one ds_read_b128 + one ds_read_b64 and the
            // operand of (j, dx) is register j + dx of that window: no VALU work beside the MFMAs on the pipe they share.
#pragma unroll
            for (int grp = 0; grp < GP / 8; ++grp) {
                const int off = 4 * (8 * grp + 4 * kb);
                const f32x4 g4 = *(const f32x4*)(G + off);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const unsigned char* A = lds + ((s - 1 + k) & 3) * SLOT_A + (wm * 32 + li) * PITCH + off;
                    const f32x4 a4 = *(const f32x4*)A;
                    const f32x2 a2 = *(const f32x2*)(A + 16);
                    const float av[6] = {a4[0], a4[1], a4[2], a4[3], a2[0], a2[1]};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int d = 0; d < 3; ++d)
                            acc[3 * k + d] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j + d], g4[j], acc[3 * k + d], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int sub = 0; sub < GP / 16; ++sub) {
            const int off = 2 * (16 * sub + 8 * kb);                       // byte offset of element px0 (pixel lx + px0 - 1)
            const u32x4 gh4 = *(const u32x4*)(G + off);
            u32x4 gl4 = gh4;
            if constexpr (SPLIT) gl4 = *(const u32x4*)(G + PLANE_G + off);
#pragma unroll
            for (int k = 0; k < KRW; ++k) {
                const int kr = SPLIT ? kw : k;                             // kernel row: activation row (output row - 1 + kr)
                const unsigned char* A = lds + ((s - 1 + kr) & 3) * SLOT_A + (wm * 32 + li) * PITCH + off;
                const u32x4 ah4 = *(const u32x4*)A;
                const unsigned ah5 = (*(const u32x2*)(A + 16))[0];            // (as 8 bytes: a ds_read_b32 of this column is a 4-way bank conflict on 80-byte rows)
                const u32x4 f1h = {__builtin_amdgcn_alignbit(ah4[1], ah4[0], 16), __builtin_amdgcn_alignbit(ah4[2], ah4[1], 16),
                                   __builtin_amdgcn_alignbit(ah4[3], ah4[2], 16), __builtin_amdgcn_alignbit(ah5, ah4[3], 16)};
                const u32x4 f2h = {ah4[1], ah4[2], ah4[3], ah5};
                if constexpr (FMT == 0) {
                    acc[3 * k + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, ah4), __builtin_bit_cast(wg_bf16x8, gh4), acc[3 * k + 0], 0, 0, 0);
                    acc[3 * k + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, f1h), __builtin_bit_cast(wg_bf16x8, gh4), acc[3 * k + 1], 0, 0, 0);
                    acc[3 * k + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, f2h), __builtin_bit_cast(wg_bf16x8, gh4), acc[3 * k + 2], 0, 0, 0);
                } else {
                    const u32x4 al4 = *(const u32x4*)(A + PLANE_A);
                    const unsigned al5 = (*(const u32x2*)(A + PLANE_A + 16))[0];
                    const u32x4 f1l = {__builtin_amdgcn_alignbit(al4[1], al4[0], 16), __builtin_amdgcn_alignbit(al4[2], al4[1], 16),
                                       __builtin_amdgcn_alignbit(al4[3], al4[2], 16), __builtin_amdgcn_alignbit(al5, al4[3], 16)};
                    const u32x4 f2l = {al4[1], al4[2], al4[3], al5};
                    const wg_f16x8 gh = __builtin_bit_cast(wg_f16x8, gh4), gl = __builtin_bit_cast(wg_f16x8, gl4);
                    const wg_f16x8 fh[3] = {__builtin_bit_cast(wg_f16x8, ah4), __builtin_bit_cast(wg_f16x8, f1h), __builtin_bit_cast(wg_f16x8, f2h)};
                    const wg_f16x8 fl[3] = {__builtin_bit_cast(wg_f16x8, al4), __builtin_bit_cast(wg_f16x8, f1l), __builtin_bit_cast(wg_f16x8, f2l)};
#pragma unroll
                    for (int d = 0; d < 3; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[d], gh, acc[d], 0, 0, 0);
#pragma unroll
                    for (int d = 0; d < 3; ++d) cor[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[d], gl, cor[d], 0, 0, 0);
#pragma unroll
                    for (int d = 0; d < 3; ++d) cor[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[d], gh, cor[d], 0, 0, 0);
                }
            }
        }
        }
    };
    if (nf > 0) {                                       // (uniform over the work-group)
        const std::integral_constant<int, 0> S0;
        const std::integral_constant<int, 1> S1;
        fetchA(S0); stashA(S0, 0);
        fetchA(S0); stashA(S0, 1);
        fetchA(S0); stashA(S0, 2);
        fetchG(S0); stashG(S0, 1);
        fetchA(S1); fetchG(S1);                         // activation position 3, gradient position 2: written to LDS in step 1
        fetchA(S0); fetchG(S0);                         // positions 4 and 3: step 2
        __syncthreads();
        const int S = nf * (H + 1) - 1;                 // steps 1 .. S; step s is an output row unless s % (H + 1) == 0
        // Step s: write the rows fetched two steps ago (activation position s + 2 into the ring slot last read - as position
        // s - 2 - in step s - 1, gradient position s + 1 into the buffer step s - 1 read), fetch positions s + 4 / s + 3 into the
        // registers that held them, multiply.  Nothing a step writes is read before the barrier that ends it.  Positions past
        // the end are zero rows (zero-sized descriptors: no memory traffic).
        auto step = [&](auto SET, int s) {
            stashA(SET, (s + 2) & 3);
            stashG(SET, (s + 1) & 1);
            fetchA(SET); fetchG(SET);
            compute(s);         // (also on the idle step between two frames: its gradient row is a zero row, the products add nothing -
                                //  and without a branch the step is one block in which the stores above interleave with the MFMAs)
            __syncthreads();
        };
        for (int s = 1; s <= S; s += 2) {
            step(S1, s);
            if (s + 1 <= S) step(S0, s + 1);
        }
    }
    const size_t slot = (size_t)fs * p.strips + strip;
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        const int tap = SPLIT ? 3 * kw + t : t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ct * CA + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
            float v = acc[t][r];
            if constexpr (SPLIT) v = fmaf(cor[t][r], 1.0f / 2048.0f, v);
            p.ws[((slot * 9 + tap) * p.cin + ci) * p.ncols + cgp * CG + wn * 32 + li] = v;
        }
    }
}

// bf16 TENSORS, cin and ncols multiples of 64: the same GEMM with DWORD loads.  A dword holds the channel pair (2l, 2l+1) of
// one pixel, so the 10 (3x3: 8 pixels + halo) dwords a lane loads for a row feed TWO M-tiles - the tile's even channels from
// the low halves, its odd channels from the high halves (one v_perm per packed pair) - and the 8 dwords of the gradient feed
// two N-tiles: a wave owns a 64 x 64 tile of ONE kernel row (three taps) and issues 18 loads per 12 MFMAs where the kernel
// above issues 38 per 9.  That ratio is what bounds these kernels (see above): every load instruction is a 64-lane gather
// through the texture addresser whatever its width.  The three kernel rows of a tile are three work items (each re-reads the
// gradient row).  Row m of an M-tile is channel 64 ct + 2 m + parity, column j of an N-tile is column 64 cg + 2 j + parity.
template <int TAPS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16x2_kernel(WgradP p) {
    constexpr int ND = TAPS == 9 ? 3 : 1, HALO = TAPS == 9 ? 1 : 0, NE = 8 + 2 * HALO, NPASS = TAPS == 9 ? 3 : 1;
    const int lane = threadIdx.x & 63, li = lane & 31, kb = lane >> 5;
    unsigned item = __builtin_amdgcn_readfirstlane(vad_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));   // see WGRAD_XCD
    if (item >= p.nitems) return;
    const int ct = item % p.ci_tiles; item /= p.ci_tiles;
    const int cgp = item % p.col_groups; item /= p.col_groups;
    const int pass = item % NPASS;
    const int split = item / NPASS;
    const int H = p.h, W = p.w, total_rows = p.n * H;
    const int r0 = split * p.rows_per_split, r1 = (r0 + p.rows_per_split < total_rows) ? r0 + p.rows_per_split : total_rows;
    const unsigned a_bytes = (unsigned)(H * W) * (unsigned)p.cin * 2u, g_bytes = (unsigned)(H * W) * (unsigned)p.ncols * 2u;
    const unsigned pix_a = (unsigned)p.cin * 2u, pix_g = (unsigned)p.ncols * 2u;
    const unsigned lane_a = (unsigned)(ct * 64 + 2 * li) * 2u, lane_g = (unsigned)(cgp * 64 + 2 * li) * 2u;
    f32x16 acc[ND][2][2];          // [dx][channel parity][column parity]
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][q >> 1][q & 1][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rzero = vad_rsrc(p.a, 0);
    const int dy = TAPS == 9 ? pass - 1 : 0;
    for (int row = r0; row < r1; ++row) {
        const int n_ = row / H, ly = row - n_ * H, yy = ly + dy;
        if (yy < 0 || yy >= H) continue;                          // (uniform) this kernel row falls outside the image: zero padding
        const __amdgpu_buffer_rsrc_t ra = vad_rsrc((const char*)p.a + (size_t)n_ * H * W * p.cin * 2u, a_bytes);
        const __amdgpu_buffer_rsrc_t rg = vad_rsrc((const char*)p.g + (size_t)n_ * H * W * p.ncols * 2u, g_bytes);
        const unsigned abase = (unsigned)(yy * W) * pix_a, gbase = (unsigned)(ly * W) * pix_g;
        for (int lx = 0; lx < W; lx += 16) {
            const int px0 = lx + 8 * kb;
            unsigned gd[8], ad[NE];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int x = px0 + e;
                gd[e] = __builtin_bit_cast(unsigned, vad_bload1(rg, x < W ? lane_g + (unsigned)x * pix_g : VAD_OOB, gbase));
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int x = px0 + e - HALO;
                ad[e] = __builtin_bit_cast(unsigned, vad_bload1(ra, (unsigned)x < (unsigned)W ? lane_a + (unsigned)x * pix_a : VAD_OOB, abase));
            }
            // low / high halves of two dwords -> one packed pair (pixels k, k+1 of one channel)
            auto lo2 = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); };
            auto hi2 = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); };
            wg_bf16x8 gb[2];
            gb[0] = wg_frag(lo2(gd[0], gd[1]), lo2(gd[2], gd[3]), lo2(gd[4], gd[5]), lo2(gd[6], gd[7]));
            gb[1] = wg_frag(hi2(gd[0], gd[1]), hi2(gd[2], gd[3]), hi2(gd[4], gd[5]), hi2(gd[6], gd[7]));
#pragma unroll
            for (int d = 0; d < ND; ++d) {                        // window of 8 pixels starting at element d
                const wg_bf16x8 fe = wg_frag(lo2(ad[d], ad[d + 1]), lo2(ad[d + 2], ad[d + 3]), lo2(ad[d + 4], ad[d + 5]), lo2(ad[d + 6], ad[d + 7]));
                const wg_bf16x8 fo = wg_frag(hi2(ad[d], ad[d + 1]), hi2(ad[d + 2], ad[d + 3]), hi2(ad[d + 4], ad[d + 5]), hi2(ad[d + 6], ad[d + 7]));
                acc[d][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fe, gb[0], acc[d][0][0], 0, 0, 0);
                acc[d][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fe, gb[1], acc[d][0][1], 0, 0, 0);
                acc[d][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo, gb[0], acc[d][1][0], 0, 0, 0);
                acc[d][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo, gb[1], acc[d][1][1], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int tap = (TAPS == 9 ? 3 * pass : 0) + d;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ct * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * kb) + pa;
                *(f32x2*)&p.ws[(((size_t)split * TAPS + tap) * p.cin + ci) * p.ncols + cgp * 64 + 2 * li] = f32x2{acc[d][pa][0][r], acc[d][pa][1][r]};
            }
    }
}

// The same paired-channel arithmetic with the operands staged ONCE per work-group through LDS (round 3).  A wave of the kernel
// above re-reads its 64 + 64 channels of every pixel from L2: 4 KB per 12 MFMAs, ~96 B/clk per CU at the matrix pipe's rate -
// more than L2 delivers, so it sits at ~0.2 of the bf16 peak waiting (SQ_WAIT_ANY 40 %).  Here WM x WN waves share a
// (64 WM) x (64 WN) tile: per group of 32 pixels the work-group fetches its A rows (34 pixels with the 3x3 halo) and G rows
// with 16-byte loads (2-5 per thread instead of 36 dword gathers per lane), writes them to LDS in NHWC order (pixel pitch
// padded by 16 bytes) and every wave reads its dword pairs from there; the next group's loads are in flight during the
// MFMAs, one barrier per group (two LDS buffers).  Same products, same per-wave accumulation order over (row, group) as the
// kernel above.
// GP = pixels per group: 32, or 16 for maps 16 pixels wide (the ConvLSTM layers: half of a 32-pixel group would be padding).
// PS = 2 (GP 32): two waves share every 64 x 64 tile and take one 16-pixel half of each group each - a second split-K factor
// inside the work-group (partial slot 2 split + half), so that a 64-channel tile still has four waves per staged group.
template <int TAPS, int WM, int WN, int GP, int PS = 1>
__global__ __launch_bounds__(64 * WM * WN * PS, 2) void conv_wgrad_bf16_lds_kernel(WgradP p) {
    static_assert(PS == 1 || GP == 32, "pixel halves need two 16-pixel sub-groups");
    constexpr int ND = TAPS == 9 ? 3 : 1, HALO = TAPS == 9 ? 1 : 0, NE = 8 + 2 * HALO, NPASS = TAPS == 9 ? 3 : 1;
    constexpr int NTH = 64 * WM * WN * PS;                           // threads
    constexpr int CA = 64 * WM, CG = 64 * WN;                        // channels / columns of the work-group tile
    constexpr int PA = CA * 2 + 16, PG = CG * 2 + 16;                // LDS pixel pitch in bytes
    constexpr int NPA = GP + 2 * HALO;                               // A pixels per group
    constexpr int QA = NPA * (CA / 8), QG = GP * (CG / 8);           // 16-byte chunks per group
    constexpr int JA = (QA + NTH - 1) / NTH, JG = (QG + NTH - 1) / NTH;
    constexpr int BUF = NPA * PA + GP * PG;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kb = lane >> 5;
    const int ph = wave / (WM * WN), wt = wave % (WM * WN), wm = wt / WN, wn = wt % WN;
    unsigned item = vad_xcd_remap(blockIdx.x, gridDim.x);             // see WGRAD_XCD
    const int ct = item % p.ci_tiles; item /= p.ci_tiles;
    const int cgp = item % p.col_groups; item /= p.col_groups;
    const int pass = item % NPASS;
    const int split = item / NPASS;
    const int H = p.h, W = p.w, total_rows = p.n * H;
    const int r0 = split * p.rows_per_split, r1 = (r0 + p.rows_per_split < total_rows) ? r0 + p.rows_per_split : total_rows;
    const int dy = TAPS == 9 ? pass - 1 : 0;
    const unsigned a_bytes = (unsigned)(H * W) * (unsigned)p.cin * 2u, g_bytes = (unsigned)(H * W) * (unsigned)p.ncols * 2u;
    const unsigned pix_a = (unsigned)p.cin * 2u, pix_g = (unsigned)p.ncols * 2u;
    f32x16 acc[ND][2][2];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[d][q >> 1][q & 1][r] = 0.f;

    // (row, group) positions of this slice whose kernel row lies inside the image, in order
    auto row_ok = [&](int row) { const int ly = row % H + dy; return ly >= 0 && ly < H; };
    int row = r0, lx = 0;
    while (row < r1 && !row_ok(row)) ++row;
    // staging registers of the group being fetched
    u32x4 sa[JA], sg[JG];
    auto fetch = [&](int frow, int flx) {
        const int n_ = frow / H, ly = frow - n_ * H;
        const __amdgpu_buffer_rsrc_t ra = vad_rsrc((const char*)p.a + (size_t)n_ * H * W * p.cin * 2u, a_bytes);
        const __amdgpu_buffer_rsrc_t rg = vad_rsrc((const char*)p.g + (size_t)n_ * H * W * p.ncols * 2u, g_bytes);
        const unsigned abase = (unsigned)((ly + dy) * W) * pix_a + (unsigned)(ct * CA) * 2u;
        const unsigned gbase = (unsigned)(ly * W) * pix_g + (unsigned)(cgp * CG) * 2u;
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            const int q = tid + NTH * j, px = q / (CA / 8), c16 = q % (CA / 8), x = flx + px - HALO;
            const bool ok = q < QA && (unsigned)x < (unsigned)W;
            sa[j] = __builtin_bit_cast(u32x4, vad_bload4(ra, ok ? (unsigned)x * pix_a + (unsigned)c16 * 16u : VAD_OOB, abase));
        }
#pragma unroll
        for (int j = 0; j < JG; ++j) {
            const int q = tid + NTH * j, px = q / (CG / 8), c16 = q % (CG / 8), x = flx + px;
            const bool ok = q < QG && x < W;
            sg[j] = __builtin_bit_cast(u32x4, vad_bload4(rg, ok ? (unsigned)x * pix_g + (unsigned)c16 * 16u : VAD_OOB, gbase));
        }
    };
    auto stash = [&](int b) {
        unsigned char* A = lds + b * BUF;
        unsigned char* G = A + NPA * PA;
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            const int q = tid + NTH * j, px = q / (CA / 8), c16 = q % (CA / 8);
            if (q < QA) *(u32x4*)(A + px * PA + c16 * 16) = sa[j];
        }
#pragma unroll
        for (int j = 0; j < JG; ++j) {
            const int q = tid + NTH * j, px = q / (CG / 8), c16 = q % (CG / 8);
            if (q < QG) *(u32x4*)(G + px * PG + c16 * 16) = sg[j];
        }
    };
    auto lo2 = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); };
    auto hi2 = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); };
    auto compute = [&](int b) {
        const unsigned char* A = lds + b * BUF + (wm * 64 + 2 * li) * 2;
        const unsigned char* G = lds + b * BUF + NPA * PA + (wn * 64 + 2 * li) * 2;
#pragma unroll
        for (int sub0 = 0; sub0 < GP / 16 / PS; ++sub0) {
            const int sub = PS == 2 ? ph : sub0;
            const int px0 = 16 * sub + 8 * kb;
            unsigned gd[8], ad[NE];
#pragma unroll
            for (int e = 0; e < 8; ++e) gd[e] = *(const unsigned*)(G + (px0 + e) * PG);
#pragma unroll
            for (int e = 0; e < NE; ++e) ad[e] = *(const unsigned*)(A + (px0 + e) * PA);
            wg_bf16x8 gb[2];
            gb[0] = wg_frag(lo2(gd[0], gd[1]), lo2(gd[2], gd[3]), lo2(gd[4], gd[5]), lo2(gd[6], gd[7]));
            gb[1] = wg_frag(hi2(gd[0], gd[1]), hi2(gd[2], gd[3]), hi2(gd[4], gd[5]), hi2(gd[6], gd[7]));
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const wg_bf16x8 fe = wg_frag(lo2(ad[d], ad[d + 1]), lo2(ad[d + 2], ad[d + 3]), lo2(ad[d + 4], ad[d + 5]), lo2(ad[d + 6], ad[d + 7]));
                const wg_bf16x8 fo = wg_frag(hi2(ad[d], ad[d + 1]), hi2(ad[d + 2], ad[d + 3]), hi2(ad[d + 4], ad[d + 5]), hi2(ad[d + 6], ad[d + 7]));
                acc[d][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fe, gb[0], acc[d][0][0], 0, 0, 0);
                acc[d][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fe, gb[1], acc[d][0][1], 0, 0, 0);
                acc[d][1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo, gb[0], acc[d][1][0], 0, 0, 0);
                acc[d][1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fo, gb[1], acc[d][1][1], 0, 0, 0);
            }
        }
    };
    if (row < r1) {                                   // (uniform over the work-group: every barrier below is reached by all waves)
        fetch(row, lx);
        stash(0);
        __syncthreads();
        int b = 0;
        while (true) {
            int nrow = row, nlx = lx + GP;
            if (nlx >= W) { nlx = 0; ++nrow; while (nrow < r1 && !row_ok(nrow)) ++nrow; }
            const bool more = nrow < r1;
            if (more) fetch(nrow, nlx);               // in flight during this group's MFMAs
            compute(b);
            if (!more) break;
            stash(b ^ 1);                             // the other buffer: nobody reads it (its readers passed the last barrier)
            __syncthreads();
            b ^= 1; row = nrow; lx = nlx;
        }
    }
    const int ct64 = ct * WM + wm, cg64 = cgp * WN + wn;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int tap = (TAPS == 9 ? 3 * pass : 0) + d;
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ct64 * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * kb) + pa;
                *(f32x2*)&p.ws[(((size_t)(split * PS + ph) * TAPS + tap) * p.cin + ci) * p.ncols + cg64 * 64 + 2 * li] = f32x2{acc[d][pa][0][r], acc[d][pa][1][r]};
            }
    }
}

// First layer (input NCHW, 3 channels): M index k = c*9 + tap (27, padded to 32), A gathered from the input planes.
struct WgradC3P {     // g: fp32 or bf16 (the kernels' storage type)
    const float* x; const void* g; float* ws;
    int n, h, w, cout, splits, rows_per_split;
    unsigned nitems;
};

template <typename T>
__global__ __launch_bounds__(256) void conv_c3_wgrad_kernel(WgradC3P p) {
    constexpr unsigned ES = sizeof(T);
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    unsigned item = __builtin_amdgcn_readfirstlane(vad_xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));   // see WGRAD_XCD
    if (item >= p.nitems) return;
    const int ctiles = p.cout / 32;
    const int cgp = item % ctiles;
    const int split = item / ctiles;
    const int H = p.h, W = p.w, total_rows = p.n * H;
    const int r0 = split * p.rows_per_split, r1 = (r0 + p.rows_per_split < total_rows) ? r0 + p.rows_per_split : total_rows;
    const int c = li / 9, tap = li - c * 9, dy = tap / 3 - 1, dx = tap % 3 - 1;
    const unsigned x_bytes = (unsigned)(3 * H * W) * 4u, g_bytes = (unsigned)(H * W) * (unsigned)p.cout * ES;
    // four independent accumulator chains (one 32x32x2 MFMA each per 8 pixels) so the matrix pipe never waits on its own
    // result; the 8 operand loads of a group are issued before its MFMAs
    constexpr int U = 4;
    f32x16 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    for (int row = r0; row < r1; ++row) {
        const int n = row / H, y = row - n * H;
        const __amdgpu_buffer_rsrc_t rx = vad_rsrc(p.x + (size_t)n * 3 * H * W, x_bytes);
        const __amdgpu_buffer_rsrc_t rg = vad_rsrc((const T*)p.g + (size_t)n * H * W * p.cout, g_bytes);
        const int yy = y + dy;
        const bool rowok = li < 27 && yy >= 0 && yy < H;
        for (int x = 0; x < W; x += 2 * U) {
            float av[U], bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int px = x + 2 * u + lh, xx = px + dx;
                bv[u] = vad_bload_e<T>(rg, px < W ? (unsigned)((y * W + px) * p.cout + cgp * 32 + li) * ES : VAD_OOB, 0);
                av[u] = vad_bload1(rx, (rowok && xx >= 0 && xx < W && px < W) ? (unsigned)(((c * H + yy) * W + xx) * 4) : VAD_OOB, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = MFMA32(av[u], bv[u], acc[u]);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
        p.ws[((size_t)split * 32 + k) * p.cout + cgp * 32 + li] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
    }
}

// The same GEMM with the A operand (the 27 taps of a pixel) read from LDS instead of gathered from global memory: a lane's
// tap of 8 consecutive pixels touches 9 different input rows (3 planes x 3 rows), i.e. every gather instruction of the
// kernel above spreads over ~9-18 cache lines and each input value is fetched ~9 times per pixel pair - the texture path,
// not the matrix pipe (0.24 busy) or the memory round trips (deeper prefetch and more waves changed nothing), was its limit.
// Here every wave keeps the 9 rows (c, y-1..y+1) of ITS current output row in its own LDS region (coalesced 16-byte loads,
// requested one row ahead, column 0 of the data at float 4 so that the writes stay 16-byte aligned, zero columns at 3 and
// 4 + W, rows outside the image written as zeros); an A value is then one ds_read_b32 at lane constant + column.  LDS
// operations of one wave execute in order, and no other wave touches the region: no barrier.  W % 8 == 0 (host-selected).
template <int MAXQ, typename T>      // 16-byte chunks per lane and staged row: W <= 256 * MAXQ (the staging registers set the occupancy)
__global__ __launch_bounds__(256) void conv_c3_wgrad_lds_kernel(WgradC3P p) {
    constexpr unsigned ES = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) float dyn_xs[];
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5, wave = threadIdx.x >> 6;
    unsigned item = __builtin_amdgcn_readfirstlane(vad_xcd_remap(blockIdx.x, gridDim.x) * 4 + wave);   // see WGRAD_XCD
    if (item >= p.nitems) return;
    const int ctiles = p.cout / 32;
    const int cgp = item % ctiles;
    const int split = item / ctiles;
    const int H = p.h, W = p.w, total_rows = p.n * H, RS = W + 8, W4 = W / 4;
    float* xs = dyn_xs + (size_t)wave * 9 * RS;
    const int r0 = split * p.rows_per_split, r1 = (r0 + p.rows_per_split < total_rows) ? r0 + p.rows_per_split : total_rows;
    const int c = li < 27 ? li / 9 : 0, tap = li < 27 ? li - (li / 9) * 9 : 0, dy = tap / 3 - 1, dx = tap % 3 - 1;   // rows k >= 27 of the
    const int lb = (c * 3 + dy + 1) * RS + 4 + dx + lh;          // result are dropped by the reduction: they may read anything finite
    const unsigned g_bytes = (unsigned)(H * W) * (unsigned)p.cout * ES;
    const unsigned gl = (unsigned)(lh * p.cout + cgp * 32 + li) * ES;
    constexpr int U = 4;
    f32x16 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    if (lane < 18) xs[(lane >> 1) * RS + ((lane & 1) ? 4 + W : 3)] = 0.f;     // the zero columns left and right of every row

    // staging: row slot s = c*3 + d holds input row (c, y - 1 + d); chunk q of a slot = its floats [4q, 4q+4)
    f32x4 st[9][MAXQ];
    auto fetch_row = [&](int row) {
        const int n = row / H, y = row - n * H;
        const float* fx = p.x + (size_t)n * 3 * H * W;
#pragma unroll
        for (int s9 = 0; s9 < 9; ++s9) {
            const int yy = y - 1 + s9 % 3;
            const bool rok = yy >= 0 && yy < H;                  // (uniform)
            const float* src = fx + ((size_t)(s9 / 3) * H + (rok ? yy : 0)) * W;
#pragma unroll
            for (int j = 0; j < MAXQ; ++j) {
                const int q = lane + 64 * j;
                st[s9][j] = (rok && q < W4) ? *(const f32x4*)(src + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_row = [&]() {
#pragma unroll
        for (int s9 = 0; s9 < 9; ++s9)
#pragma unroll
            for (int j = 0; j < MAXQ; ++j) {
                const int q = lane + 64 * j;
                if (q < W4) *(f32x4*)&xs[s9 * RS + 4 + 4 * q] = st[s9][j];
            }
    };
    if (r0 < r1) fetch_row(r0);
    for (int row = r0; row < r1; ++row) {
        store_row();                                             // (behind every read of the previous row: in order)
        if (row + 1 < r1) fetch_row(row + 1);                    // in flight during this row's groups
        const int n = row / H, y = row - n * H;
        const __amdgpu_buffer_rsrc_t rg = vad_rsrc((const T*)p.g + (size_t)n * H * W * p.cout, g_bytes);
        const unsigned grow = (unsigned)(y * W) * (unsigned)p.cout * ES;
        // g values one group ahead, in two alternating register sets (unconditional: behind the row's last group the request
        // goes to that group again - a load under `if` would make hipcc wait for it in front of the MFMAs)
        float bva[U], bvb[U];
        auto load_g = [&](int x, float (&b_)[U]) {
            const int xc = __builtin_amdgcn_readfirstlane(x < W ? x : W - 2 * U);
#pragma unroll
            for (int u = 0; u < U; ++u) b_[u] = vad_bload_e<T>(rg, gl, grow + (unsigned)((xc + 2 * u) * p.cout) * ES);
        };
        auto mma = [&](int x, const float (&b_)[U]) {
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) av[u] = xs[lb + x + 2 * u];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = MFMA32(av[u], b_[u], acc[u]);
        };
        load_g(0, bva);
        for (int x = 0; x < W; x += 4 * U) {
            load_g(x + 2 * U, bvb);
            mma(x, bva);
            load_g(x + 4 * U, bva);
            if (x + 2 * U < W) mma(x + 2 * U, bvb);      // (uniform)
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
        p.ws[((size_t)split * 32 + k) * p.cout + cgp * 32 + li] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
    }
}

// ROUTED first-layer weight gradient (bf16-tensor mode, round 4).  The first layer's conv-output gradient dy has ONE consumer -
// this weight gradient - and BatchNorm's backward makes it dense: dy = sc (dz - k1 - xhat k2), sc = gamma invstd, with dz the routed
// gradient (non-zero at the pooling argmax of each 2x2 window only).  Writing dy (1.3 GB of bf16 at 320 x 256x256) and reading it
// back was 1.2 of the step's 10.1 ms.  With X[p][k] the 27 taps of pixel p and y = W X + b the layer's own output,
//     dW[co][k] = sc ( T1[k][co] - k1 SX[k] - k2 invstd ( (W S)[co][k] + (b - mean) SX[k] ) ),
//     T1 = sum_p dz[p][co] X[p][k],   S = X^T X (the Gram matrix of the input patches),   SX[k] = sum_p X[p][k]:
// the xhat term needs no pass over y at all, and T1 is a GEMM over the POOLED gradient: this kernel reads the pooled d(out)
// (bf16), one byte of routing code per pooled element (pass A of the BatchNorm backward writes it: argmax position and sign)
// and the input planes - 0.76 GB instead of 4.6 - and never forms dy.  bf16 MFMAs (the mode's arithmetic for every other
// layer's gradients): A = the taps of 16 pixels (fp32 planes rounded as they are packed; row 27 is the constant 1 so that column
// 27 of S is SX), B = the routed gradient for T1 and A itself for S.  A wave owns a slice of ROW PAIRS (one pooled row) and keeps
// the four input rows x three planes, the pooled gradient row and its codes in its own LDS region (coalesced 16-byte loads, one
// pair ahead in registers).  vad_conv_c3_wgrad_routed reduces the partials and applies the formula above.
struct WgradC3RP {
    const float* x; const vad_bf16* dout; const unsigned char* codes; float* ws;
    int n, h, w, splits, pairs_per_split;
    unsigned nitems;
};

template <int MAXQ>      // W <= 256 MAXQ, W % 16 == 0, H even
__global__ __launch_bounds__(128) void conv_c3_wgrad_routed_kernel(WgradC3RP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    const int lane = threadIdx.x & 63, li = lane & 31, kb = lane >> 5, wave = threadIdx.x >> 6;
    const unsigned item = __builtin_amdgcn_readfirstlane(blockIdx.x * 2 + wave);
    if (item >= p.nitems) return;
    const int H = p.h, W = p.w, OW = W / 2, RS = W + 12, W4 = W / 4;           // RS: floats per staged input row (data at float 4)
    const int total_pairs = p.n * (H / 2);
    const int r0 = item * p.pairs_per_split, r1 = (r0 + p.pairs_per_split < total_pairs) ? r0 + p.pairs_per_split : total_pairs;
    const size_t region = (size_t)12 * RS * 4 + (size_t)OW * 64 + (size_t)OW * 32;
    float* xs = (float*)(dyn_lds + (size_t)wave * region);                         // [plane c][row d = 0..3 <-> 2r - 1 + d][RS]
    vad_bf16* ds = (vad_bf16*)(dyn_lds + (size_t)wave * region + (size_t)12 * RS * 4);        // [OW][32]
    unsigned char* cs = dyn_lds + (size_t)wave * region + (size_t)12 * RS * 4 + (size_t)OW * 64;   // [OW][32]
    // lane constants: tap li of the A operand (rows 28..31 are zero, row 27 is the constant one)
    const int kc = li < 27 ? li / 9 : 0, kt = li < 27 ? li - kc * 9 : 0, kdy = kt / 3 - 1, kdx = kt % 3 - 1;
    const int abase = (kc * 4 + kdy + 1) * RS + 4 + kdx + 8 * kb;                   // + yy * RS + x0 + j
    f32x16 accT, accS;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accT[r] = 0.f; accS[r] = 0.f; }
    for (int q = lane; q < 24; q += 64) xs[(q >> 1) * RS + ((q & 1) ? 4 + W : 3)] = 0.f;     // the zero columns left and right of every row

    constexpr int JD = 8 * MAXQ, JC = 4 * MAXQ;
    f32x4 sx[12][MAXQ];
    u32x4 sd[JD], sc_[JC];
    const int dchunks = OW * 4, cchunks = OW * 2;                                // 16-byte chunks of the gradient row / the code row
    auto fetch = [&](int pr) {
        const int n = pr / (H / 2), r = pr - n * (H / 2);
        const float* fx = p.x + (size_t)n * 3 * H * W;
#pragma unroll
        for (int s12 = 0; s12 < 12; ++s12) {
            const int yy = 2 * r - 1 + (s12 & 3);
            const bool rok = yy >= 0 && yy < H;                                  // (uniform)
            const float* src = fx + ((size_t)(s12 >> 2) * H + (rok ? yy : 0)) * W;
#pragma unroll
            for (int j = 0; j < MAXQ; ++j) {
                const int q = lane + 64 * j;
                sx[s12][j] = (rok && q < W4) ? *(const f32x4*)(src + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const u32x4* gd = (const u32x4*)(p.dout + ((size_t)n * (H / 2) + r) * OW * 32);
        const u32x4* gc = (const u32x4*)(p.codes + ((size_t)n * (H / 2) + r) * OW * 32);
#pragma unroll
        for (int j = 0; j < JD; ++j) { const int q = lane + 64 * j; sd[j] = q < dchunks ? gd[q] : u32x4{0u, 0u, 0u, 0u}; }
#pragma unroll
        for (int j = 0; j < JC; ++j) { const int q = lane + 64 * j; sc_[j] = q < cchunks ? gc[q] : u32x4{0u, 0u, 0u, 0u}; }
    };
    auto store = [&]() {
#pragma unroll
        for (int s12 = 0; s12 < 12; ++s12)
#pragma unroll
            for (int j = 0; j < MAXQ; ++j) { const int q = lane + 64 * j; if (q < W4) *(f32x4*)&xs[s12 * RS + 4 + 4 * q] = sx[s12][j]; }
#pragma unroll
        for (int j = 0; j < JD; ++j) { const int q = lane + 64 * j; if (q < dchunks) ((u32x4*)ds)[q] = sd[j]; }
#pragma unroll
        for (int j = 0; j < JC; ++j) { const int q = lane + 64 * j; if (q < cchunks) ((u32x4*)cs)[q] = sc_[j]; }
    };
    if (r0 < r1) fetch(r0);
    for (int pr = r0; pr < r1; ++pr) {
        store();                                                 // (behind every read of the previous pair: LDS operations of a wave execute in order)
        if (pr + 1 < r1) fetch(pr + 1);
        for (int x0 = 0; x0 < W; x0 += 16) {
            // the four windows of this lane's eight pixels: gradient and code of ITS column
            float dv[4];
            unsigned cv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int wx = (x0 + 8 * kb) / 2 + j;
                dv[j] = vad_bf16_f(ds[wx * 32 + li]);
                cv[j] = cs[wx * 32 + li];
            }
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                float av[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) av[j] = xs[abase + yy * RS + x0 + j];
                if (li >= 27) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) av[j] = li == 27 ? 1.f : 0.f;
                }
                wg_bf16x8 af, bf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    af[j] = (__bf16)av[j];
                    const unsigned c = cv[j >> 1];
                    const float g = ((c & 3u) == (unsigned)(2 * yy + (j & 1))) ? dv[j >> 1] * ((c & 4u) ? 1.f : 0.2f) : 0.f;
                    bf[j] = (__bf16)g;
                }
                accT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, accT, 0, 0, 0);
                accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, af, accS, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * kb;
        p.ws[((size_t)item * 64 + k) * 32 + li] = accT[r];                        // rows 0..31: T1[k][co]
        p.ws[((size_t)item * 64 + 32 + k) * 32 + li] = accS[r];                   // rows 32..63: S[k][k']
    }
}

// The same for fp32 tensors (the exact, split-fp16 and Winograd steps): T1 on the exact-fp32 MFMA (32x32x2: lane half lh supplies
// pixel x + lh of a pair, i.e. ONE column of a pooling window), the Gram matrix S - a property of the input frames alone - in
// split-fp16 arithmetic (22-bit products, three 32x32x16 MFMAs per 16 pixels: an exact-fp32 S would double the kernel's matrix
// work for a correction term).  d(out) is fp32 [n, h/2, w/2, 32].
struct WgradC3RFP {
    const float* x; const float* dout; const unsigned char* codes; float* ws;
    int n, h, w, splits, pairs_per_split;
    unsigned nitems;
};

template <int MAXQ>
__global__ __launch_bounds__(128) void conv_c3_wgrad_routed_f32_kernel(WgradC3RFP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5, wave = threadIdx.x >> 6;
    const unsigned item = __builtin_amdgcn_readfirstlane(blockIdx.x * 2 + wave);
    if (item >= p.nitems) return;
    const int H = p.h, W = p.w, OW = W / 2, RS = W + 12, W4 = W / 4;
    const int total_pairs = p.n * (H / 2);
    const int r0 = item * p.pairs_per_split, r1 = (r0 + p.pairs_per_split < total_pairs) ? r0 + p.pairs_per_split : total_pairs;
    // LDS of a wave: the 12 input rows + a row of ones + a row of zeros (taps 27 and 28..31 of the A operand read those: no
    // select beside the MFMAs on the pipe they share).  The pooled gradient and the codes of a 16-pixel group - 8 windows x 32
    // columns = 1 KB + 256 B, contiguous - are fetched with ONE 16-byte load per lane (+ one dword for lanes 0..63 of the codes)
    // a group ahead and turned to the (window, column) order through a small LDS buffer (two per wave): staging the whole
    // rows (16 KB per wave) left one wave per SIMD, per-lane dword / byte gathers were 16 load instructions per 16 MFMAs.
    float* xs = (float*)dyn_lds + (size_t)wave * (14 * RS + 2 * 320);
    float* gbuf = xs + 14 * RS;                                                  // [2][256 gradient floats + 64 code dwords]
    const int kc = li < 27 ? li / 9 : 0, kt = li < 27 ? li - kc * 9 : 0, kdy = kt / 3 - 1, kdx = kt % 3 - 1;
    const int abase = li < 27 ? (kc * 4 + kdy + 1) * RS + 4 + kdx : (li == 27 ? 12 : 13) * RS + 4;     // + yy * ystep + pixel
    const int ystep = li < 27 ? RS : 0;
    f32x16 accT, accS, corS;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accT[r] = 0.f; accS[r] = 0.f; corS[r] = 0.f; }
    for (int q = lane; q < 24; q += 64) xs[(q >> 1) * RS + ((q & 1) ? 4 + W : 3)] = 0.f;
    for (int q = lane; q < RS; q += 64) { xs[12 * RS + q] = 1.f; xs[13 * RS + q] = 0.f; }

    f32x4 sx[12][MAXQ];
    auto fetch = [&](int pr) {
        const int n = pr / (H / 2), r = pr - n * (H / 2);
        const float* fx = p.x + (size_t)n * 3 * H * W;
#pragma unroll
        for (int s12 = 0; s12 < 12; ++s12) {
            const int yy = 2 * r - 1 + (s12 & 3);
            const bool rok = yy >= 0 && yy < H;
            const float* src = fx + ((size_t)(s12 >> 2) * H + (rok ? yy : 0)) * W;
#pragma unroll
            for (int j = 0; j < MAXQ; ++j) {
                const int q = lane + 64 * j;
                sx[s12][j] = (rok && q < W4) ? *(const f32x4*)(src + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int s12 = 0; s12 < 12; ++s12)
#pragma unroll
            for (int j = 0; j < MAXQ; ++j) { const int q = lane + 64 * j; if (q < W4) *(f32x4*)&xs[s12 * RS + 4 + 4 * q] = sx[s12][j]; }
    };
    if (r0 < r1) fetch(r0);
    for (int pr = r0; pr < r1; ++pr) {
        store();
        if (pr + 1 < r1) fetch(pr + 1);
        const f32x4* drow = (const f32x4*)(p.dout + (size_t)pr * OW * 32);           // (pr = n * (H/2) + r: the pooled rows are contiguous)
        const unsigned* crow = (const unsigned*)(p.codes + (size_t)pr * OW * 32);
        f32x4 dq;
        unsigned cq;
        auto load_g = [&](int x0) {
            const int xc = x0 < W ? x0 : 0;                                      // (behind the last group: group 0 again, unused)
            dq = drow[(xc / 2) * 8 + lane];                                      // windows xc/2 .. xc/2+7: [8][32] floats = 64 lanes x 16 bytes
            cq = crow[(xc / 2) * 8 + lane];                                      // their codes: [8][32] bytes = 64 dwords
        };
        load_g(0);
        int gb = 0;
        for (int x0 = 0; x0 < W; x0 += 16) {
            float* gq = gbuf + gb * 320;
            *(f32x4*)&gq[4 * lane] = dq;                                         // (the buffer's last readers - two groups ago - are behind in program order)
            ((unsigned*)gq)[256 + lane] = cq;
            load_g(x0 + 16);
            float gs[8];
            unsigned cm[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned c = ((const unsigned char*)(gq + 256))[u * 32 + li];
                gs[u] = gq[u * 32 + li] * ((c & 4u) ? 1.f : 0.2f);
                cm[u] = c & 3u;
            }
            gb ^= 1;
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                // T1: exact fp32, one pixel pair (= the two columns of one window) per MFMA
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float a = xs[abase + yy * ystep + x0 + 2 * u + lh];
                    const float g = cm[u] == (unsigned)(2 * yy + lh) ? gs[u] : 0.f;
                    accT = __builtin_amdgcn_mfma_f32_32x32x2f32(a, g, accT, 0, 0, 0);
                }
                // S: eight pixels per lane half, split into fp16 (hi, lo) pairs
                unsigned hq[4], lq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    wg_split2_fast(xs[abase + yy * ystep + x0 + 8 * lh + 2 * j], xs[abase + yy * ystep + x0 + 8 * lh + 2 * j + 1], hq[j], lq[j]);
                const wg_f16x8 ah = wg_hfrag(hq[0], hq[1], hq[2], hq[3]), al = wg_hfrag(lq[0], lq[1], lq[2], lq[3]);
                accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ah, accS, 0, 0, 0);
                corS = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, al, corS, 0, 0, 0);
                corS = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, ah, corS, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * lh;
        p.ws[((size_t)item * 64 + k) * 32 + li] = accT[r];
        p.ws[((size_t)item * 64 + 32 + k) * 32 + li] = fmaf(corS[r], 1.0f / 2048.0f, accS[r]);
    }
}

// tmp[col][64]: T1[k][co] at [co][k], S[k][k'] at [k'][32 + k] (wgrad_reduce_kernel layout 4 of the [64][32] partial tiles)
__global__ __launch_bounds__(256) void c3_routed_finalize_kernel(const float* tmp, const float* w0, const float* b0, const float* stats,
                                                                 const float* gamma, const float* ksums, float* dw, int round_w) {
    for (int idx = threadIdx.x; idx < 32 * 27; idx += 256) {
        const int co = idx / 27, k = idx - co * 27;
        const float mean = stats[co], invstd = stats[32 + co], k1 = ksums[co], k2 = ksums[32 + co];
        const float sx = tmp[27 * 64 + 32 + k];                                   // S[k][27] = SX[k]
        float ws_ = 0.f;
        for (int kk = 0; kk < 27; ++kk) {          // (W S)[co][k]: S[kk][k] sits at tmp[k][32 + kk]; W as the forward used it (bf16 operands or exact)
            const float wv = round_w ? vad_bf16_f(vad_f_bf16(w0[co * 27 + kk])) : w0[co * 27 + kk];
            ws_ = fmaf(wv, tmp[k * 64 + 32 + kk], ws_);
        }
        const float v = invstd * (ws_ + (b0[co] - mean) * sx);
        dw[co * 27 + k] = gamma[co] * invstd * (tmp[co * 64 + k] - k1 * sx - k2 * v);
    }
}

// Fixed-order sum of the split-K partials, written in the torch parameter layout.
//   layout 0: Conv2d OIHW            dst[(col*cin + ci)*9 + tap]                       (taps 9)
//   layout 1: ConvTranspose2d IOHW   col = q*cout + co -> dst[(ci*cout + co)*4 + q]    (taps 1, ncols = 4*cout)
//   layout 2: first layer OIHW       rows k = c*9+tap of 32 -> dst[col*27 + k], k < 27 (taps 1, cin = 32 rows)
//   layout 3: ConvTranspose2d(->3)   col = q*3 + c < 12 -> dst[(ci*3 + c)*4 + q]       (taps 1, ncols = 32)
//   layout 4: Conv2d k1 OIHW         dst[col*cin + ci]                                 (taps 1)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, int splits, int taps, int cin, int ncols, int layout,
                                                           float* dst) {
    // 256 consecutive elements (four per lane: a wave reads 1 KB of one partial slot per load) x 4 slices of the split range per
    // work-group (slot k belongs to slice k % 4); a slice adds its slots in increasing order, the slice sums are combined in a
    // fixed order.  Eight slots in flight per thread.  (Round 4: one element per lane made every load a 256-byte request -
    // 0.68 ms per bf16 training step for ~350 MB of partials; the sums and their order are unchanged.)
    __shared__ f32x4 part[4][64];
    const long long total = (long long)taps * cin * ncols;          // a multiple of 4 (cin and ncols are multiples of 32)
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long long idx = ((long long)blockIdx.x * 64 + e) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (idx < total) {
        int k = sl;
        for (; k + 28 < splits; k += 32) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(ws + (size_t)(k + 4 * u) * total + idx);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < splits; k += 4) s += *(const f32x4*)(ws + (size_t)k * total + idx);
    }
    part[sl][e] = s;
    __syncthreads();
    if (sl != 0 || idx >= total) return;
    s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long id = idx + j;
        const int col = (int)(id % ncols), ci = (int)((id / ncols) % cin), tap = (int)(id / ((long long)ncols * cin));
        if (layout == 0) dst[((size_t)col * cin + ci) * 9 + tap] = s[j];
        else if (layout == 1) { const int cout = ncols / 4, q = col / cout, co = col - q * cout; dst[((size_t)ci * cout + co) * 4 + q] = s[j]; }
        else if (layout == 2) { if (ci < 27) dst[(size_t)col * 27 + ci] = s[j]; }
        else if (layout == 4) dst[(size_t)col * cin + ci] = s[j];
        else if (col < 12) { const int q = col / 3, c = col - q * 3; dst[((size_t)ci * 3 + c) * 4 + q] = s[j]; }
    }
}

// ------------------------------------------------------------------------------------------------ last layer + loss
// ConvTranspose2d(32->3, k2 s2) + Tanh + MSELoss, forward and backward in one pass (models/video_autoencoder.py:259-260,
// train_video.py:54-55): per input pixel the 4x3 outputs, their squared error against x, d(pre-activation) and the
// gradient with respect to the 32 input channels.  dpre is also written as the 32-column GEMM operand of the weight
// gradient (columns q*3+c, 12..31 zero).
struct To3P {         // in / din / dpre: fp32 or bf16 (the kernel's storage type)
    const void* in; const float* w; const float* bias; const float* x;
    float* recon; void* din; void* dpre; float* loss_parts;
    int n, h, w_;           // input resolution (output is 2h x 2w)
    float gscale;           // 2 / (n * 3 * 2h * 2w)
    long long total;
};

template <typename T>
__global__ __launch_bounds__(256) void convt_to3_mse_kernel(To3P p) {
    typedef vad_io4<T> io;
    __shared__ float ws[32 * 12], bs[3], red[4];
    for (int i = threadIdx.x; i < 384; i += 256) ws[i] = p.w[i];     // [ci][c][q]
    if (threadIdx.x < 3) bs[threadIdx.x] = p.bias[threadIdx.x];
    __syncthreads();
    float lsum = 0.f;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx < p.total) {
        const int x = (int)(idx % p.w_), y = (int)((idx / p.w_) % p.h);
        const long long n = idx / ((long long)p.w_ * p.h);
        float r[32];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 v = io::ld((const T*)p.in + idx * 32 + 4 * k);
            r[4 * k] = v[0]; r[4 * k + 1] = v[1]; r[4 * k + 2] = v[2]; r[4 * k + 3] = v[3];
        }
        float dp[12];       // index q*3 + c
        const int H2 = 2 * p.h, W2 = 2 * p.w_;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float pre = bs[c];
#pragma unroll
                for (int ci = 0; ci < 32; ++ci) pre = fmaf(r[ci], ws[ci * 12 + c * 4 + q], pre);
                const float rec = vad_tanh(pre);
                const size_t xo = (((size_t)n * 3 + c) * H2 + 2 * y + (q >> 1)) * W2 + 2 * x + (q & 1);
                const float d = rec - p.x[xo];
                if (p.recon) p.recon[xo] = rec;
                lsum = fmaf(d, d, lsum);
                dp[q * 3 + c] = p.gscale * d * (1.f - rec * rec);
            }
        if (p.din) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int c = 0; c < 3; ++c) s = fmaf(dp[q * 3 + c], ws[(4 * k + e) * 12 + c * 4 + q], s);
                    o[e] = s;
                }
                io::st((T*)p.din + idx * 32 + 4 * k, o);
            }
        }
        if (p.dpre) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (k < 3) o = f32x4{dp[4 * k], dp[4 * k + 1], dp[4 * k + 2], dp[4 * k + 3]};
                io::st((T*)p.dpre + idx * 32 + 4 * k, o);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) p.loss_parts[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// loss = sum(parts)/count (fixed order); optionally db[c] = sum_q colsum[q*3+c]
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* parts, int nparts, double count, float* loss,
                                                            const float* colsum32, float* dbias3) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += (double)parts[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / count);
    if (dbias3 && threadIdx.x < 3)
        dbias3[threadIdx.x] = (colsum32[threadIdx.x] + colsum32[3 + threadIdx.x]) + (colsum32[6 + threadIdx.x] + colsum32[9 + threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------ optimiser
// torch.optim.Adam (train_video.py:175): g += wd*p; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1,
                                                   float b2, float eps, float wd, float bc1, float sqrt_bc2, float gscale) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float pv = p[i];
        const float gv = g[i] * gscale + wd * pv;
        const float mv = b1 * m[i] + (1.f - b1) * gv;
        const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        p[i] = pv - (lr / bc1) * (mv / (sqrtf(vv) / sqrt_bc2 + eps));
    }
}

// ------------------------------------------------------------------------------------------------ operand packing
// torch layouts -> the kernels' MFMA operand orders, on the device (the parameters change every step)
// split != 0: the split-fp16 operand form of the same weights (csrc/pack.cpp, conv_pkernel.h): per 16 input channels and
// output channel, two halves h of [8 x hi | 8 x lo] fp16 with hi = fp16(w), lo = fp16((w - hi) * 2^11); same byte count.
// split == 2: bf16 operands in the same slots: hi = bf16(w), lo unused (conv_pkernel.h, PREC 2)
__device__ __forceinline__ void put_split(float* dst, size_t group16, int k, float v, int mode) {      // k = channel index inside the 16
    _Float16* o = (_Float16*)dst + (group16 * 2 + ((k >> 3) & 1)) * 16;
    if (mode == 2) {
        o[k & 7] = __builtin_bit_cast(_Float16, (__bf16)v);
        o[8 + (k & 7)] = (_Float16)0.f;
        return;
    }
    const _Float16 hi = (_Float16)v;
    o[k & 7] = hi;
    o[8 + (k & 7)] = (_Float16)((v - (float)hi) * 2048.0f);
}

__global__ __launch_bounds__(256) void scale_kernel(float* p, long long n, float mul) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] *= mul;
}

__global__ __launch_bounds__(256) void pack_conv3x3_kernel(const float* w, int cout, int cin, float* fwd, float* dgrad, int split) {
    const long long total = (long long)cout * cin * 9;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int tap = (int)(idx % 9), ci = (int)((idx / 9) % cin), co = (int)(idx / (9ll * cin));
        const float v = w[idx];
        // data gradient = the same convolution with the taps rotated by 180 degrees and the channel roles swapped
        if (split) {
            if (fwd) put_split(fwd, ((size_t)tap * (cin / 16) + ci / 16) * cout + co, ci & 15, v, split);
            if (dgrad) put_split(dgrad, ((size_t)(8 - tap) * (cout / 16) + co / 16) * cin + ci, co & 15, v, split);
        } else {
            if (fwd) fwd[(((size_t)tap * (cin / 8) + ci / 8) * cout + co) * 8 + (ci & 7)] = v;
            if (dgrad) dgrad[(((size_t)(8 - tap) * (cout / 8) + co / 8) * cin + ci) * 8 + (co & 7)] = v;
        }
    }
}

// Winograd F(2x2,3x3) forms of a Conv2d weight for the training step's VAD_PREC_WINO mode (csrc/conv_wino.hip): U = G g G^T per
// (output, input) channel pair in double, rounded once - the forward form [16][cin/8][cout][8] and the data-gradient form (taps
// rotated by 180 degrees, channel roles swapped) [16][cout/8][cin][8].  One thread per channel pair.
__global__ __launch_bounds__(256) void pack_conv3x3_wino_kernel(const float* w, int cout, int cin, float* fwd, float* dgrad) {
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const long long total = (long long)cout * cin;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int ci = (int)(idx % cin), co = (int)(idx / cin);
        double g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = (double)w[idx * 9 + a * 3 + b];
#pragma unroll
        for (int fr = 0; fr < 4; ++fr)
#pragma unroll
            for (int fc = 0; fc < 4; ++fc) {
                double u = 0.0, ur = 0.0;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        u += G[fr][a] * g[a][b] * G[fc][b];
                        ur += G[fr][a] * g[2 - a][2 - b] * G[fc][b];
                    }
                const int f = fr * 4 + fc;
                if (fwd) fwd[(((size_t)f * (cin / 8) + ci / 8) * cout + co) * 8 + (ci & 7)] = (float)u;
                if (dgrad) dgrad[(((size_t)f * (cout / 8) + co / 8) * cin + ci) * 8 + (co & 7)] = (float)ur;
            }
    }
}

__global__ __launch_bounds__(256) void pack_convt2x2_kernel(const float* w, int cin, int cout, float* fwd, float* dgrad, int split, int dgrad16) {
    const long long total = (long long)cin * cout * 4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int q = (int)(idx & 3), co = (int)((idx >> 2) % cout), ci = (int)(idx / (4ll * cout));
        const float v = w[idx];
        if (fwd) {
            if (split) put_split(fwd, ((size_t)q * (cin / 16) + ci / 16) * cout + co, ci & 15, v, split);
            else fwd[(((size_t)q * (cin / 8) + ci / 8) * cout + co) * 8 + (ci & 7)] = v;
        }
        // data gradient = 1x1 convolution over the space-to-depth gradient (K index q*cout+co, N index ci); that GEMM
        // runs in exact fp32 in either mode
        // (dgrad16: VAD_PREC_BF16S runs that GEMM on bf16 operands too - the gradient tensors are bf16 in memory)
        if (dgrad) {
            const int kk = q * cout + co;
            if (dgrad16) put_split(dgrad, (size_t)(kk / 16) * cin + ci, kk & 15, v, 2);
            else dgrad[(((size_t)(kk / 8)) * cin + ci) * 8 + (kk & 7)] = v;
        }
    }
}

__global__ __launch_bounds__(256) void pack_conv1x1_kernel(const float* w, int cout, int cin, float* fwd, float* dgrad, int bf16) {
    const long long total = (long long)cout * cin;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int ci = (int)(idx % cin), co = (int)(idx / cin);
        const float v = w[idx];
        if (bf16) {
            if (fwd) put_split(fwd, (size_t)(ci / 16) * cout + co, ci & 15, v, 2);
            if (dgrad) put_split(dgrad, (size_t)(co / 16) * cin + ci, co & 15, v, 2);
            continue;
        }
        if (fwd) fwd[(((size_t)(ci / 8)) * cout + co) * 8 + (ci & 7)] = v;          // K = ci, N = co
        if (dgrad) dgrad[(((size_t)(co / 8)) * cin + ci) * 8 + (co & 7)] = v;       // K = co, N = ci
    }
}

__global__ __launch_bounds__(256) void pack_conv3x3_c3_kernel(const float* w, int cout, float* fwd) {
    const int total = 28 * cout;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int co = idx % cout, k = idx / cout;
        fwd[idx] = k < 27 ? w[(size_t)co * 27 + k] : 0.f;
    }
}

// ------------------------------------------------------------------------------------ image autoencoder: last layer
// Conv2d(32->3, k3, p1) + Tanh (models/autoencoder.py:134-135) in train mode.  Forward = the scoring tail kernel on
// device-packed weights.  Backward: dpre = d(recon) * (1 - recon^2) as three NCHW planes; with the roles swapped the first
// layer's kernels do the rest - the data gradient is a 3->32 convolution of dpre with the rotated weights
// (vad_conv3x3_c3), the weight gradient is the first-layer weight gradient of (x := dpre, g := input activation), read back
// with taps mirrored.
__global__ __launch_bounds__(256) void pack_conv3x3_to3_train_kernel(const float* w, int cin, float* fwd, float* dgrad_c3) {
    const int total = 3 * cin * 9;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int tap = idx % 9, ci = (idx / 9) % cin, co = idx / (9 * cin);
        const float v = w[idx];
        if (fwd) fwd[(size_t)(ci / 4) * 108 + tap * 12 + (ci & 3) * 3 + co] = v;
        if (dgrad_c3) {
            dgrad_c3[(size_t)(co * 9 + (8 - tap)) * cin + ci] = v;           // row k = c*9 + tap' of the [28][cin] first-layer form
            if (co == 0 && tap == 0) dgrad_c3[(size_t)27 * cin + ci] = 0.f;  // padding row
        }
    }
}

// dpre[n][c][y][x] = g * (1 - recon^2), g = drecon (given) or gscale * (recon - x); per-block partial sums of dpre for the
// bias gradient: parts[(n*3 + c) * chunks + chunk]
__global__ __launch_bounds__(256) void tanh_bwd_planes_kernel(const float* recon, const float* x, const float* drecon, float gscale,
                                                              float gmul, float* dpre, float* parts, long long plane, int chunks) {
    __shared__ float red[4];
    const long long pl = blockIdx.y, base = pl * plane;
    const long long per = (plane + chunks - 1) / chunks, i0 = (long long)blockIdx.x * per, i1 = (i0 + per < plane) ? i0 + per : plane;
    float s = 0.f;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float r = recon[base + i];
        const float g = drecon ? drecon[base + i] * gmul : gscale * (r - x[base + i]);
        const float d = g * (1.f - r * r);
        dpre[base + i] = d;
        s += d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) parts[pl * chunks + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void bias3_finalize_kernel(const float* parts, int n, int chunks, float* db3) {
    const int c = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < n * chunks; i += 64) s += (double)parts[((size_t)(i / chunks) * 3 + c) * chunks + i % chunks];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) db3[c] = (float)s;
}

// dW[c][ci][tap] = tmp[ci][c][8 - tap]   (tmp = first-layer weight gradient of the swapped-role problem, OIHW (cin,3,3,3))
__global__ __launch_bounds__(256) void mirror_last_wgrad_kernel(const float* tmp, int cin, float* dw) {
    const int total = 3 * cin * 9;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int tap = idx % 9, ci = (idx / 9) % cin, c = idx / (9 * cin);
        dw[idx] = tmp[((size_t)ci * 3 + c) * 9 + (8 - tap)];
    }
}

unsigned grid_for(long long total) {
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

// ================================================================================================ host entry points
// pixels per work-group of the channel-reduction passes: short per-thread loops (memory-level parallelism comes from
// many resident groups), at most 16384 partial rows for the finalize
static long long stats_chunk(long long npix) {
    long long chunk = 512;
    if ((npix + chunk - 1) / chunk > 16384) chunk = (npix + 16383) / 16384;
    return chunk;
}

extern "C" size_t vad_chan_ws_floats(long long npix, int c) {
    if (npix <= 0 || c <= 0) return 0;
    const long long chunk = stats_chunk(npix);
    return (size_t)((npix + chunk - 1) / chunk) * 2 * c;
}

static bool chan_ok(int c) { return c >= 4 && c % 4 == 0 && c <= 1024; }

extern "C" int vad_bn_stats(const float* y, long long npix, int c, float eps, float momentum, float* stats,
                            float* running_mean, float* running_var, float* ws, void* stream) {
    VAD_REQUIRE(y && stats && ws && npix > 0 && chan_ok(c), "bn_stats: bad arguments (c=%d)", c);
    VAD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats: running_mean/var must come together");
    const long long chunk = stats_chunk(npix);
    const int nb = (int)((npix + chunk - 1) / chunk);
    // pivot = the first pixel's channel vector (y[0..c)): a sample of each channel
    hipLaunchKernelGGL(chan_sums_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, y, npix, c, chunk, y, ws);
    VAD_LAUNCH_CHECK();
    hipLaunchKernelGGL(chan_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nb, c,
                       (double)npix, 0, eps, momentum, stats, running_mean, running_var, (float*)nullptr, (float*)nullptr, y);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// BatchNorm statistics from partial sums a producer kernel wrote ([nblocks][2][c], shifted by `pivot`): the finalize half of
// vad_bn_stats (internal: conv_mfma.hip's first-layer kernel is the producer).
int vad_bn_stats_from_partials(const float* partials, int nblocks, long long npix, int c, float eps, float momentum, float* stats,
                               float* running_mean, float* running_var, const float* pivot, void* stream) {
    VAD_REQUIRE(partials && stats && pivot && nblocks > 0 && npix > 0 && chan_ok(c), "bn_stats_from_partials: bad arguments (c=%d)", c);
    VAD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats_from_partials: running_mean/var must come together");
    hipLaunchKernelGGL(chan_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, partials, nblocks, c,
                       (double)npix, 0, eps, momentum, stats, running_mean, running_var, (float*)nullptr, (float*)nullptr, pivot);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_chan_sum(const float* g, long long npix, int c, float* out, float* ws, void* stream) {
    return vad_chan_sum_t(g, 0, npix, c, out, ws, stream);
}

// io16 != 0 (here and in the *_t forms below): the activation / gradient tensors are bf16 in memory (VAD_PREC_BF16S)
int vad_chan_sum_t(const void* g, int io16, long long npix, int c, float* out, float* ws, void* stream) {
    VAD_REQUIRE(g && out && ws && npix > 0 && chan_ok(c), "chan_sum: bad arguments (c=%d)", c);
    const long long chunk = stats_chunk(npix);
    const int nb = (int)((npix + chunk - 1) / chunk);
    if (io16) hipLaunchKernelGGL(chan_sums_kernel<vad_bf16>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const vad_bf16*)g, npix, c, chunk, (const float*)nullptr, ws);
    else hipLaunchKernelGGL(chan_sums_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)g, npix, c, chunk, (const float*)nullptr, ws);
    VAD_LAUNCH_CHECK();
    hipLaunchKernelGGL(chan_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nb, c,
                       (double)npix, 2, 0.f, 0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, out, (const float*)nullptr);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_bn_act_pool_fwd(const float* y, const float* stats, const float* gamma, const float* beta, float* out,
                                   long long out_fs, int out_ps, int remap_t, int remap_b, int n, int h, int w, int c,
                                   int act, int pool, void* stream) {
    return vad_bn_act_pool_fwd_t(y, 0, stats, gamma, beta, out, out_fs, out_ps, remap_t, remap_b, n, h, w, c, act, pool, stream);
}

static std::atomic<int> g_bn_wide{1};     // debug / A-B: 0 = the bf16 BatchNorm passes at four channels per thread
extern "C" int vad_debug_set_bn_wide(int on) { g_bn_wide = on; return VAD_OK; }
int vad_bn_act_pool_fwd_t(const void* y, int io16, const float* stats, const float* gamma, const float* beta, void* out,
                          long long out_fs, int out_ps, int remap_t, int remap_b, int n, int h, int w, int c,
                          int act, int pool, void* stream) {
    VAD_REQUIRE(y && stats && gamma && beta && out && n > 0 && h > 0 && w > 0 && chan_ok(c), "bn_act_pool_fwd: bad arguments");
    VAD_REQUIRE(act >= 0 && act <= 2 && (!pool || (h % 2 == 0 && w % 2 == 0)), "bn_act_pool_fwd: bad act/pool");
    VAD_REQUIRE(remap_t == 0 || (remap_b > 0 && n == remap_t * remap_b), "bn_act_pool_fwd: n must equal T*B with a frame remap");
    const int oh = pool ? h / 2 : h, ow = pool ? w / 2 : w;
    BnFwdP p{y, stats, gamma, beta, out, out_fs ? out_fs : (long long)oh * ow * (out_ps ? out_ps : c), out_ps ? out_ps : c,
             remap_t, remap_b, n, h, w, c, act, pool, (long long)n * oh * ow * (c / 4)};
    VAD_REQUIRE(p.out_ps % 4 == 0 && p.out_fs % 4 == 0, "bn_act_pool_fwd: strides must be multiples of 4 elements");
    VAD_REQUIRE(p.total < (1ll << 31), "bn_act_pool_fwd: %lld items are too many for the kernel's 32-bit index arithmetic", p.total);
#define BN_DISPATCH(KERNEL, GRID, STREAM)                                                                                   \
    {                                                                                                                      \
        const int v_ = (pool ? 3 : 0) + act;                                                                               \
        if (io16) switch (v_) {                                                                                            \
            case 0: hipLaunchKernelGGL((KERNEL<vad_bf16, 0, 0>), GRID, dim3(256), 0, STREAM, p); break;                    \
            case 1: hipLaunchKernelGGL((KERNEL<vad_bf16, 0, 1>), GRID, dim3(256), 0, STREAM, p); break;                    \
            case 2: hipLaunchKernelGGL((KERNEL<vad_bf16, 0, 2>), GRID, dim3(256), 0, STREAM, p); break;                    \
            case 3: hipLaunchKernelGGL((KERNEL<vad_bf16, 1, 0>), GRID, dim3(256), 0, STREAM, p); break;                    \
            case 4: hipLaunchKernelGGL((KERNEL<vad_bf16, 1, 1>), GRID, dim3(256), 0, STREAM, p); break;                    \
            default: hipLaunchKernelGGL((KERNEL<vad_bf16, 1, 2>), GRID, dim3(256), 0, STREAM, p); break;                   \
        } else switch (v_) {                                                                                               \
            case 0: hipLaunchKernelGGL((KERNEL<float, 0, 0>), GRID, dim3(256), 0, STREAM, p); break;                       \
            case 1: hipLaunchKernelGGL((KERNEL<float, 0, 1>), GRID, dim3(256), 0, STREAM, p); break;                       \
            case 2: hipLaunchKernelGGL((KERNEL<float, 0, 2>), GRID, dim3(256), 0, STREAM, p); break;                       \
            case 3: hipLaunchKernelGGL((KERNEL<float, 1, 0>), GRID, dim3(256), 0, STREAM, p); break;                       \
            case 4: hipLaunchKernelGGL((KERNEL<float, 1, 1>), GRID, dim3(256), 0, STREAM, p); break;                       \
            default: hipLaunchKernelGGL((KERNEL<float, 1, 2>), GRID, dim3(256), 0, STREAM, p); break;                      \
        }                                                                                                                  \
    }
#define BN8_DISPATCH(KERNEL, GRID, STREAM)                                                                                  \
    switch ((pool ? 3 : 0) + act) {                                                                                        \
        case 0: hipLaunchKernelGGL((KERNEL<0, 0>), GRID, dim3(256), 0, STREAM, p); break;                                  \
        case 1: hipLaunchKernelGGL((KERNEL<0, 1>), GRID, dim3(256), 0, STREAM, p); break;                                  \
        case 2: hipLaunchKernelGGL((KERNEL<0, 2>), GRID, dim3(256), 0, STREAM, p); break;                                  \
        case 3: hipLaunchKernelGGL((KERNEL<1, 0>), GRID, dim3(256), 0, STREAM, p); break;                                  \
        case 4: hipLaunchKernelGGL((KERNEL<1, 1>), GRID, dim3(256), 0, STREAM, p); break;                                  \
        default: hipLaunchKernelGGL((KERNEL<1, 2>), GRID, dim3(256), 0, STREAM, p); break;                                 \
    }
    if (io16 && c % 8 == 0 && p.out_ps % 8 == 0 && p.out_fs % 8 == 0 && g_bn_wide.load(std::memory_order_relaxed)) {
        BN8_DISPATCH(bn_act_pool_fwd8_kernel, dim3(grid_for(p.total / 2)), (hipStream_t)stream)
    } else
    BN_DISPATCH(bn_act_pool_fwd_kernel, dim3(grid_for(p.total)), (hipStream_t)stream)
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// Debug: the next vad_bn_act_pool_bwd calls record their branch decisions (one byte per output pixel and channel, see
// BnBwdP::dec) consecutively into this buffer until it is full or reset with (NULL, 0).  Used by the decision-conditioned
// float64 oracle of tests/test_hip_train_step.py; vad_debug_train_decisions_used() tells how many bytes were written.
static unsigned char* g_dec_buf = nullptr;
static size_t g_dec_cap = 0, g_dec_used = 0;
extern "C" int vad_debug_set_train_decisions(void* buf, size_t bytes) { g_dec_buf = (unsigned char*)buf; g_dec_cap = buf ? bytes : 0; g_dec_used = 0; return VAD_OK; }
extern "C" size_t vad_debug_train_decisions_used(void) { return g_dec_used; }

extern "C" int vad_bn_act_pool_bwd(const float* y, const float* stats, const float* gamma, const float* beta, const float* dout,
                                   long long dout_fs, int dout_ps, int remap_t, int remap_b, float* dy, int s2d,
                                   float* dgamma, float* dbeta, float* ksums, float* ws, int n, int h, int w, int c, int act,
                                   int pool, void* stream) {
    return vad_bn_act_pool_bwd_t(y, 0, stats, gamma, beta, dout, dout_fs, dout_ps, remap_t, remap_b, dy, s2d, dgamma, dbeta, ksums, ws,
                                 n, h, w, c, act, pool, stream);
}

int vad_bn_act_pool_bwd_t(const void* y, int io16, const float* stats, const float* gamma, const float* beta, const void* dout,
                          long long dout_fs, int dout_ps, int remap_t, int remap_b, void* dy, int s2d,
                          float* dgamma, float* dbeta, float* ksums, float* ws, int n, int h, int w, int c, int act,
                          int pool, void* stream) {
    return vad_bn_act_pool_bwd_codes_t(y, io16, stats, gamma, beta, dout, dout_fs, dout_ps, remap_t, remap_b, dy, s2d, dgamma, dbeta, ksums, ws,
                                       n, h, w, c, act, pool, nullptr, stream);
}

// codes != NULL: pass A (the sums, dgamma / dbeta, k1 / k2) ONLY, and it also writes one routing byte per pooled element -
// argmax position | sign << 2 - to `codes` ([n * oh * ow][c]); dy is not touched (may be NULL).  For a layer whose dy has a
// single consumer that can work from the routed gradient: vad_conv_c3_wgrad_routed.
int vad_bn_act_pool_bwd_codes_t(const void* y, int io16, const float* stats, const float* gamma, const float* beta, const void* dout,
                                long long dout_fs, int dout_ps, int remap_t, int remap_b, void* dy, int s2d,
                                float* dgamma, float* dbeta, float* ksums, float* ws, int n, int h, int w, int c, int act,
                                int pool, unsigned char* codes, void* stream) {
    VAD_REQUIRE(y && stats && gamma && beta && dout && (dy || codes) && dgamma && dbeta && ksums && ws, "bn_act_pool_bwd: null pointer");
    VAD_REQUIRE(n > 0 && h > 0 && w > 0 && chan_ok(c) && act >= 0 && act <= 2, "bn_act_pool_bwd: bad arguments");
    VAD_REQUIRE(!pool || (h % 2 == 0 && w % 2 == 0), "bn_act_pool_bwd: pooling needs even H, W");
    VAD_REQUIRE(!s2d || (!pool && h % 2 == 0 && w % 2 == 0), "bn_act_pool_bwd: space-to-depth output is for un-pooled layers with even H, W");
    VAD_REQUIRE(dy != dout, "bn_act_pool_bwd: dy must not alias dout (pass B reads dout while writing dy)");
    VAD_REQUIRE(remap_t == 0 || (remap_b > 0 && n == remap_t * remap_b), "bn_act_pool_bwd: n must equal T*B with a frame remap");
    const int oh = pool ? h / 2 : h, ow = pool ? w / 2 : w;
    BnBwdP p{};
    p.y = y; p.stats = stats; p.gamma = gamma; p.beta = beta; p.dout = dout;
    p.dout_ps = dout_ps ? dout_ps : c;
    p.dout_fs = dout_fs ? dout_fs : (long long)oh * ow * p.dout_ps;
    p.t = remap_t; p.b = remap_b; p.ws = ws; p.k = ksums; p.dy = dy; p.s2d = s2d;
    p.n = n; p.h = h; p.w = w; p.c = c; p.act = act; p.pool = pool;
    p.opix = (long long)n * oh * ow;
    VAD_REQUIRE(p.opix * (c / 4) < (1ll << 31), "bn_act_pool_bwd: %lld items are too many for the kernels' 32-bit index arithmetic", p.opix * (c / 4));
    p.chunk = stats_chunk(p.opix);
    p.dec = nullptr;
    p.codes = codes;
    if (g_dec_buf) {
        const size_t need = (size_t)p.opix * c;
        VAD_REQUIRE(g_dec_used + need <= g_dec_cap, "bn_act_pool_bwd: decision buffer too small (%zu + %zu > %zu)", g_dec_used, need, g_dec_cap);
        p.dec = g_dec_buf + g_dec_used;
        g_dec_used += need;
    }
    VAD_REQUIRE(p.dout_ps % 4 == 0 && p.dout_fs % 4 == 0, "bn_act_pool_bwd: strides must be multiples of 4 floats");
    const int nb = (int)((p.opix + p.chunk - 1) / p.chunk);
    hipStream_t s = (hipStream_t)stream;
    BN_DISPATCH(bn_bwd_sums_kernel, dim3(nb), s)
    VAD_LAUNCH_CHECK();
    hipLaunchKernelGGL(chan_finalize_kernel, dim3(c), dim3(256), 0, s, (const float*)ws, nb, c,
                       (double)n * h * w, 1, 0.f, 0.f, ksums, (float*)nullptr, (float*)nullptr, dgamma, dbeta, (const float*)nullptr);
    VAD_LAUNCH_CHECK();
    p.dec = nullptr;
    if (codes) return VAD_OK;                     // pass A only
    if (io16 && c % 8 == 0 && p.dout_ps % 8 == 0 && p.dout_fs % 8 == 0 && g_bn_wide.load(std::memory_order_relaxed)) {
        BN8_DISPATCH(bn_bwd_apply8_kernel, dim3(grid_for(p.opix * (c / 8))), s)
    } else
    BN_DISPATCH(bn_bwd_apply_kernel, dim3(grid_for(p.opix * (c / 4))), s)
#undef BN_DISPATCH
#undef BN8_DISPATCH
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_lstm_gates_fwd(float* z, const float* c_prev, float* c_out, float* h1, long long h1_fs, int h1_ps,
                                  float* h2, long long h2_fs, int h2_ps, int nb, int hw, int hid, void* stream) {
    return vad_lstm_gates_fwd_t(z, 0, c_prev, c_out, h1, h1_fs, h1_ps, h2, h2_fs, h2_ps, nb, hw, hid, stream);
}

int vad_lstm_gates_fwd_t(void* z, int io16, const float* c_prev, float* c_out, void* h1, long long h1_fs, int h1_ps,
                         void* h2, long long h2_fs, int h2_ps, int nb, int hw, int hid, void* stream) {
    VAD_REQUIRE(z && c_out && nb > 0 && hw > 0 && hid > 0 && hid % 4 == 0, "lstm_gates_fwd: bad arguments");
    LstmFwdP p{z, c_prev, c_out, h1, h1_fs ? h1_fs : (long long)hw * (h1_ps ? h1_ps : hid), h1_ps ? h1_ps : hid,
               h2, h2_fs ? h2_fs : (long long)hw * (h2_ps ? h2_ps : hid), h2_ps ? h2_ps : hid, hw, hid, (long long)nb * hw * (hid / 4)};
    if (io16) hipLaunchKernelGGL(lstm_gates_fwd_kernel<vad_bf16>, dim3(grid_for(p.total)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(lstm_gates_fwd_kernel<float>, dim3(grid_for(p.total)), dim3(256), 0, (hipStream_t)stream, p);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_lstm_gates_bwd(const float* gates, const float* c_prev, const float* c, const float* dh1, long long dh1_fs,
                                  int dh1_ps, const float* dh2, long long dh2_fs, int dh2_ps, const float* dc_next, float* dz,
                                  float* dc_prev, int nb, int hw, int hid, void* stream) {
    return vad_lstm_gates_bwd_t(gates, 0, c_prev, c, dh1, dh1_fs, dh1_ps, dh2, dh2_fs, dh2_ps, dc_next, dz, dc_prev, nb, hw, hid, stream);
}

int vad_lstm_gates_bwd_t(const void* gates, int io16, const float* c_prev, const float* c, const void* dh1, long long dh1_fs,
                         int dh1_ps, const void* dh2, long long dh2_fs, int dh2_ps, const float* dc_next, void* dz,
                         float* dc_prev, int nb, int hw, int hid, void* stream) {
    VAD_REQUIRE(gates && c && dz && dc_prev && nb > 0 && hw > 0 && hid > 0 && hid % 4 == 0, "lstm_gates_bwd: bad arguments");
    LstmBwdP p{gates, c_prev, c, dh1, dh1_fs ? dh1_fs : (long long)hw * (dh1_ps ? dh1_ps : hid), dh1_ps ? dh1_ps : hid,
               dh2, dh2_fs ? dh2_fs : (long long)hw * (dh2_ps ? dh2_ps : hid), dh2_ps ? dh2_ps : hid, dc_next, dz, dc_prev,
               hw, hid, (long long)nb * hw * (hid / 4)};
    if (io16) hipLaunchKernelGGL(lstm_gates_bwd_kernel<vad_bf16>, dim3(grid_for(p.total)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(lstm_gates_bwd_kernel<float>, dim3(grid_for(p.total)), dim3(256), 0, (hipStream_t)stream, p);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// debug / A-B: 0 = the bf16-tensor mode uses the one-channel-per-lane kernel everywhere, 1 = paired-channel kernel (dword loads
// per wave) where cin and ncols are multiples of 64, 2 (default) = its LDS-staged work-group form where ncols is a multiple of 128
static std::atomic<int> g_wgrad_x2{3};     // 3 = the row-ring kernel for the 3x3 layers it takes (round 4)
static std::atomic<int> g_wgrad_split{3};   // debug / A-B: 0 = VAD_PREC_SPLIT weight gradients on the exact-fp32 kernel (rounds 2-3), 1 = the per-lane split-fp16 kernel, 2 = its LDS-staged form where it applies
extern "C" int vad_debug_set_wgrad_pairs(int on) { g_wgrad_x2 = on; return VAD_OK; }
extern "C" int vad_debug_set_wgrad_split(int on) { g_wgrad_split = on; return VAD_OK; }
static std::atomic<int> g_wgrad_ring_f32{1};   // debug / A-B: 0 = exact-fp32 3x3 weight gradients on the per-wave kernel (rounds 1-3)
extern "C" int vad_debug_set_wgrad_ring_f32(int on) { g_wgrad_ring_f32 = on != 0; return VAD_OK; }

// split-K factor: enough waves to fill the chip (~4096), never more splits than image rows.  Measured on both training
// steps (32 clips / 128 images): a 2048-wave target is within noise of 4096 (35.8 vs 35.6-36.0 ms, 38.5 vs 39.0 ms), 1024
// is 13 % slower; the partial buffers are small either way.
static int wgrad_splits(long long tiles, int total_rows) {
    // ~2048 wave items per launch = ONE round of the resident slots (2 work-groups of 4 waves on 256 CUs).  4096 (rounds 1-3)
    // balanced the tail better but doubled the partials the reduction reads (~150 MB per launch): measured per 32-clip step
    // 4096 / 3072 / 2048 / 1536 / 1024 items: bf16 10.54 / 10.53 / 10.46 / 10.62 / 10.84 ms, fp32 31.5 / 32.1 / 31.0 / 32.7 / 34.8.
    long long s = (2048 + tiles - 1) / tiles;
    if (s > total_rows) s = total_rows;
    if (s > 2048) s = 2048;
    if (s < 1) s = 1;
    return (int)s;
}

// work-group items of the LDS-staged split-fp16 kernel: ~two rounds of the 3 x 256 resident work-groups
static int wgrad_split_lds_splits(long long tiles, int total_rows) {
    long long s = (1536 + tiles - 1) / tiles;
    if (s > total_rows) s = total_rows;
    if (s > 2048) s = 2048;
    if (s < 1) s = 1;
    return (int)s;
}
// (3x3 layers only: the 1x1 / transposed layers have a third of the MFMAs per staged byte and measured slower than the per-lane
// kernel - 102 / 206 / 231 us against 78 / 164 / 219 us on the decoder's three)
static bool wgrad_split_lds_ok(int taps, int w, int cin, int ncols) { return taps == 9 && ncols % 64 == 0 && (cin % 64 == 0 || (cin == 32 && w > 16)); }

// Row-ring kernel: tile shape and the frames each work-group walks.  One work-group = WM x WN (x 3, split-fp16) waves; the chip holds
// `cap` of them at once; frames per item are chosen so that the launch is as few FULL rounds of that as possible, never more
// partial slots than 128 MB of fp32 (what vad_conv_wgrad_ws_floats, which does not know the map width, reserves).
static long long ring_max_slots(int cin, int ncols) { const long long s = (128ll << 20) / 4 / (9ll * cin * ncols); return s < 1 ? 1 : s; }
struct RingPlan { bool ok; int wm, wn, gp, strips, fps, fsplits, ci_tiles, col_groups; };
static RingPlan ring_plan(int fmt, int n, int h, int w, int cin, int ncols) {
    RingPlan r{};
    r.ok = ncols % 64 == 0 && (cin % 64 == 0 || cin == 32) && n > 0 && h > 0 && w > 0;
    if (!r.ok) return r;
    r.wm = cin % 64 == 0 ? 2 : 1;
    r.wn = (fmt != 1 && ncols % 128 == 0) ? 4 : 2;
    r.gp = (w <= 16 || fmt == 2) ? 16 : 32;
    r.strips = (w + r.gp - 1) / r.gp;
    r.ci_tiles = cin / (32 * r.wm); r.col_groups = ncols / (32 * r.wn);
    const int waves = r.wm * r.wn * (fmt == 1 ? 3 : 1), per_cu = (fmt == 1 ? 12 : 8) / waves;
    const long long cap = 256ll * (per_cu > 0 ? per_cu : 1);
    const long long tiles = (long long)r.ci_tiles * r.col_groups * r.strips;
    const long long max_slots = ring_max_slots(cin, ncols);                  // partial slots = fsplits * strips
    if (r.strips > max_slots) { r.ok = false; return r; }
    long long best = -1;
    for (int fps = 1; fps <= n; ++fps) {
        const long long fsplits = (n + fps - 1) / fps;
        if (fsplits * r.strips > max_slots) continue;
        const long long rounds = (tiles * fsplits + cap - 1) / cap;
        const long long cost = rounds * ((long long)fps * (h + 1) + 4);        // steps per item + the prologue
        if (best < 0 || cost < best) { best = cost; r.fps = fps; r.fsplits = (int)fsplits; }
    }
    return r;
}

extern "C" size_t vad_conv_wgrad_ws_floats(int n, int h, int taps, int cin, int ncols) {
    if (n <= 0 || h <= 0 || cin <= 0 || ncols <= 0 || cin % 32 || ncols % 32 || (taps != 9 && taps != 1)) return 0;
    const int nt = (taps == 1 && ncols % 128 == 0) ? 4 : 1;
    const long long tiles = (long long)(cin / 32) * (ncols / (32 * nt));
    int splits = wgrad_splits(tiles, n * h);       // (the split-fp16 kernel has 3x the tiles of a 3x3 layer: never more splits)
    if (cin % 64 == 0 && ncols % 64 == 0) {      // the paired-channel kernel of the bf16-tensor mode: 64 x 64 tiles, one item per kernel row
        const int s2 = wgrad_splits((long long)(cin / 64) * (ncols / 64) * (taps == 9 ? 3 : 1), n * h);
        if (s2 > splits) splits = s2;
        if (ncols % 128 == 0) {                  // its LDS-staged form: work-group tiles, 2 or 4 waves each
            const int wm = cin % 128 == 0 ? 2 : 1, ps = (taps == 9 && wm == 1) ? 2 : 1;
            const long long tiles3 = (long long)(cin / (64 * wm)) * (ncols / 128) * (taps == 9 ? 3 : 1);
            long long s3 = (4096 / (2 * wm * ps) + tiles3 - 1) / tiles3;
            if (s3 > (long long)n * h) s3 = (long long)n * h;
            if (s3 > 2048) s3 = 2048;
            if (s3 * ps > splits) splits = (int)(s3 * ps);
        }
    }
    if (taps == 9 && ncols % 64 == 0 && (cin % 64 == 0 || cin == 32)) {      // the row-ring kernels: up to ring_max_slots partial slots
        const long long s5 = ring_max_slots(cin, ncols);
        if (s5 > splits) splits = (int)s5;
    }
    if (taps == 9 && ncols % 64 == 0 && (cin % 64 == 0 || cin == 32)) {      // the LDS-staged split-fp16 kernel (any map width: upper bound)
        const int wm = cin % 64 == 0 ? 2 : 1, ps = wm == 1 ? 2 : 1;
        const int s4 = wgrad_split_lds_splits((long long)(cin / (32 * wm)) * (ncols / 64) * (taps == 9 ? 3 : 1), n * h) * ps;
        if (s4 > splits) splits = s4;
    }
    return (size_t)splits * taps * cin * ncols;
}

extern "C" int vad_conv_wgrad(const float* a, const float* g, float* dw, float* ws, int n, int h, int w, int cin, int ncols,
                              int taps, int layout, int precision, void* stream) {
    VAD_REQUIRE(a && g && dw && ws && n > 0 && h > 0 && w > 0, "conv_wgrad: bad arguments");
    VAD_REQUIRE(precision >= VAD_PREC_FP32 && precision <= VAD_PREC_BF16S, "conv_wgrad: precision=%d must be 0 (fp32), 1 (split-fp16 operands), 2 (bf16 operands) or 3 (a and g are bf16 tensors)", precision);
    VAD_REQUIRE(cin % 32 == 0 && ncols % 32 == 0 && cin > 0 && ncols > 0, "conv_wgrad: cin=%d ncols=%d must be multiples of 32", cin, ncols);
    VAD_REQUIRE((taps == 9 && layout == 0) || (taps == 1 && (layout == 1 || layout == 3 || layout == 4)), "conv_wgrad: taps/layout mismatch");
    VAD_REQUIRE(layout != 1 || ncols % 128 == 0, "conv_wgrad: convT gradient needs ncols = 4*cout");
    VAD_REQUIRE(layout != 3 || (ncols == 32), "conv_wgrad: to3 gradient needs 32 columns");
    VAD_REQUIRE((long long)h * w * cin * 4 < (1ll << 31) && (long long)h * w * ncols * 4 < (1ll << 31), "conv_wgrad: frame too large for 32-bit offsets");
    WgradP p{};
    p.a = a; p.g = g; p.ws = ws; p.n = n; p.h = h; p.w = w; p.cin = cin; p.ncols = ncols;
    // LDS-staged work-group tiles: 3x3 layers with 128-channel tiles (the 64-channel form has half the waves per tile and its
    // 192 accumulators + staging registers spill: 0.85 ms against 0.38 ms of the per-wave kernel on enc.8), 1x1 / transposed
    // layers with 64- or 128-channel tiles
    // (1x1 / transposed layers with 64-channel tiles: 136 us against 111 us of the per-wave kernel on the 64 -> 4 x 32 @ 64x64 layer,
    // which is the HBM time of its 0.5 GB - profiles/r04_wgrad_kernel_forms.txt)
    const bool lds_ok = ncols % 128 == 0 && (taps == 9 ? (cin % 128 == 0 || (cin % 64 == 0 && w > 16)) : cin % 128 == 0);
    {   // row-ring kernels (3x3 layers; bf16 tensors and split-fp16)
        const bool ring16 = precision == VAD_PREC_BF16S && g_wgrad_x2.load(std::memory_order_relaxed) >= 3;
        const bool ring32 = precision == VAD_PREC_SPLIT && g_wgrad_split.load(std::memory_order_relaxed) >= 3;
        const bool ringf = precision == VAD_PREC_FP32 && g_wgrad_ring_f32.load(std::memory_order_relaxed);
        const RingPlan rp = (taps == 9 && (ring16 || ring32 || ringf)) ? ring_plan(ring32 ? 1 : ringf ? 2 : 0, n, h, w, cin, ncols) : RingPlan{};
        if (rp.ok) {
            WgradRingP q{};
            q.a = a; q.g = g; q.ws = ws; q.n = n; q.h = h; q.w = w; q.cin = cin; q.ncols = ncols;
            q.ci_tiles = rp.ci_tiles; q.col_groups = rp.col_groups; q.strips = rp.strips; q.frames_per_split = rp.fps;
            const long long slots = (long long)rp.fsplits * rp.strips;
            const long long items5 = (long long)rp.ci_tiles * rp.col_groups * slots;
            VAD_REQUIRE(items5 < (1ll << 31), "conv_wgrad: too many work items");
            VAD_REQUIRE((size_t)slots * 9 * cin * ncols <= vad_conv_wgrad_ws_floats(n, h, taps, cin, ncols),
                        "conv_wgrad: internal error: %lld partial slots exceed the size vad_conv_wgrad_ws_floats reports", slots);
            hipStream_t s5 = (hipStream_t)stream;
            const dim3 g5((unsigned)items5);
#define WRL(F_, WM_, WN_, GP_) hipLaunchKernelGGL((conv_wgrad_ring_kernel<F_, WM_, WN_, GP_>), g5, dim3(64 * WM_ * WN_ * (F_ == 1 ? 3 : 1)), 0, s5, q)
#define WRL_GP(F_, WM_, WN_) do { if (rp.gp == 16) WRL(F_, WM_, WN_, 16); else WRL(F_, WM_, WN_, 32); } while (0)
            if (ringf) {
                if (rp.wm == 2) { if (rp.wn == 4) WRL(2, 2, 4, 16); else WRL(2, 2, 2, 16); }
                else { if (rp.wn == 4) WRL(2, 1, 4, 16); else WRL(2, 1, 2, 16); }
            }
            else if (ring32) { if (rp.wm == 2) WRL_GP(1, 2, 2); else WRL_GP(1, 1, 2); }
            else if (rp.wm == 2) { if (rp.wn == 4) WRL_GP(0, 2, 4); else WRL_GP(0, 2, 2); }
            else { if (rp.wn == 4) WRL_GP(0, 1, 4); else WRL_GP(0, 1, 2); }
#undef WRL_GP
#undef WRL
            VAD_LAUNCH_CHECK();
            const long long total5 = 9ll * cin * ncols;
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total5 + 255) / 256)), dim3(256), 0, s5, (const float*)ws, (int)slots, taps, cin, ncols, layout, dw);
            VAD_LAUNCH_CHECK();
            return VAD_OK;
        }
    }
    if (precision == VAD_PREC_BF16S && lds_ok && g_wgrad_x2.load(std::memory_order_relaxed) >= 2) {
        const int npass = taps == 9 ? 3 : 1, wm = cin % 128 == 0 ? 2 : 1;
        const int ps = (taps == 9 && wm == 1) ? 2 : 1;               // 3x3 with 64-channel tiles: pixel halves (see the kernel)
        p.ci_tiles = cin / (64 * wm); p.col_groups = ncols / 128;
        const long long tiles3 = (long long)p.ci_tiles * p.col_groups * npass;
        // (items are work-groups of 2 wm ps waves: aim at the same ~4096 waves)
        long long sp = (4096 / (2 * wm * ps) + tiles3 - 1) / tiles3;
        if (sp > n * h) sp = n * h;
        if (sp > 2048) sp = 2048;
        if (sp < 1) sp = 1;
        p.splits = (int)sp;
        p.rows_per_split = (n * h + p.splits - 1) / p.splits;
        p.splits = (n * h + p.rows_per_split - 1) / p.rows_per_split;
        const long long items3 = tiles3 * p.splits;
        VAD_REQUIRE(items3 < (1ll << 31), "conv_wgrad: too many work items");
        VAD_REQUIRE((size_t)p.splits * ps * taps * cin * ncols <= vad_conv_wgrad_ws_floats(n, h, taps, cin, ncols),
                    "conv_wgrad: internal error: %d x %d partial slots exceed the size vad_conv_wgrad_ws_floats reports", p.splits, ps);
        p.nitems = (unsigned)items3;
        hipStream_t s3 = (hipStream_t)stream;
        const dim3 g3((unsigned)items3);
        const bool narrow = w <= 16;
#define WGL(T_, WM_, GP_) hipLaunchKernelGGL((conv_wgrad_bf16_lds_kernel<T_, WM_, 2, GP_>), g3, dim3(128 * WM_), 0, s3, p)
        if (taps == 9 && wm == 1) hipLaunchKernelGGL((conv_wgrad_bf16_lds_kernel<9, 1, 2, 32, 2>), g3, dim3(256), 0, s3, p);
        else if (taps == 9) { if (narrow) WGL(9, 2, 16); else WGL(9, 2, 32); }
        else if (wm == 2) { if (narrow) WGL(1, 2, 16); else WGL(1, 2, 32); }
        else { if (narrow) WGL(1, 1, 16); else WGL(1, 1, 32); }
#undef WGL
        VAD_LAUNCH_CHECK();
        const long long total3 = (long long)taps * cin * ncols;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total3 + 255) / 256)), dim3(256), 0, s3, (const float*)ws, p.splits * ps, taps, cin, ncols, layout, dw);
        VAD_LAUNCH_CHECK();
        return VAD_OK;
    }
    if (precision == VAD_PREC_SPLIT && g_wgrad_split.load(std::memory_order_relaxed) >= 2 && wgrad_split_lds_ok(taps, w, cin, ncols)) {
        const int npass = taps == 9 ? 3 : 1, wm = cin % 64 == 0 ? 2 : 1, ps = wm == 1 ? 2 : 1;
        p.ci_tiles = cin / (32 * wm); p.col_groups = ncols / 64;
        const long long tiles4 = (long long)p.ci_tiles * p.col_groups * npass;
        p.splits = wgrad_split_lds_splits(tiles4, n * h);
        p.rows_per_split = (n * h + p.splits - 1) / p.splits;
        p.splits = (n * h + p.rows_per_split - 1) / p.rows_per_split;
        const long long items4 = tiles4 * p.splits;
        VAD_REQUIRE(items4 < (1ll << 31), "conv_wgrad: too many work items");
        VAD_REQUIRE((size_t)p.splits * ps * taps * cin * ncols <= vad_conv_wgrad_ws_floats(n, h, taps, cin, ncols),
                    "conv_wgrad: internal error: %d x %d partial slots exceed the size vad_conv_wgrad_ws_floats reports", p.splits, ps);
        p.nitems = (unsigned)items4;
        hipStream_t s4 = (hipStream_t)stream;
        const dim3 g4((unsigned)items4);
        const bool narrow = w <= 16;
#define WSL(T_, WM_, PS_, GP_) hipLaunchKernelGGL((conv_wgrad_split_lds_kernel<T_, WM_, 2, PS_, GP_>), g4, dim3(128 * WM_ * PS_), 0, s4, p)
        if (wm == 1) WSL(9, 1, 2, 32); else if (narrow) WSL(9, 2, 1, 16); else WSL(9, 2, 1, 32);
#undef WSL
        VAD_LAUNCH_CHECK();
        const long long total4 = (long long)taps * cin * ncols;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s4, (const float*)ws, p.splits * ps, taps, cin, ncols, layout, dw);
        VAD_LAUNCH_CHECK();
        return VAD_OK;
    }
    if (precision == VAD_PREC_BF16S && cin % 64 == 0 && ncols % 64 == 0 && g_wgrad_x2.load(std::memory_order_relaxed)) {
        const int npass = taps == 9 ? 3 : 1;
        p.ci_tiles = cin / 64; p.col_groups = ncols / 64;
        const long long tiles2 = (long long)p.ci_tiles * p.col_groups * npass;
        p.splits = wgrad_splits(tiles2, n * h);
        p.rows_per_split = (n * h + p.splits - 1) / p.splits;
        p.splits = (n * h + p.rows_per_split - 1) / p.rows_per_split;
        const long long items2 = tiles2 * p.splits;
        VAD_REQUIRE(items2 < (1ll << 31), "conv_wgrad: too many work items");
        VAD_REQUIRE((size_t)p.splits * taps * cin * ncols <= vad_conv_wgrad_ws_floats(n, h, taps, cin, ncols),
                    "conv_wgrad: internal error: %d partial slots exceed the size vad_conv_wgrad_ws_floats reports", p.splits);
        p.nitems = (unsigned)items2;
        hipStream_t s2 = (hipStream_t)stream;
        if (taps == 9) hipLaunchKernelGGL(conv_wgrad_bf16x2_kernel<9>, dim3((unsigned)((items2 + 3) / 4)), dim3(256), 0, s2, p);
        else hipLaunchKernelGGL(conv_wgrad_bf16x2_kernel<1>, dim3((unsigned)((items2 + 3) / 4)), dim3(256), 0, s2, p);
        VAD_LAUNCH_CHECK();
        const long long total2 = (long long)taps * cin * ncols;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total2 + 255) / 256)), dim3(256), 0, s2, (const float*)ws, p.splits, taps, cin, ncols, layout, dw);
        VAD_LAUNCH_CHECK();
        return VAD_OK;
    }
    const int nt = (taps == 1 && ncols % 128 == 0) ? 4 : 1;
    p.ci_tiles = cin / 32; p.col_groups = ncols / (32 * nt);
    const bool split16 = precision == VAD_PREC_SPLIT && g_wgrad_split.load(std::memory_order_relaxed);
    const int npass = (split16 && taps == 9) ? 3 : 1;          // the split-fp16 kernel's 3x3 items are kernel rows
    const long long tiles = (long long)p.ci_tiles * p.col_groups * npass;
    p.splits = wgrad_splits(tiles, n * h);
    p.rows_per_split = (n * h + p.splits - 1) / p.splits;
    p.splits = (n * h + p.rows_per_split - 1) / p.rows_per_split;     // no empty splits
    const long long items = tiles * p.splits;
    VAD_REQUIRE(items < (1ll << 31), "conv_wgrad: too many work items");
    VAD_REQUIRE((size_t)p.splits * taps * cin * ncols <= vad_conv_wgrad_ws_floats(n, h, taps, cin, ncols),
                "conv_wgrad: internal error: %d partial slots exceed the size vad_conv_wgrad_ws_floats reports", p.splits);
    p.nitems = (unsigned)items;
    const dim3 grid((unsigned)((items + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
#define WG16(T_, N_, IO_) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<T_, N_, IO_>), grid, dim3(256), 0, s, p);
    if (precision == VAD_PREC_BF16S) {
        if (taps == 9) { WG16(9, 1, 1) } else if (nt == 4) { WG16(1, 4, 1) } else { WG16(1, 1, 1) }
    } else if (precision == VAD_PREC_BF16) {
        if (taps == 9) { WG16(9, 1, 0) } else if (nt == 4) { WG16(1, 4, 0) } else { WG16(1, 1, 0) }
    } else if (split16) {
        if (taps == 9) hipLaunchKernelGGL((conv_wgrad_split_kernel<9, 1>), grid, dim3(256), 0, s, p);
        else if (nt == 4) hipLaunchKernelGGL((conv_wgrad_split_kernel<1, 4>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_wgrad_split_kernel<1, 1>), grid, dim3(256), 0, s, p);
    } else if (taps == 9) hipLaunchKernelGGL((conv_wgrad_kernel<9, 1>), grid, dim3(256), 0, s, p);
    else if (nt == 4) hipLaunchKernelGGL((conv_wgrad_kernel<1, 4>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<1, 1>), grid, dim3(256), 0, s, p);
#undef WG16
    VAD_LAUNCH_CHECK();
    const long long total = (long long)taps * cin * ncols;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)ws, p.splits, taps, cin, ncols, layout, dw);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" size_t vad_conv_c3_wgrad_ws_floats(int n, int h, int cout) {
    if (n <= 0 || h <= 0 || cout <= 0 || cout % 32) return 0;
    return (size_t)wgrad_splits(cout / 32, n * h) * 32 * cout;
}

extern "C" int vad_conv_c3_wgrad(const float* x_nchw, const float* g, float* dw, float* ws, int n, int h, int w, int cout,
                                 void* stream) {
    return vad_conv_c3_wgrad_t(x_nchw, g, 0, dw, ws, n, h, w, cout, stream);
}

int vad_conv_c3_wgrad_t(const float* x_nchw, const void* g, int io16, float* dw, float* ws, int n, int h, int w, int cout, void* stream) {
    VAD_REQUIRE(x_nchw && g && dw && ws && n > 0 && h > 0 && w > 0 && cout > 0 && cout % 32 == 0, "conv_c3_wgrad: bad arguments");
    VAD_REQUIRE((long long)h * w * cout * 4 < (1ll << 31), "conv_c3_wgrad: frame too large for 32-bit offsets");
    WgradC3P p{};
    p.x = x_nchw; p.g = g; p.ws = ws; p.n = n; p.h = h; p.w = w; p.cout = cout;
    p.splits = wgrad_splits(cout / 32, n * h);
    p.rows_per_split = (n * h + p.splits - 1) / p.splits;
    p.splits = (n * h + p.rows_per_split - 1) / p.rows_per_split;
    const long long items = (long long)(cout / 32) * p.splits;
    p.nitems = (unsigned)items;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)4 * 9 * (w + 8) * sizeof(float);
    if (w % 8 == 0 && w <= 1024 && lds <= 64 * 1024) {
        const dim3 g4((unsigned)((items + 3) / 4));
        if (w <= 256) { if (io16) hipLaunchKernelGGL((conv_c3_wgrad_lds_kernel<1, vad_bf16>), g4, dim3(256), lds, s, p);
                        else hipLaunchKernelGGL((conv_c3_wgrad_lds_kernel<1, float>), g4, dim3(256), lds, s, p); }
        else { if (io16) hipLaunchKernelGGL((conv_c3_wgrad_lds_kernel<4, vad_bf16>), g4, dim3(256), lds, s, p);
               else hipLaunchKernelGGL((conv_c3_wgrad_lds_kernel<4, float>), g4, dim3(256), lds, s, p); }
    } else {
        if (io16) hipLaunchKernelGGL(conv_c3_wgrad_kernel<vad_bf16>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv_c3_wgrad_kernel<float>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, p);
    }
    VAD_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((32ll * cout + 255) / 256)), dim3(256), 0, s, (const float*)ws, p.splits, 1, 32, cout, 2, dw);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// Routed first-layer weight gradient (see conv_c3_wgrad_routed_kernel): ws = [splits][64][32] partial tiles + [32][64] reduced.
static int c3_routed_splits(int n, int h) { return wgrad_splits(1, n * (h / 2)); }
size_t vad_conv_c3_wgrad_routed_ws_floats(int n, int h) {
    if (n <= 0 || h <= 0 || h % 2) return 0;
    return (size_t)c3_routed_splits(n, h) * 2048 + 2048;
}
static std::atomic<int> g_c3_routed{1};          // debug / A-B: 0 = BatchNorm backward pass B + the plain first-layer weight gradient (rounds 1-3)
extern "C" int vad_debug_set_c3_routed(int on) { g_c3_routed = on != 0; return VAD_OK; }
int vad_c3_routed_enabled(void) { return g_c3_routed.load(std::memory_order_relaxed); }
// (w <= 768: two waves' input rows + gradient row + codes fit the 160 KB of LDS in both forms; wider frames take pass B + the plain kernel)
int vad_conv_c3_wgrad_routed_ok(int h, int w, int cout) { return cout == 32 && h % 2 == 0 && w % 16 == 0 && w <= 768; }

int vad_conv_c3_wgrad_routed(const float* x_nchw, const void* dout_bf16, int io16, const unsigned char* codes, const float* w0, const float* b0,
                             const float* stats, const float* gamma, const float* ksums, float* dw, float* ws, int n, int h, int w,
                             int cout, void* stream) {
    VAD_REQUIRE(x_nchw && dout_bf16 && codes && w0 && b0 && stats && gamma && ksums && dw && ws && n > 0, "conv_c3_wgrad_routed: bad arguments");
    VAD_REQUIRE(vad_conv_c3_wgrad_routed_ok(h, w, cout), "conv_c3_wgrad_routed: needs 32 output channels, even H, W %% 16 == 0 and W <= 768 (got %dx%d, %d)", h, w, cout);
    WgradC3RP p{};
    p.x = x_nchw; p.dout = (const vad_bf16*)dout_bf16; p.codes = codes; p.ws = ws; p.n = n; p.h = h; p.w = w;
    WgradC3RFP pf{};
    pf.x = x_nchw; pf.dout = (const float*)dout_bf16; pf.codes = codes; pf.ws = ws; pf.n = n; pf.h = h; pf.w = w;
    const int total_pairs = n * (h / 2);
    p.splits = c3_routed_splits(n, h);
    p.pairs_per_split = (total_pairs + p.splits - 1) / p.splits;
    p.splits = (total_pairs + p.pairs_per_split - 1) / p.pairs_per_split;
    p.nitems = (unsigned)p.splits;
    pf.splits = p.splits; pf.pairs_per_split = p.pairs_per_split; pf.nitems = p.nitems;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = io16 ? 2 * ((size_t)12 * (w + 12) * 4 + (size_t)(w / 2) * 96) : 2 * ((size_t)14 * (w + 12) + 640) * 4;
    VAD_REQUIRE(lds <= 160 * 1024, "conv_c3_wgrad_routed: frame too wide (%zu B of LDS)", lds);
    const dim3 grid((unsigned)((p.splits + 1) / 2));
    // (> 64 KB of dynamic LDS needs the attribute, once per kernel and process)
    static std::atomic<bool> attr_set[4];        // (setting it twice from two threads is harmless; one process per GPU)
    auto big = [&](int which, const void* fn) -> int {
        if (lds > 64 * 1024 && !attr_set[which].load(std::memory_order_acquire)) {
            VAD_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[which].store(true, std::memory_order_release);
        }
        return VAD_OK;
    };
    int rc = VAD_OK;
    if (io16) {
        if (w <= 256) { rc = big(0, (const void*)conv_c3_wgrad_routed_kernel<1>); if (rc == VAD_OK) hipLaunchKernelGGL(conv_c3_wgrad_routed_kernel<1>, grid, dim3(128), lds, s, p); }
        else { rc = big(1, (const void*)conv_c3_wgrad_routed_kernel<4>); if (rc == VAD_OK) hipLaunchKernelGGL(conv_c3_wgrad_routed_kernel<4>, grid, dim3(128), lds, s, p); }
    } else {
        if (w <= 256) { rc = big(2, (const void*)conv_c3_wgrad_routed_f32_kernel<1>); if (rc == VAD_OK) hipLaunchKernelGGL(conv_c3_wgrad_routed_f32_kernel<1>, grid, dim3(128), lds, s, pf); }
        else { rc = big(3, (const void*)conv_c3_wgrad_routed_f32_kernel<4>); if (rc == VAD_OK) hipLaunchKernelGGL(conv_c3_wgrad_routed_f32_kernel<4>, grid, dim3(128), lds, s, pf); }
    }
    if (rc != VAD_OK) return rc;
    VAD_LAUNCH_CHECK();
    float* tmp = ws + (size_t)p.splits * 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((2048 + 255) / 256)), dim3(256), 0, s, (const float*)ws, p.splits, 1, 64, 32, 4, tmp);
    VAD_LAUNCH_CHECK();
    hipLaunchKernelGGL(c3_routed_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)tmp, w0, b0, stats, gamma, ksums, dw, io16 ? 1 : 0);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// ws = [nb loss partials][64: column sums of dpre][partials of that column reduction]
extern "C" size_t vad_convt_to3_mse_ws_floats(int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    const long long total = (long long)n * h * w;
    return (size_t)((total + 255) / 256) + 64 + vad_chan_ws_floats(total, 32);
}

extern "C" int vad_convt_to3_mse(const float* in_nhwc, const float* w_iohw, const float* bias3, const float* x_nchw, float* recon,
                                 float* din, float* dpre32, float* loss, float* dbias3, float* ws, int n, int h, int w,
                                 void* stream) {
    return vad_convt_to3_mse_t(in_nhwc, 0, w_iohw, bias3, x_nchw, recon, din, dpre32, loss, dbias3, ws, n, h, w, 1.f, stream);
}

int vad_convt_to3_mse_t(const void* in_nhwc, int io16, const float* w_iohw, const float* bias3, const float* x_nchw, float* recon,
                        void* din, void* dpre32, float* loss, float* dbias3, float* ws, int n, int h, int w, float grad_mul, void* stream) {
    VAD_REQUIRE(in_nhwc && w_iohw && bias3 && x_nchw && loss && ws && n > 0 && h > 0 && w > 0, "convt_to3_mse: bad arguments");
    VAD_REQUIRE(vad_is_pow2f(grad_mul), "convt_to3_mse: grad_mul=%g must be a power of two (an exact rescaling of every gradient)", (double)grad_mul);
    VAD_REQUIRE(!dbias3 || dpre32, "convt_to3_mse: the bias gradient needs the dpre buffer");
    const long long total = (long long)n * h * w;
    const long long nb = (total + 255) / 256;
    VAD_REQUIRE(nb < (1ll << 31), "convt_to3_mse: grid too large");
    const double count = (double)n * 3.0 * (2.0 * h) * (2.0 * w);
    To3P p{in_nhwc, w_iohw, bias3, x_nchw, recon, din, dpre32, ws, n, h, w, (float)(2.0 / count) * grad_mul, total};
    hipStream_t s = (hipStream_t)stream;
    if (io16) hipLaunchKernelGGL(convt_to3_mse_kernel<vad_bf16>, dim3((unsigned)nb), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(convt_to3_mse_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, p);
    VAD_LAUNCH_CHECK();
    float* colsum = ws + nb;
    if (dbias3) {
        const long long chunk = stats_chunk(total);
        const int cb = (int)((total + chunk - 1) / chunk);
        float* cws = ws + nb + 64;
        if (io16) hipLaunchKernelGGL(chan_sums_kernel<vad_bf16>, dim3(cb), dim3(256), 0, s, (const vad_bf16*)dpre32, total, 32, chunk, (const float*)nullptr, cws);
        else hipLaunchKernelGGL(chan_sums_kernel<float>, dim3(cb), dim3(256), 0, s, (const float*)dpre32, total, 32, chunk, (const float*)nullptr, cws);
        VAD_LAUNCH_CHECK();
        hipLaunchKernelGGL(chan_finalize_kernel, dim3(32), dim3(256), 0, s, (const float*)cws, cb, 32, (double)total, 2, 0.f, 0.f,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, colsum, (const float*)nullptr);
        VAD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, (int)nb, count, loss,
                       (const float*)colsum, dbias3);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, float grad_scale, void* stream) {
    VAD_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_scale_floats(float* p, long long n, float mul, void* stream) {
    VAD_REQUIRE(p && n > 0, "scale_floats: bad arguments");
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, n, mul);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_train_pack_conv3x3(const float* w_oihw, int cout, int cin, float* fwd, float* dgrad, int precision, void* stream) {
    VAD_REQUIRE(w_oihw && (fwd || dgrad) && cout > 0 && cin > 0 && cin % 8 == 0 && (!dgrad || cout % 8 == 0), "train_pack_conv3x3: bad arguments");
    if (precision == VAD_PREC_WINO) {     // Winograd forms (16 "taps": vad_pack_conv3x3_wino_floats of room each)
        hipLaunchKernelGGL(pack_conv3x3_wino_kernel, dim3(grid_for((long long)cout * cin)), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, fwd, dgrad);
        VAD_LAUNCH_CHECK();
        return VAD_OK;
    }
    VAD_REQUIRE(precision >= VAD_PREC_FP32 && precision <= VAD_PREC_BF16S, "train_pack_conv3x3: precision=%d must be 0 (fp32), 1 (split fp16), 2 or 3 (bf16) or 4 (Winograd)", precision);
    const int split = precision == VAD_PREC_BF16S ? 2 : precision;   // the packed layout follows the arithmetic mode, like the host packers (2 = bf16 in the hi slots)
    VAD_REQUIRE(!split || (cin % 16 == 0 && (!dgrad || cout % 16 == 0)), "train_pack_conv3x3: split precision needs channel counts in multiples of 16");
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid_for(9ll * cout * cin)), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, fwd, dgrad, split);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_train_pack_convt2x2(const float* w_iohw, int cin, int cout, float* fwd, float* dgrad, int precision, void* stream) {
    VAD_REQUIRE(w_iohw && (fwd || dgrad) && cout > 0 && cin > 0 && cin % 8 == 0 && (!dgrad || (4 * cout) % 8 == 0), "train_pack_convt2x2: bad arguments");
    VAD_REQUIRE(precision >= VAD_PREC_FP32 && precision <= VAD_PREC_BF16S, "train_pack_convt2x2: precision=%d must be 0 (fp32), 1 (split fp16), 2 or 3 (bf16)", precision);
    const int split = precision == VAD_PREC_BF16S ? 2 : precision;
    VAD_REQUIRE(!split || cin % 16 == 0, "train_pack_convt2x2: split precision needs cin in multiples of 16");
    hipLaunchKernelGGL(pack_convt2x2_kernel, dim3(grid_for(4ll * cout * cin)), dim3(256), 0, (hipStream_t)stream, w_iohw, cin, cout, fwd, dgrad, split,
                       precision == VAD_PREC_BF16S ? 1 : 0);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_train_pack_conv1x1(const float* w_oihw, int cout, int cin, float* fwd, float* dgrad, void* stream) {
    return vad_train_pack_conv1x1_p(w_oihw, cout, cin, fwd, dgrad, VAD_PREC_FP32, stream);
}

// precision VAD_PREC_BF16S: bf16 operand forms ([K/16][N][half][8 x bf16 | unused]) for vad_conv1x1_p; anything else: fp32
int vad_train_pack_conv1x1_p(const float* w_oihw, int cout, int cin, float* fwd, float* dgrad, int precision, void* stream) {
    VAD_REQUIRE(w_oihw && (fwd || dgrad) && cout > 0 && cin > 0 && cin % 8 == 0 && (!dgrad || cout % 8 == 0), "train_pack_conv1x1: bad arguments");
    const int bf16 = precision == VAD_PREC_BF16S;
    VAD_REQUIRE(!bf16 || (cin % 16 == 0 && cout % 16 == 0), "train_pack_conv1x1: bf16 operands need channel counts in multiples of 16");
    hipLaunchKernelGGL(pack_conv1x1_kernel, dim3(grid_for((long long)cout * cin)), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin, fwd, dgrad, bf16);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_train_pack_conv3x3_c3(const float* w_oihw, int cout, float* fwd, void* stream) {
    VAD_REQUIRE(w_oihw && fwd && cout > 0, "train_pack_conv3x3_c3: bad arguments");
    hipLaunchKernelGGL(pack_conv3x3_c3_kernel, dim3(grid_for(28ll * cout)), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, fwd);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// ------------------------------------------------------------------------------------ image autoencoder: last layer (host)
extern "C" int vad_train_pack_conv3x3_to3(const float* w_oihw, int cin, float* fwd, float* dgrad_c3, void* stream) {
    VAD_REQUIRE(w_oihw && (fwd || dgrad_c3) && cin > 0 && cin % 4 == 0, "train_pack_conv3x3_to3: bad arguments");
    hipLaunchKernelGGL(pack_conv3x3_to3_train_kernel, dim3(grid_for(27ll * cin)), dim3(256), 0, (hipStream_t)stream, w_oihw, cin, fwd, dgrad_c3);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

static int to3_chunks(int h, int w) { const long long plane = (long long)h * w; return (int)((plane + 16383) / 16384); }

// ws = [bias partials n*3*chunks][64: zero bias for the data-gradient conv][tmp weight gradient cin*27][first-layer wgrad ws]
extern "C" size_t vad_conv3x3_to3_bwd_ws_floats(int n, int h, int w, int cin) {
    if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cin % 32) return 0;
    return (size_t)n * 3 * to3_chunks(h, w) + 64 + (size_t)cin * 27 + vad_conv_c3_wgrad_ws_floats(n, h, cin);
}

extern "C" int vad_conv3x3_to3_tanh_bwd(const float* in_nhwc, const float* recon, const float* x, const float* drecon,
                                        const float* w_dgrad_c3, float* dpre, float* din, float* dw, float* db3, float* ws,
                                        int n, int h, int w, int cin, float grad_mul, void* stream) {
    VAD_REQUIRE(in_nhwc && recon && (x || drecon) && w_dgrad_c3 && dpre && din && dw && db3 && ws, "conv3x3_to3_tanh_bwd: null pointer");
    VAD_REQUIRE(vad_is_pow2f(grad_mul), "conv3x3_to3_tanh_bwd: grad_mul=%g must be a power of two", (double)grad_mul);
    VAD_REQUIRE(n > 0 && h > 0 && w > 0 && cin == 32, "conv3x3_to3_tanh_bwd: bad shape (the reference's last conv has 32 input channels)");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = to3_chunks(h, w);
    float* parts = ws;
    float* zero_bias = ws + (size_t)n * 3 * chunks;
    float* tmp = zero_bias + 64;
    float* wws = tmp + (size_t)cin * 27;
    VAD_HIP_TRY(hipMemsetAsync(zero_bias, 0, 64 * sizeof(float), s));
    const double count = (double)n * 3.0 * h * w;
    hipLaunchKernelGGL(tanh_bwd_planes_kernel, dim3(chunks, n * 3), dim3(256), 0, s, recon, x, drecon, (float)(2.0 / count) * grad_mul, grad_mul, dpre, parts,
                       (long long)h * w, chunks);
    VAD_LAUNCH_CHECK();
    hipLaunchKernelGGL(bias3_finalize_kernel, dim3(3), dim3(64), 0, s, (const float*)parts, n, chunks, db3);
    VAD_LAUNCH_CHECK();
    int rc = vad_conv3x3_c3(dpre, w_dgrad_c3, zero_bias, din, n, h, w, cin, VAD_ACT_NONE, 0, stream);      // data gradient
    if (rc != VAD_OK) return rc;
    rc = vad_conv_c3_wgrad(dpre, in_nhwc, tmp, wws, n, h, w, cin, stream);                                  // swapped-role weight gradient
    if (rc != VAD_OK) return rc;
    hipLaunchKernelGGL(mirror_last_wgrad_kernel, dim3(grid_for(27ll * cin)), dim3(256), 0, s, (const float*)tmp, cin, dw);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}
