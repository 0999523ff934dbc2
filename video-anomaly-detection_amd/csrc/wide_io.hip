// Models built with in_channels > 3 (models/autoencoder.py:161, models/video_autoencoder.py:290-296 take any width; every
// reference call site passes 3).  The first and last layers of the 3-plane path are kernels of their own (K = 27 first layer,
// Cout = 3 tails fused with the score); a wider model runs those two layers on the GENERIC kernels instead - the frames
// are copied once into a zero-padded NHWC tensor, the last convolution writes its (zero-padded) pre-activation planes - and
// the two small kernels below do what is left: the layout change on the way in, Tanh + squared error + per-frame partial
// sums (+ reconstruction / error map) on the way out.  HBM-bound element-wise work; no claim on the headline path.
#include <hip/hip_runtime.h>

#include "vad_common.h"

namespace {

// out[n][y][x][cpad] = x[n][c][y][x] for c < C, 0 for the padded channels.  One thread per (pixel, 4-channel group).
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* x, float* out, long long npix_total, int plane, int c, int cpad) {
    const int groups = cpad / 4;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= npix_total * groups) return;
    const int g = (int)(idx % groups);
    const long long pix = idx / groups;
    const long long n = pix / plane;
    const int q = (int)(pix - n * plane);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ch = 4 * g + e;
        if (ch < c) v[e] = x[((size_t)n * c + ch) * plane + q];
    }
    *(f32x4*)(out + (size_t)pix * cpad + 4 * g) = v;
}

struct WideScoreP {
    const float* pre;       // [N][H][W][cpad]: pre-activation of the last layer
    const float* x;         // [frames][C][H][W]
    float* partials;        // [N][nparts]
    float* recon;           // [N][C][H][W] or NULL
    float* errmap;          // [N][H][W] or NULL
    int plane, c, cpad, nparts, xt, xs;
};

// One work-group per 256 pixels of one frame: recon = tanh(pre), e = sum_c (recon - x)^2; the group's sum is one partial.
__global__ __launch_bounds__(256) void tanh_score_nhwc_kernel(WideScoreP p) {
    __shared__ float red[4];
    const int n = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    const int nx = p.xt ? (n / p.xt) * p.xs + n % p.xt : n;          // source frame this activation frame is scored against
    float e = 0.f;
    if (q < p.plane) {
        const float* pr = p.pre + ((size_t)n * p.plane + q) * p.cpad;
        for (int c4 = 0; c4 < p.c; c4 += 4) {
            const f32x4 v = *(const f32x4*)(pr + c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = c4 + j;
                if (ch < p.c) {
                    const float r = vad_tanh(v[j]);
                    const float d = r - p.x[((size_t)nx * p.c + ch) * p.plane + q];
                    e = fmaf(d, d, e);
                    if (p.recon) p.recon[((size_t)n * p.c + ch) * p.plane + q] = r;
                }
            }
        }
        if (p.errmap) p.errmap[(size_t)n * p.plane + q] = e / (float)p.c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) p.partials[(size_t)n * p.nparts + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

int vad_nchw_to_nhwc_pad(const float* x, float* out, long long n, int h, int w, int c, int cpad, void* stream) {
    VAD_REQUIRE(x && out && n > 0 && h > 0 && w > 0 && c > 0 && cpad >= c && cpad % 4 == 0, "nchw_to_nhwc_pad: bad arguments");
    const long long total = n * h * w * (cpad / 4);
    VAD_REQUIRE((total + 255) / 256 < (1ll << 31), "nchw_to_nhwc_pad: grid too large");
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, n * h * w, h * w, c, cpad);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

int vad_wide_score_partials(int h, int w) { return (h > 0 && w > 0) ? (h * w + 255) / 256 : 0; }

// t == clip_stride (or t == 0): activation frame n is scored against source frame n; else frame (n / t) * clip_stride + n % t
int vad_tanh_score_nhwc(const float* pre, int cpad, const float* x, int c, float* partials, float* recon, float* errmap,
                        int n, int h, int w, int t, int clip_stride, void* stream) {
    VAD_REQUIRE(pre && x && partials && n > 0 && h > 0 && w > 0 && c > 0 && cpad >= c && cpad % 4 == 0, "tanh_score_nhwc: bad arguments");
    WideScoreP p{pre, x, partials, recon, errmap, h * w, c, cpad, vad_wide_score_partials(h, w), (t == clip_stride) ? 0 : t, clip_stride};
    hipLaunchKernelGGL(tanh_score_nhwc_kernel, dim3((unsigned)p.nparts, (unsigned)n), dim3(256), 0, (hipStream_t)stream, p);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}
