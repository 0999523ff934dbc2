// SSIM / MSE criteria of the reference's utils/losses.py as one fused pass, gfx950.
//
//   SSIMLoss.forward   (utils/losses.py:51-93): five depthwise Gaussian blurs (window w, sigma 1.5, zero padding w/2) of
//                      pred, target, pred^2, target^2, pred*target -> local means / variances / covariance -> SSIM map ->
//                      1 - mean(map)
//   CombinedLoss       (utils/losses.py:96-121): (1 - alpha) * mean((pred-target)^2) + alpha * SSIMLoss
//
// The reference runs five F.conv2d calls with the 2-D window (an outer product g (x) g) plus ~12 elementwise passes; here
// a work-group owns a 32x32 tile of one (frame, channel) plane: the (32+2r)^2 halo of pred and target is read ONCE into
// LDS, the five products are blurred horizontally into LDS (g applied along x), then vertically in registers (g along y),
// the SSIM value and the squared error of the centre pixel are formed in registers and reduced to one partial per
// work-group (wave butterfly + LDS).  HBM traffic = 8 B per element (+ halo overlap), no intermediate map is written.
// A second single-block kernel adds the partials in a fixed order (deterministic) and forms the three scalars.
#include <hip/hip_runtime.h>

#include "vad_common.h"

namespace {

constexpr int ST = 32;          // output tile edge
constexpr int SR_MAX = 7;       // window radius limit (window_size <= 15; the reference's default is 11)
constexpr int SH = ST + 2 * SR_MAX;

struct SsimP {
    const float* pred; const float* target;
    float* parts;               // [2][nblocks]: SSIM-map sums, squared-error sums
    int h, w, r, tiles_x, tiles_y;
    unsigned nblocks;
    float g[2 * SR_MAX + 1];
};

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void ssim_partials_kernel(SsimP a) {
    __shared__ float sp[SH * (SH + 1)], st[SH * (SH + 1)];
    __shared__ float hz[5][SH * (ST + 1)];
    __shared__ float red[2][4];
    const int tid = threadIdx.x, r = a.r, hh = ST + 2 * r, pitch = hh + 1;
    unsigned b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const size_t plane = (size_t)(b / a.tiles_y) * a.h * a.w;
    const int y0 = ty * ST - r, x0 = tx * ST - r;

    // halo tile, zero outside the image (= F.conv2d's zero padding)
    for (int i = tid; i < hh * hh; i += 256) {
        const int yy = i / hh, xx = i - yy * hh;
        const int y = y0 + yy, x = x0 + xx;
        float p = 0.f, t = 0.f;
        if (y >= 0 && y < a.h && x >= 0 && x < a.w) {
            p = a.pred[plane + (size_t)y * a.w + x];
            t = a.target[plane + (size_t)y * a.w + x];
        }
        sp[yy * pitch + xx] = p;
        st[yy * pitch + xx] = t;
    }
    __syncthreads();

    // horizontal pass: hh rows x 32 columns x 5 quantities
    for (int i = tid; i < hh * ST; i += 256) {
        const int yy = i >> 5, xx = i & 31;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
        for (int k = 0; k <= 2 * r; ++k) {
            const float g = a.g[k], p = sp[yy * pitch + xx + k], t = st[yy * pitch + xx + k];
            m0 = fmaf(g, p, m0);
            m1 = fmaf(g, t, m1);
            m2 = fmaf(g, p * p, m2);
            m3 = fmaf(g, t * t, m3);
            m4 = fmaf(g, p * t, m4);
        }
        const int o = yy * (ST + 1) + xx;
        hz[0][o] = m0; hz[1][o] = m1; hz[2][o] = m2; hz[3][o] = m3; hz[4][o] = m4;
    }
    __syncthreads();

    // vertical pass + SSIM + squared error; thread -> column tid&31, rows (tid>>5) + 8j
    const int xx = tid & 31;
    float ssum = 0.f, esum = 0.f;
    constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;      // utils/losses.py:82-83
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = (tid >> 5) + 8 * j;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
        for (int k = 0; k <= 2 * r; ++k) {
            const float g = a.g[k];
            const int o = (yy + k) * (ST + 1) + xx;
            m0 = fmaf(g, hz[0][o], m0);
            m1 = fmaf(g, hz[1][o], m1);
            m2 = fmaf(g, hz[2][o], m2);
            m3 = fmaf(g, hz[3][o], m3);
            m4 = fmaf(g, hz[4][o], m4);
        }
        const float mpp = m0 * m0, mtt = m1 * m1, mpt = m0 * m1;
        const float num = (2.f * mpt + C1) * (2.f * (m4 - mpt) + C2);
        const float den = (mpp + mtt + C1) * ((m2 - mpp) + (m3 - mtt) + C2);
        const bool in = (ty * ST + yy) < a.h && (tx * ST + xx) < a.w;
        const float d = sp[(yy + r) * pitch + xx + r] - st[(yy + r) * pitch + xx + r];
        if (in) { ssum += num / den; esum = fmaf(d, d, esum); }
    }
    ssum = wave_sum64(ssum);
    esum = wave_sum64(esum);
    if ((tid & 63) == 0) { red[0][tid >> 6] = ssum; red[1][tid >> 6] = esum; }
    __syncthreads();
    if (tid == 0) {
        a.parts[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        a.parts[a.nblocks + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// ---------------------------------------------------------------------------------------------------- backward
// d(1 - mean S)/d pred.  Zero-padded Gaussian blurs are self-adjoint, so with the blurred moments mu_p, mu_t, m_pp = G*p^2,
// m_tt, m_pt and  S = A1*A2/(B1*B2),  A1 = 2 mu_p mu_t + C1,  A2 = 2 (m_pt - mu_p mu_t) + C2,  B1 = mu_p^2 + mu_t^2 + C1,
// B2 = (m_pp - mu_p^2) + (m_tt - mu_t^2) + C2:
//     dS/dm_pp = -S/B2      dS/dm_pt = 2 A1/(B1 B2)      dS/dmu_p = 2 mu_t (A2 - A1)/(B1 B2) - 2 mu_p S/B1 + 2 mu_p S/B2
//     d loss/d p = -(1/N) [ G*(dS/dmu_p) + 2 p G*(dS/dm_pp) + t G*(dS/dm_pt) ]
// (checked against autograd in float64 to 1e-18).  Pass 1 recomputes the moments exactly like the forward and writes the
// three adjoint maps, planar [plane][3][H][W]; pass 2 blurs them and combines with p, t, the MSE term and the upstream
// gradient (a device scalar: no host synchronisation).
__global__ __launch_bounds__(256) void ssim_adjoint_kernel(SsimP a, float* adj) {
    __shared__ float sp[SH * (SH + 1)], st[SH * (SH + 1)];
    __shared__ float hz[5][SH * (ST + 1)];
    const int tid = threadIdx.x, r = a.r, hh = ST + 2 * r, pitch = hh + 1;
    unsigned b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const size_t pl = b / a.tiles_y, plane = pl * a.h * a.w;
    const int y0 = ty * ST - r, x0 = tx * ST - r;
    for (int i = tid; i < hh * hh; i += 256) {
        const int yy = i / hh, xx = i - yy * hh;
        const int y = y0 + yy, x = x0 + xx;
        float p = 0.f, t = 0.f;
        if (y >= 0 && y < a.h && x >= 0 && x < a.w) {
            p = a.pred[plane + (size_t)y * a.w + x];
            t = a.target[plane + (size_t)y * a.w + x];
        }
        sp[yy * pitch + xx] = p;
        st[yy * pitch + xx] = t;
    }
    __syncthreads();
    for (int i = tid; i < hh * ST; i += 256) {
        const int yy = i >> 5, xx = i & 31;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
        for (int k = 0; k <= 2 * r; ++k) {
            const float g = a.g[k], p = sp[yy * pitch + xx + k], t = st[yy * pitch + xx + k];
            m0 = fmaf(g, p, m0); m1 = fmaf(g, t, m1); m2 = fmaf(g, p * p, m2); m3 = fmaf(g, t * t, m3); m4 = fmaf(g, p * t, m4);
        }
        const int o = yy * (ST + 1) + xx;
        hz[0][o] = m0; hz[1][o] = m1; hz[2][o] = m2; hz[3][o] = m3; hz[4][o] = m4;
    }
    __syncthreads();
    const int xx = tid & 31;
    constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = (tid >> 5) + 8 * j;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
        for (int k = 0; k <= 2 * r; ++k) {
            const float g = a.g[k];
            const int o = (yy + k) * (ST + 1) + xx;
            m0 = fmaf(g, hz[0][o], m0); m1 = fmaf(g, hz[1][o], m1); m2 = fmaf(g, hz[2][o], m2); m3 = fmaf(g, hz[3][o], m3); m4 = fmaf(g, hz[4][o], m4);
        }
        const int y = ty * ST + yy, x = tx * ST + xx;
        if (y < a.h && x < a.w) {
            const float mpp = m0 * m0, mtt = m1 * m1, mpt = m0 * m1;
            const float A1 = 2.f * mpt + C1, A2 = 2.f * (m4 - mpt) + C2;
            const float B1 = mpp + mtt + C1, B2 = (m2 - mpp) + (m3 - mtt) + C2;
            const float ib = 1.f / (B1 * B2), S = A1 * A2 * ib;
            const size_t o = (pl * 3) * (size_t)a.h * a.w + (size_t)y * a.w + x, hwp = (size_t)a.h * a.w;
            adj[o] = 2.f * m1 * (A2 - A1) * ib - 2.f * m0 * S / B1 + 2.f * m0 * S / B2;      // dS/dmu_p
            adj[o + hwp] = -S / B2;                                                           // dS/dm_pp
            adj[o + 2 * hwp] = 2.f * A1 * ib;                                                 // dS/dm_pt
        }
    }
}

// grad = gout * [ (1-alpha) * 2 (p - t)/N  -  alpha/N * ( G*adj0 + 2 p G*adj1 + t G*adj2 ) ]
__global__ __launch_bounds__(256) void ssim_grad_kernel(SsimP a, const float* adj, const float* gout, float alpha, float inv_n,
                                                        float* grad) {
    __shared__ float sa[3][SH * (SH + 1)];
    __shared__ float hz[3][SH * (ST + 1)];
    const int tid = threadIdx.x, r = a.r, hh = ST + 2 * r, pitch = hh + 1;
    unsigned b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const size_t pl = b / a.tiles_y, plane = pl * a.h * a.w, hwp = (size_t)a.h * a.w;
    const int y0 = ty * ST - r, x0 = tx * ST - r;
    for (int i = tid; i < hh * hh; i += 256) {
        const int yy = i / hh, xx = i - yy * hh;
        const int y = y0 + yy, x = x0 + xx;
        const bool in = y >= 0 && y < a.h && x >= 0 && x < a.w;
        const size_t o = pl * 3 * hwp + (size_t)y * a.w + x;
#pragma unroll
        for (int k = 0; k < 3; ++k) sa[k][yy * pitch + xx] = in ? adj[o + k * hwp] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < hh * ST; i += 256) {
        const int yy = i >> 5, xx = i & 31;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
        for (int k = 0; k <= 2 * r; ++k) {
            const float g = a.g[k];
            m0 = fmaf(g, sa[0][yy * pitch + xx + k], m0); m1 = fmaf(g, sa[1][yy * pitch + xx + k], m1); m2 = fmaf(g, sa[2][yy * pitch + xx + k], m2);
        }
        const int o = yy * (ST + 1) + xx;
        hz[0][o] = m0; hz[1][o] = m1; hz[2][o] = m2;
    }
    __syncthreads();
    const int xx = tid & 31;
    const float go = gout[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int yy = (tid >> 5) + 8 * j;
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
        for (int k = 0; k <= 2 * r; ++k) {
            const float g = a.g[k];
            const int o = (yy + k) * (ST + 1) + xx;
            m0 = fmaf(g, hz[0][o], m0); m1 = fmaf(g, hz[1][o], m1); m2 = fmaf(g, hz[2][o], m2);
        }
        const int y = ty * ST + yy, x = tx * ST + xx;
        if (y < a.h && x < a.w) {
            const size_t o = plane + (size_t)y * a.w + x;
            const float p = a.pred[o], t = a.target[o];
            grad[o] = go * inv_n * ((1.f - alpha) * 2.f * (p - t) - alpha * (m0 + 2.f * p * m1 + t * m2));
        }
    }
}

// out[0] = 1 - mean(SSIM map), out[1] = mean squared error, out[2] = (1-alpha)*out[1] + alpha*out[0]
__global__ __launch_bounds__(256) void ssim_finalize_kernel(const float* parts, unsigned n, double count, float alpha, float* out) {
    __shared__ double red[2][4];
    double s = 0.0, e = 0.0;
    for (unsigned i = threadIdx.x; i < n; i += 256) { s += (double)parts[i]; e += (double)parts[n + i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); e += __shfl_xor(e, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ss = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        const double ee = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const float l_ssim = (float)(1.0 - ss / count), l_mse = (float)(ee / count);
        out[0] = l_ssim;
        out[1] = l_mse;
        out[2] = (1.f - alpha) * l_mse + alpha * l_ssim;
    }
}

}  // namespace

extern "C" size_t vad_ssim_workspace_floats(long long planes, int h, int w) {
    if (planes <= 0 || h <= 0 || w <= 0) return 0;
    const long long nb = planes * ((h + ST - 1) / ST) * ((w + ST - 1) / ST);
    return nb < (1ll << 31) ? (size_t)(2 * nb) : 0;
}

extern "C" int vad_ssim_mse(const float* pred, const float* target, long long planes, int h, int w, int window_size,
                            float alpha, float* workspace, float* out3, void* stream) {
    VAD_REQUIRE(pred && target && workspace && out3, "ssim_mse: null pointer");
    VAD_REQUIRE(planes > 0 && h > 0 && w > 0, "ssim_mse: bad shape");
    VAD_REQUIRE(window_size >= 1 && (window_size & 1) && window_size <= 2 * SR_MAX + 1,
                "ssim_mse: window_size=%d must be odd and at most %d", window_size, 2 * SR_MAX + 1);
    SsimP a{};
    a.pred = pred; a.target = target; a.parts = workspace;
    a.h = h; a.w = w; a.r = window_size / 2;
    a.tiles_x = (w + ST - 1) / ST; a.tiles_y = (h + ST - 1) / ST;
    const long long nb = planes * a.tiles_x * a.tiles_y;
    VAD_REQUIRE(nb < (1ll << 31), "ssim_mse: grid too large");
    a.nblocks = (unsigned)nb;
    // 1-D Gaussian, sigma 1.5, normalised in fp32 like utils/losses.py:36-40
    float g[2 * SR_MAX + 1], gs = 0.f;
    for (int k = 0; k < window_size; ++k) {
        const float c = (float)(k - window_size / 2);
        g[k] = expf(-(c * c) / (2.f * 1.5f * 1.5f));
        gs += g[k];
    }
    for (int k = 0; k < 2 * SR_MAX + 1; ++k) a.g[k] = k < window_size ? g[k] / gs : 0.f;
    hipLaunchKernelGGL(ssim_partials_kernel, dim3(a.nblocks), dim3(256), 0, (hipStream_t)stream, a);
    VAD_LAUNCH_CHECK();
    hipLaunchKernelGGL(ssim_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, a.nblocks,
                       (double)planes * h * w, alpha, out3);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" size_t vad_ssim_grad_workspace_floats(long long planes, int h, int w) {
    if (planes <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)planes * 3 * h * w;
}

static int ssim_setup(SsimP& a, const float* pred, const float* target, long long planes, int h, int w, int window_size) {
    VAD_REQUIRE(planes > 0 && h > 0 && w > 0, "ssim: bad shape");
    VAD_REQUIRE(window_size >= 1 && (window_size & 1) && window_size <= 2 * SR_MAX + 1,
                "ssim: window_size=%d must be odd and at most %d", window_size, 2 * SR_MAX + 1);
    a.pred = pred; a.target = target; a.parts = nullptr;
    a.h = h; a.w = w; a.r = window_size / 2;
    a.tiles_x = (w + ST - 1) / ST; a.tiles_y = (h + ST - 1) / ST;
    const long long nb = planes * a.tiles_x * a.tiles_y;
    VAD_REQUIRE(nb < (1ll << 31), "ssim: grid too large");
    a.nblocks = (unsigned)nb;
    float g[2 * SR_MAX + 1], gs = 0.f;
    for (int k = 0; k < window_size; ++k) {
        const float c = (float)(k - window_size / 2);
        g[k] = expf(-(c * c) / (2.f * 1.5f * 1.5f));
        gs += g[k];
    }
    for (int k = 0; k < 2 * SR_MAX + 1; ++k) a.g[k] = k < window_size ? g[k] / gs : 0.f;
    return VAD_OK;
}

extern "C" int vad_ssim_mse_backward(const float* pred, const float* target, long long planes, int h, int w, int window_size,
                                     float alpha, const float* grad_out, float* workspace, float* grad_pred, void* stream) {
    VAD_REQUIRE(pred && target && grad_out && workspace && grad_pred, "ssim_mse_backward: null pointer");
    SsimP a{};
    const int rc = ssim_setup(a, pred, target, planes, h, w, window_size);
    if (rc != VAD_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ssim_adjoint_kernel, dim3(a.nblocks), dim3(256), 0, s, a, workspace);
    VAD_LAUNCH_CHECK();
    hipLaunchKernelGGL(ssim_grad_kernel, dim3(a.nblocks), dim3(256), 0, s, a, (const float*)workspace, grad_out, alpha,
                       (float)(1.0 / ((double)planes * h * w)), grad_pred);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}
