// One optimisation step of the convolutional image autoencoder (reference train.py:28-52: model.train(); out = model(x);
// loss = criterion(out, x); zero_grad; backward; Adam.step, criterion = nn.MSELoss (default) | SSIMLoss | CombinedLoss,
// train.py:149-158) as an explicit launch sequence over the kernels of train_ops.hip, ssim.hip and the forward convolution
// kernels.  Host orchestration only; the video counterpart is train_step.hip, whose conventions (flat parameter / gradient
// buffers in torch layouts and named_parameters() order, running statistics in module order, exact zeros for the bias
// gradients of convolutions that feed a batch-statistics BatchNorm) apply unchanged.
//
// Parameter order (models/autoencoder.py:38-79, 103-139), channels c = 3,32,64,128,latent and d = latent,128,64,32:
//   encoder.enc{i}, i = 1..4:  conv_a w (c_i,c_{i-1},3,3), b; BN_a gamma, beta; conv_b w (c_i,c_i,3,3), b; BN_b gamma, beta
//   decoder.dec{j}, j = 1..3:  convT w (d_{j-1},d_j,2,2), b; BN gamma, beta; conv w (d_j,d_j,3,3), b; BN gamma, beta
//   decoder.dec4:              convT w (32,32,2,2), b; BN gamma, beta; conv w (3,32,3,3), b
#include <hip/hip_runtime.h>

#include "vad_common.h"

namespace {

struct ImgPlan {
    int N, H, W, L;
    int c[5], d[5];
    // parameters (floats)
    size_t ea_w[4], ea_b[4], ea_g[4], ea_be[4], eb_w[4], eb_b[4], eb_g[4], eb_be[4];
    size_t dt_w[4], dt_b[4], dt_g[4], dt_be[4], dc_w[3], dc_b[3], dc_g[3], dc_be[3], last_w, last_b, nparams;
    size_t rs_ea[4], rs_eb[4], rs_dt[4], rs_dc[3], nstats;
    // workspace (floats)
    size_t pk_ea[4], pk_ea_dg[4], pk_eb[4], pk_eb_dg[4], pk_dt[4], pk_dt_dg[4], pk_dc[3], pk_dc_dg[3], pk_last, pk_last_dg;
    size_t ya[4], a[4], yb[4], p[4], st_ea[4], st_eb[4];
    size_t ut[4], rt[4], yc[3], rc[3], st_dt[4], st_dc[3];
    size_t recon, dpre, drecon, parts, g[2], ksums, zeros, ones, out3, chan_ws, wgrad_ws, to3_ws, ssim_ws;
    size_t ws_floats;
};

size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }

bool make_plan(ImgPlan& p, int N, int H, int W, int L) {
    if (N <= 0 || H <= 0 || W <= 0 || H % 16 || W % 16 || L <= 0 || L % 32 || L > 1024 || N > (1 << 20)) return false;
    p.N = N; p.H = H; p.W = W; p.L = L;
    const int c[5] = {3, 32, 64, 128, L}, d[5] = {L, 128, 64, 32, 32};
    for (int i = 0; i < 5; ++i) { p.c[i] = c[i]; p.d[i] = d[i]; }
    size_t o = 0, rs = 0;
    for (int i = 0; i < 4; ++i) {
        const int ci = c[i], co = c[i + 1];
        p.ea_w[i] = o; o += (size_t)co * ci * 9; p.ea_b[i] = o; o += co; p.ea_g[i] = o; o += co; p.ea_be[i] = o; o += co;
        p.eb_w[i] = o; o += (size_t)co * co * 9; p.eb_b[i] = o; o += co; p.eb_g[i] = o; o += co; p.eb_be[i] = o; o += co;
        p.rs_ea[i] = rs; rs += 2 * (size_t)co; p.rs_eb[i] = rs; rs += 2 * (size_t)co;
    }
    for (int j = 0; j < 4; ++j) {
        const int ci = d[j], co = d[j + 1];
        p.dt_w[j] = o; o += (size_t)ci * co * 4; p.dt_b[j] = o; o += co; p.dt_g[j] = o; o += co; p.dt_be[j] = o; o += co;
        p.rs_dt[j] = rs; rs += 2 * (size_t)co;
        if (j < 3) {
            p.dc_w[j] = o; o += (size_t)co * co * 9; p.dc_b[j] = o; o += co; p.dc_g[j] = o; o += co; p.dc_be[j] = o; o += co;
            p.rs_dc[j] = rs; rs += 2 * (size_t)co;
        }
    }
    p.last_w = o; o += 3 * 32 * 9; p.last_b = o; o += 3;
    p.nparams = o; p.nstats = rs;

    size_t w = 0, max_act = 0, max_chan = 0, max_wgrad = 0;
    auto take = [&](size_t n) { const size_t at = w; w += align64(n); return at; };
    auto chan = [&](long long npix, int cc) { const size_t v = vad_chan_ws_floats(npix, cc); if (v > max_chan) max_chan = v; };
    auto wg = [&](size_t v) { if (v > max_wgrad) max_wgrad = v; };
    const size_t n = (size_t)N;
    for (int i = 0; i < 4; ++i) {
        const int ci = c[i], co = c[i + 1], hi = H >> i, wi = W >> i;
        // (3x3 operand slots hold either form of a layer: direct, 9 taps, or Winograd, 16 - VAD_PREC_WINO)
        p.pk_ea[i] = take(i == 0 ? vad_pack_conv3x3_c3_floats(co) : vad_pack_conv3x3_wino_floats(co, ci));
        p.pk_ea_dg[i] = i == 0 ? 0 : take(vad_pack_conv3x3_wino_floats(ci, co));
        p.pk_eb[i] = take(vad_pack_conv3x3_wino_floats(co, co));
        p.pk_eb_dg[i] = take(vad_pack_conv3x3_wino_floats(co, co));
        const size_t sz = n * hi * wi * co;
        if (sz > max_act) max_act = sz;
        p.ya[i] = take(sz); p.a[i] = take(sz); p.yb[i] = take(sz); p.p[i] = take(sz / 4);
        p.st_ea[i] = take(2 * (size_t)co); p.st_eb[i] = take(2 * (size_t)co);
        chan((long long)n * hi * wi, co);
        wg(i == 0 ? vad_conv_c3_wgrad_ws_floats(N, hi, co) : vad_conv_wgrad_ws_floats(N, hi, 9, ci, co));
        wg(vad_conv_wgrad_ws_floats(N, hi, 9, co, co));
    }
    for (int j = 0; j < 4; ++j) {
        const int ci = d[j], co = d[j + 1], hj = (H / 16) << j, wj = (W / 16) << j;
        p.pk_dt[j] = take(vad_pack_convt2x2_floats(ci, co));
        p.pk_dt_dg[j] = take(vad_pack_conv1x1_floats(ci, 4 * co));
        const size_t sz = n * (size_t)(2 * hj) * (2 * wj) * co;
        if (sz > max_act) max_act = sz;
        p.ut[j] = take(sz); p.rt[j] = take(sz); p.st_dt[j] = take(2 * (size_t)co);
        chan((long long)n * 4 * hj * wj, co);
        wg(vad_conv_wgrad_ws_floats(N, hj, 1, ci, 4 * co));
        if (j < 3) {
            p.pk_dc[j] = take(vad_pack_conv3x3_wino_floats(co, co));
            p.pk_dc_dg[j] = take(vad_pack_conv3x3_wino_floats(co, co));
            p.yc[j] = take(sz); p.rc[j] = take(sz); p.st_dc[j] = take(2 * (size_t)co);
            wg(vad_conv_wgrad_ws_floats(N, 2 * hj, 9, co, co));
        }
    }
    p.pk_last = take(vad_pack_conv3x3_to3_floats(32));
    p.pk_last_dg = take(vad_pack_conv3x3_c3_floats(32));
    const size_t img = n * 3 * (size_t)H * W;
    p.recon = take(img); p.dpre = take(img); p.drecon = take(img);
    p.parts = take(n * (size_t)vad_score_partials(0, H, W));
    p.g[0] = take(max_act); p.g[1] = take(max_act);
    p.ksums = take(2 * 1024); p.zeros = take(1024); p.ones = take(64); p.out3 = take(64);
    p.chan_ws = take(max_chan);
    p.wgrad_ws = take(max_wgrad);
    p.to3_ws = take(vad_conv3x3_to3_bwd_ws_floats(N, H, W, 32));
    const size_t sf = vad_ssim_workspace_floats((long long)N * 3, H, W), sb = vad_ssim_grad_workspace_floats((long long)N * 3, H, W);
    p.ssim_ws = take(sf > sb ? sf : sb);
    p.ws_floats = w;
    return true;
}

// loss = sum(parts)/count in a fixed order (the partials of the scoring tail = per-tile sums of squared error)
__global__ __launch_bounds__(256) void mse_from_partials_kernel(const float* parts, long long nparts, double count, float* loss) {
    __shared__ double red[4];
    double s = 0.0;
    for (long long i = threadIdx.x; i < nparts; i += 256) s += (double)parts[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / count);
}

__global__ void set_scalars_kernel(float* ones, const float* out3, int which, float* loss) {
    if (threadIdx.x == 0) { ones[0] = 1.f; if (out3) loss[0] = out3[which]; }
}

}  // namespace

#define TRY(expr)                      \
    do {                               \
        const int rc_ = (expr);        \
        if (rc_ != VAD_OK) return rc_; \
    } while (0)

extern "C" size_t vad_img_train_nparams(int latent) { ImgPlan p; return make_plan(p, 1, 16, 16, latent) ? p.nparams : 0; }
extern "C" size_t vad_img_train_nstats(int latent) { ImgPlan p; return make_plan(p, 1, 16, 16, latent) ? p.nstats : 0; }
extern "C" size_t vad_img_train_workspace_bytes(int n, int h, int w, int latent) {
    ImgPlan p;
    return make_plan(p, n, h, w, latent) ? p.ws_floats * sizeof(float) : 0;
}

// loss_kind: 0 = nn.MSELoss (train.py default), 1 = SSIMLoss(window), 2 = CombinedLoss(alpha, window)
extern "C" int vad_img_train_fwd_bwd(const float* x, int n, int h, int w, int latent, const float* params, float* grads,
                                     float* running, void* workspace, size_t workspace_bytes, int loss_kind, float alpha,
                                     int window_size, int precision, float* loss, float* recon_out, void* stream) {
    VAD_REQUIRE(x && params && grads && workspace && loss, "img_train_fwd_bwd: null pointer");
    // precision VAD_PREC_SPLIT: the 3x3 / transposed convolutions (forward + data gradients) on split-fp16 operands, as in
    // the video step; first layer, last layer, weight gradients, BatchNorm, criterion and Adam stay fp32
    // VAD_PREC_WINO: fp32 everywhere, the 3x3 convolutions behind the first layer (forward + data gradients) as Winograd F(2x2,3x3)
    VAD_REQUIRE((precision >= VAD_PREC_FP32 && precision <= VAD_PREC_BF16) || precision == VAD_PREC_WINO,
                "img_train_fwd_bwd: precision=%d must be 0 (fp32), 1 (split fp16), 2 (bf16 operands) or 4 (Winograd)", precision);
    const bool wino = precision == VAD_PREC_WINO;
    const int pack_prec = precision;
    if (wino) precision = VAD_PREC_FP32;
    VAD_REQUIRE(loss_kind >= 0 && loss_kind <= 2, "img_train_fwd_bwd: loss_kind must be 0 (mse), 1 (ssim) or 2 (combined)");
    ImgPlan p;
    VAD_REQUIRE(make_plan(p, n, h, w, latent), "img_train_fwd_bwd: unsupported configuration (N=%d %dx%d latent=%d): H, W multiples "
                "of 16, latent a multiple of 32", n, h, w, latent);
    if (workspace_bytes < p.ws_floats * sizeof(float))
        return vad_fail(VAD_ERR_WS, "img_train_fwd_bwd: workspace %zu bytes < %zu needed", workspace_bytes, p.ws_floats * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const float* P = params;
    float* G = grads;
    const int N = p.N, H = p.H, W = p.W;
    const float eps = 1e-5f, mom = 0.1f;
    float *zeros = ws + p.zeros, *g0 = ws + p.g[0], *g2 = ws + p.g[1];
    float* recon = recon_out ? recon_out : ws + p.recon;
    VAD_HIP_TRY(hipMemsetAsync(zeros, 0, 1024 * sizeof(float), s));
    auto rsp = [&](size_t off) { return running ? running + off : (float*)nullptr; };
    auto bn_stats = [&](const float* y, long long npix, int c, size_t st, size_t rs) {
        float* r = rsp(rs);
        return vad_bn_stats(y, npix, c, eps, mom, ws + st, r, r ? r + c : nullptr, ws + p.chan_ws, s);
    };

    // a 3x3 convolution [n][hh][ww][ci] -> [n][hh][ww][co] without activation, in the step's mode
    auto conv3 = [&](const float* in, const float* wpk, const float* bias, float* out, int hh, int ww, int ci, int co) -> int {
        if (wino) return vad_conv3x3_wino(in, 0, wpk, bias, out, 0, N, hh, ww, ci, co, VAD_ACT_NONE, 0, s);
        return vad_conv3x3(in, 0, wpk, bias, out, 0, N, hh, ww, ci, co, VAD_ACT_NONE, 0, precision, s);
    };

    // ---- operand packing of the current parameters
    TRY(vad_train_pack_conv3x3_c3(P + p.ea_w[0], 32, ws + p.pk_ea[0], s));
    for (int i = 0; i < 4; ++i) {
        if (i > 0) TRY(vad_train_pack_conv3x3(P + p.ea_w[i], p.c[i + 1], p.c[i], ws + p.pk_ea[i], ws + p.pk_ea_dg[i], pack_prec, s));
        TRY(vad_train_pack_conv3x3(P + p.eb_w[i], p.c[i + 1], p.c[i + 1], ws + p.pk_eb[i], ws + p.pk_eb_dg[i], pack_prec, s));
    }
    for (int j = 0; j < 4; ++j) {
        TRY(vad_train_pack_convt2x2(P + p.dt_w[j], p.d[j], p.d[j + 1], ws + p.pk_dt[j], ws + p.pk_dt_dg[j], precision, s));
        if (j < 3) TRY(vad_train_pack_conv3x3(P + p.dc_w[j], p.d[j + 1], p.d[j + 1], ws + p.pk_dc[j], ws + p.pk_dc_dg[j], pack_prec, s));
    }
    TRY(vad_train_pack_conv3x3_to3(P + p.last_w, 32, ws + p.pk_last, ws + p.pk_last_dg, s));

    // ================================================================================== forward
    // encoder (models/autoencoder.py:38-79): [conv-BN-LeakyReLU, conv-BN-LeakyReLU, MaxPool2] x 4
    for (int i = 0; i < 4; ++i) {
        const int ci = p.c[i], co = p.c[i + 1], hi = H >> i, wi = W >> i;
        const long long npix = (long long)N * hi * wi;
        if (i == 0) TRY(vad_conv3x3_c3(x, ws + p.pk_ea[0], P + p.ea_b[0], ws + p.ya[0], N, hi, wi, co, VAD_ACT_NONE, 0, s));
        else TRY(conv3(ws + p.p[i - 1], ws + p.pk_ea[i], P + p.ea_b[i], ws + p.ya[i], hi, wi, ci, co));
        TRY(bn_stats(ws + p.ya[i], npix, co, p.st_ea[i], p.rs_ea[i]));
        TRY(vad_bn_act_pool_fwd(ws + p.ya[i], ws + p.st_ea[i], P + p.ea_g[i], P + p.ea_be[i], ws + p.a[i], 0, 0, 0, 0, N, hi, wi, co, VAD_ACT_LEAKY, 0, s));
        TRY(conv3(ws + p.a[i], ws + p.pk_eb[i], P + p.eb_b[i], ws + p.yb[i], hi, wi, co, co));
        TRY(bn_stats(ws + p.yb[i], npix, co, p.st_eb[i], p.rs_eb[i]));
        TRY(vad_bn_act_pool_fwd(ws + p.yb[i], ws + p.st_eb[i], P + p.eb_g[i], P + p.eb_be[i], ws + p.p[i], 0, 0, 0, 0, N, hi, wi, co, VAD_ACT_LEAKY, 1, s));
    }
    // decoder (models/autoencoder.py:103-139): [convT-BN-ReLU, conv-BN-ReLU] x 3, then convT-BN-ReLU, conv(32->3)-Tanh
    for (int j = 0; j < 4; ++j) {
        const int ci = p.d[j], co = p.d[j + 1], hj = (H / 16) << j, wj = (W / 16) << j;
        const float* in = j == 0 ? ws + p.p[3] : ws + p.rc[j - 1];
        const long long npix = (long long)N * 4 * hj * wj;
        TRY(vad_convt2x2(in, 0, ws + p.pk_dt[j], P + p.dt_b[j], ws + p.ut[j], 0, N, hj, wj, ci, co, VAD_ACT_NONE, precision, s));
        TRY(bn_stats(ws + p.ut[j], npix, co, p.st_dt[j], p.rs_dt[j]));
        TRY(vad_bn_act_pool_fwd(ws + p.ut[j], ws + p.st_dt[j], P + p.dt_g[j], P + p.dt_be[j], ws + p.rt[j], 0, 0, 0, 0, N, 2 * hj, 2 * wj, co, VAD_ACT_RELU, 0, s));
        if (j < 3) {
            TRY(conv3(ws + p.rt[j], ws + p.pk_dc[j], P + p.dc_b[j], ws + p.yc[j], 2 * hj, 2 * wj, co, co));
            TRY(bn_stats(ws + p.yc[j], npix, co, p.st_dc[j], p.rs_dc[j]));
            TRY(vad_bn_act_pool_fwd(ws + p.yc[j], ws + p.st_dc[j], P + p.dc_g[j], P + p.dc_be[j], ws + p.rc[j], 0, 0, 0, 0, N, 2 * hj, 2 * wj, co, VAD_ACT_RELU, 0, s));
        }
    }
    const int nparts = vad_score_partials(0, H, W);
    TRY(vad_conv3x3_to3_score(ws + p.rt[3], ws + p.pk_last, P + p.last_b, x, ws + p.parts, recon, nullptr, N, H, W, 32, s));
    // criterion (train.py:149-158) and the gradient of the reconstruction
    const float* drecon = nullptr;
    if (loss_kind == 0) {
        hipLaunchKernelGGL(mse_from_partials_kernel, dim3(1), dim3(256), 0, s, (const float*)(ws + p.parts), (long long)N * nparts,
                           (double)N * 3.0 * H * W, loss);
        VAD_LAUNCH_CHECK();
    } else {
        const float a = loss_kind == 1 ? 1.f : alpha;
        TRY(vad_ssim_mse(recon, x, (long long)N * 3, H, W, window_size, a, ws + p.ssim_ws, ws + p.out3, s));
        hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(64), 0, s, ws + p.ones, (const float*)(ws + p.out3), loss_kind == 1 ? 0 : 2, loss);
        VAD_LAUNCH_CHECK();
        TRY(vad_ssim_mse_backward(recon, x, (long long)N * 3, H, W, window_size, a, ws + p.ones, ws + p.ssim_ws, ws + p.drecon, s));
        drecon = ws + p.drecon;
    }

    // ================================================================================== backward
    // split-fp16 mode: the backward runs on gradients times a power of two (train_step.hip says why), unscaled at the end;
    // the SSIM / combined criteria's gradient is O(1 / count) like the MSE's
    float grad_mul = 1.f;
    if (precision == VAD_PREC_SPLIT && !wino && vad_split_grad_scale_enabled()) {
        int e = 0;
        (void)frexp((double)N * 3.0 * H * W, &e);
        grad_mul = (float)ldexp(1.0, e - 7);
    }
    TRY(vad_conv3x3_to3_tanh_bwd(ws + p.rt[3], recon, drecon ? nullptr : x, drecon, ws + p.pk_last_dg, ws + p.dpre, g0, G + p.last_w, G + p.last_b,
                                 ws + p.to3_ws, N, H, W, 32, grad_mul, s));
    for (int j = 3; j >= 0; --j) {
        const int ci = p.d[j], co = p.d[j + 1], hj = (H / 16) << j, wj = (W / 16) << j;
        if (j < 3) {      // conv-BN-ReLU: g0 = d rc_j -> g2 = d yc_j -> weight gradient, g0 = d rt_j
            TRY(vad_bn_act_pool_bwd(ws + p.yc[j], ws + p.st_dc[j], P + p.dc_g[j], P + p.dc_be[j], g0, 0, 0, 0, 0, g2, 0, G + p.dc_g[j], G + p.dc_be[j],
                                    ws + p.ksums, ws + p.chan_ws, N, 2 * hj, 2 * wj, co, VAD_ACT_RELU, 0, s));
            TRY(vad_conv_wgrad(ws + p.rt[j], g2, G + p.dc_w[j], ws + p.wgrad_ws, N, 2 * hj, 2 * wj, co, co, 9, 0, precision, s));
            VAD_HIP_TRY(hipMemsetAsync(G + p.dc_b[j], 0, (size_t)co * sizeof(float), s));       // structurally zero (train_step.hip)
            TRY(conv3(g2, ws + p.pk_dc_dg[j], zeros, g0, 2 * hj, 2 * wj, co, co));
        }
        // convT-BN-ReLU: g0 = d rt_j -> g2 = d ut_j (space-to-depth) -> weight gradient, g0 = d (input of the convT)
        const float* in = j == 0 ? ws + p.p[3] : ws + p.rc[j - 1];
        TRY(vad_bn_act_pool_bwd(ws + p.ut[j], ws + p.st_dt[j], P + p.dt_g[j], P + p.dt_be[j], g0, 0, 0, 0, 0, g2, 1, G + p.dt_g[j], G + p.dt_be[j],
                                ws + p.ksums, ws + p.chan_ws, N, 2 * hj, 2 * wj, co, VAD_ACT_RELU, 0, s));
        TRY(vad_conv_wgrad(in, g2, G + p.dt_w[j], ws + p.wgrad_ws, N, hj, wj, ci, 4 * co, 1, 1, precision, s));
        VAD_HIP_TRY(hipMemsetAsync(G + p.dt_b[j], 0, (size_t)co * sizeof(float), s));
        TRY(vad_conv1x1(g2, ws + p.pk_dt_dg[j], zeros, g0, (long long)N * hj * wj, 4 * co, ci, s));
    }
    for (int i = 3; i >= 0; --i) {      // g0 = d p_i
        const int ci = p.c[i], co = p.c[i + 1], hi = H >> i, wi = W >> i;
        TRY(vad_bn_act_pool_bwd(ws + p.yb[i], ws + p.st_eb[i], P + p.eb_g[i], P + p.eb_be[i], g0, 0, 0, 0, 0, g2, 0, G + p.eb_g[i], G + p.eb_be[i],
                                ws + p.ksums, ws + p.chan_ws, N, hi, wi, co, VAD_ACT_LEAKY, 1, s));
        TRY(vad_conv_wgrad(ws + p.a[i], g2, G + p.eb_w[i], ws + p.wgrad_ws, N, hi, wi, co, co, 9, 0, precision, s));
        VAD_HIP_TRY(hipMemsetAsync(G + p.eb_b[i], 0, (size_t)co * sizeof(float), s));
        TRY(conv3(g2, ws + p.pk_eb_dg[i], zeros, g0, hi, wi, co, co));          // g0 = d a_i
        TRY(vad_bn_act_pool_bwd(ws + p.ya[i], ws + p.st_ea[i], P + p.ea_g[i], P + p.ea_be[i], g0, 0, 0, 0, 0, g2, 0, G + p.ea_g[i], G + p.ea_be[i],
                                ws + p.ksums, ws + p.chan_ws, N, hi, wi, co, VAD_ACT_LEAKY, 0, s));
        VAD_HIP_TRY(hipMemsetAsync(G + p.ea_b[i], 0, (size_t)co * sizeof(float), s));
        if (i == 0) {
            TRY(vad_conv_c3_wgrad(x, g2, G + p.ea_w[0], ws + p.wgrad_ws, N, hi, wi, co, s));
        } else {
            TRY(vad_conv_wgrad(ws + p.p[i - 1], g2, G + p.ea_w[i], ws + p.wgrad_ws, N, hi, wi, ci, co, 9, 0, precision, s));
            TRY(conv3(g2, ws + p.pk_ea_dg[i], zeros, g0, hi, wi, co, ci));      // g0 = d p_{i-1}
        }
    }
    if (grad_mul != 1.f) TRY(vad_scale_floats(G, (long long)p.nparams, 1.f / grad_mul, s));
    return VAD_OK;
}
