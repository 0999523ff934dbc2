/* vad_oracle.c — CPU restatement of the reference's scoring path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * (video-anomaly-detection_amd/) never does.  Plain C, NCHW fp32 tensors like the reference, one
 * loop nest per torch.nn op, NO BatchNorm folding and NO layout tricks, so that it checks the HIP
 * path's restructurings (NHWC, folded BN, fused pooling, fused gates) instead of sharing them.
 * Sums are accumulated in double and rounded to fp32 once per output element.
 *
 * Pinned by tests/golden/*.npz, which were produced by importing the reference's models/ package
 * (tests/golden/make_golden.py) — see tests/test_oracle.py.
 *
 * Reference lines followed:
 *   models/autoencoder.py:38-79 (Encoder), :103-139 (Decoder), :181-193 (forward), :199-221 (error)
 *   models/video_autoencoder.py:54-85 (ConvLSTMCell), :127-172 (ConvLSTM), :191-215 (VideoEncoder),
 *   :242-261 (VideoDecoder), :329-354 (forward), :356-384 (error)
 *   torch defaults: BatchNorm2d eps 1e-5 (eval: running stats), MaxPool2d floor mode, zero padding.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(n, c, y, x, C, H, W) ((((size_t)(n) * (C) + (c)) * (H) + (y)) * (W) + (x))

/* nn.Conv2d(kernel k, padding k/2, stride 1), weight OIHW, cross-correlation */
void vo_conv2d(const float* x, const float* w, const float* b, float* y,
               int N, int Cin, int H, int W, int Cout, int k) {
    const int pad = k / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < H; ++oy)
                for (int ox = 0; ox < W; ++ox) {
                    double acc = b ? (double)b[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < k; ++ky) {
                            const int iy = oy + ky - pad;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < k; ++kx) {
                                const int ix = ox + kx - pad;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)x[IDX4(n, ci, iy, ix, Cin, H, W)] *
                                       (double)w[(((size_t)co * Cin + ci) * k + ky) * k + kx];
                            }
                        }
                    y[IDX4(n, co, oy, ox, Cout, H, W)] = (float)acc;
                }
}

/* nn.ConvTranspose2d(kernel 2, stride 2), weight IOHW: out[n,co,2i+a,2j+b] = bias + sum_ci x*W[ci,co,a,b] */
void vo_convt2x2(const float* x, const float* w, const float* b, float* y,
                 int N, int Cin, int H, int W, int Cout) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co)
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j)
                    for (int a = 0; a < 2; ++a)
                        for (int bb = 0; bb < 2; ++bb) {
                            double acc = b ? (double)b[co] : 0.0;
                            for (int ci = 0; ci < Cin; ++ci)
                                acc += (double)x[IDX4(n, ci, i, j, Cin, H, W)] *
                                       (double)w[(((size_t)ci * Cout + co) * 2 + a) * 2 + bb];
                            y[IDX4(n, co, 2 * i + a, 2 * j + bb, Cout, 2 * H, 2 * W)] = (float)acc;
                        }
}

/* nn.BatchNorm2d in eval mode, in place */
void vo_batchnorm_eval(float* x, const float* gamma, const float* beta, const float* mean, const float* var,
                       int N, int C, int HW) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const double inv = 1.0 / sqrt((double)var[c] + 1e-5);
            float* p = x + ((size_t)n * C + c) * HW;
            for (int i = 0; i < HW; ++i)
                p[i] = (float)(((double)p[i] - (double)mean[c]) * inv * (double)gamma[c] + (double)beta[c]);
        }
}

void vo_leaky_relu(float* x, size_t n, float slope) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : slope * x[i];
}
void vo_relu(float* x, size_t n) { vo_leaky_relu(x, n, 0.f); }
void vo_tanh(float* x, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) x[i] = (float)tanh((double)x[i]);
}

/* nn.MaxPool2d(2, 2) */
void vo_maxpool2(const float* x, float* y, int N, int C, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < N * C; ++nc)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const float* p = x + ((size_t)nc * H + 2 * oy) * W + 2 * ox;
                float m = p[0];
                if (p[1] > m) m = p[1];
                if (p[W] > m) m = p[W];
                if (p[W + 1] > m) m = p[W + 1];
                y[((size_t)nc * Ho + oy) * Wo + ox] = m;
            }
}

static float* falloc(size_t n) { return (float*)malloc(n * sizeof(float)); }

/* conv -> BN -> activation on a fresh buffer; act: 1 LeakyReLU(0.2), 2 ReLU */
static float* conv_bn_act(const float* x, const float* const* P, int* pi, int N, int Cin, int H, int W, int Cout, int act) {
    float* y = falloc((size_t)N * Cout * H * W);
    vo_conv2d(x, P[*pi], P[*pi + 1], y, N, Cin, H, W, Cout, 3);
    vo_batchnorm_eval(y, P[*pi + 2], P[*pi + 3], P[*pi + 4], P[*pi + 5], N, Cout, H * W);
    if (act == 1) vo_leaky_relu(y, (size_t)N * Cout * H * W, 0.2f);
    else vo_relu(y, (size_t)N * Cout * H * W);
    *pi += 6;
    return y;
}

static float* convt_bn_relu(const float* x, const float* const* P, int* pi, int N, int Cin, int H, int W, int Cout) {
    float* y = falloc((size_t)N * Cout * 4 * H * W);
    vo_convt2x2(x, P[*pi], P[*pi + 1], y, N, Cin, H, W, Cout);
    vo_batchnorm_eval(y, P[*pi + 2], P[*pi + 3], P[*pi + 4], P[*pi + 5], N, Cout, 4 * H * W);
    vo_relu(y, (size_t)N * Cout * 4 * H * W);
    *pi += 6;
    return y;
}

/* ConvAutoencoder.forward (models/autoencoder.py:181-193).  P: 92 tensors, state_dict order without
 * num_batches_tracked.  recon [N,3,H,W]; latent_out [N,latent,H/16,W/16] or NULL. */
int vo_img_forward(const float* const* P, int latent, const float* x, int N, int H, int W,
                   float* recon, float* latent_out) {
    if (H % 16 || W % 16) return -1;
    const int ch[5] = {3, 32, 64, 128, latent};
    int pi = 0, h = H, w = W;
    float* cur = falloc((size_t)N * 3 * H * W);
    memcpy(cur, x, (size_t)N * 3 * H * W * sizeof(float));
    for (int b = 0; b < 4; ++b) {                       /* Encoder.enc1..enc4 */
        float* a = conv_bn_act(cur, P, &pi, N, ch[b], h, w, ch[b + 1], 1);
        free(cur);
        float* c = conv_bn_act(a, P, &pi, N, ch[b + 1], h, w, ch[b + 1], 1);
        free(a);
        float* p = falloc((size_t)N * ch[b + 1] * (h / 2) * (w / 2));
        vo_maxpool2(c, p, N, ch[b + 1], h, w);
        free(c);
        cur = p; h /= 2; w /= 2;
    }
    if (latent_out) memcpy(latent_out, cur, (size_t)N * latent * h * w * sizeof(float));
    if (recon) {
        const int dch[5] = {latent, 128, 64, 32, 32};
        for (int b = 0; b < 4; ++b) {                   /* Decoder.dec1..dec4 */
            float* u = convt_bn_relu(cur, P, &pi, N, dch[b], h, w, dch[b + 1]);
            free(cur);
            h *= 2; w *= 2;
            if (b < 3) {
                cur = conv_bn_act(u, P, &pi, N, dch[b + 1], h, w, dch[b + 1], 2);
                free(u);
            } else {
                vo_conv2d(u, P[pi], P[pi + 1], recon, N, 32, h, w, 3, 3);
                vo_tanh(recon, (size_t)N * 3 * h * w);
                free(u);
                cur = NULL;
            }
        }
    }
    free(cur);
    return 0;
}

/* error = (x-recon)**2; map = error.mean(dim=C, keepdim) ; score = map.mean over (1,H,W)
 * (models/autoencoder.py:214-221).  frames N = B (image) or B*T (video, models/video_autoencoder.py:371-384):
 * seq[b] = mean over all T*C*H*W elements of clip b. */
void vo_error(const float* x, const float* recon, int N, int C, int H, int W, int T,
              float* errmap, float* frame_scores, float* seq_scores) {
    const size_t hw = (size_t)H * W;
    for (int n = 0; n < N; ++n) {
        double tot = 0.0;
        for (size_t i = 0; i < hw; ++i) {
            double s = 0.0;
            for (int c = 0; c < C; ++c) {
                const float d = x[((size_t)n * C + c) * hw + i] - recon[((size_t)n * C + c) * hw + i];
                s += (double)(d * d);
            }
            const float m = (float)(s / C);
            if (errmap) errmap[(size_t)n * hw + i] = m;
            tot += s;
        }
        if (frame_scores) frame_scores[n] = (float)(tot / ((double)C * hw));
    }
    if (seq_scores) {
        for (int b = 0; b < N / T; ++b) {
            double tot = 0.0;
            for (int t = 0; t < T; ++t) {
                const int n = b * T + t;
                for (size_t i = 0; i < (size_t)C * hw; ++i) {
                    const float d = x[(size_t)n * C * hw + i] - recon[(size_t)n * C * hw + i];
                    tot += (double)(d * d);
                }
            }
            seq_scores[b] = (float)(tot / ((double)T * C * hw));
        }
    }
}

/* ConvLSTMCell.forward (models/video_autoencoder.py:54-85): gates = conv(cat[x,h]); i,f,g,o split;
 * c' = sigmoid(f)*c + sigmoid(i)*tanh(g); h' = sigmoid(o)*tanh(c').  h,c updated in place. */
void vo_convlstm_cell(const float* x, float* h, float* c, const float* w, const float* b,
                      int N, int Cx, int Hd, int H, int W) {
    const int Cin = Cx + Hd;
    const size_t hw = (size_t)H * W;
    float* cat = falloc((size_t)N * Cin * hw);
    for (int n = 0; n < N; ++n) {
        memcpy(cat + (size_t)n * Cin * hw, x + (size_t)n * Cx * hw, (size_t)Cx * hw * sizeof(float));
        memcpy(cat + ((size_t)n * Cin + Cx) * hw, h + (size_t)n * Hd * hw, (size_t)Hd * hw * sizeof(float));
    }
    float* gates = falloc((size_t)N * 4 * Hd * hw);
    vo_conv2d(cat, w, b, gates, N, Cin, H, W, 4 * Hd, 3);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < Hd; ++k)
            for (size_t i = 0; i < hw; ++i) {
                const float* g0 = gates + ((size_t)n * 4 * Hd) * hw;
                const double gi = 1.0 / (1.0 + exp(-(double)g0[((size_t)0 * Hd + k) * hw + i]));
                const double gf = 1.0 / (1.0 + exp(-(double)g0[((size_t)1 * Hd + k) * hw + i]));
                const double gg = tanh((double)g0[((size_t)2 * Hd + k) * hw + i]);
                const double go = 1.0 / (1.0 + exp(-(double)g0[((size_t)3 * Hd + k) * hw + i]));
                const size_t o = ((size_t)n * Hd + k) * hw + i;
                const float cn = (float)(gf * (double)c[o] + gi * gg);
                c[o] = cn;
                h[o] = (float)(go * tanh((double)cn));
            }
    free(cat);
    free(gates);
}

/* VideoAutoencoder.forward (models/video_autoencoder.py:329-354).  x [B,T,3,H,W] -> recon same shape.
 * P: state_dict order without num_batches_tracked (see include/vad_hip.h vad_vid_pack). */
int vo_vid_forward(const float* const* P, int latent, int hid, int layers, const float* x,
                   int B, int T, int H, int W, float* recon) {
    if (H % 16 || W % 16) return -1;
    const int N = B * T;
    const int ch[5] = {3, 32, 64, 128, latent};
    int pi = 0, h = H, w = W;
    float* cur = falloc((size_t)N * 3 * H * W);
    memcpy(cur, x, (size_t)N * 3 * H * W * sizeof(float));
    for (int b = 0; b < 4; ++b) {                       /* VideoEncoder on B*T frames (:222-228) */
        float* a = conv_bn_act(cur, P, &pi, N, ch[b], h, w, ch[b + 1], 1);
        free(cur);
        float* p = falloc((size_t)N * ch[b + 1] * (h / 2) * (w / 2));
        vo_maxpool2(a, p, N, ch[b + 1], h, w);
        free(a);
        cur = p; h /= 2; w /= 2;
    }
    const size_t hw = (size_t)h * w;
    /* ConvLSTM: layers outer, time inner, zero initial state (:144-166).  cur is [B,T,C,h,w]. */
    int cx = latent;
    for (int l = 0; l < layers; ++l) {
        float* hs = (float*)calloc((size_t)B * hid * hw, sizeof(float));
        float* cs = (float*)calloc((size_t)B * hid * hw, sizeof(float));
        float* out = falloc((size_t)N * hid * hw);
        float* xt = falloc((size_t)B * cx * hw);
        for (int t = 0; t < T; ++t) {
            for (int b = 0; b < B; ++b)
                memcpy(xt + (size_t)b * cx * hw, cur + ((size_t)b * T + t) * cx * hw, (size_t)cx * hw * sizeof(float));
            vo_convlstm_cell(xt, hs, cs, P[pi], P[pi + 1], B, cx, hid, h, w);
            for (int b = 0; b < B; ++b)
                memcpy(out + ((size_t)b * T + t) * hid * hw, hs + (size_t)b * hid * hw, (size_t)hid * hw * sizeof(float));
        }
        pi += 2;
        free(hs); free(cs); free(xt); free(cur);
        cur = out;
        cx = hid;
    }
    if (hid != latent) {                                /* proj = Conv2d(hid, latent, 1) (:311, :346-349) */
        float* pr = falloc((size_t)N * latent * hw);
        vo_conv2d(cur, P[pi], P[pi + 1], pr, N, hid, h, w, latent, 1);
        pi += 2;
        free(cur);
        cur = pr;
    }
    const int dch[4] = {latent, 128, 64, 32};
    for (int b = 0; b < 3; ++b) {                       /* VideoDecoder (:242-256) */
        float* u = convt_bn_relu(cur, P, &pi, N, dch[b], h, w, dch[b + 1]);
        free(cur);
        cur = u; h *= 2; w *= 2;
    }
    vo_convt2x2(cur, P[pi], P[pi + 1], recon, N, 32, h, w, 3);   /* (:259-260) */
    vo_tanh(recon, (size_t)N * 3 * 4 * h * w);
    free(cur);
    return 0;
}
