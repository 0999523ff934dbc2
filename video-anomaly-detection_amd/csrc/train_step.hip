// One optimisation step of the ConvLSTM video autoencoder (reference train_video.py:44-65: model.train(); out = model(x);
// loss = MSELoss(out, x); zero_grad; backward; Adam.step) as an explicit launch sequence over the kernels of
// train_ops.hip and the forward convolution kernels.  Host orchestration only: no kernels in this file.
//
// Parameters live in ONE flat fp32 buffer in torch layouts and torch named_parameters() order, gradients in a buffer of
// the same shape (the Python module's nn.Parameters are views into them), so the optimiser is one launch and a
// data-parallel run needs one all-reduce.  Order (models/video_autoencoder.py:191-215, 299-316, 242-261):
//   encoder.encoder.{0,4,8,12}: conv w (OIHW), conv b, then BatchNorm gamma, beta   channels 3->32->64->128->latent
//   convlstm.cells.l.conv: w (4*hid, cin_l+hid, 3, 3), b                            cin_0 = latent, cin_l = hid
//   proj (only when hid != latent): w (latent, hid, 1, 1), b                        models/video_autoencoder.py:311
//   decoder.decoder.{0,3,6}: convT w (IOHW), b, BatchNorm gamma, beta               latent->128->64->32
//   decoder.decoder.9: convT w (32,3,2,2), b
// Running statistics: one flat buffer, {running_mean[c], running_var[c]} per BatchNorm in the order above.
//
// Activation memory (fp32, per frame of H x W): every conv output before BatchNorm is kept (the backward recomputes
// normalisation, activation and pooling from it), plus the pooled activations that feed the next conv and its weight
// gradient: 30.4 MB per 256x256 frame; the ConvLSTM keeps its operand buffers [t][b][h][w][x|h], activated gates and
// cell states for all t (BPTT).  Two scratch buffers of the largest activation size carry the gradients (the
// BatchNorm backward re-derives the routed gradient instead of storing it).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "vad_common.h"

namespace {

constexpr int ENC_C[5] = {3, 32, 64, 128, 0};    // [4] = latent
constexpr int DEC_C[4] = {0, 128, 64, 32};       // [0] = latent

struct Plan {
    int B, T, H, W, L, Hd, NL, N, h16, w16, hw;
    int encC[5], decC[4];
    // parameter offsets (floats) into the flat buffer
    size_t e_w[4], e_b[4], e_g[4], e_be[4];
    size_t l_w[8], l_b[8], pj_w, pj_b;
    bool proj;
    size_t d_w[3], d_b[3], d_g[3], d_be[3];
    size_t t_w, t_b, nparams;
    size_t e_rs[4], d_rs[3], nstats;             // running stats offsets (mean at +0, var at +c)
    // workspace offsets (floats)
    size_t pk_e[4], pk_e_dg[4], pk_l[8], pk_l_dg[8], pk_d[3], pk_d_dg[3];
    size_t y[4], a[3], st_e[4], cat[8], z[8], c[8], hseq, pseq, pk_pj, pk_pj_dg, u[3], r[3], st_d[3], dpre;
    size_t g[3], dcat[8], dzl[8], dc[8], ksums, zeros, chan_ws, wgrad_ws, to3_ws;
    size_t ws_floats;
    int lstm_cin(int l) const { return l == 0 ? L : Hd; }
};

size_t align64(size_t v) { return (v + 63) & ~(size_t)63; }

bool make_plan(Plan& p, int B, int T, int H, int W, int L, int Hd, int NL) {
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || H % 16 || W % 16 || L <= 0 || L % 32 || Hd <= 0 || Hd % 32 || Hd > 256 || NL < 1 || NL > 8) return false;
    if ((long long)B * T > (1 << 20)) return false;
    p.B = B; p.T = T; p.H = H; p.W = W; p.L = L; p.Hd = Hd; p.NL = NL; p.N = B * T;
    p.proj = Hd != L;
    p.h16 = H / 16; p.w16 = W / 16; p.hw = p.h16 * p.w16;
    for (int i = 0; i < 5; ++i) p.encC[i] = ENC_C[i];
    p.encC[4] = L;
    for (int i = 0; i < 4; ++i) p.decC[i] = DEC_C[i];
    p.decC[0] = L;
    size_t o = 0, rs = 0;
    for (int k = 0; k < 4; ++k) {
        const int ci = p.encC[k], co = p.encC[k + 1];
        p.e_w[k] = o; o += (size_t)co * ci * 9;
        p.e_b[k] = o; o += co;
        p.e_g[k] = o; o += co;
        p.e_be[k] = o; o += co;
        p.e_rs[k] = rs; rs += 2 * (size_t)co;
    }
    for (int l = 0; l < NL; ++l) {
        p.l_w[l] = o; o += (size_t)4 * Hd * (p.lstm_cin(l) + Hd) * 9;
        p.l_b[l] = o; o += (size_t)4 * Hd;
    }
    p.pj_w = p.pj_b = 0;
    if (p.proj) {
        p.pj_w = o; o += (size_t)L * Hd;
        p.pj_b = o; o += L;
    }
    for (int j = 0; j < 3; ++j) {
        const int ci = p.decC[j], co = p.decC[j + 1];
        p.d_w[j] = o; o += (size_t)ci * co * 4;
        p.d_b[j] = o; o += co;
        p.d_g[j] = o; o += co;
        p.d_be[j] = o; o += co;
        p.d_rs[j] = rs; rs += 2 * (size_t)co;
    }
    p.t_w = o; o += 32 * 3 * 4;
    p.t_b = o; o += 3;
    p.nparams = o;
    p.nstats = rs;

    // ---- workspace
    size_t w = 0;
    auto take = [&](size_t n) { const size_t at = w; w += align64(n); return at; };
    const size_t N = (size_t)p.N;
    size_t max_act = 0, max_chan = 0, max_wgrad = 0;
    auto chan = [&](long long npix, int c) { const size_t v = vad_chan_ws_floats(npix, c); if (v > max_chan) max_chan = v; };
    for (int k = 0; k < 4; ++k) {
        const int ci = p.encC[k], co = p.encC[k + 1], hk = H >> k, wk = W >> k;
        // (3x3 operand slots hold either form of a layer: direct, 9 taps, or Winograd, 16 - VAD_PREC_WINO)
        p.pk_e[k] = take(k == 0 ? vad_pack_conv3x3_c3_floats(co) : vad_pack_conv3x3_wino_floats(co, ci));
        p.pk_e_dg[k] = k == 0 ? 0 : take(vad_pack_conv3x3_wino_floats(ci, co));
        const size_t ysz = N * hk * wk * co;
        p.y[k] = take(ysz);
        if (ysz > max_act) max_act = ysz;
        if (k < 3) p.a[k] = take(ysz / 4);
        p.st_e[k] = take(2 * (size_t)co);
        chan((long long)N * hk * wk, co);
        const size_t wg = k == 0 ? vad_conv_c3_wgrad_ws_floats(p.N, hk, co) : vad_conv_wgrad_ws_floats(p.N, hk, 9, ci, co);
        if (wg > max_wgrad) max_wgrad = wg;
        if (k == 0 && vad_conv_c3_wgrad_routed_ws_floats(p.N, hk) > max_wgrad) max_wgrad = vad_conv_c3_wgrad_routed_ws_floats(p.N, hk);
    }
    for (int l = 0; l < NL; ++l) {
        const int cin = p.lstm_cin(l) + Hd;
        p.pk_l[l] = take(vad_pack_conv3x3_wino_floats(4 * Hd, cin));
        p.pk_l_dg[l] = take(vad_pack_conv3x3_wino_floats(cin, 4 * Hd));
        p.cat[l] = take(N * p.hw * cin);
        p.z[l] = take(N * p.hw * 4 * Hd);
        p.c[l] = take(N * p.hw * Hd);
        p.dcat[l] = take(N * p.hw * cin);
        p.dzl[l] = take(N * p.hw * 4 * Hd);
        chan((long long)N * p.hw, 4 * Hd);
        const size_t wg = vad_conv_wgrad_ws_floats(p.N, p.h16, 9, cin, 4 * Hd);
        if (wg > max_wgrad) max_wgrad = wg;
    }
    for (int l = 0; l < NL; ++l) p.dc[l] = take((size_t)B * p.hw * Hd);   // per layer: the layers of a small batch run as a wavefront
    p.hseq = take(N * p.hw * Hd);
    p.pseq = p.pk_pj = p.pk_pj_dg = 0;
    if (p.proj) {
        p.pseq = take(N * p.hw * L);
        p.pk_pj = take(vad_pack_conv1x1_floats(L, Hd));
        p.pk_pj_dg = take(vad_pack_conv1x1_floats(Hd, L));
        chan((long long)N * p.hw, L);
        const size_t wg = vad_conv_wgrad_ws_floats(p.N, p.h16, 1, Hd, L);
        if (wg > max_wgrad) max_wgrad = wg;
    }
    for (int j = 0; j < 3; ++j) {
        const int ci = p.decC[j], co = p.decC[j + 1], hj = p.h16 << j, wj = p.w16 << j;
        p.pk_d[j] = take(vad_pack_convt2x2_floats(ci, co));
        p.pk_d_dg[j] = take(vad_pack_conv1x1_floats(ci, 4 * co));
        const size_t usz = N * (size_t)(2 * hj) * (2 * wj) * co;
        p.u[j] = take(usz);
        p.r[j] = take(usz);
        if (usz > max_act) max_act = usz;
        p.st_d[j] = take(2 * (size_t)co);
        chan((long long)N * 4 * hj * wj, co);
        const size_t wg = vad_conv_wgrad_ws_floats(p.N, hj, 1, ci, 4 * co);
        if (wg > max_wgrad) max_wgrad = wg;
    }
    {
        const size_t wg = vad_conv_wgrad_ws_floats(p.N, H / 2, 1, 32, 32);
        if (wg > max_wgrad) max_wgrad = wg;
    }
    p.dpre = take(N * (size_t)(H / 2) * (W / 2) * 32);
    p.g[0] = take(max_act);
    p.g[2] = take(max_act);
    p.g[1] = take(max_act);   // second buffer for the gradient of a conv output: while the weight gradient of layer k reads one on the
                              // helper stream, the BatchNorm backward of layer k-1 writes the other
    p.ksums = take(2 * 1024);
    p.zeros = take(1024);
    // the first layer writes its BatchNorm partial sums itself: one [2][32] row per work-group, at most one per 32x16 tile
    { const size_t v = N * (size_t)((W + 15) / 16) * (size_t)((H + 31) / 32) * 64; if (v > max_chan) max_chan = v; }
    for (int k = 1; k < 4; ++k) { const size_t v = vad_conv3x3_stats_floats(p.encC[k + 1]); if (v > max_chan) max_chan = v; }   // so do the other encoder convolutions
    for (int j = 0; j < 3; ++j) { const size_t v = vad_convt2x2_stats_floats(p.decC[j + 1]); if (v > max_chan) max_chan = v; }  // and the decoder's transposed ones
    p.chan_ws = take(max_chan);
    p.wgrad_ws = take(max_wgrad);
    p.to3_ws = take(vad_convt_to3_mse_ws_floats(p.N, H / 2, W / 2));
    p.ws_floats = w;
    return true;
}

}  // namespace

// Helper streams of the ConvLSTM layer wavefront (small batches: a step's launch fills a fraction of the chip, so layer l's step t
// runs beside layer l-1's step t+1 - forward - and layer l's step t beside layer l+1's step t-1 - backward; the reference's loops
// are layers-outer, models/video_autoencoder.py:153-160, the data dependences allow the diagonal order).  Created once per thread
// and device; fork / join through events on the caller's stream, so the step stays asynchronous.
namespace {
struct TrainStreams {
    int dev = -1, n = 0;
    hipStream_t st[7] = {};
    hipEvent_t done[8] = {};
    hipEvent_t fork = nullptr;
    hipStream_t wg = nullptr;          // the weight-gradient GEMMs of the backward pass (beside the BatchNorm / data-gradient chain)
    hipEvent_t wg_in = nullptr, wg_read[2] = {};
};
thread_local TrainStreams t_ts;
int train_streams(int layers, TrainStreams** out) {
    int dev = 0;
    VAD_HIP_TRY(hipGetDevice(&dev));
    TrainStreams& S = t_ts;
    if (S.dev != dev) {
        S = TrainStreams{};
        S.dev = dev;
        VAD_HIP_TRY(hipEventCreateWithFlags(&S.fork, hipEventDisableTiming));
        for (int l = 0; l < 8; ++l) VAD_HIP_TRY(hipEventCreateWithFlags(&S.done[l], hipEventDisableTiming));
        VAD_HIP_TRY(hipStreamCreateWithFlags(&S.wg, hipStreamNonBlocking));
        VAD_HIP_TRY(hipEventCreateWithFlags(&S.wg_in, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) VAD_HIP_TRY(hipEventCreateWithFlags(&S.wg_read[i], hipEventDisableTiming));
    }
    while (S.n < layers - 1) { VAD_HIP_TRY(hipStreamCreateWithFlags(&S.st[S.n], hipStreamNonBlocking)); ++S.n; }
    *out = &S;
    return VAD_OK;
}
}  // namespace
extern "C" int vad_lstm_wavefront_mode(void);   // vad_api.hip: vad_debug_set_lstm_wavefront (0 = never, 1 = small launch groups, 2 = always)

// Debug: return from vad_vid_train_fwd_bwd right after the backward of decoder stage `j` (2, 1, 0), 10 + l after ConvLSTM
// layer l, leaving the gradient scratch (g0 = gradient of that stage's input, g2 = gradient of its conv output)
// in the workspace for inspection (tools/diag_stream.py).  -1 = run the whole step (default).
static int g_vad_train_stop = -1;
extern "C" int vad_debug_set_train_stop(int stage) { g_vad_train_stop = stage; return VAD_OK; }

// A/B switch for the split-fp16 step's gradient scaling (tests/test_hip_train_step.py shows what it buys); default on.
static int g_vad_split_grad_scale = 1;
extern "C" int vad_debug_set_split_grad_scale(int on) { g_vad_split_grad_scale = on != 0; return VAD_OK; }
extern "C" int vad_split_grad_scale_enabled(void) { return g_vad_split_grad_scale; }

// per-group timing (vad_prof_*, model 2 of vad_prof_slot_name): every launch of the step belongs to one of these groups
enum { TS_C3_FWD = 0, TS_CONV_FWD, TS_BN_FWD, TS_LSTM_CONV_FWD, TS_LSTM_GATES_FWD, TS_CONVT_FWD, TS_LOSS, TS_WGRAD, TS_BN_BWD,
       TS_CONVT_DGRAD, TS_LSTM_GATES_BWD, TS_LSTM_CONV_DGRAD, TS_CONV_DGRAD, TS_C3_WGRAD, TS_PACK, TS_STATS_MISC };
#define PS(slot) VadProfScope ps_(slot, s)
#define PSS(slot, st_) VadProfScope ps_(slot, st_)

#define TRY(expr)                    \
    do {                             \
        const int rc_ = (expr);      \
        if (rc_ != VAD_OK) return rc_; \
    } while (0)

extern "C" size_t vad_vid_train_nparams(int latent, int hid, int layers) {
    Plan p;
    return make_plan(p, 1, 1, 16, 16, latent, hid, layers) ? p.nparams : 0;
}

extern "C" size_t vad_vid_train_nstats(int latent, int hid, int layers) {
    Plan p;
    return make_plan(p, 1, 1, 16, 16, latent, hid, layers) ? p.nstats : 0;
}

extern "C" size_t vad_vid_train_workspace_bytes(int b, int t, int h, int w, int latent, int hid, int layers) {
    Plan p;
    return make_plan(p, b, t, h, w, latent, hid, layers) ? p.ws_floats * sizeof(float) : 0;
}

// Debug: float offsets of the saved forward buffers inside the workspace, in the order
//   y[0..3], a[0..2], st_e[0..3], cat[0..NL), z[0..NL), c[0..NL), hseq, u[0..2], r[0..2], st_d[0..2], dpre, g[0..2], ws_floats
// (tools/diag_saved.py compares them with a float64 forward after a real step).  Returns the number written or < 0.
extern "C" int vad_vid_train_debug_layout(int b, int t, int h, int w, int latent, int hid, int layers, long long* out, int cap) {
    Plan p;
    VAD_REQUIRE(out && make_plan(p, b, t, h, w, latent, hid, layers), "vid_train_debug_layout: unsupported configuration");
    std::vector<long long> v;
    for (int k = 0; k < 4; ++k) v.push_back((long long)p.y[k]);
    for (int k = 0; k < 3; ++k) v.push_back((long long)p.a[k]);
    for (int k = 0; k < 4; ++k) v.push_back((long long)p.st_e[k]);
    for (int l = 0; l < layers; ++l) v.push_back((long long)p.cat[l]);
    for (int l = 0; l < layers; ++l) v.push_back((long long)p.z[l]);
    for (int l = 0; l < layers; ++l) v.push_back((long long)p.c[l]);
    v.push_back((long long)p.hseq);
    for (int j = 0; j < 3; ++j) v.push_back((long long)p.u[j]);
    for (int j = 0; j < 3; ++j) v.push_back((long long)p.r[j]);
    for (int j = 0; j < 3; ++j) v.push_back((long long)p.st_d[j]);
    v.push_back((long long)p.dpre);
    for (int i = 0; i < 3; ++i) v.push_back((long long)p.g[i]);
    v.push_back((long long)p.ws_floats);
    VAD_REQUIRE((int)v.size() <= cap, "vid_train_debug_layout: need room for %d entries", (int)v.size());
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}

extern "C" int vad_vid_train_fwd_bwd(const float* x, int b, int t, int h, int w, int latent, int hid, int layers,
                                     const float* params, float* grads, float* running, void* workspace, size_t workspace_bytes,
                                     int precision, float* loss, float* recon, void* stream) {
    VAD_REQUIRE(x && params && grads && workspace && loss, "vid_train_fwd_bwd: null pointer");
    VAD_REQUIRE(precision >= VAD_PREC_FP32 && precision <= VAD_PREC_WINO, "vid_train_fwd_bwd: precision=%d must be 0 (fp32), 1 (split fp16), 2 (bf16 operands), 3 (bf16 tensors) or 4 (Winograd)", precision);
    // VAD_PREC_WINO: the 3x3 convolutions behind the first layer - forward and data gradients, the ConvLSTM gate convolutions
    // included - in Winograd F(2x2,3x3) form on the exact-fp32 matrix pipe (csrc/conv_wino.hip); everything else is the
    // VAD_PREC_FP32 arithmetic (`precision` below).  All-fp32, another rounding order than mode 0.
    const bool wino = precision == VAD_PREC_WINO;
    const int pack_prec = precision;
    if (wino) precision = VAD_PREC_FP32;
    // a 3x3 convolution [n][hh][ww][ci] -> [n][hh][ww][co] without activation, in the step's mode
    auto conv3 = [&](const float* in, const float* wpk, const float* bias, float* out, int n, int hh, int ww, int ci, int co, float* stats, int* srows,
                     hipStream_t st) -> int {
        if (wino) { if (srows) *srows = 0; return vad_conv3x3_wino(in, 0, wpk, bias, out, 0, n, hh, ww, ci, co, VAD_ACT_NONE, 0, st); }
        return vad_conv3x3_stats(in, 0, wpk, bias, out, 0, n, hh, ww, ci, co, VAD_ACT_NONE, 0, precision, stats, srows, st);
    };
    // Arithmetic mode (argument `precision`): 0 = exact fp32 everywhere (the parity path); 1 = the 3x3 and transposed
    // convolutions (forward and data gradients) take split-fp16 operands (22-bit products, fp32 accumulate), everything
    // else - first layer, weight gradients, 1x1 data gradients, BatchNorm, gates, loss, Adam - stays fp32; 2 = bf16 operands
    // in the same places plus the weight gradients; 3 (VAD_PREC_BF16S) = 2 with every activation / activation-gradient
    // tensor of the workspace stored as bf16 (the buffers keep their fp32-sized slots and use the first half), the 1x1
    // data gradients and the first layer's forward on bf16 operands too.
    Plan p;
    VAD_REQUIRE(make_plan(p, b, t, h, w, latent, hid, layers),
                "vid_train_fwd_bwd: unsupported configuration (B=%d T=%d %dx%d latent=%d hid=%d layers=%d): H, W multiples of 16, "
                "latent and hidden multiples of 32, hidden <= 256, 1..8 layers", b, t, h, w, latent, hid, layers);
    if (workspace_bytes < p.ws_floats * sizeof(float))
        return vad_fail(VAD_ERR_WS, "vid_train_fwd_bwd: workspace %zu bytes < %zu needed", workspace_bytes, p.ws_floats * sizeof(float));
    // bf16 tensors: the BatchNorm statistics come from the convolutions' accumulators and from nowhere else (there is no
    // bf16 form of the stand-alone statistics pass).  Say so BEFORE anything is launched - a refusal in the middle of the
    // step would leave the workspace and the running statistics half-updated.
    VAD_REQUIRE(precision != VAD_PREC_BF16S || vad_conv_stats_available(),
                "vid_train_fwd_bwd: VAD_PREC_BF16S needs the persistent convolution kernels' own BatchNorm partial sums "
                "(vad_debug_set_conv_variant bit 0 is cleared)");
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const float* P = params;
    float* G = grads;
    const int N = p.N, B = p.B, T = p.T, H = p.H, W = p.W, L = p.L, Hd = p.Hd, NL = p.NL, hw = p.hw;
    const float eps = 1e-5f, mom = 0.1f;     // nn.BatchNorm2d defaults (models/video_autoencoder.py:193)
    const int io = precision == VAD_PREC_BF16S;          // activation tensors are bf16
    const size_t es = io ? 2 : 4;
    // element `elem` of the activation buffer at workspace offset `off` (floats), in the mode's storage type
    auto A = [&](size_t off, size_t elem = 0) -> float* { return (float*)((char*)(ws + off) + elem * es); };
    auto at = [&](const float* base, size_t elem) -> float* { return (float*)((char*)base + elem * es); };
    float* zeros = ws + p.zeros;
    VAD_HIP_TRY(hipMemsetAsync(zeros, 0, 1024 * sizeof(float), s));

    // ---- operand packing of the current parameters
    { PS(TS_PACK);
    TRY(vad_train_pack_conv3x3_c3(P + p.e_w[0], 32, ws + p.pk_e[0], s));
    for (int k = 1; k < 4; ++k)
        TRY(vad_train_pack_conv3x3(P + p.e_w[k], p.encC[k + 1], p.encC[k], ws + p.pk_e[k], ws + p.pk_e_dg[k], pack_prec, s));
    for (int l = 0; l < NL; ++l)
        TRY(vad_train_pack_conv3x3(P + p.l_w[l], 4 * Hd, p.lstm_cin(l) + Hd, ws + p.pk_l[l], ws + p.pk_l_dg[l], pack_prec, s));
    for (int j = 0; j < 3; ++j)
        TRY(vad_train_pack_convt2x2(P + p.d_w[j], p.decC[j], p.decC[j + 1], ws + p.pk_d[j], ws + p.pk_d_dg[j], precision, s));
    if (p.proj) TRY(vad_train_pack_conv1x1_p(P + p.pj_w, L, Hd, ws + p.pk_pj, ws + p.pk_pj_dg, precision, s));
    }

    // ================================================================================== forward
    // encoder (models/video_autoencoder.py:191-215): conv -> BatchNorm(batch stats) -> LeakyReLU(0.2) -> MaxPool2
    for (int k = 0; k < 4; ++k) {
        const int ci = p.encC[k], co = p.encC[k + 1], hk = H >> k, wk = W >> k;
        float* y = A(p.y[k]);
        int sblocks = 0;      // > 0: the convolution wrote the BatchNorm partial sums itself (first layer: no second pass over y)
        // (bf16 tensors: the first layer on bf16 MFMA operands too - out16 = 2 - so that it runs at the rate of its stores)
        if (k == 0) { PS(TS_C3_FWD); TRY(vad_conv3x3_c3_stats_t(x, VAD_X_F32_NCHW, ws + p.pk_e[0], P + p.e_b[0], y, io ? 2 : 0, N, hk, wk, co, VAD_ACT_NONE, 0, ws + p.chan_ws, &sblocks, s)); }
        else { PS(TS_CONV_FWD); TRY(conv3(A(p.a[k - 1]), ws + p.pk_e[k], P + p.e_b[k], y, N, hk, wk, ci, co, ws + p.chan_ws, &sblocks, s)); }
        float* rs = running ? running + p.e_rs[k] : nullptr;
        if (sblocks > 0) { PS(TS_STATS_MISC); TRY(vad_bn_stats_from_partials(ws + p.chan_ws, sblocks, (long long)N * hk * wk, co, eps, mom, ws + p.st_e[k], rs, rs ? rs + co : nullptr, P + p.e_b[k], s)); }
        else {
            VAD_REQUIRE(!io, "vid_train_fwd_bwd: bf16 tensors need the convolutions' own BatchNorm partial sums (persistent kernels)");
            TRY(vad_bn_stats(y, (long long)N * hk * wk, co, eps, mom, ws + p.st_e[k], rs, rs ? rs + co : nullptr, ws + p.chan_ws, s));
        }
        PS(TS_BN_FWD);
        if (k < 3)
            TRY(vad_bn_act_pool_fwd_t(y, io, ws + p.st_e[k], P + p.e_g[k], P + p.e_be[k], A(p.a[k]), 0, 0, 0, 0, N, hk, wk, co, VAD_ACT_LEAKY, 1, s));
        else   // latent features go straight into layer 0's operand buffers: frame b*T+t -> slot t*B+b, x-part
            TRY(vad_bn_act_pool_fwd_t(y, io, ws + p.st_e[k], P + p.e_g[k], P + p.e_be[k], A(p.cat[0]), 0, L + Hd, T, B, N, hk, wk, co, VAD_ACT_LEAKY, 1, s));
    }
    // ConvLSTM (models/video_autoencoder.py:153-163); h(-1) = c(-1) = 0.  Step (l, t) needs (l, t-1) and (l-1, t) only.
    // A step's launch covers B*h16*w16 pixels: at 32 clips of 16x16 that is 256 (forward) or 128 (data gradient) work-groups on
    // 512 resident slots, so with more than one layer the layers run as a WAVEFRONT on helper streams (layer l on stream l-1,
    // step t of layer l behind step t of layer l-1) and two launches share the chip; large batches keep the layers-outer order.
    const int wf_mode = vad_lstm_wavefront_mode();
    const bool wavefront = NL > 1 && g_vad_train_stop < 10 && (wf_mode == 2 || (wf_mode == 1 && (long long)B * hw <= 12288));
    // The weight-gradient GEMMs run on a helper stream BESIDE the chain BatchNorm backward -> data gradient -> BatchNorm backward
    // of the next layer: they are matrix- / latency-bound, the chain is HBM-bound, and nothing downstream needs them before the
    // optimiser.  The gradient of a conv output alternates between two buffers so that the next layer's BatchNorm backward does
    // not overwrite what a weight gradient is still reading.  Same launches, same operands: bit-identical to the serial order.
    const bool overlap = wf_mode != 0 && g_vad_train_stop < 0;
    TrainStreams* TS = nullptr;
    if (wavefront || overlap) TRY(train_streams(NL, &TS));
    for (int l = 0; l < NL; ++l) {    // h-part of every layer's t = 0 operand is the zero initial state (the x-part is written by the producer)
        const int cx = p.lstm_cin(l), cin = cx + Hd;
        VAD_HIP_TRY(hipMemset2DAsync(A(p.cat[l], cx), (size_t)cin * es, 0, (size_t)Hd * es, (size_t)B * hw, s));
    }
    auto lstm_fwd_step = [&](int l, int tt, hipStream_t st) -> int {
        const int cx = p.lstm_cin(l), cin = cx + Hd;
        const size_t slab = (size_t)B * hw * cin;
        float* zt = A(p.z[l], (size_t)tt * B * hw * 4 * Hd);
        float* ct = ws + p.c[l] + (size_t)tt * B * hw * Hd;
        { PSS(TS_LSTM_CONV_FWD, st); TRY(conv3(A(p.cat[l], tt * slab), ws + p.pk_l[l], P + p.l_b[l], zt, B, p.h16, p.w16, cin, 4 * Hd, nullptr, nullptr, st)); }
        float* h1 = tt + 1 < T ? A(p.cat[l], (tt + 1) * slab + cx) : nullptr;
        float* h2; long long h2_fs; int h2_ps;
        if (l + 1 < NL) { h2 = A(p.cat[l + 1], (size_t)tt * B * hw * 2 * Hd); h2_ps = 2 * Hd; h2_fs = (long long)hw * h2_ps; }
        else { h2 = A(p.hseq, (size_t)tt * hw * Hd); h2_ps = Hd; h2_fs = (long long)T * hw * Hd; }
        PSS(TS_LSTM_GATES_FWD, st);
        return vad_lstm_gates_fwd_t(zt, io, tt ? ct - (size_t)B * hw * Hd : nullptr, ct, h1, (long long)hw * cin, cin, h2, h2_fs, h2_ps, B, hw, Hd, st);
    };
    if (wavefront) {
        VAD_HIP_TRY(hipEventRecord(TS->fork, s));                                   // the latent features and the zeroed states are ready
        for (int l = 1; l < NL; ++l) VAD_HIP_TRY(hipStreamWaitEvent(TS->st[l - 1], TS->fork, 0));
        for (int tt = 0; tt < T; ++tt)
            for (int l = 0; l < NL; ++l) {
                hipStream_t st = l ? TS->st[l - 1] : s;
                if (l) VAD_HIP_TRY(hipStreamWaitEvent(st, TS->done[l - 1], 0));       // (l-1, tt) finished
                TRY(lstm_fwd_step(l, tt, st));
                if (l + 1 < NL || tt + 1 == T) VAD_HIP_TRY(hipEventRecord(TS->done[l], st));
            }
        for (int l = 1; l < NL; ++l) VAD_HIP_TRY(hipStreamWaitEvent(s, TS->done[l], 0));   // join
    } else {
        for (int l = 0; l < NL; ++l)
            for (int tt = 0; tt < T; ++tt) TRY(lstm_fwd_step(l, tt, s));
    }
    // proj (models/video_autoencoder.py:311-312, 346-349): Conv2d k1 hidden -> latent when the two differ, else Identity
    const float* dec_in = A(p.hseq);
    if (p.proj) {
        PS(TS_CONVT_FWD);
        TRY(vad_conv1x1_p(A(p.hseq), ws + p.pk_pj, P + p.pj_b, A(p.pseq), (long long)N * hw, Hd, L, precision, s));
        dec_in = A(p.pseq);
    }
    // decoder (models/video_autoencoder.py:242-256): convT -> BatchNorm -> ReLU, three times
    for (int j = 0; j < 3; ++j) {
        const int ci = p.decC[j], co = p.decC[j + 1], hj = p.h16 << j, wj = p.w16 << j;
        const float* in = j == 0 ? dec_in : A(p.r[j - 1]);
        float* u = A(p.u[j]);
        int srows = 0;        // > 0: the transposed convolution wrote the BatchNorm partial sums itself
        { PS(TS_CONVT_FWD); TRY(vad_convt2x2_stats(in, 0, ws + p.pk_d[j], P + p.d_b[j], u, 0, N, hj, wj, ci, co, VAD_ACT_NONE, precision, ws + p.chan_ws, &srows, s)); }
        float* rs = running ? running + p.d_rs[j] : nullptr;
        if (srows > 0) { PS(TS_STATS_MISC); TRY(vad_bn_stats_from_partials(ws + p.chan_ws, srows, (long long)N * 4 * hj * wj, co, eps, mom, ws + p.st_d[j], rs, rs ? rs + co : nullptr, P + p.d_b[j], s)); }
        else {
            VAD_REQUIRE(!io, "vid_train_fwd_bwd: bf16 tensors need the transposed convolutions' own BatchNorm partial sums");
            TRY(vad_bn_stats(u, (long long)N * 4 * hj * wj, co, eps, mom, ws + p.st_d[j], rs, rs ? rs + co : nullptr, ws + p.chan_ws, s));
        }
        PS(TS_BN_FWD);
        TRY(vad_bn_act_pool_fwd_t(u, io, ws + p.st_d[j], P + p.d_g[j], P + p.d_be[j], A(p.r[j]), 0, 0, 0, 0, N, 2 * hj, 2 * wj, co, VAD_ACT_RELU, 0, s));
    }
    // last layer + loss, forward and backward (models/video_autoencoder.py:259-260, train_video.py:55)
    float *g0 = A(p.g[0]), *g2 = A(p.g[2]);
    // Split-fp16 mode: the criterion's gradient is 2 (recon - x) / count - ~1e-8 at 32 x 10 x 256x256 -, below the fp16 range the
    // data-gradient and weight-gradient kernels split their operands into (hi = 0, lo subnormal: 8-10 bits left, measured
    // with tools/split_range.py).  The backward is linear in that gradient, so it runs on gradients times a power of two that
    // puts 2 / count at 2^-6 .. 2^-5 (exact in fp32; 2^20 of headroom to the fp16 maximum) and the parameter gradients are
    // scaled back at the end.  The other modes have the fp32 exponent range in every operand: no scaling.
    float grad_mul = 1.f;
    if (precision == VAD_PREC_SPLIT && !wino && g_vad_train_stop < 0 && g_vad_split_grad_scale) {
        int e = 0;
        (void)frexp((double)N * 3.0 * H * W, &e);            // count = m 2^e, m in [0.5, 1)
        grad_mul = (float)ldexp(1.0, e - 7);                  // 2 / count * 2^(e-7) = 2^-6 / m
    }
    { PS(TS_LOSS);
    TRY(vad_convt_to3_mse_t(A(p.r[2]), io, P + p.t_w, P + p.t_b, x, recon, g0, A(p.dpre), loss, G + p.t_b, ws + p.to3_ws, N, H / 2, W / 2, grad_mul, s)); }

    if (g_vad_train_stop == 20) return VAD_OK;      // debug: g0 = gradient of the last decoder activation, dpre intact

    // ================================================================================== backward
    float* dyb[2] = {g2, overlap ? A(p.g[1]) : g2};
    bool dy_pending[2] = {false, false};
    int dy_turn = 0, dy_cur = 0;
    hipStream_t wgs = overlap ? TS->wg : s;
    // the buffer the next BatchNorm backward writes: behind the weight gradient that last read it
    auto acquire_dy = [&]() -> float* {
        dy_cur = dy_turn; dy_turn ^= 1;
        if (dy_pending[dy_cur]) { (void)hipStreamWaitEvent(s, TS->wg_read[dy_cur], 0); dy_pending[dy_cur] = false; }
        return dyb[dy_cur];
    };
    // the helper stream picks up behind everything the caller's stream has launched so far
    auto wg_begin = [&]() -> int {
        if (overlap) { VAD_HIP_TRY(hipEventRecord(TS->wg_in, s)); VAD_HIP_TRY(hipStreamWaitEvent(wgs, TS->wg_in, 0)); }
        return VAD_OK;
    };
    auto wg_end_reads_dy = [&]() -> int {
        if (overlap) { VAD_HIP_TRY(hipEventRecord(TS->wg_read[dy_cur], wgs)); dy_pending[dy_cur] = true; }
        return VAD_OK;
    };
    TRY(wg_begin());
    { PSS(TS_WGRAD, wgs); TRY(vad_conv_wgrad(A(p.r[2]), A(p.dpre), G + p.t_w, ws + p.wgrad_ws, N, H / 2, W / 2, 32, 32, 1, 3, precision, wgs)); }
    for (int j = 2; j >= 0; --j) {
        const int ci = p.decC[j], co = p.decC[j + 1], hj = p.h16 << j, wj = p.w16 << j;
        const float* in = j == 0 ? dec_in : A(p.r[j - 1]);
        // g0 = d r_j (dense, 2hj x 2wj) -> g2 = d u_j in the space-to-depth view [N][hj][wj][4*co]
        g2 = acquire_dy();
        { PS(TS_BN_BWD);
        TRY(vad_bn_act_pool_bwd_t(A(p.u[j]), io, ws + p.st_d[j], P + p.d_g[j], P + p.d_be[j], g0, 0, 0, 0, 0, g2, 1, G + p.d_g[j], G + p.d_be[j],
                                  ws + p.ksums, ws + p.chan_ws, N, 2 * hj, 2 * wj, co, VAD_ACT_RELU, 0, s)); }
        TRY(wg_begin());
        { PSS(TS_WGRAD, wgs); TRY(vad_conv_wgrad(in, g2, G + p.d_w[j], ws + p.wgrad_ws, N, hj, wj, ci, 4 * co, 1, 1, precision, wgs)); }
        TRY(wg_end_reads_dy());
        // bias of a conv that feeds a batch-statistics BatchNorm: sum(dy) = gamma*invstd*(sum(dz) - M*k1 - k2*sum(xhat)) = 0
        // exactly (the batch mean removes any constant).  Autograd returns ~1e-9 rounding noise there, which Adam turns
        // into a +-lr random walk; an exact zero costs no pass over the tensor and leaves the bias where it is.
        VAD_HIP_TRY(hipMemsetAsync(G + p.d_b[j], 0, (size_t)co * sizeof(float), s));
        { PS(TS_CONVT_DGRAD); TRY(vad_conv1x1_p(g2, ws + p.pk_d_dg[j], zeros, g0, (long long)N * hj * wj, 4 * co, ci, precision, s)); }     // g0 = d (input of convT j)
        if (g_vad_train_stop == j) return VAD_OK;
    }
    // g0 = gradient of the decoder input [b*T+t][hw][L]; through proj when present
    const float* dhseq = g0;
    if (p.proj) {
        // (the split-K scratch belongs to the helper stream; this one reads g0, which the chain reuses: it is waited for at once)
        TRY(wg_begin());
        { PSS(TS_WGRAD, wgs); TRY(vad_conv_wgrad(A(p.hseq), g0, G + p.pj_w, ws + p.wgrad_ws, N, p.h16, p.w16, Hd, L, 1, 4, precision, wgs)); }
        if (overlap) { VAD_HIP_TRY(hipEventRecord(TS->wg_in, wgs)); VAD_HIP_TRY(hipStreamWaitEvent(s, TS->wg_in, 0)); }
        { PS(TS_STATS_MISC); TRY(vad_chan_sum_t(g0, io, (long long)N * hw, L, G + p.pj_b, ws + p.chan_ws, s)); }
        g2 = acquire_dy();
        { PS(TS_CONVT_DGRAD); TRY(vad_conv1x1_p(g0, ws + p.pk_pj_dg, zeros, g2, (long long)N * hw, L, Hd, precision, s)); }
        dhseq = g2;
    }
    // dhseq = d hseq [b*T+t][hw][Hd].  BPTT: step (l, t) needs (l, t+1) and (l+1, t) only - the same wavefront, top layer on the
    // caller's stream.
    auto lstm_bwd_step = [&](int l, int tt, hipStream_t st) -> int {
        const int cx = p.lstm_cin(l), cin = cx + Hd;
        const size_t slab = (size_t)B * hw * cin;
        float* dc = ws + p.dc[l];
        const float* dh1; long long dh1_fs; int dh1_ps;
        if (l == NL - 1) { dh1 = at(dhseq, (size_t)tt * hw * Hd); dh1_ps = Hd; dh1_fs = (long long)T * hw * Hd; }
        else { dh1 = A(p.dcat[l + 1], (size_t)tt * B * hw * 2 * Hd); dh1_ps = 2 * Hd; dh1_fs = (long long)hw * dh1_ps; }
        const float* dh2 = tt + 1 < T ? A(p.dcat[l], (tt + 1) * slab + cx) : nullptr;
        float* dzt = A(p.dzl[l], (size_t)tt * B * hw * 4 * Hd);
        const float* ct = ws + p.c[l] + (size_t)tt * B * hw * Hd;
        { PSS(TS_LSTM_GATES_BWD, st);
        TRY(vad_lstm_gates_bwd_t(A(p.z[l], (size_t)tt * B * hw * 4 * Hd), io, tt ? ct - (size_t)B * hw * Hd : nullptr, ct, dh1, dh1_fs, dh1_ps,
                                 dh2, (long long)hw * cin, cin, tt + 1 < T ? dc : nullptr, dzt, dc, B, hw, Hd, st)); }
        PSS(TS_LSTM_CONV_DGRAD, st);
        return conv3(dzt, ws + p.pk_l_dg[l], zeros, A(p.dcat[l], tt * slab), B, p.h16, p.w16, 4 * Hd, cin, nullptr, nullptr, st);
    };
    // weight / bias gradients of a cell's convolution over all its steps at once (frames = T*B)
    auto lstm_wgrad = [&](int l) -> int {
        const int cin = p.lstm_cin(l) + Hd;
        TRY(wg_begin());           // (reads cat[l] / dzl[l] only: nothing later overwrites them)
        { PSS(TS_WGRAD, wgs); TRY(vad_conv_wgrad(A(p.cat[l]), A(p.dzl[l]), G + p.l_w[l], ws + p.wgrad_ws, N, p.h16, p.w16, cin, 4 * Hd, 9, 0, precision, wgs)); }
        PS(TS_STATS_MISC);
        return vad_chan_sum_t(A(p.dzl[l]), io, (long long)N * hw, 4 * Hd, G + p.l_b[l], ws + p.chan_ws, s);
    };
    if (wavefront) {
        VAD_HIP_TRY(hipEventRecord(TS->fork, s));                                   // d hseq is ready
        for (int l = 0; l + 1 < NL; ++l) VAD_HIP_TRY(hipStreamWaitEvent(TS->st[l], TS->fork, 0));
        for (int tt = T - 1; tt >= 0; --tt)
            for (int l = NL - 1; l >= 0; --l) {
                hipStream_t st = (l == NL - 1) ? s : TS->st[l];
                if (l + 1 < NL) VAD_HIP_TRY(hipStreamWaitEvent(st, TS->done[l + 1], 0));   // (l+1, tt) finished
                TRY(lstm_bwd_step(l, tt, st));
                if (l > 0 || tt == 0) VAD_HIP_TRY(hipEventRecord(TS->done[l], st));
            }
        for (int l = 0; l + 1 < NL; ++l) VAD_HIP_TRY(hipStreamWaitEvent(s, TS->done[l], 0));   // join
        for (int l = NL - 1; l >= 0; --l) TRY(lstm_wgrad(l));                       // (shared scratch: on the caller's stream, in order)
    } else {
        for (int l = NL - 1; l >= 0; --l) {
            for (int tt = T - 1; tt >= 0; --tt) TRY(lstm_bwd_step(l, tt, s));
            TRY(lstm_wgrad(l));
            if (g_vad_train_stop == 10 + l) return VAD_OK;
        }
    }
    // encoder, last stage first; the x-part of layer 0's operand gradient is d(latent features)
    for (int k = 3; k >= 0; --k) {
        const int ci = p.encC[k], co = p.encC[k + 1], hk = H >> k, wk = W >> k;
        g2 = acquire_dy();
        // The first layer's dy has one consumer, its weight gradient, which works from the POOLED gradient and one
        // routing byte per pooled element instead (conv_c3_wgrad_routed_kernel): pass A only, no dy (its buffer holds the bytes)
        const bool routed = k == 0 && vad_c3_routed_enabled() && vad_conv_c3_wgrad_routed_ok(hk, wk, co);
        { PS(TS_BN_BWD);
        if (routed)
            TRY(vad_bn_act_pool_bwd_codes_t(A(p.y[k]), io, ws + p.st_e[k], P + p.e_g[k], P + p.e_be[k], g0, 0, 0, 0, 0, nullptr, 0,
                                            G + p.e_g[k], G + p.e_be[k], ws + p.ksums, ws + p.chan_ws, N, hk, wk, co, VAD_ACT_LEAKY, 1, (unsigned char*)g2, s));
        else if (k == 3)
            TRY(vad_bn_act_pool_bwd_t(A(p.y[k]), io, ws + p.st_e[k], P + p.e_g[k], P + p.e_be[k], A(p.dcat[0]), 0, L + Hd, T, B, g2, 0,
                                      G + p.e_g[k], G + p.e_be[k], ws + p.ksums, ws + p.chan_ws, N, hk, wk, co, VAD_ACT_LEAKY, 1, s));
        else
            TRY(vad_bn_act_pool_bwd_t(A(p.y[k]), io, ws + p.st_e[k], P + p.e_g[k], P + p.e_be[k], g0, 0, 0, 0, 0, g2, 0,
                                      G + p.e_g[k], G + p.e_be[k], ws + p.ksums, ws + p.chan_ws, N, hk, wk, co, VAD_ACT_LEAKY, 1, s));
        }
        VAD_HIP_TRY(hipMemsetAsync(G + p.e_b[k], 0, (size_t)co * sizeof(float), s));      // structurally zero, see the decoder loop
        TRY(wg_begin());
        if (routed) {
            PSS(TS_C3_WGRAD, wgs);
            TRY(vad_conv_c3_wgrad_routed(x, g0, io, (const unsigned char*)g2, P + p.e_w[0], P + p.e_b[0], ws + p.st_e[0], P + p.e_g[0], ws + p.ksums,
                                         G + p.e_w[0], ws + p.wgrad_ws, N, hk, wk, co, wgs));
        } else if (k == 0) {
            PSS(TS_C3_WGRAD, wgs);
            TRY(vad_conv_c3_wgrad_t(x, g2, io, G + p.e_w[0], ws + p.wgrad_ws, N, hk, wk, co, wgs));
        } else {
            { PSS(TS_WGRAD, wgs); TRY(vad_conv_wgrad(A(p.a[k - 1]), g2, G + p.e_w[k], ws + p.wgrad_ws, N, hk, wk, ci, co, 9, 0, precision, wgs)); }
            TRY(wg_end_reads_dy());
            PS(TS_CONV_DGRAD);
            TRY(conv3(g2, ws + p.pk_e_dg[k], zeros, g0, N, hk, wk, co, ci, nullptr, nullptr, s));
        }
    }
    if (overlap) {             // join: the optimiser reads every gradient
        VAD_HIP_TRY(hipEventRecord(TS->wg_in, wgs));
        VAD_HIP_TRY(hipStreamWaitEvent(s, TS->wg_in, 0));
    }
    if (grad_mul != 1.f) TRY(vad_scale_floats(G, (long long)vad_vid_train_nparams(latent, hid, layers), 1.f / grad_mul, s));
    return VAD_OK;
}
