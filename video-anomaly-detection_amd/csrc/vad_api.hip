// Whole-model launch sequences (image autoencoder, ConvLSTM video autoencoder) and the per-layer
// hipEvent timing used by bench.py.  No allocation, no synchronisation on the launch path.
//
// Launch order follows ConvAutoencoder.forward / get_reconstruction_error
// (reference models/autoencoder.py:181-221) and VideoAutoencoder.forward /
// get_reconstruction_error (reference models/video_autoencoder.py:329-384).
#include <atomic>
#include <mutex>
#include <vector>
#include "vad_common.h"
#include "vad_layout.h"

// ------------------------------------------------------------------------------ profiling
// The record list and the event pool are shared by every thread that scores while profiling is on: one mutex guards
// them, and a scope keeps its OWN end event (another thread may append records between its two ends).
namespace {
struct ProfRec { int slot; hipEvent_t a, b; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;
std::vector<ProfRec> g_recs;      // records since the last reset
std::vector<hipEvent_t> g_pool;   // recycled events
size_t g_pool_used = 0;

hipEvent_t prof_event() {         // g_prof_mu held
    if (g_pool_used == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_used++];
}
}  // namespace

VadProfScope::VadProfScope(int slot_, hipStream_t stream_) : slot(-1), stream(stream_), end(nullptr) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;     // timing events have no place inside a captured graph
    if (hipStreamIsCapturing(stream_, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return;
    hipEvent_t a, b;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        a = prof_event(); b = prof_event();
        if (!a || !b) return;
        g_recs.push_back({slot_, a, b});
    }
    slot = slot_;
    end = b;
    (void)hipEventRecord(a, stream);
}
VadProfScope::~VadProfScope() {
    if (slot >= 0) (void)hipEventRecord(end, stream);
}

extern "C" int vad_prof_enable(int on) { g_prof_on = on != 0; return VAD_OK; }
extern "C" int vad_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_recs.clear(); g_pool_used = 0;
    return VAD_OK;
}
extern "C" int vad_prof_read(float* ms, int* launches) {
    VAD_REQUIRE(ms && launches, "prof_read: null pointer");
    for (int i = 0; i < VAD_PROF_SLOTS; ++i) { ms[i] = 0.f; launches[i] = 0; }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (const ProfRec& r : g_recs) {
        VAD_HIP_TRY(hipEventSynchronize(r.b));
        float t = 0.f;
        VAD_HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        if (r.slot >= 0 && r.slot < VAD_PROF_SLOTS) { ms[r.slot] += t; launches[r.slot] += 1; }
    }
    return VAD_OK;
}

static const char* kImgSlots[] = {"enc1.0", "enc1.0+enc1.3+pool", "enc2.0", "enc2.3+pool", "enc3.0", "enc3.3+pool",
                                  "enc4.0", "enc4.3+pool", "dec1.0", "dec1.3", "dec2.0", "dec2.3", "dec3.0",
                                  "dec3.3", "dec4.0", "dec4.3+score", "finalize", "latent_nchw", "dec4.0+dec4.3+score"};
static const char* kVidSlots[] = {"enc.0+pool", "enc.4+pool", "enc.8+pool", "enc.12+pool", "convlstm", "proj",
                                  "dec.0", "dec.3", "dec.6", "dec.9+score", "finalize"};
static const char* kTrainSlots[] = {"first layer fwd", "conv3x3 fwd", "BatchNorm fwd", "ConvLSTM conv fwd", "ConvLSTM gates fwd", "convT / proj fwd",
                                    "last layer + loss", "weight gradients", "BatchNorm bwd", "convT / proj dgrad (1x1)", "ConvLSTM gates bwd",
                                    "ConvLSTM conv dgrad", "conv3x3 dgrad", "first layer wgrad", "operand packing", "statistics finalize + bias sums"};
extern "C" const char* vad_prof_slot_name(int model, int slot) {
    if (model == 2 && slot >= 0 && slot < (int)(sizeof kTrainSlots / sizeof *kTrainSlots)) return kTrainSlots[slot];
    if (model == 0 && slot >= 0 && slot < (int)(sizeof kImgSlots / sizeof *kImgSlots)) return kImgSlots[slot];
    if (model == 1 && slot >= 0 && slot < (int)(sizeof kVidSlots / sizeof *kVidSlots)) return kVidSlots[slot];
    return "";
}

#define TRY(call) do { int rc_ = (call); if (rc_ != VAD_OK) return rc_; } while (0)

// ------------------------------------------------------------------------------ hipGraph capture / replay
// The scoring entry points only launch kernels (and record / wait events between `stream` and the library's helper
// streams), so ONE call can be captured into a hipGraph and replayed: at the reference's call sizes (batch 16 images,
// evaluate.py:240; 4 clips x 16 frames, evaluate_video.py:416; one window, evaluate_video.py:344) a call is 16-45 short
// launches and the host-side launch cost is a visible part of it.
extern "C" int vad_graph_begin(void* stream) {
    VAD_HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return VAD_OK;
}
extern "C" int vad_graph_end(void* stream, void** exec_out) {
    VAD_REQUIRE(exec_out, "graph_end: null pointer");
    *exec_out = nullptr;
    hipGraph_t graph = nullptr;
    VAD_HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &graph));
    VAD_REQUIRE(graph, "graph_end: the capture produced no graph (was it invalidated by a synchronising call?)");
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return vad_fail(VAD_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    *exec_out = (void*)exec;
    return VAD_OK;
}
extern "C" int vad_graph_launch(void* exec, void* stream) {
    VAD_REQUIRE(exec, "graph_launch: null graph");
    VAD_HIP_TRY(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream));
    return VAD_OK;
}
extern "C" int vad_graph_destroy(void* exec) {
    if (exec) VAD_HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)exec));
    return VAD_OK;
}

static std::atomic<int> g_vad_tail_group{0};   // debug: frames per dec4.0 -> tail sub-group (0 = the whole launch group)
extern "C" int vad_debug_set_tail_group(int frames) { g_vad_tail_group = frames; return VAD_OK; }
#define REQ_PREC(who) VAD_REQUIRE(precision == VAD_PREC_FP32 || precision == VAD_PREC_SPLIT || precision == VAD_PREC_WINO, who ": precision=%d must be VAD_PREC_FP32 (0), VAD_PREC_SPLIT (1) or VAD_PREC_WINO (4)", precision)

// A 3x3 convolution behind the first layer in the model's arithmetic mode: VAD_PREC_WINO blobs hold the Winograd form of these
// layers (csrc/conv_wino.hip); every other kernel of such a model runs the VAD_PREC_FP32 arithmetic (`bprec`).
static int conv3x3_mode(const float* in, const float* w, const float* bias, float* out, int n, int h, int wd, int cin, int cout, int act, int pool,
                        int precision, hipStream_t s) {
    if (precision == VAD_PREC_WINO) return vad_conv3x3_wino(in, 0, w, bias, out, 0, n, h, wd, cin, cout, act, pool, s);
    return vad_conv3x3(in, 0, w, bias, out, 0, n, h, wd, cin, cout, act, pool, precision, s);
}

static size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

// ------------------------------------------------------------------------------ image model
static size_t img_act_floats(int h, int w, int latent) {
    size_t m = (size_t)h * w * 32;                                   // enc1.0 / dec4.0 outputs
    const size_t e4 = (size_t)(h / 8) * (w / 8) * latent;            // enc4.0 output
    return e4 > m ? e4 : m;
}

// per-frame partial sums: the larger of the two tails' counts (unfused 32x32 tiles; fused: one per output row, strip and wave)
static size_t img_partials(int h, int w) {
    const size_t a = (size_t)vad_score_partials(0, h, w), b = (size_t)vad_dec4_score_partials(h, w), c = (size_t)vad_wide_score_partials(h, w);
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

static std::atomic<int> g_vad_dec4_fused{1};   // debug / A-B: 0 = dec4.0 and the scoring tail as two launches (the round-2 path)
extern "C" int vad_debug_set_dec4_fused(int on) { g_vad_dec4_fused = on; return VAD_OK; }

extern "C" size_t vad_img_workspace_bytes_c(int chunk, int h, int w, int latent, int in_ch) {
    if (in_ch < 3 || in_ch > VAD_MAX_IN_CH) return 0;
    return vad_img_workspace_bytes(chunk, h, w, latent);          // (the planes of a wide model pad to 32 channels: the size of the largest map either way)
}

extern "C" size_t vad_img_workspace_bytes(int chunk, int h, int w, int latent) {
    if (chunk <= 0 || h <= 0 || w <= 0 || h % 16 || w % 16 || latent <= 0 || latent > VAD_MAX_WIDTH) return 0;
    latent = vad_img_latent_p(latent);
    const size_t act = up256(sizeof(float) * chunk * img_act_floats(h, w, latent));
    const size_t parts = up256(sizeof(float) * chunk * img_partials(h, w));
    return 2 * act + parts;
}

extern "C" int vad_img_score(const float* x, long long b, int h, int w, int latent, const float* packed,
                             void* ws, size_t ws_bytes, int chunk, float* scores, float* errmap,
                             float* recon, float* latent_out, void* stream) {
    return vad_img_score_x(x, VAD_X_F32_NCHW, VAD_PREC_FP32, b, h, w, latent, packed, ws, ws_bytes, chunk, scores, errmap, recon,
                           latent_out, stream);
}

extern "C" int vad_img_score_x(const void* xv, int x_format, int precision, long long b, int h, int w, int latent_real,
                               const float* packed, void* ws, size_t ws_bytes, int chunk, float* scores, float* errmap,
                               float* recon, float* latent_out, void* stream) {
    return vad_img_score_c(xv, x_format, precision, 3, b, h, w, latent_real, packed, ws, ws_bytes, chunk, scores, errmap, recon, latent_out, stream);
}

extern "C" int vad_img_score_c(const void* xv, int x_format, int precision, int in_ch, long long b, int h, int w, int latent_real,
                               const float* packed, void* ws, size_t ws_bytes, int chunk, float* scores, float* errmap,
                               float* recon, float* latent_out, void* stream) {
    VAD_REQUIRE(xv && packed && ws, "img_score: null pointer");
    REQ_PREC("img_score");
    VAD_REQUIRE(x_format == VAD_X_F32_NCHW || x_format == VAD_X_U8_NHWC, "img_score: unknown input format %d", x_format);
    VAD_REQUIRE(in_ch >= 3 && in_ch <= VAD_MAX_IN_CH, "img_score: in_channels=%d out of range [3,%d]", in_ch, VAD_MAX_IN_CH);
    VAD_REQUIRE(in_ch == 3 || x_format == VAD_X_F32_NCHW, "img_score: uint8 frames are 3-channel images (in_channels=%d)", in_ch);
    const size_t xelem = x_format == VAD_X_U8_NHWC ? 1 : 4;
    const char* x = (const char*)xv;
    VAD_REQUIRE(b > 0 && chunk > 0, "img_score: batch=%lld chunk=%d must be positive", b, chunk);
    VAD_REQUIRE(h > 0 && w > 0 && h % 16 == 0 && w % 16 == 0,
                "img_score: H=%d W=%d must be positive multiples of 16 (4 MaxPool2d(2) stages)", h, w);
    VAD_REQUIRE(latent_real > 0 && latent_real <= VAD_MAX_WIDTH, "img_score: latent_dim=%d out of range [1,%d]", latent_real, VAD_MAX_WIDTH);
    VAD_REQUIRE(scores || errmap || recon || latent_out, "img_score: no output requested");
    const size_t need = vad_img_workspace_bytes(chunk, h, w, latent_real);
    if (ws_bytes < need) return vad_fail(VAD_ERR_WS, "img_score: workspace %zu B < required %zu B", ws_bytes, need);
    VAD_REQUIRE(((uintptr_t)ws & 255) == 0 && ((uintptr_t)packed & 15) == 0, "img_score: workspace must be 256-B and weights 16-B aligned");

    hipStream_t s = (hipStream_t)stream;
    const ImgLayout L = img_layout(in_ch, latent_real);
    const int latent = L.latent_p;                  // the width the kernels see (zero-padded to a multiple of 32, vad_layout.h)
    const int wide = L.wide;                        // in_channels > 3: generic first / last layers over planes padded to `wide` channels
    const size_t act = up256(sizeof(float) * chunk * img_act_floats(h, w, latent));
    float* A = (float*)ws;
    float* B = (float*)((char*)ws + act);
    float* parts = (float*)((char*)ws + 2 * act);
    const int nparts_tail = vad_score_partials(0, h, w);
    const bool need_decoder = scores || errmap || recon;
    const int ch[5] = {3, 32, 64, 128, latent};
    const int dch[5] = {latent, 128, 64, 32, 32};
    const int bprec = precision == VAD_PREC_WINO ? VAD_PREC_FP32 : precision;   // arithmetic of the kernels that are not 3x3 convolutions
#define W_(i) (packed + L.layer[i].w)
#define B_(i) (packed + L.layer[i].b)

    for (long long f0 = 0; f0 < b; f0 += chunk) {
        const int n = (int)((b - f0 < chunk) ? (b - f0) : chunk);
        const char* xin = x + (size_t)f0 * in_ch * h * w * xelem;
        int hh = h, ww = w;
        // encoder: 4 x [conv-BN-LeakyReLU, conv-BN-LeakyReLU-MaxPool] (models/autoencoder.py:38-79)
        if (wide) {                         // csrc/wide_io.hip: planes -> zero-padded NHWC, then two generic layers
            { VadProfScope ps(0, s);
              TRY(vad_nchw_to_nhwc_pad((const float*)xin, B, n, hh, ww, in_ch, wide, s));
              TRY(conv3x3_mode(B, W_(0), B_(0), A, n, hh, ww, wide, 32, VAD_ACT_LEAKY, 0, precision, s)); }
            { VadProfScope ps(1, s); TRY(conv3x3_mode(A, W_(1), B_(1), B, n, hh, ww, 32, 32, VAD_ACT_LEAKY, 1, precision, s)); }
        } else if (precision == VAD_PREC_WINO) {   // first layer on its own (K = 27: nothing to gain from Winograd), then enc1.3 + pool in Winograd form
            { VadProfScope ps(0, s); TRY(vad_conv3x3_c3_fmt(xin, x_format, W_(0), B_(0), A, n, hh, ww, 32, VAD_ACT_LEAKY, 0, s)); }
            { VadProfScope ps(1, s); TRY(conv3x3_mode(A, W_(1), B_(1), B, n, hh, ww, 32, 32, VAD_ACT_LEAKY, 1, precision, s)); }
        } else {
            VadProfScope ps(1, s); TRY(vad_conv3x3_c3_fused_fmt(xin, x_format, W_(0), B_(0), W_(1), B_(1), B, n, hh, ww, precision, s));
        }
        for (int blk = 1; blk < 4; ++blk) {
            hh /= 2; ww /= 2;
            { VadProfScope ps(2 * blk, s);
              TRY(conv3x3_mode(B, W_(2 * blk), B_(2 * blk), A, n, hh, ww, ch[blk], ch[blk + 1], VAD_ACT_LEAKY, 0, precision, s)); }
            { VadProfScope ps(2 * blk + 1, s);
              TRY(conv3x3_mode(A, W_(2 * blk + 1), B_(2 * blk + 1), B, n, hh, ww, ch[blk + 1], ch[blk + 1], VAD_ACT_LEAKY, 1, precision, s)); }
        }
        hh /= 2; ww /= 2;   // B = latent code [n, H/16, W/16, latent]
        if (latent_out) {
            VadProfScope ps(17, s);
            TRY(vad_nhwc_to_nchw_ld(B, latent, latent_out + (size_t)f0 * latent_real * hh * ww, n, hh, ww, latent_real, s));
        }
        if (!need_decoder) continue;
        // decoder: 3 x [convT-BN-ReLU, conv-BN-ReLU] + [convT-BN-ReLU, conv-Tanh] (models/autoencoder.py:103-139)
        for (int blk = 0; blk < 3; ++blk) {
            { VadProfScope ps(8 + 2 * blk, s);
              TRY(vad_convt2x2(B, 0, W_(8 + 2 * blk), B_(8 + 2 * blk), A, 0, n, hh, ww, dch[blk], dch[blk + 1], VAD_ACT_RELU, bprec, s)); }
            hh *= 2; ww *= 2;
            { VadProfScope ps(9 + 2 * blk, s);
              TRY(conv3x3_mode(A, W_(9 + 2 * blk), B_(9 + 2 * blk), B, n, hh, ww, dch[blk + 1], dch[blk + 1], VAD_ACT_RELU, 0, precision, s)); }
        }
        // dec4.0 + dec4.3 + score: ONE kernel, the 8.4 MB per frame map between the two layers never exists in memory
        // (dec4_fused.hip), in BOTH arithmetic modes: these layers are HBM-bound, so the split mode keeps them exact fp32
        // (dec4.0 is packed as fp32 in every blob).  A/B switch: two launches, convT 32->32 then the VALU tail; those
        // can run in sub-groups (vad_debug_set_tail_group) so that the map stays in the 256 MiB Infinity Cache - measured
        // on MI355X this does NOT pay (15.2 k frames/s whole group vs 14.9 k at 16 frames), so the default is the whole group.
        const size_t in_f = (size_t)hh * ww * 32;
        int nparts = nparts_tail;
        if (wide) {                         // convT 32 -> 32, Conv2d(32 -> in_ch) into padded planes, Tanh + score (wide_io.hip)
            nparts = vad_wide_score_partials(h, w);
            { VadProfScope ps(14, s); TRY(vad_convt2x2(B, 0, W_(14), B_(14), A, 0, n, hh, ww, 32, 32, VAD_ACT_RELU, VAD_PREC_FP32, s)); }
            { VadProfScope ps(15, s);
              TRY(conv3x3_mode(A, W_(15), B_(15), B, n, h, w, 32, wide, VAD_ACT_NONE, 0, precision, s));
              TRY(vad_tanh_score_nhwc(B, wide, (const float*)xin, in_ch, parts, recon ? recon + (size_t)f0 * in_ch * h * w : nullptr,
                                      errmap ? errmap + (size_t)f0 * h * w : nullptr, n, h, w, 0, 0, s)); }
        } else if (g_vad_dec4_fused.load(std::memory_order_relaxed)) {
            nparts = vad_dec4_score_partials(h, w);
            VadProfScope ps(18, s);
            TRY(vad_dec4_score_fmt(B, W_(14), B_(14), W_(15) + 8 * 108, B_(15), xin, x_format, parts,
                                   recon ? recon + (size_t)f0 * 3 * h * w : nullptr, errmap ? errmap + (size_t)f0 * h * w : nullptr,
                                   n, hh, ww, s));
        } else {
        const int tg = g_vad_tail_group.load(std::memory_order_relaxed);
        const int sub = tg > 0 ? tg : n;
        for (int f1 = 0; f1 < n; f1 += sub) {
            const int m = (n - f1 < sub) ? (n - f1) : sub;
            { VadProfScope ps(14, s);
              TRY(vad_convt2x2(B + (size_t)f1 * in_f, 0, W_(14), B_(14), A, 0, m, hh, ww, 32, 32, VAD_ACT_RELU, VAD_PREC_FP32, s)); }
            { VadProfScope ps(15, s);
              const size_t fo = (size_t)(f0 + f1);
              TRY(vad_conv3x3_to3_score_fmt(A, W_(15), B_(15), xin + (size_t)f1 * 3 * h * w * xelem, x_format, parts + (size_t)f1 * nparts,
                                        recon ? recon + fo * 3 * h * w : nullptr,
                                        errmap ? errmap + fo * h * w : nullptr, m, h, w, 32, s)); }
        }
        }
        if (scores) {
            VadProfScope ps(16, s);
            TRY(vad_score_finalize_tagged(parts, nparts, n, h, w, in_ch, scores + f0, nullptr, 1, (const unsigned*)packed,
                                          vad_blob_tag(VAD_BLOB_IMG, precision), s));
        }
    }
#undef W_
#undef B_
    return VAD_OK;
}

// ------------------------------------------------------------------------------ video model
// One code path serves both clip batches and dense sliding windows: "clip" c covers source frames
// [c*cs, c*cs + T).  cs == T: independent clips [B,T,...] (evaluate_video.py:138-154).  cs == stride < T:
// overlapping windows of ONE video (evaluate_video.py:322-352, VideoFileDataset(sequence_length, stride)); every
// source frame is then encoded ONCE instead of once per window that contains it, and the ConvLSTM reads window
// c's frame t straight from that shared feature buffer (its state still restarts from zero per window, as in
// models/video_autoencoder.py:144-145).
namespace {
struct VidWs {
    size_t act, enc, hseq, cst, proj, parts, zx0, zxl, zw, total;
};
// work-groups of one ConvLSTM step in the large (32x32x2) tiling: below one per CU the layers run as a wavefront on helper
// streams and the steps' x halves are computed ahead of the recurrence
long long vid_lstm_groups(int nc, int h16, int w16, int hid) { return (long long)nc * ((w16 + 15) / 16) * ((h16 + 3) / 4) * (hid / 64); }
VidWs vid_ws(int chunk, int t, int cs, int h, int w, int latent_real, int hid_real, int layers, int in_ch) {
    VidWs z{};
    const bool wide = in_ch > 3;
    const int latent = vad_vid_latent_p(latent_real, hid_real), hid = vad_vid_hid_p(latent_real, hid_real);
    const size_t n = (size_t)chunk * t, nf = (size_t)(chunk - 1) * cs + t, p16 = (size_t)(h / 16) * (w / 16);
    const size_t nmax = n > nf ? n : nf;
    z.act = up256(sizeof(float) * nmax * (size_t)h * w * (wide ? 32 : 8));       // [frames, H/2, W/2, 32]; wide models: [frames, H, W, 32] padded planes
    z.enc = up256(sizeof(float) * nf * p16 * latent);
    z.hseq = up256(sizeof(float) * n * p16 * hid);
    z.cst = up256(sizeof(float) * (size_t)chunk * p16 * hid);
    z.proj = (hid_real != latent_real) ? up256(sizeof(float) * n * p16 * latent) : 0;
    z.parts = up256(sizeof(float) * n * (size_t)(wide ? vad_wide_score_partials(h, w) : vad_score_partials(1, h, w)));
    // small launch groups: bias + x half of every step's gate pre-activations, [frames][h/16][w/16][4*hid] - layer 0 per SOURCE
    // frame (overlapping windows share them), the layers above per (clip, t)
    const bool small = vid_lstm_groups(chunk, h / 16, w / 16, hid) < 256;
    z.zx0 = small ? up256(sizeof(float) * nf * p16 * 4 * hid) : 0;
    z.zxl = small ? up256(sizeof(float) * n * p16 * 4 * hid) : 0;
    z.zw = up256(sizeof(float) * (size_t)chunk * p16 * 4 * hid);   // VAD_PREC_WINO: one step's gate pre-activations (vad_convlstm_step_wino)
    z.total = 2 * z.act + z.enc + (size_t)layers * (z.hseq + z.cst + z.zw) + z.proj + z.parts + z.zx0 + (size_t)(layers - 1) * z.zxl;   // h sequence + cell state (+ z) per layer
    return z;
}

// Helper streams for the ConvLSTM layer wavefront (small batches): layer l runs on side stream l-1 and step t of layer l
// waits only for step t of layer l-1, so layer 1 step t overlaps layer 0 step t+1 (the reference's loop is strictly
// layers-outer, models/video_autoencoder.py:153-160; the data dependences allow the diagonal order).  Created once per
// thread and device on the first small-batch call; fork / join through events on the caller's stream, so the call stays
// asynchronous and capturable into a hipGraph (after one eager call has created the streams).
struct VadSideStreams {
    int dev = -1, n = 0;
    hipStream_t st[7] = {};       // st[l-1]: the steps of layer l
    hipStream_t xs[7] = {};       // xs[l-1]: the x halves of layer l's steps (one launch per step, as soon as layer l-1 has produced its input)
    hipEvent_t done[8] = {};      // done[l]: layer l finished its latest step
    hipEvent_t xdone[8] = {};     // xdone[l]: the x half of layer l's next step is ready
    hipEvent_t fork = nullptr;
};
static thread_local VadSideStreams t_side;

static int side_streams(int layers, VadSideStreams** out) {
    int dev = 0;
    VAD_HIP_TRY(hipGetDevice(&dev));
    VadSideStreams& S = t_side;
    if (S.dev != dev) {            // first use on this device by this thread (streams of another device are left to the runtime)
        S = VadSideStreams{};
        S.dev = dev;
        VAD_HIP_TRY(hipEventCreateWithFlags(&S.fork, hipEventDisableTiming));
        for (int l = 0; l < 8; ++l) VAD_HIP_TRY(hipEventCreateWithFlags(&S.done[l], hipEventDisableTiming));
        for (int l = 0; l < 8; ++l) VAD_HIP_TRY(hipEventCreateWithFlags(&S.xdone[l], hipEventDisableTiming));
    }
    while (S.n < layers - 1) {
        VAD_HIP_TRY(hipStreamCreateWithFlags(&S.st[S.n], hipStreamNonBlocking));
        VAD_HIP_TRY(hipStreamCreateWithFlags(&S.xs[S.n], hipStreamNonBlocking));
        ++S.n;
    }
    *out = &S;
    return VAD_OK;
}

static std::atomic<int> g_vad_lstm_wavefront{1};   // debug: 0 = always the sequential layers-outer order, 2 = wavefront at any size
extern "C" int vad_debug_set_lstm_wavefront(int on) { g_vad_lstm_wavefront = on; return VAD_OK; }
extern "C" int vad_lstm_wavefront_mode(void) { return g_vad_lstm_wavefront.load(std::memory_order_relaxed); }   // the training step shares the switch

// clips [c0, c0+nc) of a stream whose clip c starts at source frame c*cs; x points at source frame 0 of the stream
int vid_run(const void* xv, int x_format, int precision, int in_ch, long long nclips, int t, int cs, int h, int w, int latent_real, int hid_real, int layers,
            const float* packed, void* ws, size_t ws_bytes, int chunk, float* seq_scores, float* frame_scores,
            float* errmap, float* recon, hipStream_t s, const char* who) {
    VAD_REQUIRE(x_format == VAD_X_F32_NCHW || x_format == VAD_X_U8_NHWC, "%s: unknown input format %d", who, x_format);
    VAD_REQUIRE(precision == VAD_PREC_FP32 || precision == VAD_PREC_SPLIT || precision == VAD_PREC_WINO, "%s: precision=%d must be VAD_PREC_FP32 (0), VAD_PREC_SPLIT (1) or VAD_PREC_WINO (4)", who, precision);
    const int mprec = precision;                                                  // the blob's mode (device-side tag check)
    const bool wino = precision == VAD_PREC_WINO;
    if (wino) precision = VAD_PREC_FP32;                                          // everything but the encoder's 3x3 convolutions
    const size_t xelem = x_format == VAD_X_U8_NHWC ? 1 : 4;
    const char* x = (const char*)xv;
    VAD_REQUIRE(in_ch >= 3 && in_ch <= VAD_MAX_IN_CH, "%s: in_channels=%d out of range [3,%d]", who, in_ch, VAD_MAX_IN_CH);
    VAD_REQUIRE(in_ch == 3 || x_format == VAD_X_F32_NCHW, "%s: uint8 frames are 3-channel images (in_channels=%d)", who, in_ch);
    const VidWs Z = vid_ws(chunk, t, cs, h, w, latent_real, hid_real, layers, in_ch);
    if (ws_bytes < Z.total) return vad_fail(VAD_ERR_WS, "%s: workspace %zu B < required %zu B", who, ws_bytes, Z.total);
    VAD_REQUIRE(((uintptr_t)ws & 255) == 0 && ((uintptr_t)packed & 15) == 0, "%s: workspace must be 256-B and weights 16-B aligned", who);
    const VidLayout L = vid_layout(in_ch, latent_real, hid_real, layers);
    const int latent = L.latent_p, hid = L.hid_p;   // the widths the kernels see (zero-padded, vad_layout.h)
    const int wide = L.wide;                        // in_channels > 3: generic first / last layers over planes padded to `wide` channels
    char* base = (char*)ws;
    float* A = (float*)base; base += Z.act;
    float* Bf = (float*)base; base += Z.act;
    float* E = (float*)base; base += Z.enc;
    float* HS[8];
    float* CS[8];
    for (int l = 0; l < layers; ++l) { HS[l] = (float*)base; base += Z.hseq; }
    for (int l = 0; l < layers; ++l) { CS[l] = (float*)base; base += Z.cst; }
    float* P = nullptr;
    if (L.has_proj) { P = (float*)base; base += Z.proj; }
    float* parts = (float*)base; base += Z.parts;
    float* ZX[8] = {};
    if (Z.zx0) {
        ZX[0] = (float*)base; base += Z.zx0;
        for (int l = 1; l < layers; ++l) { ZX[l] = (float*)base; base += Z.zxl; }
    }
    float* ZW[8];                                   // per layer: the layers of a small launch group run concurrently (wavefront)
    for (int l = 0; l < layers; ++l) { ZW[l] = (float*)base; base += Z.zw; }
    const int nparts = wide ? vad_wide_score_partials(h, w) : vad_score_partials(1, h, w);
    const int h16 = h / 16, w16 = w / 16;
    const long long fs_lat = (long long)h16 * w16 * latent, fs_hid = (long long)h16 * w16 * hid;
#define W_(i) (packed + L.layer[i].w)
#define B_(i) (packed + L.layer[i].b)

    for (long long c0 = 0; c0 < nclips; c0 += chunk) {
        const int nc = (int)((nclips - c0 < chunk) ? (nclips - c0) : chunk);
        const int n = nc * t;                       // (clip, t) frames that are decoded and scored
        const int nf = (nc - 1) * cs + t;           // distinct source frames that are encoded
        const char* xin = x + (size_t)c0 * cs * in_ch * h * w * xelem;
        // VideoEncoder: 4 x conv-BN-LeakyReLU-MaxPool on the flattened frames
        // (models/video_autoencoder.py:191-215, :222-228)
        if (wide) {
            VadProfScope ps(0, s);
            TRY(vad_nchw_to_nhwc_pad((const float*)xin, Bf, nf, h, w, in_ch, wide, s));
            TRY(conv3x3_mode(Bf, W_(0), B_(0), A, nf, h, w, wide, 32, VAD_ACT_LEAKY, 1, mprec, s));
        } else { VadProfScope ps(0, s); TRY(vad_conv3x3_c3_fmt(xin, x_format, W_(0), B_(0), A, nf, h, w, 32, VAD_ACT_LEAKY, 1, s)); }
        { VadProfScope ps(1, s); TRY(conv3x3_mode(A, W_(1), B_(1), Bf, nf, h / 2, w / 2, 32, 64, VAD_ACT_LEAKY, 1, mprec, s)); }
        { VadProfScope ps(2, s); TRY(conv3x3_mode(Bf, W_(2), B_(2), A, nf, h / 4, w / 4, 64, 128, VAD_ACT_LEAKY, 1, mprec, s)); }
        { VadProfScope ps(3, s); TRY(conv3x3_mode(A, W_(3), B_(3), E, nf, h / 8, w / 8, 128, latent, VAD_ACT_LEAKY, 1, mprec, s)); }
        // ConvLSTM, zero initial state (models/video_autoencoder.py:144-166).  Step (l, t) needs (l, t-1) and (l-1, t) only.
        const long long fs_zx = (long long)h16 * w16 * 4 * hid;
        // x halves ahead of the recurrence (small launch groups, exact fp32): a step's accumulator chain runs over the x chunks
        // first and the h chunks second, so bias + x half can be computed for ALL steps of layer 0 in one batched convolution
        // (per source frame: overlapping windows share it) and per step for the layers above, stored as fp32 and resumed by
        // the step kernel - bit-identical, and the serial K loop of a step halves (models/video_autoencoder.py:67-70 multiplies
        // cat([x, h]) inside the recurrence).
        // VAD_PREC_WINO: every step's gate convolution in Winograd form (both sources of every layer have one width), whatever
        // the launch group's size - the arithmetic of a model never depends on the batch
        // (layer 0 stays direct when latent_dim and lstm_hidden_dim pad to different widths - the packer makes the same test)
        auto wino_layer = [&](int l) { return wino && (l > 0 || latent == hid); };
        const bool hoist = !wino && precision == VAD_PREC_FP32 && ZX[0] && vad_convlstm_hoist_ok();
        bool hoist_upper = hoist;
        auto lstm_step = [&](int l, int ti, hipStream_t st) -> int {
            const float* xin_l = (l == 0) ? E : HS[l - 1];
            const long long fs_in = (l == 0) ? fs_lat : fs_hid;
            const long long clip_in = (l == 0) ? (long long)cs * fs_lat : (long long)t * fs_hid;   // layer 0 reads the shared features
            const long long clip_zx = (l == 0) ? (long long)cs * fs_zx : (long long)t * fs_zx;
            VadProfScope ps(4, st);
            if (wino_layer(l))
                return vad_convlstm_step_wino(xin_l + (size_t)ti * fs_in, clip_in, ti ? HS[l] + (size_t)(ti - 1) * fs_hid : nullptr, (long long)t * fs_hid,
                                              ti ? CS[l] : nullptr, W_(4 + l), B_(4 + l), HS[l] + (size_t)ti * fs_hid, (long long)t * fs_hid, CS[l], ZW[l],
                                              nc, h16, w16, hid, hid, st);
            return vad_convlstm_step_zx(xin_l + (size_t)ti * fs_in, clip_in, (hoist && (l == 0 || hoist_upper)) ? ZX[l] + (size_t)ti * fs_zx : nullptr, clip_zx,
                                        ti ? HS[l] + (size_t)(ti - 1) * fs_hid : nullptr, (long long)t * fs_hid,
                                        ti ? CS[l] : nullptr, W_(4 + l), B_(4 + l),
                                        HS[l] + (size_t)ti * fs_hid, (long long)t * fs_hid, CS[l],
                                        nc, h16, w16, (l == 0) ? latent : hid, hid, precision, st);
        };
        // bias + x half of layer l's step ti for the nc clips (l >= 1), or of every source frame (l == 0, ti < 0)
        auto lstm_xhalf = [&](int l, int ti, hipStream_t st) -> int {
            VadProfScope ps(4, st);
            if (l == 0)
                return vad_conv3x3_kpart(E, 0, W_(4), B_(4), ZX[0], 0, nf, h16, w16, latent, latent + hid, 4 * hid, VAD_ACT_NONE, 0,
                                         VAD_PREC_FP32, nullptr, nullptr, st);
            return vad_conv3x3_kpart(HS[l - 1] + (size_t)ti * fs_hid, (long long)t * fs_hid, W_(4 + l), B_(4 + l), ZX[l] + (size_t)ti * fs_zx,
                                     (long long)t * fs_zx, nc, h16, w16, hid, 2 * hid, 4 * hid, VAD_ACT_NONE, 0, VAD_PREC_FP32, nullptr, nullptr, st);
        };
        // Large launch groups fill the chip with one step: layers outer, time inner on the caller's stream (the reference's
        // order).  Small ones (the reference's batch sizes; a step is then one wave's serial K loop on a fraction of the
        // CUs) run the layers as a wavefront on helper streams.
        const long long lstm_groups = vid_lstm_groups(nc, h16, w16, hid);
        const int wf = g_vad_lstm_wavefront.load(std::memory_order_relaxed);
        if (hoist) TRY(lstm_xhalf(0, -1, s));
        if (layers > 1 && ((lstm_groups < 256 && wf) || wf == 2)) {   // (Winograd steps share ONE z buffer: layers strictly in order)
            // per-step x halves of the layers above 0 (a helper-stream launch and two event hops per step) pay while a step is a
            // long serial K loop; with the gate-split kernel (one window: ~11 us per step) they cost more than they save - measured
            // 0.78 -> 0.70 ms for one 16-frame window, 1.05 -> 0.97 for two clips - so those steps run their whole K loop
            hoist_upper = hoist && !vad_convlstm_gate_wins(nc, h16, w16, hid);
            VadSideStreams* S = nullptr;
            TRY(side_streams(layers, &S));
            VAD_HIP_TRY(hipEventRecord(S->fork, s));                          // the encoder's output (and layer 0's x halves) are ready
            for (int l = 1; l < layers; ++l) {
                VAD_HIP_TRY(hipStreamWaitEvent(S->st[l - 1], S->fork, 0));
                if (hoist_upper) VAD_HIP_TRY(hipStreamWaitEvent(S->xs[l - 1], S->fork, 0));
            }
            for (int ti = 0; ti < t; ++ti)
                for (int l = 0; l < layers; ++l) {
                    hipStream_t st = l ? S->st[l - 1] : s;
                    if (l && hoist_upper) {                                            // (l-1, ti) finished -> x half of (l, ti) -> step (l, ti)
                        hipStream_t xs = S->xs[l - 1];
                        VAD_HIP_TRY(hipStreamWaitEvent(xs, S->done[l - 1], 0));
                        TRY(lstm_xhalf(l, ti, xs));
                        VAD_HIP_TRY(hipEventRecord(S->xdone[l], xs));
                        VAD_HIP_TRY(hipStreamWaitEvent(st, S->xdone[l], 0));
                    } else if (l) {
                        VAD_HIP_TRY(hipStreamWaitEvent(st, S->done[l - 1], 0));     // (l-1, ti) finished
                    }
                    TRY(lstm_step(l, ti, st));
                    if (l + 1 < layers || ti + 1 == t) VAD_HIP_TRY(hipEventRecord(S->done[l], st));
                }
            for (int l = 1; l < layers; ++l) VAD_HIP_TRY(hipStreamWaitEvent(s, S->done[l], 0));   // join (the x streams end before their steps)
        } else {
            for (int l = 0; l < layers; ++l) {
                if (l && hoist) {                        // one batched launch per layer: its whole input sequence exists
                    VadProfScope ps(4, s);
                    TRY(vad_conv3x3_kpart(HS[l - 1], 0, W_(4 + l), B_(4 + l), ZX[l], 0, nc * t, h16, w16, hid, 2 * hid, 4 * hid, VAD_ACT_NONE, 0,
                                          VAD_PREC_FP32, nullptr, nullptr, s));
                }
                for (int ti = 0; ti < t; ++ti) TRY(lstm_step(l, ti, s));
            }
        }
        const float* dec_in = HS[layers - 1];
        int li = 4 + layers;
        if (L.has_proj) {   // models/video_autoencoder.py:346-349
            VadProfScope ps(5, s);
            TRY(vad_conv1x1(dec_in, W_(li), B_(li), P, (long long)n * h16 * w16, hid, latent, s));
            dec_in = P;
            ++li;
        }
        // VideoDecoder: 3 x convT-BN-ReLU + convT-Tanh (models/video_autoencoder.py:242-261)
        { VadProfScope ps(6, s); TRY(vad_convt2x2(dec_in, 0, W_(li), B_(li), A, 0, n, h16, w16, latent, 128, VAD_ACT_RELU, precision, s)); }
        { VadProfScope ps(7, s); TRY(vad_convt2x2(A, 0, W_(li + 1), B_(li + 1), Bf, 0, n, h / 8, w / 8, 128, 64, VAD_ACT_RELU, precision, s)); }
        { VadProfScope ps(8, s); TRY(vad_convt2x2(Bf, 0, W_(li + 2), B_(li + 2), A, 0, n, h / 4, w / 4, 64, 32, VAD_ACT_RELU, precision, s)); }
        if (wide) {                         // ConvTranspose2d(32 -> in_ch) into padded planes, Tanh + score (wide_io.hip)
            VadProfScope ps(9, s);
            TRY(vad_convt2x2(A, 0, W_(li + 3), B_(li + 3), Bf, 0, n, h / 2, w / 2, 32, wide, VAD_ACT_NONE, VAD_PREC_FP32, s));
            TRY(vad_tanh_score_nhwc(Bf, wide, (const float*)xin, in_ch, parts, recon ? recon + (size_t)c0 * t * in_ch * h * w : nullptr,
                                    errmap ? errmap + (size_t)c0 * t * h * w : nullptr, n, h, w, t, cs, s));
        } else { VadProfScope ps(9, s);
          TRY(vad_convt2x2_to3_score_fmt(A, W_(li + 3), B_(li + 3), xin, x_format, parts,
                                     recon ? recon + (size_t)c0 * t * 3 * h * w : nullptr,
                                     errmap ? errmap + (size_t)c0 * t * h * w : nullptr, n, h / 2, w / 2, 32,
                                     t, cs, s)); }
        if (seq_scores || frame_scores) {
            VadProfScope ps(10, s);
            TRY(vad_score_finalize_tagged(parts, nparts, n, h, w, in_ch, frame_scores ? frame_scores + (size_t)c0 * t : nullptr,
                                          seq_scores ? seq_scores + c0 : nullptr, t, (const unsigned*)packed,
                                          vad_blob_tag(VAD_BLOB_VID, mprec), s));
        }
    }
#undef W_
#undef B_
    return VAD_OK;
}
}  // namespace

extern "C" size_t vad_vid_workspace_bytes(int chunk, int t, int h, int w, int latent, int hid, int layers) {
    return vad_vid_workspace_bytes_c(chunk, t, h, w, latent, hid, layers, 3);
}
extern "C" size_t vad_vid_workspace_bytes_c(int chunk, int t, int h, int w, int latent, int hid, int layers, int in_ch) {
    if (chunk <= 0 || t <= 0 || h <= 0 || w <= 0 || h % 16 || w % 16) return 0;
    if (vad_vid_packed_floats_c(in_ch, latent, hid, layers) == 0) return 0;
    return vid_ws(chunk, t, t, h, w, latent, hid, layers, in_ch).total;
}

extern "C" int vad_vid_score(const float* x, long long b, int t, int h, int w, int latent, int hid, int layers,
                             const float* packed, void* ws, size_t ws_bytes, int chunk,
                             float* seq_scores, float* frame_scores, float* errmap, float* recon, void* stream) {
    return vad_vid_score_x(x, VAD_X_F32_NCHW, VAD_PREC_FP32, b, t, h, w, latent, hid, layers, packed, ws, ws_bytes, chunk, seq_scores,
                           frame_scores, errmap, recon, stream);
}

extern "C" int vad_vid_score_x(const void* x, int x_format, int precision, long long b, int t, int h, int w, int latent, int hid, int layers,
                               const float* packed, void* ws, size_t ws_bytes, int chunk,
                               float* seq_scores, float* frame_scores, float* errmap, float* recon, void* stream) {
    return vad_vid_score_c(x, x_format, precision, 3, b, t, h, w, latent, hid, layers, packed, ws, ws_bytes, chunk, seq_scores, frame_scores, errmap, recon, stream);
}

extern "C" int vad_vid_score_c(const void* x, int x_format, int precision, int in_ch, long long b, int t, int h, int w, int latent, int hid, int layers,
                               const float* packed, void* ws, size_t ws_bytes, int chunk,
                               float* seq_scores, float* frame_scores, float* errmap, float* recon, void* stream) {
    VAD_REQUIRE(x && packed && ws, "vid_score: null pointer");
    VAD_REQUIRE(b > 0 && t > 0 && chunk > 0, "vid_score: clips=%lld T=%d chunk=%d must be positive", b, t, chunk);
    VAD_REQUIRE(h > 0 && w > 0 && h % 16 == 0 && w % 16 == 0,
                "vid_score: H=%d W=%d must be positive multiples of 16 (4 MaxPool2d(2) stages)", h, w);
    if (vad_vid_packed_floats(latent, hid, layers) == 0) return VAD_ERR_ARG;   // message already set
    VAD_REQUIRE(seq_scores || frame_scores || errmap || recon, "vid_score: no output requested");
    return vid_run(x, x_format, precision, in_ch, b, t, t, h, w, latent, hid, layers, packed, ws, ws_bytes, chunk, seq_scores, frame_scores, errmap,
                   recon, (hipStream_t)stream, "vid_score");
}

extern "C" long long vad_vid_num_windows(long long frames, int t, int stride) {
    if (frames < t || t <= 0 || stride <= 0) return 0;
    return (frames - t) / stride + 1;
}

extern "C" size_t vad_vid_windows_workspace_bytes(int chunk, int t, int stride, int h, int w, int latent, int hid, int layers) {
    return vad_vid_windows_workspace_bytes_c(chunk, t, stride, h, w, latent, hid, layers, 3);
}
extern "C" size_t vad_vid_windows_workspace_bytes_c(int chunk, int t, int stride, int h, int w, int latent, int hid, int layers, int in_ch) {
    if (chunk <= 0 || t <= 0 || stride <= 0 || stride > t || h <= 0 || w <= 0 || h % 16 || w % 16) return 0;
    if (vad_vid_packed_floats_c(in_ch, latent, hid, layers) == 0) return 0;
    return vid_ws(chunk, t, stride, h, w, latent, hid, layers, in_ch).total;
}

extern "C" int vad_vid_score_windows(const float* frames, long long nframes, int t, int stride, int h, int w,
                                     int latent, int hid, int layers, const float* packed, void* ws, size_t ws_bytes,
                                     int chunk, float* seq_scores, float* frame_scores, float* errmap, float* recon,
                                     void* stream) {
    return vad_vid_score_windows_x(frames, VAD_X_F32_NCHW, VAD_PREC_FP32, nframes, t, stride, h, w, latent, hid, layers, packed, ws, ws_bytes,
                                   chunk, seq_scores, frame_scores, errmap, recon, stream);
}

extern "C" int vad_vid_score_windows_x(const void* frames, int x_format, int precision, long long nframes, int t, int stride, int h, int w,
                                       int latent, int hid, int layers, const float* packed, void* ws, size_t ws_bytes,
                                       int chunk, float* seq_scores, float* frame_scores, float* errmap, float* recon,
                                       void* stream) {
    return vad_vid_score_windows_c(frames, x_format, precision, 3, nframes, t, stride, h, w, latent, hid, layers, packed, ws, ws_bytes, chunk,
                                   seq_scores, frame_scores, errmap, recon, stream);
}

extern "C" int vad_vid_score_windows_c(const void* frames, int x_format, int precision, int in_ch, long long nframes, int t, int stride, int h, int w,
                                       int latent, int hid, int layers, const float* packed, void* ws, size_t ws_bytes,
                                       int chunk, float* seq_scores, float* frame_scores, float* errmap, float* recon,
                                       void* stream) {
    VAD_REQUIRE(frames && packed && ws, "vid_score_windows: null pointer");
    VAD_REQUIRE(t > 0 && stride > 0 && stride <= t && chunk > 0, "vid_score_windows: need 0 < stride <= T (got T=%d stride=%d) and chunk > 0", t, stride);
    VAD_REQUIRE(nframes >= t, "vid_score_windows: %lld frames are fewer than one window of %d", nframes, t);
    VAD_REQUIRE(h > 0 && w > 0 && h % 16 == 0 && w % 16 == 0, "vid_score_windows: H=%d W=%d must be positive multiples of 16", h, w);
    if (vad_vid_packed_floats(latent, hid, layers) == 0) return VAD_ERR_ARG;
    VAD_REQUIRE(seq_scores || frame_scores || errmap || recon, "vid_score_windows: no output requested");
    return vid_run(frames, x_format, precision, in_ch, vad_vid_num_windows(nframes, t, stride), t, stride, h, w, latent, hid, layers, packed, ws,
                   ws_bytes, chunk, seq_scores, frame_scores, errmap, recon, (hipStream_t)stream, "vid_score_windows");
}
