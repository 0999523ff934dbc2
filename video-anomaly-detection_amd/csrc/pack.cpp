// Host-side weight packing: eval-mode BatchNorm folding (fp64, one rounding to fp32) and the
// re-ordering the kernels' B-operand loads expect.  Pure CPU code, no HIP calls.
//
// BatchNorm2d eval semantics restated (torch defaults, eps = 1e-5; used after every conv/convT of
// reference models/autoencoder.py:38-131 and models/video_autoencoder.py:191-256):
//   y = (conv(x) - running_mean) / sqrt(running_var + eps) * gamma + beta
//     = conv_{w * s}(x) + ((b - running_mean) * s + beta),   s = gamma / sqrt(running_var + eps)
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>
#include "vad_layout.h"

static thread_local char g_err[512] = "";

int vad_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* vad_last_error(void) { return g_err; }
extern "C" int vad_abi_version(void) { return VAD_ABI_VERSION; }

#define REQ(cond, ...) do { if (!(cond)) return vad_fail(VAD_ERR_ARG, __VA_ARGS__); } while (0)

static void bn_scale_shift(const float* bias, const float* const* bn, int cout,
                           std::vector<double>& s, float* bias_out) {
    s.assign(cout, 1.0);
    for (int co = 0; co < cout; ++co) {
        double b = bias ? (double)bias[co] : 0.0;
        if (bn) {
            const double sc = (double)bn[0][co] / std::sqrt((double)bn[3][co] + 1e-5);
            s[co] = sc;
            b = (b - (double)bn[2][co]) * sc + (double)bn[1][co];
        }
        bias_out[co] = (float)b;
    }
}

// The arithmetic mode is an ARGUMENT of every packer and launcher (ABI 2): there is no process-wide switch.
#define REQ_PREC(who) REQ(precision == VAD_PREC_FP32 || precision == VAD_PREC_SPLIT || precision == VAD_PREC_WINO, who ": precision=%d must be VAD_PREC_FP32 (0), VAD_PREC_SPLIT (1) or VAD_PREC_WINO (4)", precision)
// arithmetic of everything that is not a Winograd 3x3 convolution in VAD_PREC_WINO blobs
static int base_prec(int precision) { return precision == VAD_PREC_WINO ? VAD_PREC_FP32 : precision; }

extern "C" size_t vad_pack_conv3x3_floats(int cout, int cin) { return (size_t)9 * ((cin + 7) / 8) * cout * 8; }

// Split-fp16 operand form of the same weights (same byte count): [tap][cin/16][cout][half h][8 x hi | 8 x lo] fp16,
// element j of half h = input channel 16*c16 + 8*h + j; hi = fp16(w), lo = fp16((w - hi) * 2^11).
static void pack_conv3x3_split(const float* w, const std::vector<double>& s, int cout, int cin, float* out) {
    _Float16* o = (_Float16*)out;
    const int c16n = cin / 16;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < 9; ++tap) {
                const float v = (float)((double)w[((size_t)co * cin + ci) * 9 + tap] * s[co]);
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
                const size_t base = ((((size_t)tap * c16n + ci / 16) * cout + co) * 2 + ((ci >> 3) & 1)) * 16;
                o[base + (ci & 7)] = hi;
                o[base + 8 + (ci & 7)] = lo;
            }
}

extern "C" int vad_pack_conv3x3(const float* w, const float* bias, const float* const* bn,
                                int cout, int cin, int precision, float* out, float* bias_out) {
    REQ(w && out && bias_out && cout > 0 && cin > 0, "pack_conv3x3: bad arguments");
    REQ_PREC("pack_conv3x3");
    REQ(precision != VAD_PREC_WINO, "pack_conv3x3: the Winograd form of a layer is packed by vad_pack_conv3x3_wino");
    REQ(precision != VAD_PREC_SPLIT || cin % 16 == 0, "pack_conv3x3: split precision needs cin %% 16 == 0 (got %d)", cin);
    std::vector<double> s;
    bn_scale_shift(bias, bn, cout, s, bias_out);
    const int c8n = (cin + 7) / 8;
    memset(out, 0, vad_pack_conv3x3_floats(cout, cin) * sizeof(float));
    if (precision == VAD_PREC_SPLIT) {
        pack_conv3x3_split(w, s, cout, cin, out);
        return VAD_OK;
    }
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < 9; ++tap)
                out[(((size_t)tap * c8n + ci / 8) * cout + co) * 8 + (ci & 7)] =
                    (float)((double)w[((size_t)co * cin + ci) * 9 + tap] * s[co]);
    return VAD_OK;
}

// Winograd F(2x2,3x3) form (csrc/conv_wino.hip; opt-in arithmetic): U = G g G^T per (cout, cin) with the BatchNorm scale folded,
// computed in double and rounded once -> [16 frequencies f = 4 fr + fc][cin/8][cout][8], the layout of the direct form with 16
// "taps".  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].
extern "C" size_t vad_pack_conv3x3_wino_floats(int cout, int cin) { return (size_t)16 * ((cin + 7) / 8) * cout * 8; }

extern "C" int vad_pack_conv3x3_wino(const float* w, const float* bias, const float* const* bn, int cout, int cin, float* out, float* bias_out) {
    REQ(w && out && bias_out && cout > 0 && cin > 0, "pack_conv3x3_wino: bad arguments");
    std::vector<double> s;
    bn_scale_shift(bias, bn, cout, s, bias_out);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int c8n = (cin + 7) / 8;
    memset(out, 0, vad_pack_conv3x3_wino_floats(cout, cin) * sizeof(float));
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float* g = w + ((size_t)co * cin + ci) * 9;
            for (int fr = 0; fr < 4; ++fr)
                for (int fc = 0; fc < 4; ++fc) {
                    double u = 0.0;
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) u += G[fr][a] * (double)g[a * 3 + b] * G[fc][b];
                    out[(((size_t)(fr * 4 + fc) * c8n + ci / 8) * cout + co) * 8 + (ci & 7)] = (float)(u * s[co]);
                }
        }
    return VAD_OK;
}

// [28][cout] fp32 (K = 27 padded to 28) followed by the split-fp16 form for the fused first stage in split mode:
// [2 k-steps][cout][half h][8 x hi | 8 x lo] fp16 (K padded to 32), 32*cout halves*2 = 32*cout floats.
extern "C" size_t vad_pack_conv3x3_c3_floats(int cout) { return (size_t)(28 + 32) * cout; }

extern "C" int vad_pack_conv3x3_c3(const float* w, const float* bias, const float* const* bn,
                                   int cout, float* out, float* bias_out) {
    REQ(w && out && bias_out && cout > 0, "pack_conv3x3_c3: bad arguments");
    std::vector<double> s;
    bn_scale_shift(bias, bn, cout, s, bias_out);
    for (int k = 0; k < 28; ++k)
        for (int co = 0; co < cout; ++co)
            out[(size_t)k * cout + co] = k < 27 ? (float)((double)w[(size_t)co * 27 + k] * s[co]) : 0.f;
    _Float16* o = (_Float16*)(out + (size_t)28 * cout);
    for (int k = 0; k < 32; ++k)
        for (int co = 0; co < cout; ++co) {
            const float v = k < 27 ? (float)((double)w[(size_t)co * 27 + k] * s[co]) : 0.f;
            const _Float16 hi = (_Float16)v, lo = (_Float16)((v - (float)hi) * 2048.0f);
            const size_t base = ((((size_t)(k / 16)) * cout + co) * 2 + ((k >> 3) & 1)) * 16;
            o[base + (k & 7)] = hi;
            o[base + 8 + (k & 7)] = lo;
        }
    return VAD_OK;
}

extern "C" size_t vad_pack_convt2x2_floats(int cin, int cout) { return (size_t)4 * (cin / 8) * cout * 8; }

extern "C" int vad_pack_convt2x2(const float* w, const float* bias, const float* const* bn,
                                 int cin, int cout, int precision, float* out, float* bias_out) {
    REQ(w && out && bias_out && cout > 0 && cin > 0 && cin % 8 == 0, "pack_convt2x2: bad arguments");
    REQ_PREC("pack_convt2x2");
    REQ(precision != VAD_PREC_WINO, "pack_convt2x2: transposed convolutions have no Winograd form (pack them as VAD_PREC_FP32)");
    REQ(precision != VAD_PREC_SPLIT || cin % 16 == 0, "pack_convt2x2: split precision needs cin %% 16 == 0 (got %d)", cin);
    std::vector<double> s;
    bn_scale_shift(bias, bn, cout, s, bias_out);
    if (precision == VAD_PREC_SPLIT) {   // split-fp16 operands: [q][cin/16][cout][half h][8 x hi | 8 x lo]
        _Float16* o = (_Float16*)out;
        for (int ci = 0; ci < cin; ++ci)
            for (int co = 0; co < cout; ++co)
                for (int q = 0; q < 4; ++q) {
                    const float v = (float)((double)w[((size_t)ci * cout + co) * 4 + q] * s[co]);
                    const _Float16 hi = (_Float16)v, lo = (_Float16)((v - (float)hi) * 2048.0f);
                    const size_t base = ((((size_t)q * (cin / 16) + ci / 16) * cout + co) * 2 + ((ci >> 3) & 1)) * 16;
                    o[base + (ci & 7)] = hi;
                    o[base + 8 + (ci & 7)] = lo;
                }
        return VAD_OK;
    }
    for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < cout; ++co)
            for (int q = 0; q < 4; ++q)
                out[(((size_t)q * (cin / 8) + ci / 8) * cout + co) * 8 + (ci & 7)] =
                    (float)((double)w[((size_t)ci * cout + co) * 4 + q] * s[co]);
    return VAD_OK;
}

extern "C" size_t vad_pack_conv1x1_floats(int cout, int cin) { return (size_t)(cin / 8) * cout * 8; }

extern "C" int vad_pack_conv1x1(const float* w, const float* bias, int cout, int cin, float* out, float* bias_out) {
    REQ(w && out && bias_out && cout > 0 && cin > 0 && cin % 8 == 0, "pack_conv1x1: bad arguments");
    for (int co = 0; co < cout; ++co) {
        bias_out[co] = bias ? bias[co] : 0.f;
        for (int ci = 0; ci < cin; ++ci)
            out[((size_t)(ci / 8) * cout + co) * 8 + (ci & 7)] = w[(size_t)co * cin + ci];
    }
    return VAD_OK;
}

// two forms of the same weights: the per-lane rows of the VALU tail kernel, then (cin == 32 only) the A operand of the fused
// dec4 kernel's second GEMM (csrc/dec4_fused.hip): [m 32][lane half 2][r 16], m = tap * 3 + co (rows 27..31 zero),
// element = W[co][ci = (r & 3) + 8 (r >> 2) + 4 half][tap]
extern "C" size_t vad_pack_conv3x3_to3_floats(int cin) { return (size_t)(cin / 4) * 108 + (cin == 32 ? 1024 : 0); }

// [cin/4][9 taps][4 channels][3 outputs]: lane cg of the tail kernel reads its 108 weights contiguously
extern "C" int vad_pack_conv3x3_to3(const float* w, int cin, float* out) {
    REQ(w && out && cin > 0 && cin % 4 == 0, "pack_conv3x3_to3: bad arguments");
    for (int ci = 0; ci < cin; ++ci)
        for (int tap = 0; tap < 9; ++tap)
            for (int co = 0; co < 3; ++co)
                out[(size_t)(ci / 4) * 108 + tap * 12 + (ci & 3) * 3 + co] = w[((size_t)co * cin + ci) * 9 + tap];
    if (cin == 32) {
        float* g = out + (size_t)(cin / 4) * 108;
        for (int m = 0; m < 32; ++m)
            for (int half = 0; half < 2; ++half)
                for (int r = 0; r < 16; ++r) {
                    const int ci = (r & 3) + 8 * (r >> 2) + 4 * half, tap = m / 3, co = m % 3;
                    g[(m * 2 + half) * 16 + r] = m < 27 ? w[((size_t)co * cin + ci) * 9 + tap] : 0.f;
                }
    }
    return VAD_OK;
}

// --------------------------------------------------------------------------- zero-padded channel dimensions
// (vad_layout.h: why padding is exact.)  A torch-layout copy of one layer's parameters with its channel dimensions
// zero-padded; `co_at(co)` / `ci_at(ci)` give the padded position of a real channel (the ConvLSTM cell pads each of its
// four gate blocks and its x / h input halves separately).  Padded BatchNorm channels get gamma = beta = mean = 0,
// var = 1: scale 0, shift 0.
namespace {
struct PaddedLayer {
    std::vector<float> w, bias, bnv[4];
    const float* bn[4] = {nullptr, nullptr, nullptr, nullptr};
};
template <class CoAt, class CiAt>
void pad_layer(const float* w, const float* bias, const float* const* bn, int cout, int cin, int kk, bool transposed,
               int cout_p, int cin_p, CoAt co_at, CiAt ci_at, PaddedLayer& o) {
    o.w.assign((size_t)cout_p * cin_p * kk, 0.f);
    o.bias.assign(cout_p, 0.f);
    for (int co = 0; co < cout; ++co) {
        const int cop = co_at(co);
        if (bias) o.bias[cop] = bias[co];
        for (int ci = 0; ci < cin; ++ci) {
            const int cip = ci_at(ci);
            const float* src = transposed ? w + ((size_t)ci * cout + co) * kk : w + ((size_t)co * cin + ci) * kk;
            float* dst = transposed ? o.w.data() + ((size_t)cip * cout_p + cop) * kk : o.w.data() + ((size_t)cop * cin_p + cip) * kk;
            for (int k = 0; k < kk; ++k) dst[k] = src[k];
        }
    }
    if (bn) {
        for (int i = 0; i < 4; ++i) {
            o.bnv[i].assign(cout_p, i == 3 ? 1.f : 0.f);
            for (int co = 0; co < cout; ++co) o.bnv[i][co_at(co)] = bn[i][co];
            o.bn[i] = o.bnv[i].data();
        }
    }
}
const auto same = [](int c) { return c; };
}  // namespace

// conv3x3 / convT / conv1x1 whose real widths (cout, cin) sit in slots of (s.cout, s.cin) channels
static int pack_conv3x3_any(const float* w, const float* b, const float* const* bn, int cout, int cin, int precision, float* wo, float* bo) {
    if (precision == VAD_PREC_WINO) return vad_pack_conv3x3_wino(w, b, bn, cout, cin, wo, bo);     // U = G g G^T, 16 "taps"
    return vad_pack_conv3x3(w, b, bn, cout, cin, precision, wo, bo);
}
static int pack_conv3x3_slot(const float* w, const float* b, const float* const* bn, int cout, int cin, const LayerSlot& s,
                             int precision, float* out) {
    if (cout == s.cout && cin == s.cin) return pack_conv3x3_any(w, b, bn, cout, cin, precision, out + s.w, out + s.b);
    PaddedLayer P;
    pad_layer(w, b, bn, cout, cin, 9, false, s.cout, s.cin, same, same, P);
    return pack_conv3x3_any(P.w.data(), P.bias.data(), bn ? P.bn : nullptr, s.cout, s.cin, precision, out + s.w, out + s.b);
}
static int pack_convt2x2_slot(const float* w, const float* b, const float* const* bn, int cin, int cout, const LayerSlot& s,
                              int precision, float* out) {
    precision = base_prec(precision);
    if (cout == s.cout && cin == s.cin) return vad_pack_convt2x2(w, b, bn, cin, cout, precision, out + s.w, out + s.b);
    PaddedLayer P;
    pad_layer(w, b, bn, cout, cin, 4, true, s.cout, s.cin, same, same, P);
    return vad_pack_convt2x2(P.w.data(), P.bias.data(), bn ? P.bn : nullptr, s.cin, s.cout, precision, out + s.w, out + s.b);
}

// --------------------------------------------------------------------------- image autoencoder
static size_t align4(size_t x) { return (x + 3) & ~(size_t)3; }

// Every model blob starts with a 4-word header {magic, tag, dim0, dim1} (uint32 bit patterns in the float array): the tag
// names the model kind and the arithmetic mode the operands were packed for.  The score entry points take the mode as an
// argument; the kernel that finalises the scores compares it with the tag ON THE DEVICE and writes NaN scores on a
// mismatch, so a blob launched under the wrong mode cannot pass as a result.
static void put_header(float* out, int kind, int precision, int d0, int d1) {
    const unsigned h[4] = {VAD_BLOB_MAGIC, vad_blob_tag(kind, precision), (unsigned)d0, (unsigned)d1};
    memcpy(out, h, sizeof h);
}

extern "C" int vad_blob_precision(const float* packed_host) {
    if (!packed_host) return vad_fail(VAD_ERR_ARG, "blob_precision: null pointer");
    unsigned h[2];
    memcpy(h, packed_host, sizeof h);
    if (h[0] != VAD_BLOB_MAGIC || (h[1] >> 16) != VAD_ABI_VERSION) return vad_fail(VAD_ERR_ARG, "blob_precision: not a packed model blob of ABI %d", VAD_ABI_VERSION);
    return (int)((h[1] >> 8) & 0xff);
}

// A 3x3 layer's slot holds either form of its weights: direct [9][cin/8][cout][8] or Winograd [16][cin/8][cout][8]
// (VAD_PREC_WINO blobs); one layout for every arithmetic mode keeps the size queries of the C ABI mode-free.
static size_t conv3x3_slot_floats(int cout, int cin) { return vad_pack_conv3x3_wino_floats(cout, cin); }

ImgLayout img_layout(int in_ch, int latent_real) {
    ImgLayout L{};
    const int latent = L.latent_p = vad_img_latent_p(latent_real);
    const int wide = L.wide = in_ch > 3 ? vad_wide_p(in_ch) : 0;
    const int ch[5] = {3, 32, 64, 128, latent};
    size_t off = VAD_BLOB_HEADER_FLOATS;
    int li = 0;
    auto add = [&](int kind, int cin, int cout, size_t wfloats) {
        LayerSlot& s = L.layer[li++];
        s.kind = kind; s.cin = cin; s.cout = cout;
        s.w = off; off = align4(off + wfloats);
        s.b = off; off = align4(off + (size_t)cout);
    };
    for (int b = 0; b < 4; ++b) {   // encoder blocks (models/autoencoder.py:38-79)
        if (b == 0 && wide) add(LK_CONV, wide, 32, conv3x3_slot_floats(32, wide));
        else if (b == 0) add(LK_CONV_C3, 3, 32, vad_pack_conv3x3_c3_floats(32));
        else add(LK_CONV, ch[b], ch[b + 1], conv3x3_slot_floats(ch[b + 1], ch[b]));
        add(LK_CONV, ch[b + 1], ch[b + 1], conv3x3_slot_floats(ch[b + 1], ch[b + 1]));
    }
    const int dch[5] = {latent, 128, 64, 32, 32};
    for (int b = 0; b < 4; ++b) {   // decoder blocks (models/autoencoder.py:103-139)
        add(LK_CONVT, dch[b], dch[b + 1], vad_pack_convt2x2_floats(dch[b], dch[b + 1]));
        if (b < 3) add(LK_CONV, dch[b + 1], dch[b + 1], conv3x3_slot_floats(dch[b + 1], dch[b + 1]));
        else if (wide) add(LK_CONV, 32, wide, conv3x3_slot_floats(wide, 32));     // Conv2d(32 -> in_ch) un-activated; Tanh + score: wide_io.hip
        else add(LK_TAIL_CONV, 32, 3, vad_pack_conv3x3_to3_floats(32));
    }
    L.nlayers = li;
    L.total = off;
    return L;
}

extern "C" size_t vad_img_packed_floats(int in_ch, int latent) {
    if (in_ch < 3 || in_ch > VAD_MAX_IN_CH || latent <= 0 || latent > VAD_MAX_WIDTH) return 0;
    return img_layout(in_ch, latent).total;
}

extern "C" int vad_img_pack(const float* const* P, int nparams, int in_ch, int latent, int precision, float* out) {
    REQ(P && out, "img_pack: null pointer");
    REQ_PREC("img_pack");
    REQ(in_ch >= 3 && in_ch <= VAD_MAX_IN_CH, "img_pack: in_channels=%d out of range [3,%d] (1- and 2-channel models: widen to 3 planes with zero weights)", in_ch, VAD_MAX_IN_CH);
    REQ(latent > 0 && latent <= VAD_MAX_WIDTH, "img_pack: latent_dim=%d out of range [1,%d]", latent, VAD_MAX_WIDTH);
    REQ(nparams == VAD_IMG_NPARAMS, "img_pack: expected %d parameter tensors, got %d", VAD_IMG_NPARAMS, nparams);
    for (int i = 0; i < nparams; ++i) REQ(P[i], "img_pack: parameter %d is NULL", i);
    const ImgLayout L = img_layout(in_ch, latent);
    memset(out, 0, L.total * sizeof(float));
    put_header(out, VAD_BLOB_IMG, precision, latent, in_ch);
    int pi = 0, rc = VAD_OK;
    for (int li = 0; li < L.nlayers && rc == VAD_OK; ++li) {
        const LayerSlot& s = L.layer[li];
        const float* w = P[pi], *b = P[pi + 1];
        if (s.kind == LK_TAIL_CONV) {
            rc = vad_pack_conv3x3_to3(w, s.cin, out + s.w);
            for (int c = 0; c < 3; ++c) out[s.b + c] = b[c];
            pi += 2;
            continue;
        }
        if (L.wide && li == 15) {                  // last layer of a wide model: Conv2d(32 -> in_ch) + bias, no BatchNorm
            rc = pack_conv3x3_slot(w, b, nullptr, in_ch, 32, s, precision, out);
            pi += 2;
            continue;
        }
        const float* bn[4] = {P[pi + 2], P[pi + 3], P[pi + 4], P[pi + 5]};
        if (L.wide && li == 0) {                   // first layer of a wide model: a generic 3x3 layer over the zero-padded planes
            rc = pack_conv3x3_slot(w, b, bn, 32, in_ch, s, precision, out);
            pi += 6;
            continue;
        }
        // real widths of this layer: only enc4.0 (cout), enc4.3 (both) and dec1.0 (cin) carry latent_dim
        const int cin = s.cin == L.latent_p && (li == 7 || li == 8) ? latent : s.cin;
        const int cout = s.cout == L.latent_p && (li == 6 || li == 7) ? latent : s.cout;
        if (s.kind == LK_CONV_C3) rc = vad_pack_conv3x3_c3(w, b, bn, s.cout, out + s.w, out + s.b);
        else if (s.kind == LK_CONV) rc = pack_conv3x3_slot(w, b, bn, cout, cin, s, precision, out);
        // dec4.0 (li 14) is always packed for - and run on - the exact-fp32 path: it shares ONE kernel with dec4.3 and the
        // score (dec4_fused.hip), HBM-bound in either arithmetic, and that kernel's first GEMM is exact fp32
        else rc = pack_convt2x2_slot(w, b, bn, cin, cout, s, li == 14 ? VAD_PREC_FP32 : precision, out);
        pi += 6;
    }
    if (rc == VAD_OK && pi != nparams) return vad_fail(VAD_ERR_ARG, "img_pack: consumed %d of %d parameters", pi, nparams);
    return rc;
}

// --------------------------------------------------------------------------- video autoencoder
VidLayout vid_layout(int in_ch, int latent_real, int hid_real, int layers) {
    VidLayout L{};
    const int wide = L.wide = in_ch > 3 ? vad_wide_p(in_ch) : 0;
    const int latent = L.latent_p = vad_vid_latent_p(latent_real, hid_real);
    const int hid = L.hid_p = vad_vid_hid_p(latent_real, hid_real);
    size_t off = VAD_BLOB_HEADER_FLOATS;
    int li = 0;
    auto add = [&](int kind, int cin, int cout, size_t wfloats) {
        LayerSlot& s = L.layer[li++];
        s.kind = kind; s.cin = cin; s.cout = cout;
        s.w = off; off = align4(off + wfloats);
        s.b = off; off = align4(off + (size_t)cout);
    };
    const int ch[5] = {3, 32, 64, 128, latent};
    for (int b = 0; b < 4; ++b) {   // VideoEncoder (models/video_autoencoder.py:191-215)
        if (b == 0 && wide) add(LK_CONV, wide, 32, conv3x3_slot_floats(32, wide));
        else if (b == 0) add(LK_CONV_C3, 3, 32, vad_pack_conv3x3_c3_floats(32));
        else add(LK_CONV, ch[b], ch[b + 1], conv3x3_slot_floats(ch[b + 1], ch[b]));
    }
    for (int l = 0; l < layers; ++l) {   // ConvLSTM cells (models/video_autoencoder.py:118-125)
        const int cin = (l == 0 ? latent : hid) + hid;
        add(LK_LSTM, cin, 4 * hid, conv3x3_slot_floats(4 * hid, cin));
    }
    L.has_proj = hid_real != latent_real;          // models/video_autoencoder.py:311-312
    if (L.has_proj) add(LK_PROJ, hid, latent, vad_pack_conv1x1_floats(latent, hid));
    const int dch[4] = {latent, 128, 64, 32};
    for (int b = 0; b < 3; ++b) add(LK_CONVT, dch[b], dch[b + 1], vad_pack_convt2x2_floats(dch[b], dch[b + 1]));
    if (wide) add(LK_CONVT, 32, wide, vad_pack_convt2x2_floats(32, wide));     // ConvTranspose2d(32 -> in_ch) un-activated; Tanh + score: wide_io.hip
    else add(LK_TAIL_CONVT, 32, 3, (size_t)32 * 12);   // VideoDecoder (models/video_autoencoder.py:242-261)
    L.nlayers = li;
    L.total = off;
    return L;
}

extern "C" int vad_vid_nparams(int layers, int has_proj) { return 24 + 2 * layers + (has_proj ? 2 : 0) + 18 + 2; }

static int vid_dims_ok(int latent, int hid, int layers) {
    REQ(latent > 0 && latent <= VAD_MAX_WIDTH, "vid: latent_dim=%d out of range [1,%d]", latent, VAD_MAX_WIDTH);
    REQ(hid > 0 && hid <= VAD_MAX_WIDTH, "vid: lstm_hidden_dim=%d out of range [1,%d]", hid, VAD_MAX_WIDTH);
    REQ(layers >= 1 && layers <= 8, "vid: lstm_num_layers=%d out of range [1,8]", layers);
    return VAD_OK;
}

extern "C" size_t vad_vid_packed_floats(int latent, int hid, int layers) { return vad_vid_packed_floats_c(3, latent, hid, layers); }
extern "C" size_t vad_vid_packed_floats_c(int in_ch, int latent, int hid, int layers) {
    if (in_ch < 3 || in_ch > VAD_MAX_IN_CH || vid_dims_ok(latent, hid, layers) != VAD_OK) return 0;
    return vid_layout(in_ch, latent, hid, layers).total;
}

extern "C" int vad_vid_pack(const float* const* P, int nparams, int latent, int hid, int layers, int precision, float* out) {
    return vad_vid_pack_c(P, nparams, 3, latent, hid, layers, precision, out);
}

extern "C" int vad_vid_pack_c(const float* const* P, int nparams, int in_ch, int latent, int hid, int layers, int precision, float* out) {
    REQ(P && out, "vid_pack: null pointer");
    REQ_PREC("vid_pack");
    REQ(in_ch >= 3 && in_ch <= VAD_MAX_IN_CH, "vid_pack: in_channels=%d out of range [3,%d] (1- and 2-channel models: widen to 3 planes with zero weights)", in_ch, VAD_MAX_IN_CH);
    int rc = vid_dims_ok(latent, hid, layers);
    if (rc != VAD_OK) return rc;
    const VidLayout L = vid_layout(in_ch, latent, hid, layers);
    REQ(nparams == vad_vid_nparams(layers, L.has_proj), "vid_pack: expected %d parameter tensors, got %d",
        vad_vid_nparams(layers, L.has_proj), nparams);
    for (int i = 0; i < nparams; ++i) REQ(P[i], "vid_pack: parameter %d is NULL", i);
    memset(out, 0, L.total * sizeof(float));
    put_header(out, VAD_BLOB_VID, precision, latent, hid | (layers << 16));
    const int first_convt = 4 + layers + (L.has_proj ? 1 : 0);
    int pi = 0;
    for (int li = 0; li < L.nlayers && rc == VAD_OK; ++li) {
        const LayerSlot& s = L.layer[li];
        const float* w = P[pi], *b = P[pi + 1];
        const float* bn[4] = {nullptr, nullptr, nullptr, nullptr};
        switch (s.kind) {
        case LK_CONV_C3: for (int i = 0; i < 4; ++i) bn[i] = P[pi + 2 + i];
            rc = vad_pack_conv3x3_c3(w, b, bn, s.cout, out + s.w, out + s.b); pi += 6; break;
        case LK_CONV: for (int i = 0; i < 4; ++i) bn[i] = P[pi + 2 + i];     // encoder.12 is the one whose cout is latent_dim; encoder.0 of a wide model reads in_ch planes
            rc = pack_conv3x3_slot(w, b, bn, li == 3 ? latent : s.cout, (li == 0 && L.wide) ? in_ch : s.cin, s, precision, out); pi += 6; break;
        case LK_CONVT:
            if (L.wide && li == L.nlayers - 1) {   // last layer of a wide model: ConvTranspose2d(32 -> in_ch) + bias, no BatchNorm; exact fp32 like the 3-plane tail
                rc = pack_convt2x2_slot(w, b, nullptr, 32, in_ch, s, VAD_PREC_FP32, out); pi += 2; break;
            }
            for (int i = 0; i < 4; ++i) bn[i] = P[pi + 2 + i];    // decoder.0 is the one whose cin is latent_dim
            rc = pack_convt2x2_slot(w, b, bn, li == first_convt ? latent : s.cin, s.cout, s, precision, out); pi += 6; break;
        case LK_LSTM: {   // weight (4*hid, x + hid, 3, 3): gate blocks i,f,g,o and the x / h input halves are padded separately
            const int xr = (li == 4) ? latent : hid, xp = (li == 4) ? L.latent_p : L.hid_p, hp = L.hid_p;
            // VAD_PREC_WINO: the gate convolution in Winograd form when both sources have one width (vad_convlstm_step_wino's
            // requirement; vid_run makes the same test), else the direct form
            const int lprec = (precision == VAD_PREC_WINO && xp == hp) ? VAD_PREC_WINO : base_prec(precision);
            if (xr == xp && hid == hp) { rc = pack_conv3x3_any(w, b, nullptr, s.cout, s.cin, lprec, out + s.w, out + s.b); pi += 2; break; }
            PaddedLayer Q;
            pad_layer(w, b, nullptr, 4 * hid, xr + hid, 9, false, s.cout, s.cin,
                      [=](int co) { return (co / hid) * hp + co % hid; }, [=](int ci) { return ci < xr ? ci : xp + (ci - xr); }, Q);
            rc = pack_conv3x3_any(Q.w.data(), Q.bias.data(), nullptr, s.cout, s.cin, lprec, out + s.w, out + s.b); pi += 2; break; }
        case LK_PROJ: {
            if (latent == s.cout && hid == s.cin) { rc = vad_pack_conv1x1(w, b, s.cout, s.cin, out + s.w, out + s.b); pi += 2; break; }
            PaddedLayer Q;
            pad_layer(w, b, nullptr, latent, hid, 1, false, s.cout, s.cin, same, same, Q);
            rc = vad_pack_conv1x1(Q.w.data(), Q.bias.data(), s.cout, s.cin, out + s.w, out + s.b); pi += 2; break; }
        case LK_TAIL_CONVT:
            memcpy(out + s.w, w, (size_t)32 * 12 * sizeof(float));
            for (int c = 0; c < 3; ++c) out[s.b + c] = b[c];
            pi += 2; break;
        default: return vad_fail(VAD_ERR_ARG, "vid_pack: internal layout error");
        }
    }
    if (rc == VAD_OK && pi != nparams) return vad_fail(VAD_ERR_ARG, "vid_pack: consumed %d of %d parameters", pi, nparams);
    return rc;
}
