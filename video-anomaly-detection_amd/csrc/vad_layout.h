// Packed-weight blob layouts shared by pack.cpp (host packing) and vad_api.hip (launch order).
#pragma once
#include <stddef.h>
#include "vad_hip.h"

int vad_fail(int code, const char* fmt, ...);

// model blob header (4 uint32 words at float offset 0): {VAD_BLOB_MAGIC, tag, dim0, dim1}
constexpr unsigned VAD_BLOB_MAGIC = 0x42444156u;   // "VADB"
constexpr int VAD_BLOB_HEADER_FLOATS = 4;
enum { VAD_BLOB_IMG = 1, VAD_BLOB_VID = 2 };
inline unsigned vad_blob_tag(int kind, int precision) {
    return ((unsigned)VAD_ABI_VERSION << 16) | ((unsigned)precision << 8) | (unsigned)kind;
}

enum { LK_CONV_C3 = 0, LK_CONV, LK_CONVT, LK_TAIL_CONV, LK_LSTM, LK_PROJ, LK_TAIL_CONVT };

struct LayerSlot {
    int kind, cin, cout;
    size_t w, b;   // float offsets into the blob
};
// The reference's constructors take ANY positive latent_dim / lstm_hidden_dim (models/autoencoder.py:161,
// models/video_autoencoder.py:290-296); the kernels tile channels in blocks of 32 (64 hidden channels per ConvLSTM block).
// The packers therefore zero-pad those dimensions: a padded channel has zero weights, zero bias and zero outgoing weights, so
// it carries exactly 0 through LeakyReLU / ReLU / MaxPool and through the ConvLSTM cell (gates 0 -> c' = 0.5*0 + 0.5*0, h' = 0)
// and adds `+ 0*w` terms to the sums it feeds: the results are those of the unpadded network.  Entry points take the REAL
// dimensions; layouts, workspaces and launches use the padded ones below.
inline int vad_pad_up(int x, int m) { return (x + m - 1) / m * m; }
inline int vad_img_latent_p(int latent) { return vad_pad_up(latent, 32); }
// without `proj` (hid == latent) the decoder reads the ConvLSTM output directly: one common padded width
inline int vad_vid_hid_p(int latent, int hid) { (void)latent; return vad_pad_up(hid, 64); }
inline int vad_vid_latent_p(int latent, int hid) { return hid == latent ? vad_pad_up(hid, 64) : vad_pad_up(latent, 32); }

struct ImgLayout {
    LayerSlot layer[16];
    int nlayers;
    int wide;              // in_ch > 3: padded channel count of the input / output planes, else 0
    int latent_p;          // padded latent width (layer[].cin / cout hold padded widths too)
    size_t total;
};
struct VidLayout {
    LayerSlot layer[4 + 8 + 1 + 4];
    int nlayers;
    int wide;              // as in ImgLayout
    int has_proj;          // from the REAL dimensions: lstm_hidden_dim != latent_dim (models/video_autoencoder.py:311-312)
    int latent_p, hid_p;
    size_t total;
};
// in_ch <= 3: the 3-plane path (K = 27 first layer, Cout = 3 tails; 1- and 2-channel models are widened with zero planes by the
// caller).  in_ch > 3 (csrc/wide_io.hip): first and last layer in generic slots of vad_wide_p(in_ch) channels.
constexpr int VAD_MAX_IN_CH = VAD_MAX_IN_CHANNELS;
inline int vad_wide_p(int in_ch) { return vad_pad_up(in_ch, 32); }
ImgLayout img_layout(int in_ch, int latent);            // real dimensions in, padded slots out
VidLayout vid_layout(int in_ch, int latent, int hid, int layers);
