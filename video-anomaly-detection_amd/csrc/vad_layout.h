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
struct ImgLayout {
    LayerSlot layer[16];
    int nlayers;
    size_t total;
};
struct VidLayout {
    LayerSlot layer[4 + 8 + 1 + 4];
    int nlayers;
    int has_proj;
    size_t total;
};
ImgLayout img_layout(int latent);
VidLayout vid_layout(int latent, int hid, int layers);
