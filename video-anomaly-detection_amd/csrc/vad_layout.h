// Packed-weight blob layouts shared by pack.cpp (host packing) and vad_api.hip (launch order).
#pragma once
#include <stddef.h>
#include "vad_hip.h"

int vad_fail(int code, const char* fmt, ...);

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
