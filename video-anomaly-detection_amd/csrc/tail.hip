// Memory-bound tails of the scoring path (VALU kernels): the Cout = 3 output layers fused with
// Tanh, the squared reconstruction error and a deterministic per-frame reduction; layout
// transposes; the on-device synthetic frame generator.
//
// Reference ops restated: Conv2d(32->3)+Tanh (models/autoencoder.py:134-135),
// ConvTranspose2d(32->3)+Tanh (models/video_autoencoder.py:259-260), error = (x-recon)**2,
// channel mean, spatial / temporal means (models/autoencoder.py:214-221,
// models/video_autoencoder.py:371-384).
#include "vad_common.h"

struct TailP {
    const float* in;      // NHWC activations feeding the last layer
    const float* w;       // packed weights
    const float* bias;    // [3]
    const float* x;       // original input, NCHW [N,3,H2,W2]
    float* partials;      // [N][nparts]
    float* recon;         // NCHW or NULL
    float* errmap;        // [N,H2,W2] or NULL
    int h, w_;            // spatial size of `in`
    int tiles_x, tiles_y;
    int xt, xs;           // x frame of activation frame n is (n / xt) * xs + n % xt (xt == 0: n)
    int xu8;              // x is uint8 NHWC [N,H2,W2,3]
    unsigned nitems;      // convT tail: n * h * tiles_x wave items
};

// original-input value of (frame nx, channel c, pixel y,x)
__device__ __forceinline__ float tail_x(const TailP& p, size_t nx, int c, int y, int x, int h2, int w2) {
    if (p.xu8) return vad_norm_u8(((const unsigned char*)p.x)[(nx * h2 * w2 + (size_t)y * w2 + x) * 3 + c]);
    return p.x[(nx * 3 + c) * (size_t)h2 * w2 + (size_t)y * w2 + x];
}

// the same value in two halves: the raw 32-bit load (issued early, so that waiting for it does not drain the younger
// activation prefetches behind it - vmcnt retires in order) and its conversion at the point of use
// The input format is a template argument there: with a run-time branch hipcc joins the two load paths on one register and
// puts a vmcnt(0) in front of the load, which drains the queue just the same.
template <bool XU8>
__device__ __forceinline__ void tail_x_raw(const TailP& p, size_t nx, int y, int x, int h2, int w2, unsigned (&raw)[3]) {
    if (XU8) {
        // bytes 0,1 as ONE (unaligned) 16-bit load, split at the use: hipcc merges two byte loads into this load anyway, and
        // then splits it right behind the load - a wait in the wrong place again
        const unsigned char* q = (const unsigned char*)p.x + (nx * h2 * w2 + (size_t)y * w2 + x) * 3;
        raw[0] = *(const unsigned short*)q;
        raw[1] = q[2];
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) raw[c] = __float_as_uint(p.x[(nx * 3 + c) * (size_t)h2 * w2 + (size_t)y * w2 + x]);
    }
}
template <bool XU8>
__device__ __forceinline__ float tail_x_cvt(const unsigned (&raw)[3], int c) {
    if (XU8) return vad_norm_u8((unsigned char)(c == 0 ? raw[0] & 0xffu : c == 1 ? raw[0] >> 8 : raw[1]));
    return __uint_as_float(raw[c]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// Fixed-shape block reduction: wave butterfly, then waves 0..3 added in order by thread 0.
__device__ __forceinline__ void block_partial(float e, float* red, float* dst) {
    const float s = wave_sum(e);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *dst = ((red[0] + red[1]) + red[2]) + red[3];
}

// Conv2d(32->3)+Tanh+error, VALU.  A wave is 8 row-groups x 8 channel-groups: lane (g = lane>>3,
// cg = lane&7) owns output row y0 + 8*wave + g and input channels 4cg..4cg+3, keeps its 108 weights
// (9 taps x 4 channels x 3 outputs) in registers for the whole tile and slides a 3x3 window along x,
// loading ONE new column (3 x 16 B; the 8 lanes of a group read one full 128-B pixel row) per step,
// two columns ahead of its use.  Every 8 steps the 8x3 partial sums are transpose-reduced across
// the 8 channel-group lanes (7 adds, 21 shuffles for 24 values), which leaves lane cg holding the 3
// finished outputs of pixel x0 + cg: Tanh, error and the optional stores then run on all 64 lanes
// with 32-B contiguous accesses per row.
constexpr int TAIL_T = 32;   // tile side (output pixels) of the conv3x3 tail

template <bool XU8>
__global__ __launch_bounds__(256, 2) void conv3x3_to3_score_kernel(TailP p) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 3, cg = lane & 7;
    unsigned L = blockIdx.x;
    const int tx = L % p.tiles_x; L /= p.tiles_x;
    const int ty = L % p.tiles_y;
    const int n = L / p.tiles_y;
    const int y = ty * TAIL_T + wave * 8 + g, x0 = tx * TAIL_T;
    const int H = p.h, W = p.w_;

    float w[9][4][3];
    {
        const f32x4* wv = (const f32x4*)(p.w + cg * 108);
#pragma unroll
        for (int i = 0; i < 27; ++i) {
            const f32x4 q = wv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = i * 4 + e;
                w[f / 12][(f / 3) % 4][f % 3] = q[e];
            }
        }
    }
    const float b0 = p.bias[0], b1 = p.bias[1], b2 = p.bias[2];

    // Zero padding in y is folded into the per-lane weights (taps of an outside row are zeroed and the
    // row address is clamped), so loads need no per-lane predicate; padding in x is wave-uniform.
    if (y == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) w[t][j][0] = w[t][j][1] = w[t][j][2] = 0.f;
    }
    if (y >= H - 1) {
#pragma unroll
        for (int t = 6; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) w[t][j][0] = w[t][j][1] = w[t][j][2] = 0.f;
    }
    const char* base = (const char*)(p.in + (size_t)n * H * W * 32);   // wave-uniform
    int voff[3];                                                       // per-lane byte offset of column x0+15
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        int yy = y - 1 + r;
        yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
        voff[r] = ((yy * W + x0 + 15) * 32 + cg * 4) * 4;
    }
    constexpr int D = 2;                       // prefetch distance in columns
    f32x4 col[TAIL_T + 2][3];                  // col[k] = input column x0 - 1 + k (registers; static indices)
    // Columns 1..16 of a tile are always inside the image (W % 16 == 0): immediate offsets, no checks.
    // Column 0 (x0-1) and columns >= 17 may fall outside: the address is clamped (wave-uniform) and the
    // two columns that can carry zero padding into a kept output (x = -1 and x = W) are multiplied by 0 - at their FIRST USE
    // (MASK_COL, step k-2), not behind the load: a multiply there waits for the column just requested, i.e. for everything.
#define LOAD_COL(k)                                                                          \
    {                                                                                        \
        if ((k) >= 1 && (k) <= 16) {                                                         \
            _Pragma("unroll") for (int r = 0; r < 3; ++r)                                    \
                col[k][r] = *(const f32x4*)(base + voff[r] + ((k) - 16) * 128);              \
        } else {                                                                             \
            const int xx_ = x0 - 1 + (k);                                                    \
            const int xc_ = xx_ < 0 ? 0 : (xx_ > W - 1 ? W - 1 : xx_);                       \
            _Pragma("unroll") for (int r = 0; r < 3; ++r)                                    \
                col[k][r] = *(const f32x4*)(base + voff[r] + (xc_ - x0 - 15) * 128);         \
        }                                                                                    \
    }
#define MASK_COL(k)                                                                          \
    if ((k) == 0 || (k) == 17 || (k) == 33) {                                                \
        const int xx_ = x0 - 1 + (k);                                                        \
        const float mk_ = (xx_ >= 0 && xx_ < W) ? 1.f : 0.f;                                 \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) col[k][r] *= mk_;                      \
    }
#pragma unroll
    for (int k = 0; k < 2 + D; ++k) LOAD_COL(k);

    const size_t plane = (size_t)H * W;
    float esum = 0.f;
    float v[8][3];
    unsigned xraw[3] = {0u, 0u, 0u};
#pragma unroll
    for (int s = 0; s < TAIL_T; ++s) {
        if ((s & 7) == 0) {
            // original-input values of the pixel this lane finishes 7 steps from now: requested BEFORE the columns below, so
            // the wait at their use leaves the column prefetch in flight (loaded at the use they cost a vmcnt(0) - the whole
            // queue drained plus one exposed HBM round trip - four times per tile)
            const int x = x0 + s + cg;
            if (y < H && x < W) tail_x_raw<XU8>(p, n, y, x, H, W, xraw);
        }
        if (s + 2 + D < TAIL_T + 2) LOAD_COL(s + 2 + D);
        __builtin_amdgcn_sched_barrier(0);
        if (s == 0) MASK_COL(0);
        MASK_COL(s + 2);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float in = col[s + dx][dy][j];
                    a0 = fmaf(in, w[dy * 3 + dx][j][0], a0);
                    a1 = fmaf(in, w[dy * 3 + dx][j][1], a1);
                    a2 = fmaf(in, w[dy * 3 + dx][j][2], a2);
                }
        v[s & 7][0] = a0; v[s & 7][1] = a1; v[s & 7][2] = a2;
        if ((s & 7) == 7) {
            // transpose-reduce over the 8 channel-group lanes: lane cg ends with pixel x0 + (s-7) + cg
            float r4[4][3], r2[2][3], r1[3];
            const bool h4 = cg & 4, h2 = cg & 2, h1 = cg & 1;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float keep = h4 ? v[q + 4][c] : v[q][c], send = h4 ? v[q][c] : v[q + 4][c];
                    r4[q][c] = keep + __shfl_xor(send, 4);
                }
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float keep = h2 ? r4[q + 2][c] : r4[q][c], send = h2 ? r4[q][c] : r4[q + 2][c];
                    r2[q][c] = keep + __shfl_xor(send, 2);
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float keep = h1 ? r2[1][c] : r2[0][c], send = h1 ? r2[0][c] : r2[1][c];
                r1[c] = keep + __shfl_xor(send, 1);
            }
            const int x = x0 + (s - 7) + cg;
            if (y < H && x < W) {
                const float rc[3] = {vad_tanh(r1[0] + b0), vad_tanh(r1[1] + b1), vad_tanh(r1[2] + b2)};
                const size_t o = (size_t)n * 3 * plane + (size_t)y * W + x;
                float e = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float d = tail_x_cvt<XU8>(xraw, c) - rc[c];
                    e += d * d;
                    if (p.recon) p.recon[o + c * plane] = rc[c];
                }
                if (p.errmap) p.errmap[(size_t)n * plane + (size_t)y * W + x] = e / 3.0f;
                esum += e;
            }
        }
    }
#undef LOAD_COL
#undef MASK_COL
    block_partial(esum, red, &p.partials[(size_t)n * (p.tiles_x * p.tiles_y) + ty * p.tiles_x + tx]);
}

// ConvTranspose2d(32->3, k2, s2) + Tanh + error: a 32 -> 12 GEMV per input pixel (12 = 3 channels x 2x2 output pixels), no
// reuse between pixels - so no LDS tile and no barrier.  One WAVE owns 64 consecutive input pixels of one row and reads them
// fully coalesced: load i (of 8) takes pixels 8i + lane/8, channel quad lane%8, i.e. 1 KiB contiguous per instruction.  A
// lane therefore holds a 4-channel slice of 8 pixels; it multiplies them by ITS 4 x 12 weights (48 registers, read once),
// and the 8 lanes of a pixel group transpose-reduce the partial sums (as in the conv3x3 tail above) so that lane c4 ends up
// with the 12 finished outputs of pixel 8 c4 + lane/8: Tanh, error against the original input (loaded up front with the
// activations, so both streams are in flight together) and the optional stores run on all 64 lanes.  History: an 8x32-pixel
// LDS tile behind a barrier, one thread per pixel, x loaded after the FMAs ran at 2.26 TB/s of algorithmic bytes; one lane
// per pixel reading its own 128 B (64 different lines per load instruction) at 3.6 TB/s.  Per-(row, segment) sums go to
// `partials` (h * ceil(w/64) per frame); score_finalize_kernel adds them in a fixed order.
template <int CIN>
__global__ __launch_bounds__(256) void convt2x2_to3_score_kernel(TailP p) {
    static_assert(CIN == 32, "eight channel quads per pixel, eight lanes per pixel group");
    const int lane = threadIdx.x & 63, g = lane >> 3, c4 = lane & 7;
    const unsigned item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (item >= p.nitems) return;                                   // wave-uniform
    const int segs = p.tiles_x, H = p.h, W = p.w_;
    unsigned r = item;
    const int sx = r % segs; r /= segs;
    const int y = r % H;
    const int n = r / H;
    const int x = sx * 64 + 8 * c4 + g;                             // the pixel this lane finishes
    const bool ok = x < W;
    const int nx = p.xt ? (n / p.xt) * p.xs + n % p.xt : n;         // source frame this activation frame is scored against
    const int h2 = 2 * H, w2 = 2 * W;
    const size_t plane = (size_t)h2 * w2;

    f32x4 a[8];
    const float* row = p.in + ((size_t)n * H + y) * W * CIN + 4 * c4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int xi = sx * 64 + 8 * i + g;                         // (pixels past the row end: read pixel 0, result discarded)
        a[i] = *(const f32x4*)(row + (size_t)(xi < W ? xi : 0) * CIN);
    }
    float2 xv[2][3];
    if (!p.xu8) {
        const float* xs = p.x + (size_t)nx * 3 * plane + (size_t)(2 * y) * w2 + 2 * (ok ? x : 0);
#pragma unroll
        for (int ra = 0; ra < 2; ++ra)
#pragma unroll
            for (int co = 0; co < 3; ++co) xv[ra][co] = *(const float2*)(xs + co * plane + (size_t)ra * w2);
    } else {
#pragma unroll
        for (int ra = 0; ra < 2; ++ra)
#pragma unroll
            for (int co = 0; co < 3; ++co)
                xv[ra][co] = make_float2(tail_x(p, nx, co, 2 * y + ra, 2 * (ok ? x : 0), h2, w2), tail_x(p, nx, co, 2 * y + ra, 2 * (ok ? x : 0) + 1, h2, w2));
    }
    f32x4 wq[12];                                                   // rows 4 c4 .. 4 c4 + 3 of the IOHW weight: [j][k = (co, a, b)]
#pragma unroll
    for (int q = 0; q < 12; ++q) wq[q] = *(const f32x4*)(p.w + c4 * 48 + 4 * q);

    float acc[12];
    const bool h4 = c4 & 4, h2b = c4 & 2, h1 = c4 & 1;
#pragma unroll
    for (int co = 0; co < 3; ++co) {                                // one output channel (4 of the 12 columns) at a time: 32 live partials
        float v[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = co * 4 + kk;                          // weight (j, k) = wq[(12 j + k) / 4][(12 j + k) % 4]
                float sacc = a[i][0] * wq[k / 4][k % 4];
#pragma unroll
                for (int j = 1; j < 4; ++j) sacc = fmaf(a[i][j], wq[(12 * j + k) / 4][(12 * j + k) % 4], sacc);
                v[i][kk] = sacc;
            }
        // transpose-reduce over the 8 channel-quad lanes: lane c4 ends with pixel index i == c4
        float r4[4][4], r2[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float keep = h4 ? v[q + 4][kk] : v[q][kk], send = h4 ? v[q][kk] : v[q + 4][kk];
                r4[q][kk] = keep + __shfl_xor(send, 4);
            }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float keep = h2b ? r4[q + 2][kk] : r4[q][kk], send = h2b ? r4[q][kk] : r4[q + 2][kk];
                r2[q][kk] = keep + __shfl_xor(send, 2);
            }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float keep = h1 ? r2[1][kk] : r2[0][kk], send = h1 ? r2[0][kk] : r2[1][kk];
            acc[co * 4 + kk] = keep + __shfl_xor(send, 1) + p.bias[co];
        }
    }
    float e = 0.f;
    if (ok) {
#pragma unroll
        for (int ra = 0; ra < 2; ++ra) {
            float ea[2] = {0.f, 0.f};
            const size_t o = (size_t)n * 3 * plane + (size_t)(2 * y + ra) * w2 + 2 * x;
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                const float r0 = vad_tanh(acc[co * 4 + ra * 2 + 0]), r1 = vad_tanh(acc[co * 4 + ra * 2 + 1]);
                const float d0 = xv[ra][co].x - r0, d1 = xv[ra][co].y - r1;
                ea[0] += d0 * d0;
                ea[1] += d1 * d1;
                if (p.recon) *(float2*)(p.recon + o + co * plane) = make_float2(r0, r1);
            }
            if (p.errmap)
                *(float2*)(p.errmap + (size_t)n * plane + (size_t)(2 * y + ra) * w2 + 2 * x) =
                    make_float2(ea[0] / 3.0f, ea[1] / 3.0f);
            e += ea[0] + ea[1];
        }
    }
    const float s = wave_sum(e);
    if (lane == 0) p.partials[(size_t)n * ((size_t)H * segs) + (size_t)y * segs + sx] = s;
}

// One 64-lane block per clip (t frames): lane l adds partials l, l+64, ... in order, then the
// wave butterfly; the result depends only on (frame data, tile grid), never on batch or rank.
// hdr / want: header of the packed model blob and the tag the launch was made for (vad_layout.h); a blob packed for
// another arithmetic mode or model kind turns every score into NaN instead of a plausible-looking number.
__global__ __launch_bounds__(256) void score_finalize_kernel(const float* partials, int nparts, float denom,
                                                             float* frame_scores, float* seq_scores, int t,
                                                             const unsigned* hdr, unsigned want) {
    // One work-group per clip; wave w reduces frames w, w+4, ... (a frame is one wave's job, lane i adding partials i, i+64, ...
    // and a butterfly over the lanes - the arithmetic of the one-wave form this replaces, which walked the frames one memory
    // round trip after the other: 20 us for 16 frames), thread 0 then adds the frame scores in frame order.
    __shared__ float fs[256];
    const int clip = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float poison = (hdr && hdr[1] != want) ? __builtin_nanf("") : 0.f;
    float seq = 0.f;
    for (int f0 = 0; f0 < t; f0 += 256) {
        const int cnt = t - f0 < 256 ? t - f0 : 256;
        for (int f = wave; f < cnt; f += 4) {
            const float* pp = partials + ((size_t)clip * t + f0 + f) * nparts;
            float s = 0.f;
            for (int i = lane; i < nparts; i += 64) s += pp[i];
            s = wave_sum(s) / denom + poison;
            if (lane == 0) {
                fs[f] = s;
                if (frame_scores) frame_scores[(size_t)clip * t + f0 + f] = s;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int f = 0; f < cnt; ++f) seq += fs[f];
        __syncthreads();
    }
    if (threadIdx.x == 0 && seq_scores) seq_scores[clip] = seq / (float)t;
}

extern "C" int vad_score_partials(int kind, int h2, int w2) {
    if (kind == 0) return ((h2 + TAIL_T - 1) / TAIL_T) * ((w2 + TAIL_T - 1) / TAIL_T);
    if (kind == 1) return (h2 / 2) * ((w2 / 2 + 63) / 64);          // one partial per (input row, 64-pixel segment)
    return vad_fail(VAD_ERR_ARG, "score_partials: kind must be 0 (conv3x3 tail) or 1 (convT tail)");
}

extern "C" int vad_conv3x3_to3_score(const float* in, const float* w_packed, const float* bias3,
                                     const float* x, float* partials, float* recon, float* errmap,
                                     int n, int h2, int w2, int cin, void* stream) {
    return vad_conv3x3_to3_score_fmt(in, w_packed, bias3, x, VAD_X_F32_NCHW, partials, recon, errmap, n, h2, w2, cin, stream);
}

int vad_conv3x3_to3_score_fmt(const float* in, const float* w_packed, const float* bias3, const void* x, int fmt,
                              float* partials, float* recon, float* errmap, int n, int h2, int w2, int cin, void* stream) {
    VAD_REQUIRE(in && w_packed && bias3 && x && partials, "conv3x3_to3_score: null pointer");
    VAD_REQUIRE(fmt == VAD_X_F32_NCHW || fmt == VAD_X_U8_NHWC, "conv3x3_to3_score: bad input format %d", fmt);
    VAD_REQUIRE(cin == 32, "conv3x3_to3_score: cin=%d unsupported (the reference's dec4.3 has 32)", cin);
    VAD_REQUIRE(n > 0 && h2 > 0 && w2 > 0 && w2 % 16 == 0, "conv3x3_to3_score: bad shape (W=%d must be a positive multiple of 16)", w2);
    TailP p{in, w_packed, bias3, (const float*)x, partials, recon, errmap, h2, w2, (w2 + TAIL_T - 1) / TAIL_T, (h2 + TAIL_T - 1) / TAIL_T, 0, 0, fmt == VAD_X_U8_NHWC};
    const long long nb = (long long)n * p.tiles_x * p.tiles_y;
    VAD_REQUIRE(nb < (1ll << 31), "conv3x3_to3_score: grid too large");
    if (p.xu8) hipLaunchKernelGGL(conv3x3_to3_score_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(conv3x3_to3_score_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_convt2x2_to3_score(const float* in, const float* w_iohw, const float* bias3,
                                      const float* x, float* partials, float* recon, float* errmap,
                                      int n, int h, int w, int cin, int t, int clip_stride, void* stream) {
    return vad_convt2x2_to3_score_fmt(in, w_iohw, bias3, x, VAD_X_F32_NCHW, partials, recon, errmap, n, h, w, cin, t, clip_stride, stream);
}

int vad_convt2x2_to3_score_fmt(const float* in, const float* w_iohw, const float* bias3, const void* x, int fmt,
                               float* partials, float* recon, float* errmap, int n, int h, int w, int cin, int t,
                               int clip_stride, void* stream) {
    VAD_REQUIRE(in && w_iohw && bias3 && x && partials, "convt2x2_to3_score: null pointer");
    VAD_REQUIRE(fmt == VAD_X_F32_NCHW || fmt == VAD_X_U8_NHWC, "convt2x2_to3_score: bad input format %d", fmt);
    VAD_REQUIRE(cin == 32, "convt2x2_to3_score: cin=%d unsupported (the reference's decoder.9 has 32)", cin);
    VAD_REQUIRE(n > 0 && h > 0 && w > 0, "convt2x2_to3_score: bad shape");
    VAD_REQUIRE(t >= 0 && clip_stride >= 0 && (t == 0 || clip_stride > 0), "convt2x2_to3_score: bad window mapping");
    TailP p{in, w_iohw, bias3, (const float*)x, partials, recon, errmap, h, w, (w + 63) / 64, h, (t == clip_stride) ? 0 : t, clip_stride, fmt == VAD_X_U8_NHWC, 0u};
    const long long items = (long long)n * h * p.tiles_x;
    VAD_REQUIRE(items < (1ll << 31), "convt2x2_to3_score: grid too large");
    p.nitems = (unsigned)items;
    hipLaunchKernelGGL((convt2x2_to3_score_kernel<32>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_score_finalize(const float* partials, int nparts, int n, int h2, int w2,
                                  float* frame_scores, float* seq_scores, int t, void* stream) {
    return vad_score_finalize_tagged(partials, nparts, n, h2, w2, 3, frame_scores, seq_scores, t, nullptr, 0u, stream);
}

int vad_score_finalize_tagged(const float* partials, int nparts, int n, int h2, int w2, int channels, float* frame_scores,
                              float* seq_scores, int t, const unsigned* hdr, unsigned want_tag, void* stream) {
    VAD_REQUIRE(partials && nparts > 0 && n > 0 && t > 0 && n % t == 0 && channels > 0, "score_finalize: bad arguments");
    VAD_REQUIRE(frame_scores || seq_scores, "score_finalize: no output requested");
    const float denom = (float)channels * (float)h2 * (float)w2;
    hipLaunchKernelGGL(score_finalize_kernel, dim3(n / t), dim3(256), 0, (hipStream_t)stream,
                       partials, nparts, denom, frame_scores, seq_scores, t, hdr, want_tag);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// --------------------------------------------------------------------------------- transposes
// Per frame [P][C] <-> [C][P] through a 32x33 LDS tile (both sides coalesced).
// in_ld: row pitch of the source (>= cols; the latent code of a zero-padded latent_dim keeps its padded pitch)
__global__ __launch_bounds__(256) void transpose_kernel(const float* in, float* out, int rows, int cols, int in_ld) {
    __shared__ float t[32][33];
    const size_t base = (size_t)blockIdx.z * rows * cols, ibase = (size_t)blockIdx.z * rows * in_ld;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int j = ly; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + lx;
        if (r < rows && c < cols) t[j][lx] = in[ibase + (size_t)r * in_ld + c];
    }
    __syncthreads();
    for (int j = ly; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + lx;
        if (r < rows && c < cols) out[base + (size_t)c * rows + r] = t[lx][j];
    }
}

static int launch_transpose(const float* in, float* out, int n, int rows, int cols, void* stream, int in_ld = 0) {
    if (in_ld == 0) in_ld = cols;
    VAD_REQUIRE(in && out && n > 0 && rows > 0 && cols > 0 && n < 65536 && in_ld >= cols, "transpose: bad arguments");
    dim3 g((cols + 31) / 32, (rows + 31) / 32, n);
    VAD_REQUIRE(g.y < 65536, "transpose: too many row tiles");
    hipLaunchKernelGGL(transpose_kernel, g, dim3(256), 0, (hipStream_t)stream, in, out, rows, cols, in_ld);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// the first `c` channels of an NHWC tensor whose pixels are `in_c` floats apart
int vad_nhwc_to_nchw_ld(const float* in, int in_c, float* out, int n, int h, int w, int c, void* stream) {
    return launch_transpose(in, out, n, h * w, c, stream, in_c);
}
extern "C" int vad_nhwc_to_nchw(const float* in, float* out, int n, int h, int w, int c, void* stream) {
    return launch_transpose(in, out, n, h * w, c, stream);
}
extern "C" int vad_nchw_to_nhwc(const float* in, float* out, int n, int h, int w, int c, void* stream) {
    return launch_transpose(in, out, n, c, h * w, stream);
}

// ----------------------------------------------------------------------- synthetic frames
// Must stay bit-identical to synth.py: splitmix64 finaliser of (element index + seed*golden),
// top byte -> u8 -> (u8/255 - 0.5)/0.5 in fp32 (mirrors ToTensor+Normalize, reference
// utils/dataset.py:67-69).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_frames_kernel(float* out, unsigned long long seed, long long first_frame,
                                                           long long total, int c, int h, int w, int anomalies) {
    const unsigned long long golden = 0x9E3779B97F4A7C15ull;
    const long long per = (long long)c * h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const unsigned long long gidx = (unsigned long long)(first_frame * per + i);
        unsigned u8 = (unsigned)(mix64(gidx + seed * golden) >> 56);
        if (anomalies) {
            const long long f = first_frame + i / per;
            const unsigned long long lab = mix64((unsigned long long)f + (seed ^ 0x5DEECE66Dull) * golden) >> 63;
            if (lab) {
                const unsigned long long z = mix64((unsigned long long)f + (seed ^ 0xB5297A4Dull) * golden);
                const int side = anomalies == 1 ? 32 : anomalies;     // synth.patch_side
                const int ph = h - side + 1 > 1 ? h - side + 1 : 1, pw = w - side + 1 > 1 ? w - side + 1 : 1;
                const int py = (int)((z & 0xFFFF) % (unsigned)ph), px = (int)(((z >> 16) & 0xFFFF) % (unsigned)pw);
                const long long rem = i % per;
                const int yy = (int)((rem / w) % h), xx = (int)(rem % w);
                if (yy >= py && yy < py + side && xx >= px && xx < px + side) u8 = 255;
            }
        }
        float f32 = (float)u8;
        f32 = f32 / 255.0f;
        f32 = f32 - 0.5f;
        out[i] = f32 / 0.5f;
    }
}

extern "C" int vad_synth_frames(float* out, unsigned long long seed, long long first_frame, long long n,
                                int c, int h, int w, int anomalies, void* stream) {
    VAD_REQUIRE(out && n > 0 && c > 0 && h > 0 && w > 0 && first_frame >= 0 && anomalies >= 0, "synth_frames: bad arguments");
    const long long total = n * c * h * w;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(synth_frames_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       out, seed, first_frame, total, c, h, w, anomalies);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}
