// dec4.0 + dec4.3 + score in ONE kernel: ConvTranspose2d(32->32, k2 s2) + BatchNorm + ReLU, Conv2d(32->3, k3 p1) + Tanh,
// (x - recon)^2, channel mean and the per-frame reduction (reference models/autoencoder.py:131-139, 211-221).
//
// Unfused, the 256x256x32 map between the two layers is written and read back: 2 x 8.39 MB of the 19.7 MB the two launches
// move per frame.  Here it never exists in memory, and neither in LDS: it lives for 16 MFMAs in the accumulator registers.
//
//   GEMM 1 (the transposed convolution), TRANSPOSED orientation: A = weights (lane = output channel), B = input pixels
//     (lane = pixel), so the result tile has lane = pixel and register r = channel (r&3) + 8 (r>>2) + 4 (lane>>5).  The k
//     order per output is the one convt2x2_pkernel uses (channel pairs (j, 4+j) of each 8-group, j = 0..3): the activation
//     is bit-identical to the unfused dec4.0.
//   GEMM 2 (the 3x3 convolution as a per-pixel product): P[pixel][(tap, co)] = sum_ci act[pixel][ci] W[co][ci][tap],
//     M = 27 (tap, co) rows padded to 32, K = 32 channels, N = pixels.  In that orientation the B operand of step r is
//     lane = pixel, k = (channel c_r, c_r + 4) - exactly register r of GEMM 1's result in its two lane halves: bias and ReLU
//     are applied in place and the 16 registers are fed straight back into the matrix pipe.  No LDS transpose.
//   The 3x3 neighbourhood is then 27 additions per output pixel: out[y][x][co] = sum_{dy,dx} P[y+dy-1][x+dx-1][(dy,dx),co].
//     A work-group owns a band of rows over the whole width (no x halo to recompute; the y halo is one input row per band
//     end) and walks it one OUTPUT row at a time: the four waves put that row of P (27 x 256 floats) into LDS, and after one
//     barrier thread x adds its nine row sums s[dy][co] = sum_dx P[(dy,dx,co)][x+dx-1] into three running output rows held
//     in registers: row y is complete when P rows y-1, y, y+1 have passed.  Tanh, error, optional recon / error-map stores
//     and the per-row partial sum follow in the same thread.  P rows are double-buffered: one barrier per output row.
//
// Per 256x256 frame: 2 x 0.134 GFLOP on the fp32 matrix pipe (GEMM 2 pads 27 -> 32 rows; the unfused tail ran 0.113 GFLOP on
// the VALU, the same pipe) and 2.1 MB + 0.79 MB read from HBM: bound by the exact-fp32 matrix pipe at ~1.8 us per frame.
// Sums are taken in a fixed order that depends on the pixel only (never on the band height, batch or rank).
#include <atomic>
#include "vad_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {
constexpr int D4_PITCH = 260;                 // P row in LDS: column jl = 2 + j, j = output column - 2 * xs0 in [-1, 256]
constexpr int D4_ROWS = 27;                   // (tap, co) rows, m = tap * 3 + co
constexpr int D4_BUF = D4_ROWS * D4_PITCH;    // floats per P buffer

struct Dec4P {
    const float* in;       // [n][H][W][32] NHWC: dec3.3 output
    const float* wt;       // transposed-conv weights, fp32 pack [q][cin/8][cout][8] (BatchNorm folded)
    const float* bt;       // [32] folded bias
    const float* w2;       // GEMM form of the 3x3 weights: [m 32][lane half 2][r 16]
    const float* b3;       // [3]
    const void* x;         // original frames (format per template argument)
    float* partials;       // [n][2H * nstrips * 4]
    float* recon;          // NCHW or NULL
    float* errmap;         // [n][2H][2W] or NULL
    int n, H, W;           // input map size; the output is 2H x 2W
    int R, nbands, nstrips;
    unsigned nitems;       // n * nbands * nstrips
};

__device__ __forceinline__ float d4_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

template <bool XU8>
__global__ __launch_bounds__(256, 2) void dec4_score_kernel(Dec4P p) {
    __shared__ __attribute__((aligned(16))) float P[2 * D4_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int H = p.H, W = p.W, H2 = 2 * H, W2 = 2 * W;

    // ---- per-lane constants, resident for the whole kernel
    // GEMM 1 A fragments: lane (co = li, half lh) holds channels 8 ks + 4 lh + {0..3} of quadrant q = 2a + b
    f32x4 wA[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                wA[a][b][ks] = *(const f32x4*)(p.wt + ((((2 * a + b) * 4 + ks) * 32 + li) * 8 + 4 * lh));
    float bias1[16], w2f[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bias1[r] = p.bt[(r & 3) + 8 * (r >> 2) + 4 * lh];
        w2f[r] = p.w2[(li * 2 + lh) * 16 + r];
    }
    const float c3b0 = p.b3[0], c3b1 = p.b3[1], c3b2 = p.b3[2];

    // zero both P buffers once: columns no lane ever writes (the x halo, columns past the image) are the zero padding
    for (int i = tid; i < 2 * D4_BUF / 4; i += 256) ((f32x4*)P)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const unsigned in_bytes = (unsigned)(H * W) * 32u * 4u;
    const size_t plane = (size_t)H2 * W2;
    unsigned phase = 0;                                                  // P buffer parity, carried across items

    for (unsigned item = blockIdx.x; item < p.nitems; item += gridDim.x) {
        unsigned t_ = item;
        const int strip = t_ % p.nstrips; t_ /= p.nstrips;
        const int band = t_ % p.nbands;
        const int n = t_ / p.nbands;
        const int r0 = band * p.R, r1 = (r0 + p.R < H) ? r0 + p.R : H;
        const int xs0 = 126 * strip;                                     // first input column of the strip
        const int jlo = strip ? 2 : 0;
        int jhi = W2 - 2 * xs0;
        const int jcap = (strip == p.nstrips - 1) ? 256 : 254;
        if (jhi > jcap) jhi = jcap;
        if (p.nstrips > 1) {                                             // the never-written columns differ between strips
            __syncthreads();
            for (int i = tid; i < 2 * D4_BUF / 4; i += 256) ((f32x4*)P)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
        }
        const int cx = xs0 + 32 * wave + li;                             // this lane's input column (GEMM phases)
        const bool wave_on = xs0 + 32 * wave < W;                        // (wave-uniform)
        const bool col_ok = cx < W;
        const __amdgpu_buffer_rsrc_t rin = vad_rsrc(p.in + (size_t)n * H * W * 32, in_bytes);
        const int jl_w = 2 + 2 * (32 * wave + li);                       // LDS column of this lane's b = 0 output

        // this thread's output column (combine phases)
        const int j = tid, ox = 2 * xs0 + j;
        const bool own = j >= jlo && j < jhi;
        const int oxc = own ? ox : 0;

        f32x4 fr[4], nf[4];                                              // input fragments of the current / next input row
        auto load_row = [&](f32x4 (&dst)[4], int iy) {
            const bool ok = iy >= 0 && iy < H && col_ok;
            const unsigned off = ok ? (unsigned)(__mul24(__mul24(iy, W) + cx, 32) + 4 * lh) * 4u : VAD_OOB;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dst[ks] = vad_bload4(rin, off, (unsigned)ks * 32u);
        };
        if (wave_on) load_row(fr, r0 - 1);

        unsigned xraw[3] = {0u, 0u, 0u};
        auto load_x = [&](int yo) {                                      // original-input values of (yo, ox), raw
            if (!own) return;
            if (XU8) {
                const unsigned char* q = (const unsigned char*)p.x + (((size_t)n * H2 + yo) * W2 + oxc) * 3;
                xraw[0] = q[0]; xraw[1] = q[1]; xraw[2] = q[2];
            } else {
                const float* q = (const float*)p.x + (size_t)n * 3 * plane + (size_t)yo * W2 + oxc;
#pragma unroll
                for (int c = 0; c < 3; ++c) xraw[c] = __float_as_uint(q[c * plane]);
            }
        };
        load_x(2 * r0);

        float Oa[3] = {0.f, 0.f, 0.f}, Ob[3] = {0.f, 0.f, 0.f};          // rows yp-1 (two of three terms) and yp (one term)

        for (int iy = r0 - 1; iy <= r1; ++iy) {
            const bool row_ok = iy >= 0 && iy < H;                       // (uniform)
            if (wave_on && iy + 1 <= r1) load_row(nf, iy + 1);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (a == 0 ? iy < r0 : iy >= r1) continue;               // band ends: only a = 1 of row r0-1, only a = 0 of row r1
                const int yp = 2 * iy + a;
                float* buf = P + (phase & 1u) * D4_BUF;
                if (row_ok && wave_on) {
                    f32x16 acc1[2], acc2[2];
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { acc1[b][r] = bias1[r]; acc2[b][r] = 0.f; }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int b = 0; b < 2; ++b) acc1[b] = MFMA32(wA[a][b][ks][jj], fr[ks][jj], acc1[b]);
#pragma unroll
                    // ReLU as plain fmaxf, NOT the inline-asm v_max of vad_act: hipcc's hazard recogniser does not look inside
                    // inline asm, so it would put no wait states between the MFMA that writes acc1 and an asm statement that
                    // reads it (the other kernels' epilogues have address arithmetic in between; here the read follows at once)
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc2[b] = MFMA32(w2f[r], fmaxf(acc1[b][r], 0.f), acc2[b]);
                    // P rows m = (r&3) + 8 (r>>2) + 4 lh of this lane's two output pixels (b = 0, 1): one 8-byte store each
                    if (col_ok) {
                        float* dst = buf + 4 * lh * D4_PITCH + jl_w;
#pragma unroll
                        for (int r = 0; r < 12; ++r)
                            *(f32x2*)(dst + ((r & 3) + 8 * (r >> 2)) * D4_PITCH) = f32x2{acc2[0][r], acc2[1][r]};
                        if (lh == 0) {
#pragma unroll
                            for (int r = 12; r < 15; ++r)
                                *(f32x2*)(dst + ((r & 3) + 8 * (r >> 2)) * D4_PITCH) = f32x2{acc2[0][r], acc2[1][r]};
                        }
                    }
                }
                __syncthreads();
                // ---- combine: thread j adds P row yp into its running output rows
                float s[3][3];
                if (row_ok) {
                    const float* q = buf + 1 + j;                        // column jl - 1
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int co = 0; co < 3; ++co) {
                            const int m0 = (dy * 3) * 3 + co;
                            s[dy][co] = (q[m0 * D4_PITCH] + q[(m0 + 3) * D4_PITCH + 1]) + q[(m0 + 6) * D4_PITCH + 2];
                        }
                } else {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int co = 0; co < 3; ++co) s[dy][co] = 0.f;
                }
                float fin[3];
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    fin[co] = Oa[co] + s[2][co];
                    Oa[co] = Ob[co] + s[1][co];
                    Ob[co] = s[0][co];
                }
                const int yo = yp - 1;
                if (yo >= 2 * r0 && yo < 2 * r1) {                       // (uniform) output row yo is complete
                    float e = 0.f;
                    if (own) {
                        const float rc[3] = {vad_tanh(fin[0] + c3b0), vad_tanh(fin[1] + c3b1), vad_tanh(fin[2] + c3b2)};
                        const size_t o = (size_t)n * 3 * plane + (size_t)yo * W2 + ox;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float xv = XU8 ? vad_norm_u8(xraw[c]) : __uint_as_float(xraw[c]);
                            const float d = xv - rc[c];
                            e += d * d;
                            if (p.recon) p.recon[o + c * plane] = rc[c];
                        }
                        if (p.errmap) p.errmap[(size_t)n * plane + (size_t)yo * W2 + ox] = e / 3.0f;
                    }
                    if (yo + 1 < 2 * r1) load_x(yo + 1);
                    const float ws = d4_wave_sum(e);
                    if (lane == 0)
                        p.partials[(size_t)n * ((size_t)H2 * p.nstrips * 4) + ((size_t)yo * p.nstrips + strip) * 4 + wave] = ws;
                }
                ++phase;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fr[ks] = nf[ks];
        }
    }
}

int d4_num_cus() {
    static std::atomic<int> cached{0};
    int ncu = cached.load(std::memory_order_relaxed);
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
        cached = ncu;
    }
    return ncu;
}
}  // namespace

static std::atomic<int> g_vad_dec4_band{0};     // debug: rows per band (0 = chosen from the batch size)
extern "C" int vad_debug_set_dec4_band(int rows) { g_vad_dec4_band = rows; return VAD_OK; }

// strips of the output width: one up to 256 columns, else 254 + 252 k
static int d4_strips(int w_in) { return w_in <= 128 ? 1 : 1 + (2 * w_in - 254 + 251) / 252; }

extern "C" int vad_dec4_score_partials(int h2, int w2) {
    if (h2 <= 0 || w2 <= 0 || (h2 & 1) || (w2 & 1)) return vad_fail(VAD_ERR_ARG, "dec4_score_partials: bad size %dx%d", h2, w2);
    return h2 * d4_strips(w2 / 2) * 4;
}

int vad_dec4_score_fmt(const float* in, const float* wt_packed, const float* bt, const float* w2_gemm, const float* bias3,
                       const void* x, int fmt, float* partials, float* recon, float* errmap, int n, int h, int w, void* stream) {
    VAD_REQUIRE(in && wt_packed && bt && w2_gemm && bias3 && x && partials, "dec4_score: null pointer");
    VAD_REQUIRE(fmt == VAD_X_F32_NCHW || fmt == VAD_X_U8_NHWC, "dec4_score: bad input format %d", fmt);
    VAD_REQUIRE(n > 0 && h > 0 && w > 0 && w % 8 == 0, "dec4_score: bad shape n=%d %dx%d (W must be a positive multiple of 8)", n, h, w);
    VAD_REQUIRE((long long)h * w * 32 * 4 < (1ll << 31), "dec4_score: frame %dx%d too large for 32-bit offsets", h, w);
    VAD_REQUIRE(((uintptr_t)wt_packed & 15) == 0 && ((uintptr_t)in & 15) == 0, "dec4_score: weights and activations must be 16-B aligned");
    Dec4P p{};
    p.in = in; p.wt = wt_packed; p.bt = bt; p.w2 = w2_gemm; p.b3 = bias3; p.x = x;
    p.partials = partials; p.recon = recon; p.errmap = errmap;
    p.n = n; p.H = h; p.W = w;
    p.nstrips = d4_strips(w);
    // Band height: the y halo costs one input row per band end ((R + 1) / R of the matrix work), but the grid needs about
    // two work-groups per CU.  The result does not depend on it (every output pixel's sums are ordered by pixel only).
    static std::atomic<int> per_cu_cached{0};
    int per_cu = per_cu_cached.load(std::memory_order_relaxed);
    if (!per_cu) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dec4_score_kernel<false>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        per_cu_cached = per_cu;
    }
    const long long slots = (long long)d4_num_cus() * per_cu;
    int R = g_vad_dec4_band.load(std::memory_order_relaxed);
    if (R <= 0) {
        R = 16;
        while (R > 1 && (long long)n * p.nstrips * ((h + R - 1) / R) < 2 * slots) R >>= 1;
    }
    if (R > h) R = h;
    p.R = R;
    p.nbands = (h + R - 1) / R;
    const long long items = (long long)n * p.nbands * p.nstrips;
    VAD_REQUIRE(items < (1ll << 31), "dec4_score: %lld work items out of range", items);
    p.nitems = (unsigned)items;
    const unsigned grid = (unsigned)(items < slots ? items : slots);
    if (fmt == VAD_X_U8_NHWC) hipLaunchKernelGGL(dec4_score_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(dec4_score_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_dec4_score(const float* in_nhwc, const float* wt_packed, const float* bt, const float* w2_gemm,
                              const float* bias3, const float* x_nchw, float* partials, float* recon_nchw, float* errmap,
                              int n, int h, int w, void* stream) {
    return vad_dec4_score_fmt(in_nhwc, wt_packed, bt, w2_gemm, bias3, x_nchw, VAD_X_F32_NCHW, partials, recon_nchw, errmap, n, h, w, stream);
}
