// Winograd F(2x2, 3x3) form of the 3x3 convolution on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32), gfx950.
// OPT-IN arithmetic (model.precision = "winograd"): every multiply and add is fp32, but a 2x2 block of outputs is computed from
// 16 products per input channel instead of 36, so the results are NOT bit-identical to the direct kernels of conv_pkernel.h
// (they differ by fp32 rounding, ~1e-6 relative on a layer's outputs); `value` of bench.py is always the direct path.
//
// Reference ops restated: nn.Conv2d(k3,p1)+BatchNorm2d(eval, folded)+LeakyReLU/ReLU(+MaxPool2d) for the layers with
// cin = cout >= 64 or cin >= 32 (models/autoencoder.py:49-79,103-128; models/video_autoencoder.py:197-215).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A,   d = 4x4 input patch, g = 3x3 filter, Y = 2x2 outputs (Lavin & Gray)
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
// The 16 element-wise products, summed over input channels, are 16 independent GEMMs
//   M_f[tile][cout] = sum_cin V_f[tile][cin] * U_f[cin][cout],   f = (fr, fc) = 4 x 4 "frequencies"
// with U = G g G^T packed on the host (pack.cpp, [16][cin/8][cout][8]) and V = B^T d B computed on the fly.
//
// Work split.  A work-group (4 waves) owns 4 x 8 Winograd tiles = 8 rows x 16 columns of outputs and 32*NT output channels and
// walks the frames of the batch (persistent, like conv3x3_mfma_pkernel).  Wave r owns frequency ROW fr = r for all 32 tiles:
//   * its A operands need only TWO of a patch's four rows (fr 0: d0-d2, 1: d1+d2, 2: d2-d1, 3: d1-d3): per 4 channels 8
//     ds_read_b128 + 16 row adds + 16 column adds give the 16 A values (4 fc x 4 channels) of 16*NT MFMAs - the transform is
//     ~6 % of the wave's matrix-pipe time at NT = 2 (fp32 VALU work shares the pipe with the fp32 MFMAs);
//   * its accumulators are 4 fc x NT tiles x 16 registers = 128 registers at NT = 2;
//   * the output transform's column half (over fc) happens in registers; the row half (over fr = over the four waves) goes
//     through LDS once per tile: 64 KB that alias the input tile, XOR-swizzled 16-byte slots, conflict-free both ways.  Wave w
//     then finishes tile row w: bias, activation, (the 2x2 outputs of a tile ARE one MaxPool2d window) pooling, stores.
// B operands come straight from L2 (buffer_load_dwordx4, a ring of NB register sets, PB steps ahead) as in the direct kernels.
#include <atomic>

#include "vad_common.h"

#define MFMA32W(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

struct ConvWP {
    const float* in;  long long in_fs;     // NHWC fp32 [n][h][w][cin_a], frame stride in floats: input channels [0, cin_a)
    const float* in2; long long in2_fs;    // channels [cin_a, cin) (the h half of a ConvLSTM step; NULL: the layer has cin_a channels)
    int cin_a;
    const float* w;                        // vad_pack_conv3x3_wino: [16][cin/8][cout][8]
    const float* bias;                     // [cout] (BatchNorm folded)
    float* out;       long long out_fs;    // NHWC [n][h or h/2][w or w/2][cout]
    int n, h, w_, cin, cout;
    int tiles_x, tiles_y, cblocks;
    // POOL == 2 (ConvLSTM cell fused into the epilogue): cout = 4 * hid gate pre-activations, `out` = h' [n][h][w][hid];
    const float* c_prev; float* c_out; int hid;   // c_prev NULL = zero initial state; dense [n][h][w][hid]
};

// POOL: 0 = plain, 1 = MaxPool2d(2,2) fused (a tile's 2x2 outputs are one window), 2 = ConvLSTM cell fused (NT == 2): a work-group
// owns 16 hidden channels - N-tile 0 = their i | f gates (lanes 0-15 | 16-31), N-tile 1 = g | o - so that after the row half of the
// output transform the lane pair (li, li ^ 16) holds all four gates of a hidden channel; the pair swaps half of its values
// (ds_swizzle, xor 16) and each lane finishes the cells of one column parity: models/video_autoencoder.py:73-83, vad_lstm_cell.
template <int NT, int POOL, int ACT>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_pkernel(ConvWP p) {
    static_assert(POOL != 2 || NT == 2, "fused ConvLSTM cell: two N-tiles (i|f and g|o) per wave");
    constexpr int CK = 32, LW = 18, LH = 10, PS = CK + 4, NPIX = LW * LH;
    constexpr int TOT = NPIX * (CK / 4), NPF = (TOT + 255) / 256;
    constexpr int XF = 4 * 2 * NT * 64 * 16;                        // exchange floats
    // TWO input tiles (chunk c is read from tile c & 1 while chunk c + 1 is written to the other): one barrier per chunk instead
    // of two.  The exchange records alias both.
    constexpr int TILE_F = NPIX * PS;
    constexpr int SMEM = (2 * TILE_F > XF) ? 2 * TILE_F : XF;
    constexpr int NS = 16;                                          // (k-group, fc) steps per 32-channel chunk
    constexpr int PB = 1, NB = 2;                                   // B ring: PB steps ahead, NB register sets (NS % NB == 0)
    __shared__ __attribute__((aligned(16))) float smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

    // ---- geometry of this work-group (fixed for its whole life)
    const unsigned per_frame = (unsigned)(p.tiles_x * p.tiles_y * p.cblocks);
    unsigned L = vad_xcd_remap(blockIdx.x % per_frame, per_frame);
    const int fg = blockIdx.x / per_frame, fgroups = gridDim.x / per_frame;
    const int cb = L % p.cblocks; L /= p.cblocks;
    const int x0 = (L % p.tiles_x) * 16, y0 = (L / p.tiles_x) * 8;
    const int H = p.h, W = p.w_;
    const int nch_a = p.cin_a / CK;
    const int nch = (p.in2 ? p.cin : p.cin_a) / CK;     // in2 == NULL with cin > cin_a: the missing channels are zeros (ConvLSTM at t = 0), skipped

    // ---- staging slots (input tile + 1-pixel halo, NHWC chunk of 32 channels): slot i = float4 number tid + 256 i
    unsigned svo[NPF];
    const int c4 = tid & 7, pix0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int pix = pix0 + 32 * i;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        const bool ok = pix < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
        svo[i] = ok ? (unsigned)(__mul24(__mul24(gy, W) + gx, p.cin_a) + c4 * 4) * 4u : VAD_OOB;
    }
    // both sources of a two-source launch have the same channel count (host-checked), so svo serves both
    const unsigned in_bytes = (unsigned)(H * W) * (unsigned)p.cin_a * 4u;
    f32x4 pf[NPF];
#define ISSUE(n_, ch_)                                                                                   \
    {                                                                                                    \
        const float* src_ = ((ch_) < nch_a) ? p.in + (size_t)(n_) * p.in_fs + (ch_) * CK                 \
                                            : p.in2 + (size_t)(n_) * p.in2_fs + ((ch_) - nch_a) * CK;    \
        const unsigned skip_ = (unsigned)(((ch_) < nch_a) ? (ch_) : (ch_) - nch_a) * CK * 4u;            \
        const __amdgpu_buffer_rsrc_t r_ = vad_rsrc(src_, in_bytes - skip_);                              \
        _Pragma("unroll") for (int i_ = 0; i_ < NPF; ++i_) pf[i_] = vad_bload4(r_, svo[i_], 0);          \
    }

    // ---- A side: this wave's frequency row needs patch rows (ia, ib) combined as d[ia] + sgn * d[ib]
    const int tr = li >> 3, tc = li & 7;
    const int ia = (wave == 0) ? 0 : (wave == 2) ? 2 : 1;
    const int ib = (wave == 0) ? 2 : (wave == 1) ? 2 : (wave == 2) ? 1 : 3;
    const float sgn = (wave == 1) ? 1.f : -1.f;
    const int rowa = ((2 * tr + ia) * LW + 2 * tc) * PS + 4 * lh;
    const int rowb = ((2 * tr + ib) * LW + 2 * tc) * PS + 4 * lh;

    // ---- B side
    const unsigned wf = (unsigned)p.cout * 32u;                     // bytes per (frequency, 8-channel group) slab
    const unsigned ftap = (unsigned)(p.cin / 8) * wf;               // bytes per frequency
    const __amdgpu_buffer_rsrc_t rw = vad_rsrc(p.w, 16u * ftap);
    unsigned wl[NT];
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (POOL == 2) ? (2 * nt + (li >> 4)) * p.hid + cb * 16 + (li & 15) : (cb * NT + nt) * 32 + li;
        wl[nt] = (unsigned)(wave * 4) * ftap + (unsigned)co * 32u + 16u * lh;
        bv[nt] = p.bias[co];
    }
    f32x4 b[NB][NT];
#define LOAD_B(buf, chunk, step)   /* step = kg * 4 + fc */                                              \
    {                                                                                                    \
        const unsigned soff_ = (unsigned)((step) & 3) * ftap + (unsigned)((chunk) * 4 + ((step) >> 2)) * wf; \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) b[buf][nt] = vad_bload4(rw, wl[nt], soff_);    \
    }

    // ---- epilogue geometry: wave w finishes tile row w; lane = (channel li, column half lh), kk = tile column 4 lh + kk
    const int ech = (POOL == 2) ? p.hid : p.cout;
    const int ow = (POOL == 1) ? (W >> 1) : W, oh = (POOL == 1) ? (H >> 1) : H;
    const int ey0 = (POOL == 1) ? (y0 >> 1) + wave : y0 + 2 * wave;
    const int ex0 = (POOL == 1) ? (x0 >> 1) + 4 * lh : x0 + 8 * lh;
    const unsigned erow = (unsigned)__mul24(ow, ech) * 4u, ecol = (unsigned)ech * 4u;
    const unsigned eoff = (POOL == 2) ? (unsigned)(__mul24(__mul24(ey0, ow) + ex0 + (li >> 4), ech) + cb * 16 + (li & 15)) * 4u   // this lane's column parity
                                      : (unsigned)(__mul24(__mul24(ey0, ow) + ex0, ech) + cb * NT * 32 + li) * 4u;
    const unsigned out_bytes = (unsigned)__mul24(oh, ow) * (unsigned)ech * 4u;
    const bool full_tile = (y0 + 8 <= H) && (x0 + 16 <= W);
    const int sw = (lane >> 2) & 3;                                 // 16-byte slot swizzle of the exchange records

    int n = fg;
    if (n >= p.n) return;
    ISSUE(n, 0);
#pragma unroll
    for (int s0 = 0; s0 < PB; ++s0) LOAD_B(s0, 0, s0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(bv[nt]));

    // One 32-channel chunk: publish the staged tile, request the next stage, 16 (k-group, fc) steps.  FIRST (a tile's first chunk):
    // the first MFMA of every accumulator chain takes C = 0 as an inline constant - no accumulator is ever zeroed (128 v_mov per tile
    // on the pipe the MFMAs use).
    f32x16 acc[4][NT];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // A operands of one 8-channel group: the two patch rows this wave's frequency row needs are REQUESTED one group ahead (8
    // ds_read_b128 into `raw`, issued in front of the previous group's 16*NT MFMAs, so their latency passes under ~2,000 cycles of
    // matrix work) and TRANSFORMED between two groups' MFMAs (32 VALU instructions: the row combination d[ia] + sgn d[ib], then the
    // column transform; fp32 VALU work does not overlap with fp32 MFMAs anyway).
#define LOAD_RAW(kg)                                                                                     \
    {                                                                                                    \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                               \
            raw[0][j_] = *(const f32x4*)&tile_[rowa + j_ * PS + (kg) * 8];                               \
            raw[1][j_] = *(const f32x4*)&tile_[rowb + j_ * PS + (kg) * 8];                               \
        }                                                                                                \
    }
#define TRANSFORM_V()                                                                                    \
    {                                                                                                    \
        f32x4 t_[4];                                                                                     \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                 \
            _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) t_[j_][e_] = __builtin_fmaf(sgn, raw[1][j_][e_], raw[0][j_][e_]); \
        v[0] = t_[0] - t_[2];                                                                            \
        v[1] = t_[1] + t_[2];                                                                            \
        v[2] = t_[2] - t_[1];                                                                            \
        v[3] = t_[1] - t_[3];                                                                            \
    }
#define CHUNK(FIRST, ch, n_, nn_, has_next_)                                                             \
    {                                                                                                    \
        /* the tile this chunk is written to was last read two chunks ago (a barrier lies between), or held exchange records \
           whose readers have passed the barrier that ends the epilogue */                               \
        float* tile_ = smem + (((ch) & 1) ? TILE_F : 0);                                                 \
        _Pragma("unroll") for (int i = 0; i < NPF; ++i)                                                  \
            if (pix0 + 32 * i < NPIX) *(f32x4*)&tile_[(pix0 + 32 * i) * PS + c4 * 4] = pf[i];            \
        __syncthreads();                                                                                 \
        f32x4 raw[2][4], v[4];                 /* v[fc][channel j of this lane's quad] */                \
        LOAD_RAW(0);                                                                                     \
        if ((ch) + 1 < nch) { ISSUE(n_, (ch) + 1); }                                                     \
        else if (has_next_) { ISSUE(nn_, 0); }                                                           \
        TRANSFORM_V();                                                                                   \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg) {                                               \
            if (kg + 1 < 4) { LOAD_RAW(kg + 1); }                                                        \
            _Pragma("unroll") for (int fc = 0; fc < 4; ++fc) {                                           \
                const int s_ = kg * 4 + fc, bcur = s_ % NB, bnxt = (s_ + PB) % NB;                       \
                if (s_ + PB < NS) { LOAD_B(bnxt, ch, s_ + PB); }                                         \
                else if ((ch) + 1 < nch) { LOAD_B(bnxt, (ch) + 1, s_ + PB - NS); }                       \
                __builtin_amdgcn_sched_barrier(0);                                                       \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                            \
                    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                    \
                        acc[fc][nt] = MFMA32W(v[fc][j], b[bcur][nt][j], ((FIRST) && kg == 0 && j == 0) ? zero16 : acc[fc][nt]); \
                __builtin_amdgcn_sched_barrier(0);                                                       \
            }                                                                                            \
            if (kg + 1 < 4) { TRANSFORM_V(); }                                                           \
        }                                                                                                \
    }

    while (true) {
        const int nn = n + fgroups;
        const bool has_next = nn < p.n;
        CHUNK(1, 0, n, nn, has_next);
        for (int ch = 1; ch < nch; ++ch) CHUNK(0, ch, n, nn, has_next);

        // ---- output transform, column half (over fc) in registers: P0 = M0 + M1 + M2, P1 = M1 - M2 - M3
        f32x16 P[2][NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            P[0][nt] = acc[0][nt] + acc[1][nt] + acc[2][nt];
            P[1][nt] = acc[1][nt] - acc[2][nt] - acc[3][nt];
        }
        // next frame's first B fragments go out BEFORE this frame's stores (vmcnt retires in order)
        if (has_next) {
#pragma unroll
            for (int s0 = 0; s0 < PB; ++s0) LOAD_B(s0, 0, s0);
        }
        __syncthreads();                           // every wave is done reading the input tile: its space takes the records
        // record of (frequency row r, j, nt, lane): 16 floats = the lane's 16 tiles, as four 16-byte slots q = tile row
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(f32x4*)&smem[((((wave * 2 + j) * NT + nt) * 64 + lane) * 4 + (q ^ sw)) * 4] =
                        f32x4{P[j][nt][4 * q], P[j][nt][4 * q + 1], P[j][nt][4 * q + 2], P[j][nt][4 * q + 3]};
        __syncthreads();
        // ---- row half (over the four waves' frequency rows) for tile row `wave`: Y0 = P0 + P1 + P2, Y1 = P1 - P2 - P3
        const __amdgpu_buffer_rsrc_t ro = vad_rsrc(p.out + (size_t)n * p.out_fs, out_bytes);
        f32x4 yy[NT][2][2];                        // [nt][dy][dx], components kk = tile column 4 lh + kk
        float cpv[POOL == 2 ? 8 : 1];              // fused cell: previous cell state of this lane's 8 pixels (dy, kk), requested before the records are read
        if constexpr (POOL == 2) {
            const __amdgpu_buffer_rsrc_t rc = vad_rsrc(p.c_prev ? p.c_prev + (size_t)n * (out_bytes / 4) : p.c_out, p.c_prev ? out_bytes : 0u);   // zero-sized -> 0
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bool ok = full_tile || ((ey0 + dy) < oh && (ex0 + 2 * kk + (li >> 4)) < ow);
                    cpv[dy * 4 + kk] = vad_bload1(rc, ok ? eoff : VAD_OOB, dy * erow + 2 * kk * ecol);
                }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 pr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[r] = *(const f32x4*)&smem[((((r * 2 + j) * NT + nt) * 64 + lane) * 4 + (wave ^ sw)) * 4];
                yy[nt][0][j] = pr[0] + pr[1] + pr[2];
                yy[nt][1][j] = pr[1] - pr[2] - pr[3];
            }
        if constexpr (POOL == 2) {
            // lanes li < 16 hold (i, g), lanes li >= 16 hold (f, o) of hidden channel cb*16 + (li & 15), both column parities dx.  The
            // low lane keeps dx = 0 and hands over its dx = 1 values, the high lane the other way round: afterwards each lane has all
            // four gates of ITS parity's 8 pixels.
            const bool hi = (li >> 4) != 0;
            const __amdgpu_buffer_rsrc_t rco = vad_rsrc(p.c_out + (size_t)n * (out_bytes / 4), out_bytes);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    float own[2], got[2];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const float keep = hi ? yy[nt][dy][1][kk] : yy[nt][dy][0][kk];
                        const float give = hi ? yy[nt][dy][0][kk] : yy[nt][dy][1][kk];
                        own[nt] = keep + bv[nt];
                        got[nt] = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, give + bv[nt]), 0x401f));   // lane ^ 16
                    }
                    // low lane: own = (i, g), got = (f, o); high lane: own = (f, o), got = (i, g)
                    const float zi = hi ? got[0] : own[0], zf = hi ? own[0] : got[0], zg = hi ? got[1] : own[1], zo = hi ? own[1] : got[1];
                    float cn, hn;
                    vad_lstm_cell(zi, zf, zg, zo, cpv[dy * 4 + kk], cn, hn);
                    const bool ok = full_tile || ((ey0 + dy) < oh && (ex0 + 2 * kk + (li >> 4)) < ow);
                    const unsigned so = dy * erow + 2 * kk * ecol;
                    vad_bstore1(cn, rco, ok ? eoff : VAD_OOB, so);
                    vad_bstore1(hn, ro, ok ? eoff : VAD_OOB, so);
                }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (POOL == 1) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const bool ok = full_tile || (ey0 < oh && (ex0 + kk) < ow);
                        const float m = fmaxf(fmaxf(yy[nt][0][0][kk], yy[nt][0][1][kk]), fmaxf(yy[nt][1][0][kk], yy[nt][1][1][kk]));
                        vad_bstore1(vad_act(m + bv[nt], ACT), ro, ok ? eoff : VAD_OOB, kk * ecol + nt * 128u);
                    }
                } else {
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx) {
                                const bool ok = full_tile || ((ey0 + dy) < oh && (ex0 + 2 * kk + dx) < ow);
                                vad_bstore1(vad_act(yy[nt][dy][dx][kk] + bv[nt], ACT), ro, ok ? eoff : VAD_OOB, dy * erow + (2 * kk + dx) * ecol + nt * 128u);
                            }
                }
            }
        }
        if (!has_next) break;
        n = nn;
        __syncthreads();                           // every wave has read its records: the next tile may overwrite them
    }
#undef ISSUE
#undef LOAD_B
#undef CHUNK
#undef LOAD_RAW
#undef TRANSFORM_V
}

// ------------------------------------------------------------------------------------------------ host side
static int wino_num_cus() {
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

template <int NT, int POOL, int ACT>
static void launch_wino(const ConvWP& p, hipStream_t s) {
    static std::atomic<unsigned> cap_{0};
    unsigned cap = cap_.load(std::memory_order_relaxed);
    if (!cap) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv3x3_wino_pkernel<NT, POOL, ACT>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        cap_ = cap = (unsigned)wino_num_cus() * (unsigned)per_cu;
    }
    const unsigned per_frame = (unsigned)(p.tiles_x * p.tiles_y * p.cblocks);
    unsigned fgroups = cap / per_frame;
    if (fgroups < 1) fgroups = 1;
    if (fgroups > (unsigned)p.n) fgroups = (unsigned)p.n;
    hipLaunchKernelGGL((conv3x3_wino_pkernel<NT, POOL, ACT>), dim3(per_frame * fgroups), dim3(256), 0, s, p);
}

template <int NT>
static int launch_wino_nt(const ConvWP& p, int act, int pool, hipStream_t s) {
#define W_(P_, A_) launch_wino<NT, P_, A_>(p, s)
    if (pool) { if (act == VAD_ACT_LEAKY) W_(1, VAD_ACT_LEAKY); else if (act == VAD_ACT_RELU) W_(1, VAD_ACT_RELU); else W_(1, VAD_ACT_NONE); }
    else { if (act == VAD_ACT_LEAKY) W_(0, VAD_ACT_LEAKY); else if (act == VAD_ACT_RELU) W_(0, VAD_ACT_RELU); else W_(0, VAD_ACT_NONE); }
#undef W_
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

#define TRYW(call) do { int rc_ = (call); if (rc_ != VAD_OK) return rc_; } while (0)

// include/vad_hip.h: vad_conv3x3_wino
extern "C" int vad_conv3x3_wino(const float* in, long long in_fs, const float* w, const float* bias, float* out, long long out_fs,
                                int n, int h, int wd, int cin, int cout, int act, int pool, void* stream) {
    VAD_REQUIRE(in && w && bias && out, "conv3x3_wino: null pointer");
    VAD_REQUIRE(n > 0 && h > 0 && wd > 0, "conv3x3_wino: bad shape n=%d h=%d w=%d", n, h, wd);
    VAD_REQUIRE(cin % 32 == 0 && cout % 32 == 0 && cin > 0 && cout > 0, "conv3x3_wino: cin=%d cout=%d must be positive multiples of 32", cin, cout);
    VAD_REQUIRE(!pool || (h % 2 == 0 && wd % 2 == 0), "conv3x3_wino: pooling needs even H, W (got %dx%d)", h, wd);   // (odd sizes: the last 2x2 tiles are partial)
    VAD_REQUIRE(act >= 0 && act <= 2, "conv3x3_wino: bad act %d", act);
    const long long chmax = cin > cout ? cin : cout;
    VAD_REQUIRE((long long)h * wd * chmax * 4 < (1ll << 31) && 16ll * cin * cout * 4 < (1ll << 31),
                "conv3x3_wino: frame %dx%dx%lld or weights %dx%d too large for 32-bit offsets", h, wd, chmax, cin, cout);
    ConvWP p{};
    p.in = in; p.in_fs = in_fs ? in_fs : (long long)h * wd * cin; p.cin_a = cin;
    p.w = w; p.bias = bias; p.out = out;
    const int ho = pool ? h / 2 : h, wo = pool ? wd / 2 : wd;
    p.out_fs = out_fs ? out_fs : (long long)ho * wo * cout;
    p.n = n; p.h = h; p.w_ = wd; p.cin = cin; p.cout = cout;
    p.tiles_x = (wd + 15) / 16; p.tiles_y = (h + 7) / 8;
    // Two N-tiles per wave (64 channels per work-group) halve the input transform per MFMA; when that grid would leave CUs idle
    // (the reference's own call sizes) one N-tile per wave gives twice the work-groups and half the serial K loop per wave.  The
    // accumulation order of an output does not depend on NT: the same bits either way.
    const bool two = cout % 64 == 0 && (long long)n * p.tiles_x * p.tiles_y * (cout / 64) >= wino_num_cus();
    p.cblocks = cout / (two ? 64 : 32);
    VAD_REQUIRE((long long)p.tiles_x * p.tiles_y * p.cblocks < (1ll << 24), "conv3x3_wino: %d x %d x %d work-group positions out of range", p.tiles_x, p.tiles_y, p.cblocks);
    return two ? launch_wino_nt<2>(p, act, pool, (hipStream_t)stream) : launch_wino_nt<1>(p, act, pool, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ ConvLSTM step, Winograd form
// ConvLSTMCell (reference models/video_autoencoder.py:54-85): cat([x, h]) -> Conv3x3(-> 4*hid, bias) -> i, f, g, o -> cell.  The
// gate convolution runs as ONE Winograd launch over the two sources (x: channels [0, cin_x), h: [cin_x, cin_x + hid); at t = 0
// the h half is zeros and skipped) writing the pre-activations z [n][h][w][4*hid] (bias added); the cell is a pointwise pass
// (vad_lstm_cell of vad_common.h: the one every fused form uses).  z costs 2 x 2 KB per pixel of HBM traffic per step: ~10 % of
// the step at 64 clips - the price of keeping every accumulator of a hidden channel's four gates out of one wave.
struct CellP {
    const float* z; const float* c_prev; float* c_out; float* h_out;
    long long h_fs;        // frame stride of h_out in floats
    int hw, hid;           // pixels per frame, hidden channels
    long long total;       // n * hw * hid / 4
};

__global__ __launch_bounds__(256) void wino_lstm_cell_kernel(CellP p) {
    const int hq = p.hid >> 2;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < p.total; idx += (long long)gridDim.x * 256) {
        const long long pix = idx / hq;
        const int c4 = (int)(idx - pix * hq) * 4;
        const long long n = pix / p.hw;
        const int q = (int)(pix - n * p.hw);
        const float* z = p.z + pix * 4 * p.hid + c4;
        const f32x4 zi = *(const f32x4*)z, zf = *(const f32x4*)(z + p.hid), zg = *(const f32x4*)(z + 2 * p.hid), zo = *(const f32x4*)(z + 3 * p.hid);
        const f32x4 cp = p.c_prev ? *(const f32x4*)(p.c_prev + pix * p.hid + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 cn, hn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float c1, h1;
            vad_lstm_cell(zi[e], zf[e], zg[e], zo[e], cp[e], c1, h1);
            cn[e] = c1; hn[e] = h1;
        }
        *(f32x4*)(p.c_out + pix * p.hid + c4) = cn;
        *(f32x4*)(p.h_out + n * p.h_fs + (long long)q * p.hid + c4) = hn;
    }
}

// include/vad_hip.h: vad_convlstm_step_wino.  z_ws: n*h*w*4*hid floats of scratch.
extern "C" int vad_convlstm_step_wino(const float* x, long long x_fs, const float* h_prev, long long h_prev_fs, const float* c_prev,
                                      const float* w, const float* bias, float* h_out, long long h_out_fs, float* c_out, float* z_ws,
                                      int n, int h, int wd, int cin_x, int hid, void* stream) {
    VAD_REQUIRE(x && w && bias && h_out && c_out && z_ws, "convlstm_step_wino: null pointer");
    VAD_REQUIRE((h_prev == nullptr) == (c_prev == nullptr), "convlstm_step_wino: h_prev and c_prev must both be given or both NULL");
    VAD_REQUIRE(n > 0 && h > 0 && wd > 0, "convlstm_step_wino: bad shape %d x %dx%d", n, h, wd);
    VAD_REQUIRE(cin_x > 0 && hid > 0 && hid % 16 == 0 && cin_x % 32 == 0 && cin_x == hid,
                "convlstm_step_wino: cin_x=%d and hid=%d must be equal multiples of 32 (one set of staging offsets serves both sources)", cin_x, hid);
    const int cin = cin_x + hid, cout = 4 * hid;
    VAD_REQUIRE((long long)h * wd * cout * 4 < (1ll << 31) && 16ll * cin * cout * 4 < (1ll << 31), "convlstm_step_wino: frame or weights too large for 32-bit offsets");
    ConvWP p{};
    p.in = x; p.in_fs = x_fs ? x_fs : (long long)h * wd * cin_x; p.cin_a = cin_x;
    p.in2 = h_prev; p.in2_fs = h_prev_fs ? h_prev_fs : (long long)h * wd * hid;
    p.w = w; p.bias = bias; p.out = z_ws; p.out_fs = (long long)h * wd * cout;
    p.n = n; p.h = h; p.w_ = wd; p.cin = cin; p.cout = cout;
    p.tiles_x = (wd + 15) / 16; p.tiles_y = (h + 7) / 8;
    const bool two = (long long)n * p.tiles_x * p.tiles_y * (cout / 64) >= wino_num_cus();     // see vad_conv3x3_wino
    if (two) {
        // large launch groups: the cell in the convolution's epilogue - z never exists (2 x 2 KB per pixel and step of HBM traffic
        // and one launch less per step).  Same values as the two-launch form below: the same accumulation order, the same
        // bias add, the same vad_lstm_cell.
        p.cblocks = hid / 16; p.hid = hid;
        p.out = h_out; p.out_fs = h_out_fs ? h_out_fs : (long long)h * wd * hid;
        p.c_prev = c_prev; p.c_out = c_out;
        VAD_REQUIRE((long long)p.tiles_x * p.tiles_y * p.cblocks < (1ll << 24), "convlstm_step_wino: grid out of range");
        launch_wino<2, 2, VAD_ACT_NONE>(p, (hipStream_t)stream);
        VAD_LAUNCH_CHECK();
        return VAD_OK;
    }
    p.cblocks = cout / 32;
    TRYW(launch_wino_nt<1>(p, VAD_ACT_NONE, 0, (hipStream_t)stream));
    CellP c{};
    c.z = z_ws; c.c_prev = c_prev; c.c_out = c_out; c.h_out = h_out;
    c.h_fs = h_out_fs ? h_out_fs : (long long)h * wd * hid;
    c.hw = h * wd; c.hid = hid; c.total = (long long)n * h * wd * (hid / 4);
    const long long blocks = (c.total + 255) / 256;
    hipLaunchKernelGGL(wino_lstm_cell_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, c);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}
