// Implicit-GEMM convolutions on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32), gfx950.
//
// Orientation used by every kernel here: GEMM M = pixels, N = output channels, K = (tap, cin).
//   A (pixels x k)   : NHWC input tile staged in LDS with a 1-pixel halo, padded pixel stride
//                      (CK+4 floats, so consecutive pixels land on different 16-B bank slots);
//                      lane (i = lane&31, h = lane>>5) reads 4 consecutive channels with one
//                      ds_read_b128 and feeds them to 4 consecutive MFMAs (k-pair j = {4h+j}).
//   B (k x cout)     : weights pre-packed on the host as [tap][cin/8][cout][8] so that the same
//                      lane reads its 4 k-values for its cout with ONE coalesced 16-B global load
//                      (a wave reads 1 KiB contiguous); weights stay L2 resident, no LDS copy.
//   C/D              : column = lane&31 = cout, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = pixel.
// The 32 pixels of an M-tile are 2 rows x 16 columns ordered as 8 pooling windows x 4 positions
// (i -> window i>>2, dy (i>>1)&1, dx i&1), so the 4 registers reg&3 of one lane are exactly one
// MaxPool2d(2,2) window: pooling is 3 v_max in registers, and every store instruction writes two
// full 128-B channel rows.
//
// Reference ops restated: nn.Conv2d(k3,p1)+BatchNorm2d(eval, folded)+LeakyReLU/ReLU(+MaxPool2d)
// (models/autoencoder.py:38-79,103-139; models/video_autoencoder.py:191-215), ConvLSTMCell
// (models/video_autoencoder.py:54-85), nn.ConvTranspose2d(k2,s2)+BN+ReLU
// (models/autoencoder.py:104-131; models/video_autoencoder.py:244-256).
#include <atomic>
#include <type_traits>
#include "vad_common.h"

enum { MODE_PLAIN = 0, MODE_POOL = 1, MODE_LSTM = 2 };

struct Conv3P {
    const float* in;  long long in_fs;  int cin_a;   // channels [0, cin_a)
    const float* in2; long long in2_fs;              // channels [cin_a, cin); NULL = zeros
    const float* w;   const float* bias;
    float* out;       long long out_fs;
    const float* c_prev; float* c_out;               // MODE_LSTM
    const float* zx; long long zx_fs;                // small-grid ConvLSTM kernel: bias + the x half of the gate pre-activations
                                                     // [n][h][w][4*hid], computed ahead for all time steps (NULL: computed here)
    const float* w0; const float* b0;                // FUSE_C3: first-layer (3->32) weights [28][32], bias
    int xu8;                                         // FUSE_C3: `in` is uint8 NHWC [N,H,W,3] (normalised in the kernel)
    int n, h, w_, cin, cout, hid;
    int tiles_x, tiles_y, cblocks;
    unsigned nblocks;
    unsigned long long* dbg;                         // VAD_STAMPS diagnostic build only
    int stagger;                                     // persistent kernel: start delay of the upper half of the grid (x 8128 cycles)
    float* stats;                                    // nullable; persistent MODE_PLAIN / VAD_ACT_NONE kernels (training forward): BatchNorm
                                                     // partial sums [row][2][cout], row = frame group * tiles + tile, shifted by the bias
};

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// K order inside an 8-channel group, shared by every exact-fp32 3x3 kernel so that they stay bit-identical to each other:
// a 32x32x2 step j multiplies channels (j, 4+j) (its two lane halves), and the steps run in the order j = 0, 2, 1, 3, i.e.
// channels 0,4,2,6 | 1,5,3,7.  That is the one order both this instruction AND v_mfma_f32_16x16x4_f32 can feed with plain
// contiguous reads: the 16x16x4 kernel's four k lanes read the adjacent channel pairs (0,1) (4,5) (2,3) (6,7) and issue
// (0,4,2,6) then (1,5,3,7) (conv_small.h) - no per-lane element selects, no duplicated weight bytes through L1.
#define VAD_KORDER(jj) (((jj) == 1) ? 2 : ((jj) == 2) ? 1 : (jj))

template <int CK, int MT, int NT, int WM, int WN, int MODE, int ACT, int FUSE_C3 = 0>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(Conv3P p) {
    static_assert(WM * WN == 4, "4 waves per work-group");
    static_assert(!FUSE_C3 || CK == 32, "fused first layer produces exactly one 32-channel chunk");
    static_assert(MODE != MODE_LSTM || NT == 4, "LSTM mode: one N-tile per gate");
    constexpr int TH = 2 * MT * WM, LH = TH + 2, LW = 18, PS = CK + 4;
    __shared__ __attribute__((aligned(16))) float tile[LH * LW * PS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    unsigned L = vad_xcd_remap(blockIdx.x, p.nblocks);
    const int cb = L % p.cblocks; L /= p.cblocks;
    const int tx = L % p.tiles_x; L /= p.tiles_x;
    const int ty = L % p.tiles_y;
    const int n = L / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * 16;

    // per-lane pixel inside an M-tile
    const int prow = (li >> 1) & 1, pcol = 2 * (li >> 2) + (li & 1);
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        abase[mt] = ((2 * (wm * MT + mt) + prow) * LW + pcol) * PS + 4 * lh;

    // per-lane weight row pointers and bias
    const float* wp[NT];
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int co;
        if (MODE == MODE_LSTM) co = nt * p.hid + (cb * WN + wn) * 32 + li;
        else co = ((cb * WN + wn) * NT + nt) * 32 + li;
        wp[nt] = p.w + (size_t)co * 8 + 4 * lh;
        bv[nt] = p.bias[co];
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bv[nt];

    const int nch_a = p.cin_a / CK;
    const int nch = (p.in2 ? p.cin : p.cin_a) / CK;
    const size_t wstep = (size_t)p.cout * 8;           // floats per (tap, k8) slab
    const size_t wtap = (size_t)(p.cin / 8) * wstep;   // floats per tap

    // Software pipeline: fragments for step s+1 are loaded (A: ds_read_b128, B: global_load_dwordx4
    // from L2) before the MFMAs of step s issue; B also runs ahead across the chunk barrier.
    constexpr int NS = 9 * (CK / 8);                   // (tap, k8) steps per channel chunk
    static_assert(NS % 2 == 0, "double-buffer parity must be the same in every chunk");
    f32x4 a[2][MT], b[2][NT];
#define LOAD_B(buf, chunk, step)                                                                  \
    {                                                                                             \
        const size_t woff_ = (size_t)((step) / (CK / 8)) * wtap +                                 \
                             (size_t)((chunk) * (CK / 8) + (step) % (CK / 8)) * wstep;            \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) b[buf][nt] = *(const f32x4*)(wp[nt] + woff_); \
    }
#define LOAD_A(buf, step)                                                                         \
    {                                                                                             \
        const int toff_ = ((((step) / (CK / 8)) / 3) * LW + (((step) / (CK / 8)) % 3)) * PS +     \
                          ((step) % (CK / 8)) * 8;                                                \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) a[buf][mt] = *(const f32x4*)&tile[abase[mt] + toff_]; \
    }
    LOAD_B(0, 0, 0);

    for (int ch = 0; ch < nch; ++ch) {
        const float* src;
        int pstride, coff;
        if (ch < nch_a) { src = p.in + (size_t)n * p.in_fs; pstride = p.cin_a; coff = ch * CK; }
        else { src = p.in2 + (size_t)n * p.in2_fs; pstride = p.cin - p.cin_a; coff = ch * CK - p.cin_a; }

        __syncthreads();
        if constexpr (FUSE_C3) {
            // Fused first layer: Conv2d(3->32)+BN+LeakyReLU of the tile AND its halo is computed here from
            // the NCHW input planes (K = 27 padded to 28, 14 MFMAs per 32 pixels) straight into the LDS
            // tile the 32->32 convolution reads, so the 32-channel full-resolution map never touches HBM.
            constexpr int XH = LH + 2, XS = 24, NPIX = LH * LW, NT0 = (NPIX + 31) / 32;
            __shared__ float xin[3 * XH * XS];
            const float* xs = p.in + (size_t)n * 3 * p.h * p.w_;
            for (int idx = tid; idx < 3 * XH * 20; idx += 256) {
                const int lx = idx % 20, t = idx / 20, ly = t % XH, c = t / XH;
                const int gy = y0 - 2 + ly, gx = x0 - 2 + lx;
                float v = 0.f;
                if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_) v = xs[((size_t)c * p.h + gy) * p.w_ + gx];
                xin[(c * XH + ly) * XS + lx] = v;
            }
            __syncthreads();
            float b0[14];
            int koff[14];
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                b0[s] = p.w0[(s * 2 + lh) * 32 + li];
                const int k0 = 2 * s, k1 = 2 * s + 1;
                const int o0 = ((k0 / 9) * XH + (k0 % 9) / 3) * XS + (k0 % 3);
                const int o1 = (k1 < 27) ? ((k1 / 9) * XH + (k1 % 9) / 3) * XS + (k1 % 3) : 0;
                koff[s] = lh ? o1 : o0;
            }
            const float bias0 = p.b0[li];
            for (int t = wave; t < NT0; t += 4) {
                const int q = t * 32 + li, qc = q < NPIX ? q : NPIX - 1;
                const int abase0 = (qc / LW) * XS + (qc % LW);
                f32x16 c0;
#pragma unroll
                for (int r = 0; r < 16; ++r) c0[r] = bias0;
#pragma unroll
                for (int s = 0; s < 14; ++s) c0 = MFMA32(xin[abase0 + koff[s]], b0[s], c0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q2 = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (q2 < NPIX) {
                        const int gy = y0 - 1 + q2 / LW, gx = x0 - 1 + q2 % LW;
                        const bool inside = gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_;
                        tile[q2 * PS + li] = inside ? vad_act(c0[r], VAD_ACT_LEAKY) : 0.f;   // zero padding of conv #2
                    }
                }
            }
        } else {
            for (int idx = tid; idx < LH * LW * (CK / 4); idx += 256) {
                const int c4 = idx % (CK / 4), pix = idx / (CK / 4);
                const int ly = pix / LW, lx = pix - ly * LW;
                const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_)
                    v = *(const f32x4*)(src + ((size_t)gy * p.w_ + gx) * pstride + coff + c4 * 4);
                *(f32x4*)&tile[pix * PS + c4 * 4] = v;
            }
        }
        __syncthreads();

        LOAD_A(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < NS) {
                LOAD_A(nxt, s + 1);
                LOAD_B(nxt, ch, s + 1);
            } else if (ch + 1 < nch) {
                LOAD_B(nxt, ch + 1, 0);
            }
            // keep the prefetch ABOVE this step's MFMAs (hipcc otherwise sinks each load to its first use)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = VAD_KORDER(jj);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = MFMA32(a[cur][mt][j], b[cur][nt][j], acc[mt][nt]);
            }
        }
    }
#undef LOAD_A
#undef LOAD_B

    // ---------------------------------------------------------------- epilogue
    if (MODE == MODE_LSTM) {
        const int hc = (cb * WN + wn) * 32 + li;
        const size_t cfs = (size_t)p.h * p.w_ * p.hid;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int wq = 2 * (r >> 2) + lh, pos = r & 3;
                const int y = y0 + 2 * (wm * MT + mt) + (pos >> 1), x = x0 + 2 * wq + (pos & 1);
                if (y < p.h && x < p.w_) {
                    const size_t pix = (size_t)y * p.w_ + x;
                    const float cp = p.c_prev ? p.c_prev[(size_t)n * cfs + pix * p.hid + hc] : 0.f;
                    float cn, hn;
                    vad_lstm_cell(acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r], cp, cn, hn);
                    p.c_out[(size_t)n * cfs + pix * p.hid + hc] = cn;
                    p.out[(size_t)n * p.out_fs + pix * p.hid + hc] = hn;
                }
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = ((cb * WN + wn) * NT + nt) * 32 + li;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int wq = 2 * q + lh;
                    float v[4];
#pragma unroll
                    for (int pos = 0; pos < 4; ++pos) v[pos] = vad_act(acc[mt][nt][4 * q + pos], ACT);
                    if (MODE == MODE_POOL) {
                        const float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                        const int oy = (y0 >> 1) + (wm * MT + mt), ox = (x0 >> 1) + wq;
                        if (oy < (p.h >> 1) && ox < (p.w_ >> 1))
                            p.out[(size_t)n * p.out_fs + ((size_t)oy * (p.w_ >> 1) + ox) * p.cout + co] = m;
                    } else {
#pragma unroll
                        for (int pos = 0; pos < 4; ++pos) {
                            const int y = y0 + 2 * (wm * MT + mt) + (pos >> 1), x = x0 + 2 * wq + (pos & 1);
                            if (y < p.h && x < p.w_)
                                p.out[(size_t)n * p.out_fs + ((size_t)y * p.w_ + x) * p.cout + co] = v[pos];
                        }
                    }
                }
            }
        }
    }
}

#include "conv_pkernel.h"
#include "conv_small.h"
#include "convt_pkernel.h"

// Developer A/B switches (vad_debug_*): written only by an explicit debug call, read once per launch; the library
// itself never writes them, and the arithmetic mode is NOT among them (it is an argument of every entry point).
static std::atomic<unsigned long long*> g_vad_dbg{nullptr};
extern "C" int vad_debug_set_stamp_buffer(void* p) { g_vad_dbg = (unsigned long long*)p; return VAD_OK; }
static std::atomic<int> g_vad_conv_bits{1};   // bit 0: 1 = persistent + register prefetch (default), 0 = one tile per work-group;
                                              // bit 1: pricing runs, results invalid (exact: drop epilogue stores; split: no weight reads);
                                              // bit 2: alternative cout-64 tiling;
                                              // bit 3: never use the small-grid (16x16x4) ConvLSTM kernel; bit 4: always use it;
                                              // bit 5: never compute the ConvLSTM x halves ahead of the recurrence;
                                              // bit 6: never use the gate-split small-grid kernel; bit 7: use its 8-wave form wherever the small-grid form runs
extern "C" int vad_debug_set_conv_variant(int v) { g_vad_conv_bits = v & 255; return VAD_OK; }
// do un-activated convolution launches return their own BatchNorm partial sums (the persistent kernels' statistics
// instantiations)?  The bf16-tensor training step has no other source of statistics and asks BEFORE its first launch.
bool vad_conv_stats_available(void) { return (g_vad_conv_bits.load(std::memory_order_relaxed) & 1) != 0; }
// may the model-level launch sequence split the ConvLSTM steps of small launch groups into x halves (ahead) + h halves?
bool vad_convlstm_hoist_ok(void) {
    const int b = g_vad_conv_bits.load(std::memory_order_relaxed);
    return (b & 1) && !(b & 8) && !(b & 32);
}
struct ConvKnobs {
    int variant, stagger, conv64, no_small, all_small, no_gate, all_gate;
    ConvKnobs() {
        const int b = g_vad_conv_bits.load(std::memory_order_relaxed);
        variant = b & 1; stagger = (b >> 1) & 1; conv64 = (b >> 2) & 1; no_small = (b >> 3) & 1; all_small = (b >> 4) & 1;
        no_gate = (b >> 6) & 1; all_gate = (b >> 7) & 1;
    }
};
#define VAD_REQUIRE_PREC(who) VAD_REQUIRE(precision >= VAD_PREC_FP32 && precision <= VAD_PREC_BF16S, who ": precision=%d must be VAD_PREC_FP32 (0), VAD_PREC_SPLIT (1), VAD_PREC_BF16 (2) or VAD_PREC_BF16S (3)", precision)

static int vad_num_cus() {
    static std::atomic<int> ncu_cached{0};
    int ncu = ncu_cached.load(std::memory_order_relaxed);
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
        ncu_cached = ncu;
    }
    return ncu;
}

// does the gate-split kernel (conv_small.h) serve n frames best?  cq = output columns / 4 (hid for a ConvLSTM step).  Measured
// per step on a 16x16 map, hid 128 (tools/gpu_lstm_slope.sh): see DESIGN.md section 4.5.
static int vad_gate_kernel_wins(int n, int h, int wd, int cq, const ConvKnobs& kn) {   // -> M-tiles per work-group (1, 2) or 0 = no
    if (kn.no_gate || cq % 16 != 0) return 0;
    if (kn.all_gate) return 2;                                                            // (debug: the 8-wave form everywhere)
    const int ncu = vad_num_cus();
    const long long nb1 = (long long)n * ((wd + 7) / 8) * ((h + 1) / 2) * (cq / 16);      // work-groups of 4 waves
    // (measured per step, 16x16 map, hid 128, tools/gpu_lstm_variants.sh: 4 waves per work-group win or tie up to 6 work-groups
    // per CU - 5 clips 44.8 -> 36.3 us, 10 clips 78.6 -> 62.8 - and lose at 8: 16 clips 88.6 -> 99.8)
    if (nb1 <= 6ll * ncu) return 1;
    const long long nb2 = (long long)n * ((wd + 15) / 16) * ((h + 1) / 2) * (cq / 16);    // of 8 waves
    return 2 * nb2 <= 3ll * ncu ? 2 : 0;
}


template <typename K>
static unsigned persistent_grid(K kernel, unsigned nblocks, int max_per_cu = 2) {
    int per_cu = 0;
    const int ncu = vad_num_cus();
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    // Use EVERY resident slot the hardware offers: with fewer work-groups than slots the dispatcher packs some CUs
    // to their limit and leaves others with one group, and a persistent grid then runs at the pace of the fullest CU.
    (void)max_per_cu;
    const unsigned g = (unsigned)ncu * (unsigned)per_cu;
    return nblocks < g ? nblocks : g;
}

// Persistent grid: (tiles per frame x cout blocks) work-group positions, replicated over `fgroups` frame groups
// so that about (CUs x resident work-groups per CU) groups run; group g of a position walks frames g, g+fgroups, ...
static thread_local unsigned t_vad_last_pgrid = 0;      // grid of this thread's latest persistent conv3x3 launch (0: not persistent)
static unsigned persistent_grid_for(const Conv3P& p, unsigned cap) {
    const unsigned per_frame = (unsigned)(p.tiles_x * p.tiles_y * p.cblocks);
    unsigned fgroups = cap / per_frame;
    if (fgroups < 1) fgroups = 1;
    if (fgroups > (unsigned)p.n) fgroups = (unsigned)p.n;
    return t_vad_last_pgrid = per_frame * fgroups;
}

// `variant`: 1 = persistent kernel, 0 = one tile per work-group (exact fp32 only); grid caps are per instantiation and
// written once with the same value by whichever thread gets there first.
// one persistent launch; ST = 1: the instantiation that also writes the BatchNorm partial sums (p.stats)
template <int CK, int MT, int NT, int WM, int WN, int MODE, int ACT, int PREC, int ST, int IO16 = 0>
static void launch_conv3_persistent(const Conv3P& p, const Conv3P& q, hipStream_t s) {
    static std::atomic<unsigned> grid_cap{0};
    unsigned cap = grid_cap.load(std::memory_order_relaxed);
    if (!cap) grid_cap = cap = persistent_grid(conv3x3_mfma_pkernel<CK, MT, NT, WM, WN, MODE, ACT, 0, PREC, ST, IO16>, ~0u);
    hipLaunchKernelGGL((conv3x3_mfma_pkernel<CK, MT, NT, WM, WN, MODE, ACT, 0, PREC, ST, IO16>), dim3(persistent_grid_for(p, cap)), dim3(256), 0, s, q);
}
template <int CK, int MT, int NT, int WM, int WN, int MODE, int ACT, int PREC, int IO16 = 0>
static void launch_conv3_p(const Conv3P& p, const Conv3P& q, hipStream_t s) {
    if constexpr (MODE == MODE_PLAIN && ACT == VAD_ACT_NONE) {
        if (p.stats) { launch_conv3_persistent<CK, MT, NT, WM, WN, MODE, ACT, PREC, 1, IO16>(p, q, s); return; }
    }
    launch_conv3_persistent<CK, MT, NT, WM, WN, MODE, ACT, PREC, 0, IO16>(p, q, s);
}

template <int CK, int MT, int NT, int WM, int WN, int MODE, int ACT>
static void launch_conv3_act(const Conv3P& p, hipStream_t s, int precision, int variant, const ConvKnobs& kn) {
    if (precision == VAD_PREC_SPLIT) {   // split-fp16 operands (persistent kernel only)
        Conv3P q = p;
        q.dbg = g_vad_dbg;
        q.stagger = kn.stagger;   // debug (variant bit 1): zero-sized weight descriptor = price the weight traffic
        launch_conv3_p<CK, MT, NT, WM, WN, MODE, ACT, 1>(p, q, s);
        return;
    }
    if constexpr (MODE == MODE_PLAIN && ACT == VAD_ACT_NONE) {   // bf16 tensors (host-checked: plain un-activated launches only)
        if (precision == VAD_PREC_BF16S) {
            Conv3P q = p;
            q.dbg = g_vad_dbg;
            launch_conv3_p<CK, MT, NT, WM, WN, MODE, ACT, 2, 1>(p, q, s);
            return;
        }
    }
    if constexpr (MODE != MODE_LSTM) {   // bf16 operands (training convolutions; the ConvLSTM step is not offered in bf16)
        if (precision == VAD_PREC_BF16) {
            Conv3P q = p;
            q.dbg = g_vad_dbg;
            launch_conv3_p<CK, MT, NT, WM, WN, MODE, ACT, 2>(p, q, s);
            return;
        }
    }
    if (variant == 0) {
        t_vad_last_pgrid = 0;
        hipLaunchKernelGGL((conv3x3_mfma_kernel<CK, MT, NT, WM, WN, MODE, ACT>), dim3(p.nblocks), dim3(256), 0, s, p);
    } else {
        Conv3P q = p;
        q.dbg = g_vad_dbg;
        q.stagger = kn.stagger * 2;   // debug (variant bit 1, exact kernels): bit 1 = drop the epilogue's stores (prices them)
        launch_conv3_p<CK, MT, NT, WM, WN, MODE, ACT, 0>(p, q, s);
    }
}

template <int CK, int MT, int NT, int WM, int WN, int MODE>
static int launch_conv3(Conv3P& p, int n, int act, hipStream_t s, int precision, int variant, const ConvKnobs& kn) {
    constexpr int TH = 2 * MT * WM;
    p.tiles_x = (p.w_ + 15) / 16;
    p.tiles_y = (p.h + TH - 1) / TH;
    p.cblocks = (MODE == MODE_LSTM) ? p.hid / (32 * WN) : p.cout / (32 * NT * WN);
    const long long nb = (long long)n * p.tiles_x * p.tiles_y * p.cblocks;
    VAD_REQUIRE(nb > 0 && nb < (1ll << 31), "conv3x3: grid of %lld blocks out of range", nb);
    p.nblocks = (unsigned)nb;
    p.n = n;
    // 32-bit addressing inside the kernels: one frame of any tensor and the weight blob must stay below 2^31 bytes
    const long long chmax = p.cin > p.cout ? p.cin : p.cout;
    VAD_REQUIRE((long long)p.h * p.w_ * chmax * 4 < (1ll << 31) && 9ll * p.cin * p.cout * 4 < (1ll << 31),
                "conv3x3: frame %dx%dx%lld or weights %dx%d too large for 32-bit offsets", p.h, p.w_, chmax, p.cin, p.cout);
    if (MODE == MODE_LSTM || act == VAD_ACT_NONE) launch_conv3_act<CK, MT, NT, WM, WN, MODE, VAD_ACT_NONE>(p, s, precision, variant, kn);
    else if (act == VAD_ACT_LEAKY) launch_conv3_act<CK, MT, NT, WM, WN, MODE, VAD_ACT_LEAKY>(p, s, precision, variant, kn);
    else launch_conv3_act<CK, MT, NT, WM, WN, MODE, VAD_ACT_RELU>(p, s, precision, variant, kn);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// upper bound of the partial-sum rows x 2 x cout floats a stats launch writes (persistent grid <= 4 work-groups per CU)
size_t vad_conv3x3_stats_floats(int cout) { return (size_t)vad_num_cus() * 4 * 2 * (size_t)cout; }

static int conv3_stats_rows(int rc, const Conv3P& p, bool with_stats, int* stats_rows) {
    if (rc == VAD_OK && with_stats && t_vad_last_pgrid) *stats_rows = (int)(t_vad_last_pgrid / (unsigned)p.cblocks);
    return rc;
}

extern "C" int vad_conv3x3(const float* in, long long in_fs, const float* w, const float* bias,
                           float* out, long long out_fs, int n, int h, int wd, int cin, int cout,
                           int act, int pool, int precision, void* stream) {
    return vad_conv3x3_stats(in, in_fs, w, bias, out, out_fs, n, h, wd, cin, cout, act, pool, precision, nullptr, nullptr, stream);
}

// stats / stats_rows (both or neither; un-pooled, un-activated launches = the training forward): the persistent kernel also
// writes BatchNorm partial sums [rows][2][cout] (shifted by the bias) and *stats_rows = their row count; 0 when the launch
// could not provide them (the caller then makes its own pass over `out`).  Room needed: vad_conv3x3_stats_floats().
int vad_conv3x3_stats(const float* in, long long in_fs, const float* w, const float* bias,
                      float* out, long long out_fs, int n, int h, int wd, int cin, int cout,
                      int act, int pool, int precision, float* stats, int* stats_rows, void* stream) {
    return vad_conv3x3_kpart(in, in_fs, w, bias, out, out_fs, n, h, wd, cin, cin, cout, act, pool, precision, stats, stats_rows, stream);
}

// cin_w >= cin: the packed weights have cin_w input channels and the convolution uses their first `cin` (the x half of a
// ConvLSTM cell's weight (4*hid, x + hid, 3, 3): exact fp32, persistent kernels only)
int vad_conv3x3_kpart(const float* in, long long in_fs, const float* w, const float* bias,
                      float* out, long long out_fs, int n, int h, int wd, int cin, int cin_w, int cout,
                      int act, int pool, int precision, float* stats, int* stats_rows, void* stream) {
    if (stats_rows) *stats_rows = 0;
    VAD_REQUIRE((stats == nullptr) == (stats_rows == nullptr), "conv3x3: stats and stats_rows come together");
    VAD_REQUIRE(in && w && bias && out, "conv3x3: null pointer");
    VAD_REQUIRE_PREC("conv3x3");
    const ConvKnobs kn;
    VAD_REQUIRE(n > 0 && h > 0 && wd > 0, "conv3x3: bad shape n=%d h=%d w=%d", n, h, wd);
    VAD_REQUIRE(cin % 32 == 0 && cout % 32 == 0 && cin > 0 && cout > 0,
                "conv3x3: cin=%d cout=%d must be positive multiples of 32", cin, cout);
    VAD_REQUIRE(!pool || (h % 2 == 0 && wd % 2 == 0), "conv3x3: pooling needs even H,W (got %dx%d)", h, wd);
    VAD_REQUIRE(act >= 0 && act <= 2, "conv3x3: bad act %d", act);
    VAD_REQUIRE(precision != VAD_PREC_BF16S || (act == VAD_ACT_NONE && !pool), "conv3x3: VAD_PREC_BF16S (bf16 tensors) is the training form: no activation, no pooling");
    Conv3P p{};
    p.in = in; p.in_fs = in_fs ? in_fs : (long long)h * wd * cin; p.cin_a = cin;
    p.in2 = nullptr; p.in2_fs = 0;
    p.w = w; p.bias = bias; p.out = out;
    const int ho = pool ? h / 2 : h, wo = pool ? wd / 2 : wd;
    p.out_fs = out_fs ? out_fs : (long long)ho * wo * cout;
    p.h = h; p.w_ = wd; p.cin = cin_w; p.cout = cout; p.hid = 0;
    VAD_REQUIRE(cin_w == cin || (cin_w > cin && cin_w % 32 == 0 && precision == VAD_PREC_FP32 && kn.variant != 0),
                "conv3x3: a channel sub-range of the weights (cin %d of %d) is offered by the exact-fp32 persistent kernels only", cin, cin_w);
    hipStream_t s = (hipStream_t)stream;
    const bool with_stats = stats && !pool && act == VAD_ACT_NONE && kn.variant != 0;
    p.stats = with_stats ? stats : nullptr;
    t_vad_last_pgrid = 0;
    // (every return below goes through L3; the row count is read back from the launch's grid)
#define L3(CK, MT, NT, WM, WN, MODE) conv3_stats_rows(launch_conv3<CK, MT, NT, WM, WN, MODE>(p, n, act, s, precision, kn.variant, kn), p, with_stats, stats_rows)
    if (precision != VAD_PREC_FP32) {
        // split-fp16 operands double the accumulators (main + cross terms): keep one N-tile per wave (bf16 shares the tilings)
        // (a 64-column tiling for grids of the 128-column one that leave CUs idle - the per-step ConvLSTM data gradients of the
        // bf16 training step: 128 work-groups - measured no gain: 56 vs 53 us per launch; not kept)
        if (cout % 128 == 0 && !kn.conv64)   // one B fragment per 4 M-tiles: halves the weight traffic through L1
            return pool ? L3(32, 4, 1, 1, 4, MODE_POOL) : L3(32, 4, 1, 1, 4, MODE_PLAIN);
        if (cout % 64 == 0)
            return pool ? L3(32, 2, 1, 2, 2, MODE_POOL) : L3(32, 2, 1, 2, 2, MODE_PLAIN);
        return pool ? L3(32, 2, 1, 4, 1, MODE_POOL) : L3(32, 2, 1, 4, 1, MODE_PLAIN);
    }
    // Latency tiling (round 3; the reference's own call sizes: one image in main.py:274, 16 in evaluate.py:240): a launch takes
    // as long as ONE wave's serial K loop when its grid does not fill the chip - enc4.3 on one 256x256 frame was 16
    // work-groups and 123 us.  When the throughput tiling gives fewer work-groups than CUs, a work-group takes 4 rows x 16
    // columns x 64 channels instead (one 32x32 MFMA tile per wave: a quarter of the serial work, 4x the work-groups; same K
    // order, bit-identical results).
    // The x half of a ConvLSTM step computed ahead of the recurrence (a channel sub-range of the cell's weights, no activation)
    // on the smallest grids: the gate-split kernel's K loop without the cell (conv_small.h), same values bit for bit.
    // Gate-split small-grid kernel without the cell (conv_small.h; same values bit for bit): the x half of a ConvLSTM step
    // computed ahead of the recurrence (a channel sub-range of the cell's weights), and - the reference's one-image / 16-image
    // calls - any layer whose throughput tiling leaves CUs idle and whose K loop is long enough to pay for 4x the waves.
    int mtw = 0;
    if (!with_stats && cout % 64 == 0 && kn.variant != 0 && (long long)h * wd * cin * 4 < (1ll << 31) && 9ll * cin_w * cout * 4 < (1ll << 31)) {
        if (cin_w > cin && !pool && act == VAD_ACT_NONE) mtw = vad_gate_kernel_wins(n, h, wd, cout / 4, kn);
        else if (cin_w == cin && cin >= 64 && !kn.no_gate) {
            const long long px = (long long)n * ((wd + 15) / 16);
            const long long nb_thr = cout % 128 == 0 ? px * ((h + 7) / 8) * (cout / 128) : px * ((h + 15) / 16) * (cout / 64);
            if (nb_thr < vad_num_cus() || kn.all_gate) mtw = vad_gate_kernel_wins(n, h, wd, cout / 4, kn);
        }
    }
    if (mtw) {
        p.hid = cout / 4;
        p.tiles_x = (wd + 8 * mtw - 1) / (8 * mtw); p.tiles_y = (h + 1) / 2; p.cblocks = p.hid / 16; p.n = n;
        p.nblocks = (unsigned)((long long)n * p.tiles_x * p.tiles_y * p.cblocks);
#define GATE_LAUNCH(A_, P_)                                                                                                          \
        { if (mtw == 1) hipLaunchKernelGGL((convlstm_gate_kernel<0, 1, A_, P_>), dim3(p.nblocks), dim3(256), 0, s, p);                \
          else hipLaunchKernelGGL((convlstm_gate_kernel<0, 2, A_, P_>), dim3(p.nblocks), dim3(512), 0, s, p); }
        if (pool) { if (act == VAD_ACT_LEAKY) GATE_LAUNCH(VAD_ACT_LEAKY, 1) else if (act == VAD_ACT_RELU) GATE_LAUNCH(VAD_ACT_RELU, 1) else GATE_LAUNCH(VAD_ACT_NONE, 1) }
        else { if (act == VAD_ACT_LEAKY) GATE_LAUNCH(VAD_ACT_LEAKY, 0) else if (act == VAD_ACT_RELU) GATE_LAUNCH(VAD_ACT_RELU, 0) else GATE_LAUNCH(VAD_ACT_NONE, 0) }
#undef GATE_LAUNCH
        VAD_LAUNCH_CHECK();
        return VAD_OK;
    }
    if (cout % 64 == 0) {
        const long long px = (long long)n * ((wd + 15) / 16);
        const long long nb_thr = cout % 128 == 0 ? px * ((h + 7) / 8) * (cout / 128) : px * ((h + 15) / 16) * (cout / 64);
        // (not for the training forward's statistics launches: the convolution's values do not depend on the tiling, but its
        // BatchNorm partial sums are grouped by work-group, and the multi-step loss-curve gate of tests/test_hip_train_step.py was
        // calibrated - 1.1e-4 against a 5e-4 bound, chaotic in the last digits of the statistics - on the throughput tilings)
        if (nb_thr < vad_num_cus() && !with_stats) return pool ? L3(32, 1, 1, 2, 2, MODE_POOL) : L3(32, 1, 1, 2, 2, MODE_PLAIN);
    }
    if (cout % 128 == 0) {
        // Small grids (the per-step ConvLSTM gate convolutions of the training path: 16x16 maps, a few dozen frames): the
        // 8-row tile yields 8 work-groups per 16x16x512 frame, too few to fill 256 CUs twice; a 4-row tile doubles them
        // (same K order, identical results).  Measured at 32 clips: 24 -> see DESIGN.md section 9.
        const long long nb8 = (long long)n * ((wd + 15) / 16) * ((h + 7) / 8) * (cout / 128);
        if (!pool && nb8 < 768) return L3(32, 1, 2, 2, 2, MODE_PLAIN);
        return pool ? L3(32, 2, 2, 2, 2, MODE_POOL) : L3(32, 2, 2, 2, 2, MODE_PLAIN);
    } else if (cout % 64 == 0) {
        if (kn.conv64) return pool ? L3(32, 2, 1, 2, 2, MODE_POOL) : L3(32, 2, 1, 2, 2, MODE_PLAIN);
        return pool ? L3(32, 2, 2, 4, 1, MODE_POOL) : L3(32, 2, 2, 4, 1, MODE_PLAIN);
    } else {
        return pool ? L3(32, 2, 1, 4, 1, MODE_POOL) : L3(32, 2, 1, 4, 1, MODE_PLAIN);
    }
#undef L3
}

// Fused enc1: Conv2d(3->32)+BN+LeakyReLU -> Conv2d(32->32)+BN+LeakyReLU -> MaxPool2d(2,2)
// (reference models/autoencoder.py:38-46) in one launch; x is NCHW [N,3,H,W], out NHWC [N,H/2,W/2,32].
extern "C" int vad_conv3x3_c3_fused(const float* x, const float* w0, const float* b0, const float* w1,
                                    const float* b1, float* out, int n, int h, int wd, int precision, void* stream) {
    return vad_conv3x3_c3_fused_fmt(x, VAD_X_F32_NCHW, w0, b0, w1, b1, out, n, h, wd, precision, stream);
}

int vad_conv3x3_c3_fused_fmt(const void* x, int fmt, const float* w0, const float* b0, const float* w1,
                             const float* b1, float* out, int n, int h, int wd, int precision, void* stream) {
    VAD_REQUIRE(x && w0 && b0 && w1 && b1 && out, "conv3x3_c3_fused: null pointer");
    VAD_REQUIRE(precision == VAD_PREC_FP32 || precision == VAD_PREC_SPLIT, "conv3x3_c3_fused: precision=%d must be VAD_PREC_FP32 (0) or VAD_PREC_SPLIT (1)", precision);
    const ConvKnobs kn;
    VAD_REQUIRE(fmt == VAD_X_F32_NCHW || (fmt == VAD_X_U8_NHWC && (kn.variant != 0 || precision == VAD_PREC_SPLIT)),
                "conv3x3_c3_fused: input format %d unsupported (uint8 input needs the persistent kernel)", fmt);
    VAD_REQUIRE(n > 0 && h > 0 && wd > 0 && h % 2 == 0 && wd % 2 == 0, "conv3x3_c3_fused: bad shape %dx%d", h, wd);
    Conv3P p{};
    p.in = (const float*)x; p.in_fs = (long long)3 * h * wd; p.cin_a = 32;   // frame stride in ELEMENTS of the input type
    p.w = w1; p.bias = b1; p.out = out; p.out_fs = (long long)(h / 2) * (wd / 2) * 32;
    p.w0 = w0; p.b0 = b0; p.xu8 = fmt == VAD_X_U8_NHWC;
    p.h = h; p.w_ = wd; p.cin = 32; p.cout = 32; p.hid = 0;
    constexpr int TH = 16;
    p.tiles_x = (wd + 15) / 16; p.tiles_y = (h + TH - 1) / TH; p.cblocks = 1;
    const long long nb = (long long)n * p.tiles_x * p.tiles_y;
    VAD_REQUIRE(nb < (1ll << 31), "conv3x3_c3_fused: grid too large");
    p.nblocks = (unsigned)nb;
    p.n = n;
    if (precision == VAD_PREC_SPLIT) {
        static std::atomic<unsigned> grid_cap1{0};
        unsigned cap = grid_cap1.load(std::memory_order_relaxed);
        if (!cap) grid_cap1 = cap = persistent_grid(conv3x3_mfma_pkernel<32, 2, 1, 4, 1, MODE_POOL, VAD_ACT_LEAKY, 1, 1>, ~0u);
        p.dbg = g_vad_dbg;
        hipLaunchKernelGGL((conv3x3_mfma_pkernel<32, 2, 1, 4, 1, MODE_POOL, VAD_ACT_LEAKY, 1, 1>), dim3(persistent_grid_for(p, cap)), dim3(256), 0,
                           (hipStream_t)stream, p);
    } else if (kn.variant == 0) {
        hipLaunchKernelGGL((conv3x3_mfma_kernel<32, 2, 1, 4, 1, MODE_POOL, VAD_ACT_LEAKY, 1>), dim3((unsigned)nb), dim3(256), 0,
                           (hipStream_t)stream, p);
    } else {
        static std::atomic<unsigned> grid_cap{0};
        unsigned cap = grid_cap.load(std::memory_order_relaxed);
        if (!cap) grid_cap = cap = persistent_grid(conv3x3_mfma_pkernel<32, 2, 1, 4, 1, MODE_POOL, VAD_ACT_LEAKY, 1>, ~0u);
        p.dbg = g_vad_dbg;
        hipLaunchKernelGGL((conv3x3_mfma_pkernel<32, 2, 1, 4, 1, MODE_POOL, VAD_ACT_LEAKY, 1>), dim3(persistent_grid_for(p, cap)), dim3(256), 0,
                           (hipStream_t)stream, p);
    }
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

bool vad_convlstm_gate_wins(int n, int h, int wd, int hid) { const ConvKnobs kn; return kn.variant != 0 && !kn.no_small && vad_gate_kernel_wins(n, h, wd, hid, kn) != 0; }

// the cost model of vad_convlstm_step: does the small-grid (16x16x4) form serve n frames of h x w best?
bool vad_convlstm_small_wins(int n, int h, int wd, int hid) {
    const long long nb_big = (long long)n * ((wd + 15) / 16) * ((h + 3) / 4) * (hid / 64);
    const int ncu = vad_num_cus();
    const long long m_big = (nb_big + ncu - 1) / ncu;
    const double cost_big = 2.0 * (double)(m_big / 2) + 1.107 * (double)(m_big % 2);     // in units of half a pair's time
    const double cost_small = 1.178 * (double)nb_big / (double)ncu;
    return nb_big < ncu || cost_small < cost_big;
}

extern "C" int vad_convlstm_step(const float* x, long long x_fs, const float* h_prev, long long h_prev_fs, const float* c_prev,
                                 const float* w, const float* bias, float* h_out, long long h_out_fs,
                                 float* c_out, int n, int h, int wd, int cin_x, int hid, int precision, void* stream) {
    return vad_convlstm_step_zx(x, x_fs, nullptr, 0, h_prev, h_prev_fs, c_prev, w, bias, h_out, h_out_fs, c_out, n, h, wd, cin_x, hid, precision, stream);
}

// zx != NULL (exact fp32, small-grid kernel): bias + x half of the gate pre-activations [n][h][w][4*hid] (frame stride zx_fs,
// 0 = dense) computed ahead by vad_conv3x3_kpart on the cell's weight; the step then multiplies the h half only (nothing at
// all when h_prev == NULL).  Bit-identical to the un-split step: the accumulator chain is cut at a chunk boundary, stored as
// fp32 and resumed.  x is unused then.
int vad_convlstm_step_zx(const float* x, long long x_fs, const float* zx, long long zx_fs, const float* h_prev, long long h_prev_fs,
                         const float* c_prev, const float* w, const float* bias, float* h_out, long long h_out_fs,
                         float* c_out, int n, int h, int wd, int cin_x, int hid, int precision, void* stream) {
    VAD_REQUIRE((x || zx) && w && bias && h_out && c_out, "convlstm_step: null pointer");
    VAD_REQUIRE(precision == VAD_PREC_FP32 || precision == VAD_PREC_SPLIT, "convlstm_step: precision=%d must be VAD_PREC_FP32 (0) or VAD_PREC_SPLIT (1)", precision);
    VAD_REQUIRE((h_prev == nullptr) == (c_prev == nullptr), "convlstm_step: h_prev and c_prev must both be given or both NULL");
    VAD_REQUIRE(n > 0 && h > 0 && wd > 0, "convlstm_step: bad shape");
    VAD_REQUIRE(cin_x % 32 == 0 && hid % 64 == 0 && cin_x > 0 && hid > 0,
                "convlstm_step: cin_x=%d must be a multiple of 32 and hid=%d a multiple of 64", cin_x, hid);
    Conv3P p{};
    p.in = x; p.in_fs = x_fs ? x_fs : (long long)h * wd * cin_x; p.cin_a = cin_x;
    p.in2 = h_prev; p.in2_fs = h_prev_fs ? h_prev_fs : (long long)h * wd * hid;
    p.w = w; p.bias = bias; p.out = h_out;
    p.out_fs = h_out_fs ? h_out_fs : (long long)h * wd * hid;
    p.c_prev = c_prev; p.c_out = c_out;
    p.zx = zx; p.zx_fs = zx_fs ? zx_fs : (long long)h * wd * 4 * hid;
    p.h = h; p.w_ = wd; p.cin = cin_x + hid; p.cout = 4 * hid; p.hid = hid;
    // the persistent kernel shares one set of staging offsets between x and h: needs cin_x == hid
    VAD_REQUIRE(!(precision == VAD_PREC_SPLIT && cin_x != hid), "convlstm_step: split precision needs cin_x == hid (got %d, %d)", cin_x, hid);
    const ConvKnobs kn;
    // Small grids (the reference's own batch sizes: 4 clips, or 1 window): the 32x32x2 tiling yields 8 work-groups per clip of
    // a 16x16 map, each a 123 us serial K loop; below one work-group per CU the 16x16x4 form (4x the waves, a quarter of the
    // latency, bit-identical results: conv_small.h) wins - 133 -> ~35 us per step at B <= 8.
    // Above that the small form still wins where the large one quantises badly: a launch takes as long as its fullest CU, which
    // runs m = ceil(groups / CUs) of the large work-groups two at a time (measured per step, 16x16 map, hid 128: 0.166 ms alone,
    // 0.30 ms for a pair), while the small form's time is proportional to the work (0.354 ms at 64 clips).  40 clips: 0.281 ms
    // large (64 CUs hold a pair, 192 one group and wait) vs 0.228 ms small; 64 clips: 0.320 vs 0.354.
    const bool small_wins = vad_convlstm_small_wins(n, h, wd, hid);
    VAD_REQUIRE(!zx || (precision == VAD_PREC_FP32 && kn.variant != 0), "convlstm_step: precomputed x halves go with the exact-fp32 small-grid kernel");
    if (precision == VAD_PREC_FP32 && kn.variant != 0 && ((!kn.no_small && (small_wins || kn.all_small)) || zx)) {
        p.tiles_x = (wd + 15) / 16; p.tiles_y = (h + 1) / 2; p.cblocks = hid / 32;
        const long long nb = (long long)n * p.tiles_x * p.tiles_y * p.cblocks;
        p.nblocks = (unsigned)nb; p.n = n;
        VAD_REQUIRE((long long)h * wd * (cin_x > hid ? cin_x : hid) * 4 < (1ll << 31) && 9ll * p.cin * p.cout * 4 < (1ll << 31),
                    "convlstm_step: frame or weights too large for 32-bit offsets");
        // the smallest grids: one gate per wave, 8 waves per work-group, twice the work-groups (conv_small.h)
        if (const int mtw = vad_gate_kernel_wins(n, h, wd, hid, kn)) {
            p.cblocks = hid / 16; p.tiles_x = (wd + 8 * mtw - 1) / (8 * mtw);
            p.nblocks = (unsigned)((long long)n * p.tiles_x * p.tiles_y * p.cblocks);
            if (mtw == 1) hipLaunchKernelGGL((convlstm_gate_kernel<1, 1>), dim3(p.nblocks), dim3(256), 0, (hipStream_t)stream, p);
            else hipLaunchKernelGGL((convlstm_gate_kernel<1, 2>), dim3(p.nblocks), dim3(512), 0, (hipStream_t)stream, p);
        } else {
            hipLaunchKernelGGL(convlstm_small_kernel<4>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
        }
        VAD_LAUNCH_CHECK();
        return VAD_OK;
    }
    return launch_conv3<32, 1, 4, 2, 2, MODE_LSTM>(p, n, VAD_ACT_NONE, (hipStream_t)stream, precision, cin_x != hid ? 0 : kn.variant, kn);
}

// ---------------------------------------------------------------------------------------------
// First layer: Conv2d(3 -> Cout) straight from the NCHW input planes.  K = 27 padded to 28 (14
// k-pairs); the B operand (14 floats per lane) lives in registers for the whole work-group.
struct ConvC3P {
    const float* x; const float* w; const float* bias; float* out;
    int h, w_, cout, tiles_x, tiles_y;
    unsigned nblocks;
    int xu8;                 // x is uint8 NHWC [N,H,W,3]
    float* stats;            // nullable (persistent un-pooled kernel, cout == 32): per-work-group partial sums for BatchNorm,
                             // stats[block][0][c] = sum(y - bias[c]), stats[block][1][c] = sum((y - bias[c])^2)
};

template <int MT, int POOL, int ACT>
__global__ __launch_bounds__(256) void conv3x3_c3_kernel(ConvC3P p) {
    constexpr int TH = 2 * MT * 4, LH = TH + 2, RS = 20;
    __shared__ float tile[3 * LH * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    unsigned L = vad_xcd_remap(blockIdx.x, p.nblocks);
    const int tx = L % p.tiles_x; L /= p.tiles_x;
    const int ty = L % p.tiles_y;
    const int n = L / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * 16;

    const float* xin = p.x + (size_t)n * 3 * p.h * p.w_;
    const unsigned char* xin8 = (const unsigned char*)p.x + (size_t)n * 3 * p.h * p.w_;
    for (int idx = tid; idx < 3 * LH * 18; idx += 256) {
        const int lx = idx % 18, t = idx / 18, ly = t % LH, c = t / LH;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        float v = 0.f;
        if (gy >= 0 && gy < p.h && gx >= 0 && gx < p.w_)
            v = p.xu8 ? vad_norm_u8(xin8[((size_t)gy * p.w_ + gx) * 3 + c]) : xin[((size_t)c * p.h + gy) * p.w_ + gx];
        tile[(c * LH + ly) * RS + lx] = v;
    }
    __syncthreads();

    const int prow = (li >> 1) & 1, pcol = 2 * (li >> 2) + (li & 1);
    // per-lane tap offsets: k = 2s + h -> (c, dy, dx) = (k/9, (k%9)/3, k%3); k = 27 is padding
    int koff[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int k0 = 2 * s, k1 = 2 * s + 1;
        const int o0 = ((k0 / 9) * LH + (k0 % 9) / 3) * RS + (k0 % 3);
        const int o1 = (k1 < 27) ? ((k1 / 9) * LH + (k1 % 9) / 3) * RS + (k1 % 3) : 0;
        koff[s] = lh ? o1 : o0;
    }

    for (int nt = 0; nt < p.cout / 32; ++nt) {
        const int co = nt * 32 + li;
        float b[14];
#pragma unroll
        for (int s = 0; s < 14; ++s) b[s] = p.w[(size_t)(s * 2 + lh) * p.cout + co];
        const float bv = p.bias[co];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int base = ((2 * (wave * MT + mt) + prow) * RS + pcol);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bv;
#pragma unroll
            for (int s = 0; s < 14; ++s) acc = MFMA32(tile[base + koff[s]], b[s], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int wq = 2 * q + lh;
                float v[4];
#pragma unroll
                for (int pos = 0; pos < 4; ++pos) v[pos] = vad_act(acc[4 * q + pos], ACT);
                if (POOL) {
                    const float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    const int oy = (y0 >> 1) + (wave * MT + mt), ox = (x0 >> 1) + wq;
                    if (oy < (p.h >> 1) && ox < (p.w_ >> 1))
                        p.out[(((size_t)n * (p.h >> 1) + oy) * (p.w_ >> 1) + ox) * p.cout + co] = m;
                } else {
#pragma unroll
                    for (int pos = 0; pos < 4; ++pos) {
                        const int y = y0 + 2 * (wave * MT + mt) + (pos >> 1), x = x0 + 2 * wq + (pos & 1);
                        if (y < p.h && x < p.w_)
                            p.out[(((size_t)n * p.h + y) * p.w_ + x) * p.cout + co] = v[pos];
                    }
                }
            }
        }
    }
}

// Persistent form of the first layer.  The exact-fp32 floor of this layer is the matrix pipe, not HBM: K = 27 -> 28 costs 14
// v_mfma_f32_32x32x2_f32 per 32-pixel M-tile (0.85 us per 256x256 frame when the pipe never idles), so the kernel is built
// around keeping it fed: a work-group walks tiles of 32 rows x 16 columns (grid = resident slots), the 14 B fragments and
// the bias live in registers for its whole life, every wave runs FOUR independent accumulator chains (its 4 M-tiles of
// 2 rows x 16 columns) interleaved k-step by k-step, and the input halo of the NEXT tile is fetched into registers while
// the current one is computed (3 planes x 34 x 18 values = 8 per thread).
// OUT16 (un-pooled, un-activated = the training forward of VAD_PREC_BF16S): the output tensor is bf16 in memory (the
// arithmetic and the statistics are unchanged).
// BF16OP (with OUT16; the training forward of VAD_PREC_BF16S since round 4): bf16 MFMA OPERANDS as well - K = 27 padded to 32 is
// two v_mfma_f32_32x32x16_bf16 per M-tile instead of fourteen exact-fp32 ones (a lane gathers its 2 x 8 taps from the fp32 tile in
// LDS and rounds them to bf16, nearest even; the weights are rounded once per work-group), fp32 accumulation from the bias; the
// epilogue and the statistics are the fp32 kernel's.  The layer then runs at the rate of its bf16 stores instead of the fp32 pipe.
template <int POOL, int ACT, int OUT16 = 0, int BF16OP = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_c3_pkernel(ConvC3P p) {
    static_assert(!OUT16 || (!POOL && ACT == VAD_ACT_NONE), "bf16 output: training forward only");
    static_assert(!BF16OP || OUT16, "bf16 operands: the bf16-tensor training forward");
    constexpr unsigned OS = OUT16 ? 2u : 4u;
    constexpr int MTW = 4, TH = 2 * MTW * 4, LH = TH + 2, RS = 20, NE = 3 * LH * 18, NST = (NE + 255) / 256;
    __shared__ float tile[3 * LH * RS + 1];              // + one dummy slot: staging slots past the tile write there
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int prow = (li >> 1) & 1, pcol = 2 * (li >> 2) + (li & 1);
    int koff[BF16OP ? 16 : 14];
    if constexpr (BF16OP) {        // MFMA s of an M-tile multiplies k = 16 s + 8 lh + j, j = 0..7 (k >= 27: weight 0, any finite tile value)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k0 = 16 * (i >> 3) + (i & 7), k1 = k0 + 8;
            const int o0 = ((k0 / 9) * LH + (k0 % 9) / 3) * RS + (k0 % 3);
            const int o1 = (k1 < 27) ? ((k1 / 9) * LH + (k1 % 9) / 3) * RS + (k1 % 3) : 0;
            koff[i] = lh ? o1 : (k0 < 27 ? o0 : 0);
        }
    } else {
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int o0 = ((k0 / 9) * LH + (k0 % 9) / 3) * RS + (k0 % 3);
            const int o1 = (k1 < 27) ? ((k1 / 9) * LH + (k1 % 9) / 3) * RS + (k1 % 3) : 0;
            koff[s] = lh ? o1 : o0;
        }
    }
    const int ctiles = p.cout / 32;            // output-channel tiles, one after the other (one for every reference layer)
    float pre[NST];
    bool oob[NST];              // uint8 input, border tiles only: element is padding
    // VALU work shares the pipe with the exact-fp32 MFMAs, and this kernel has only 56 MFMAs per tile to
    // hide it behind: everything that does not change from tile to tile is computed ONCE per work-group, and what changes is
    // wave-uniform and rides in the scalar offset of the buffer instructions.
    //   staging element e = tid + 256*j -> (plane c, halo row ly, halo column lx): its BYTE offset relative to the tile's
    //   first halo pixel (rel[j], VAD_OOB past the tile), its (ly, lx) for the border tiles, its LDS slot;
    //   epilogue: the lane's part of an output offset (column half lh, channel li).
    const unsigned esz = p.xu8 ? 1u : 4u;
    const unsigned plane_e = p.xu8 ? 1u : (unsigned)(p.h * p.w_), px_e = p.xu8 ? 3u : 1u;   // u8 NHWC: pixel stride 3, channel stride 1
    unsigned rel[NST];
    unsigned lyx[NST];          // ly << 8 | lx
    int lds_pos[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int e = tid + 256 * j, lx = e % 18, t = e / 18, ly = t % LH, c = t / LH;
        rel[j] = e < NE ? ((unsigned)((ly * p.w_ + lx) * (int)px_e) + (unsigned)c * plane_e) * esz : VAD_OOB;
        lyx[j] = (unsigned)(ly << 8 | lx);
        lds_pos[j] = e < NE ? (c * LH + ly) * RS + lx : 3 * LH * RS;
    }
    const unsigned frame_elems = 3u * (unsigned)(p.h * p.w_);
    auto fetch = [&](unsigned L) {
        const int tx = L % p.tiles_x; L /= p.tiles_x;
        const int ty = L % p.tiles_y;
        const int n = L / p.tiles_y;
        const int gy0 = ty * TH - 1, gx0 = tx * 16 - 1;                       // first halo pixel of the tile
        const bool inner = gy0 >= 0 && gx0 >= 0 && gy0 + LH <= p.h && gx0 + 18 <= p.w_;      // wave-uniform
        const int origin = (gy0 * p.w_ + gx0) * (int)px_e * (int)esz;         // bytes; negative on top / left border tiles
        const __amdgpu_buffer_rsrc_t r = vad_rsrc((const char*)p.x + (size_t)n * frame_elems * esz, frame_elems * esz);
        // one uniform branch selects the input format, one the tile kind; an interior tile (most of them) costs no vector
        // arithmetic at all: per-lane offset rel[j] + scalar offset origin
        if (inner) {
            if (p.xu8) {
#pragma unroll
                for (int j = 0; j < NST; ++j) {
                    // the RAW byte stays in flight; it is normalised where it is written to LDS - arithmetic here would wait
                    // for each load in turn, eight exposed round trips per tile
                    pre[j] = __uint_as_float((unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r, (int)rel[j], origin, 0));
                    oob[j] = false;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NST; ++j) pre[j] = vad_bload1(r, rel[j], (unsigned)origin);
            }
        } else {
            // border tile: elements outside the image get the out-of-range offset and the buffer load returns 0 by itself
#pragma unroll
            for (int j = 0; j < NST; ++j) {
                const int gy = gy0 + (int)(lyx[j] >> 8), gx = gx0 + (int)(lyx[j] & 255u);
                const bool ok = rel[j] != VAD_OOB && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w_;
                const unsigned off = ok ? (unsigned)(origin + (int)rel[j]) : VAD_OOB;
                if (p.xu8) {
                    pre[j] = __uint_as_float((unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r, (int)off, 0, 0));
                    oob[j] = !ok;                                              // padding is 0.0 AFTER normalisation
                } else {
                    pre[j] = vad_bload1(r, off, 0);
                }
            }
        }
    };
    // B fragments + bias: ONE register set, (re)loaded only when the channel tile changes - once per work-group for every
    // reference layer.  (Without restrict the compiler cannot hoist loads out of the tile loop past the stores.)
    float b[BF16OP ? 1 : 14], bv = 0.f;
    f16x8 bq[BF16OP ? 2 : 1];   // bf16 operands: the lane's 2 x 8 weights (bit patterns), k = 16 s + 8 lh + j
    f32x16 bias16;              // bv in every register: the C operand of a tile's first k-step
    int have = -1;
    // BatchNorm statistics of the training forward, taken from the accumulators instead of a second pass over the 8.4 MB per
    // frame this kernel has just written: two levels (per tile, then per work-group) so that a lane never adds more than 64 +
    // ~100 terms in a row; shifted by the bias (the channel's mean is close to it), as chan_sums_kernel shifts by a sample
    float st_s = 0.f, st_q = 0.f;
    const int oh = POOL ? p.h >> 1 : p.h, ow = POOL ? p.w_ >> 1 : p.w_;
    const unsigned orow = (unsigned)(ow * p.cout) * OS, ocol = (unsigned)p.cout * OS;      // bytes per output row / pixel
    // XCD-aware walk: in every round of gridDim.x tiles, the work-groups of one XCD take a CONTIGUOUS run of tiles (four tile
    // rows at 256x256), so neighbours that share 128-byte input lines (a tile row is 18 floats of a line; a line spans two
    // tiles) share an L2.  Dealt round-robin, every XCD fetched the lines for itself: 1.91 GB per 640 frames for 0.50 GB of
    // input (FETCH_SIZE, profiles/r02_pmc_traffic_video.json before this change).
    unsigned L = vad_xcd_remap(blockIdx.x, gridDim.x);
    if (L < p.nblocks) fetch(L);
    for (; L < p.nblocks; L += gridDim.x) {
        __syncthreads();                                   // every wave is done reading the previous tile
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            float v = pre[j];
            if (p.xu8) v = oob[j] ? 0.f : vad_norm_u8(__float_as_uint(v));
            tile[lds_pos[j]] = v;
        }
        __syncthreads();
        const unsigned Ln = L + gridDim.x;
        if (Ln < p.nblocks) fetch(Ln);                     // in flight during the MFMAs below
        unsigned t_ = L;
        const int tx = t_ % p.tiles_x; t_ /= p.tiles_x;
        const int ty = t_ % p.tiles_y;
        const int n = t_ / p.tiles_y;
        const int y0 = ty * TH, x0 = tx * 16;
        const __amdgpu_buffer_rsrc_t rout = vad_rsrc((const char*)p.out + (size_t)n * (oh * ow) * p.cout * OS, (unsigned)(oh * ow * p.cout) * OS);
        // this wave's first output row / the tile's first output column, as a byte offset inside the frame (scalar)
        const int oy0 = POOL ? (y0 >> 1) + wave * MTW : y0 + 2 * wave * MTW, ox0 = POOL ? x0 >> 1 : x0;
        const unsigned ubase = (unsigned)oy0 * orow + (unsigned)ox0 * ocol;
        const bool full = POOL ? (oy0 + MTW <= oh && ox0 + 8 <= ow) : (oy0 + 2 * MTW <= oh && ox0 + 16 <= ow);   // wave-uniform
        for (int nt = 0; nt < ctiles; ++nt) {
            const int co = nt * 32 + li;
            const unsigned lanepart = (unsigned)((POOL ? lh : 2 * lh) * p.cout + co) * OS;
            if (have != nt) {
                if constexpr (BF16OP) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        unsigned pk[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int k0 = 16 * s2 + 8 * lh + 2 * j, k1 = k0 + 1;       // (rows 27..31 of the operand are zero)
                            const float w0 = k0 < 27 ? p.w[(size_t)k0 * p.cout + co] : 0.f, w1 = k1 < 27 ? p.w[(size_t)k1 * p.cout + co] : 0.f;
                            pk[j] = vad_pack_bf16(w0, w1);
                        }
                        bq[s2] = __builtin_bit_cast(f16x8, u32x4{pk[0], pk[1], pk[2], pk[3]});
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 14; ++s) b[s] = p.w[(size_t)(s * 2 + lh) * p.cout + co];
                }
                bv = p.bias[co];
#pragma unroll
                for (int r = 0; r < 16; ++r) bias16[r] = bv;
                have = nt;
                // retire these loads HERE, inside the branch: left pending, hipcc guards the first MFMA below with a
                // vmcnt(0) that runs on every tile - and, vmcnt being in order, drains the next tile's prefetch issued just
                // above, tile after tile
                if constexpr (BF16OP) { asm volatile("" ::"v"(bq[0])); asm volatile("" ::"v"(bq[1])); }
                else {
#pragma unroll
                    for (int s = 0; s < 14; ++s) asm volatile("" ::"v"(b[s]));
                }
                asm volatile("" ::"v"(bv));
            }
            // (the accumulators are never initialised: the FIRST k-step takes the bias splat `bias16` as its C operand - the same
            // arithmetic as starting from the bias, without 64 v_mov per tile on the pipe the MFMAs use)
            f32x16 acc[MTW];
            // A operands through a two-deep register pipeline (4 LDS values per k-step), fenced per step: left alone the
            // compiler hoists all 56 reads of a tile into registers and spills
            const int abase = (2 * wave * MTW + prow) * RS + pcol;
            if constexpr (BF16OP) {
                // per M-tile: 16 taps from LDS, rounded to bf16 in pairs, two MFMAs; the next M-tile's reads are issued first
                float g[2][16];
#pragma unroll
                for (int i = 0; i < 16; ++i) g[0][i] = tile[abase + koff[i]];
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
                    if (mt + 1 < MTW) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) g[(mt + 1) & 1][i] = tile[abase + 2 * (mt + 1) * RS + koff[i]];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const float* gm = g[mt & 1];
                    const f16x8 a0 = __builtin_bit_cast(f16x8, u32x4{vad_pack_bf16(gm[0], gm[1]), vad_pack_bf16(gm[2], gm[3]), vad_pack_bf16(gm[4], gm[5]), vad_pack_bf16(gm[6], gm[7])});
                    const f16x8 a1 = __builtin_bit_cast(f16x8, u32x4{vad_pack_bf16(gm[8], gm[9]), vad_pack_bf16(gm[10], gm[11]), vad_pack_bf16(gm[12], gm[13]), vad_pack_bf16(gm[14], gm[15])});
                    acc[mt] = MFMAB16(a0, bq[0], bias16);
                    acc[mt] = MFMAB16(a1, bq[1], acc[mt]);
                }
            } else {
            float av[2][MTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) av[0][mt] = tile[abase + 2 * mt * RS + koff[0]];
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                if (s + 1 < 14) {
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) av[(s + 1) & 1][mt] = tile[abase + 2 * mt * RS + koff[s + 1]];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) acc[mt] = MFMA32(av[s & 1][mt], b[s], s == 0 ? bias16 : acc[mt]);
            }
            }
            float ts = 0.f, tq = 0.f;        // this tile's share of the statistics
            // The epilogue in TWO copies behind ONE scalar branch: `full` is wave-uniform, but written as `if (!full) vo = ...` per
            // store hipcc if-converted it - 2 v_cmp + 2 v_cndmask in front of every one of the 64 stores of a tile, ~8 of the
            // kernel's ~13 VALU instructions per MFMA, on the pipe the exact-fp32 MFMAs use (profiles/r04_pmc_insts_winograd_image.txt).
            auto epilogue = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if constexpr (POOL) {
                            // MaxPool2d(act(.)) == act(MaxPool2d(.)) bit for bit (ReLU / LeakyReLU are non-decreasing): one activation
                            // per window instead of four; vad_vmax = one v_max_f32 (fmaxf adds a canonicalising max per operand)
                            const float m = vad_act(vad_vmax(vad_vmax(acc[mt][4 * q], acc[mt][4 * q + 1]), vad_vmax(acc[mt][4 * q + 2], acc[mt][4 * q + 3])), ACT);
                            // lane part (column half, channel) + scalar part (row, column pair); a window outside the pooled image
                            // (partial tiles only) gets the out-of-range offset and the store is dropped
                            unsigned vo = lanepart;
                            if constexpr (!FULL) vo = ((oy0 + mt) < oh && (ox0 + 2 * q + lh) < ow) ? lanepart : VAD_OOB;
                            vad_bstore1(m, rout, vo, ubase + (unsigned)mt * orow + (unsigned)(2 * q) * ocol);
                        } else {
                            // un-pooled (training forward): the window's four pixels (dy = pos >> 1, dx = pos & 1) at column 4q + 2 lh + dx
#pragma unroll
                            for (int pos = 0; pos < 4; ++pos) {
                                unsigned vo = lanepart;
                                if constexpr (!FULL) vo = ((oy0 + 2 * mt + (pos >> 1)) < oh && (ox0 + 4 * q + 2 * lh + (pos & 1)) < ow) ? lanepart : VAD_OOB;
                                if constexpr (OUT16) vad_bstore_h(vad_f_bf16(acc[mt][4 * q + pos]), rout, vo,
                                                                  ubase + (unsigned)(2 * mt + (pos >> 1)) * orow + (unsigned)(4 * q + (pos & 1)) * ocol);
                                else vad_bstore1(vad_act(acc[mt][4 * q + pos], ACT), rout, vo,
                                                 ubase + (unsigned)(2 * mt + (pos >> 1)) * orow + (unsigned)(4 * q + (pos & 1)) * ocol);
                            }
                        }
                    }
                }
                // the statistics of the training forward (un-activated launches only) in a pass of their own behind ONE uniform
                // branch per tile - tested per element the branch sat in front of every store; same terms, same order
                if constexpr (!POOL && ACT == VAD_ACT_NONE) {
                    if (p.stats) {
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int pos = 0; pos < 4; ++pos) {
                                    const bool in = FULL || ((oy0 + 2 * mt + (pos >> 1)) < oh && (ox0 + 4 * q + 2 * lh + (pos & 1)) < ow);
                                    const float d = in ? acc[mt][4 * q + pos] - bv : 0.f;
                                    ts += d;
                                    tq = fmaf(d, d, tq);
                                }
                    }
                }
            };
            if (full) epilogue(std::true_type{});
            else epilogue(std::false_type{});
            st_s += ts;
            st_q += tq;
        }
    }
    if constexpr (!POOL) {
        if (p.stats) {       // (uniform; host: cout == 32, so one channel tile and lane li = channel li)
            __shared__ float red[4][2][32];
            st_s += __shfl_xor(st_s, 32);
            st_q += __shfl_xor(st_q, 32);
            if (lh == 0) { red[wave][0][li] = st_s; red[wave][1][li] = st_q; }
            __syncthreads();
            if (tid < 64) {
                const int j = tid >> 5;
                p.stats[(size_t)blockIdx.x * 2 * p.cout + j * p.cout + li] = ((red[0][j][li] + red[1][j][li]) + red[2][j][li]) + red[3][j][li];
            }
        }
    }
}

extern "C" int vad_conv3x3_c3(const float* x, const float* w, const float* bias, float* out,
                              int n, int h, int wd, int cout, int act, int pool, void* stream) {
    return vad_conv3x3_c3_fmt(x, VAD_X_F32_NCHW, w, bias, out, n, h, wd, cout, act, pool, stream);
}

int vad_conv3x3_c3_fmt(const void* x, int fmt, const float* w, const float* bias, float* out,
                       int n, int h, int wd, int cout, int act, int pool, void* stream) {
    return vad_conv3x3_c3_stats(x, fmt, w, bias, out, n, h, wd, cout, act, pool, nullptr, nullptr, stream);
}

// stats / stats_blocks (both or neither): the persistent un-pooled kernel also writes per-work-group BatchNorm partial sums
// [blocks][2][cout] (shifted by the bias) and *stats_blocks = the number of work-groups; *stats_blocks = 0 when the launch
// could not provide them (the caller then makes its own pass over `out`).
int vad_conv3x3_c3_stats(const void* x, int fmt, const float* w, const float* bias, float* out,
                         int n, int h, int wd, int cout, int act, int pool, float* stats, int* stats_blocks, void* stream) {
    return vad_conv3x3_c3_stats_t(x, fmt, w, bias, out, 0, n, h, wd, cout, act, pool, stats, stats_blocks, stream);
}

extern "C" int vad_conv3x3_c3_bf16(const float* x, const float* w, const float* bias, void* out, int n, int h, int wd, int cout, void* stream) {
    return vad_conv3x3_c3_stats_t(x, VAD_X_F32_NCHW, w, bias, out, 1, n, h, wd, cout, VAD_ACT_NONE, 0, nullptr, nullptr, stream);
}
// ... with bf16 MFMA operands as well (the form the bf16-tensor training step runs)
extern "C" int vad_conv3x3_c3_bf16op(const float* x, const float* w, const float* bias, void* out, int n, int h, int wd, int cout, void* stream) {
    return vad_conv3x3_c3_stats_t(x, VAD_X_F32_NCHW, w, bias, out, 2, n, h, wd, cout, VAD_ACT_NONE, 0, nullptr, nullptr, stream);
}

// out16 != 0: `out` is a bf16 tensor (VAD_PREC_BF16S; persistent un-pooled un-activated launches only)
int vad_conv3x3_c3_stats_t(const void* x, int fmt, const float* w, const float* bias, void* outv, int out16,
                           int n, int h, int wd, int cout, int act, int pool, float* stats, int* stats_blocks, void* stream) {
    float* out = (float*)outv;
    if (stats_blocks) *stats_blocks = 0;
    VAD_REQUIRE((stats == nullptr) == (stats_blocks == nullptr), "conv3x3_c3: stats and stats_blocks come together");
    VAD_REQUIRE(x && w && bias && out, "conv3x3_c3: null pointer");
    VAD_REQUIRE(fmt == VAD_X_F32_NCHW || fmt == VAD_X_U8_NHWC, "conv3x3_c3: bad input format %d", fmt);
    VAD_REQUIRE(n > 0 && h > 0 && wd > 0 && cout > 0 && cout % 32 == 0, "conv3x3_c3: bad shape");
    VAD_REQUIRE(!pool || (h % 2 == 0 && wd % 2 == 0), "conv3x3_c3: pooling needs even H,W");
    VAD_REQUIRE(act == VAD_ACT_LEAKY || act == VAD_ACT_RELU || act == VAD_ACT_NONE, "conv3x3_c3: bad act");
    hipStream_t s = (hipStream_t)stream;
    const ConvKnobs kn;
    VAD_REQUIRE(!out16 || (kn.variant != 0 && !pool && act == VAD_ACT_NONE), "conv3x3_c3: bf16 output is offered by the persistent un-pooled un-activated kernel only");
    // persistent kernel (tiles of 32 rows x 16 columns) for both forms.  The un-pooled one (training forward) writes 8.4 MB per
    // 256x256 frame: 2.6 us/frame = 3.5 TB/s with the scalar-offset buffer stores (the one-tile-per-work-group kernel, with a
    // 64-bit address computed per store, needs 3.8 us; an earlier persistent form with the same address arithmetic 6.5 us).
    if (kn.variant != 0) {
        VAD_REQUIRE(12ll * h * wd < (1ll << 31) && (long long)h * wd * cout < (1ll << 31),
                    "conv3x3_c3: frame %dx%d (cout %d) too large for the 32-bit offsets inside one frame", h, wd, cout);
        ConvC3P p{(const float*)x, w, bias, out, h, wd, cout, (wd + 15) / 16, (h + 31) / 32, 0, fmt == VAD_X_U8_NHWC, nullptr};
        const long long nb = (long long)n * p.tiles_x * p.tiles_y;
        VAD_REQUIRE(nb < (1ll << 31), "conv3x3_c3: grid too large");
        p.nblocks = (unsigned)nb;
        const bool with_stats = stats && !pool && act == VAD_ACT_NONE && cout == 32;
        if (with_stats) p.stats = stats;
#define C3P_LAUNCH(POOL, ACT)                                                                              \
    {                                                                                                      \
        static std::atomic<unsigned> cap_{0};                                                              \
        unsigned cap = cap_.load(std::memory_order_relaxed);                                               \
        if (!cap) cap_ = cap = persistent_grid(conv3x3_c3_pkernel<POOL, ACT>, ~0u);                        \
        hipLaunchKernelGGL((conv3x3_c3_pkernel<POOL, ACT>), dim3(p.nblocks < cap ? p.nblocks : cap), dim3(256), 0, s, p); \
        if (with_stats) *stats_blocks = (int)(p.nblocks < cap ? p.nblocks : cap);                          \
    }
        if (pool) {
            if (act == VAD_ACT_LEAKY) C3P_LAUNCH(1, VAD_ACT_LEAKY)
            else if (act == VAD_ACT_RELU) C3P_LAUNCH(1, VAD_ACT_RELU)
            else C3P_LAUNCH(1, VAD_ACT_NONE)
        } else {
            if (act == VAD_ACT_LEAKY) C3P_LAUNCH(0, VAD_ACT_LEAKY)
            else if (act == VAD_ACT_RELU) C3P_LAUNCH(0, VAD_ACT_RELU)
            else if (out16 == 2) {       // bf16 output AND bf16 MFMA operands (float input only)
                VAD_REQUIRE(fmt == VAD_X_F32_NCHW, "conv3x3_c3: the bf16-operand form takes float frames");
                static std::atomic<unsigned> cap_{0};
                unsigned cap = cap_.load(std::memory_order_relaxed);
                if (!cap) cap_ = cap = persistent_grid(conv3x3_c3_pkernel<0, VAD_ACT_NONE, 1, 1>, ~0u);
                hipLaunchKernelGGL((conv3x3_c3_pkernel<0, VAD_ACT_NONE, 1, 1>), dim3(p.nblocks < cap ? p.nblocks : cap), dim3(256), 0, s, p);
                if (with_stats) *stats_blocks = (int)(p.nblocks < cap ? p.nblocks : cap);
            }
            else if (out16) {
                static std::atomic<unsigned> cap_{0};
                unsigned cap = cap_.load(std::memory_order_relaxed);
                if (!cap) cap_ = cap = persistent_grid(conv3x3_c3_pkernel<0, VAD_ACT_NONE, 1>, ~0u);
                hipLaunchKernelGGL((conv3x3_c3_pkernel<0, VAD_ACT_NONE, 1>), dim3(p.nblocks < cap ? p.nblocks : cap), dim3(256), 0, s, p);
                if (with_stats) *stats_blocks = (int)(p.nblocks < cap ? p.nblocks : cap);
            }
            else C3P_LAUNCH(0, VAD_ACT_NONE)
        }
#undef C3P_LAUNCH
        VAD_LAUNCH_CHECK();
        return VAD_OK;
    }
    ConvC3P p{(const float*)x, w, bias, out, h, wd, cout, (wd + 15) / 16, (h + 15) / 16, 0, fmt == VAD_X_U8_NHWC, nullptr};
    const long long nb = (long long)n * p.tiles_x * p.tiles_y;
    VAD_REQUIRE(nb < (1ll << 31), "conv3x3_c3: grid too large");
    p.nblocks = (unsigned)nb;
    dim3 g((unsigned)nb), b(256);
#define C3_LAUNCH(POOL, ACT) hipLaunchKernelGGL((conv3x3_c3_kernel<2, POOL, ACT>), g, b, 0, s, p)
    if (pool) {
        if (act == VAD_ACT_LEAKY) C3_LAUNCH(1, VAD_ACT_LEAKY);
        else if (act == VAD_ACT_RELU) C3_LAUNCH(1, VAD_ACT_RELU);
        else C3_LAUNCH(1, VAD_ACT_NONE);
    } else {
        if (act == VAD_ACT_LEAKY) C3_LAUNCH(0, VAD_ACT_LEAKY);
        else if (act == VAD_ACT_RELU) C3_LAUNCH(0, VAD_ACT_RELU);
        else C3_LAUNCH(0, VAD_ACT_NONE);
    }
#undef C3_LAUNCH
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// ---------------------------------------------------------------------------------------------
// ConvTranspose2d k2 s2 (and 1x1 conv when UPS == 0) as a GEMM over flattened input pixels:
// out[n, 2y+a, 2x+b, co] = bias[co] + sum_ci in[n,y,x,ci] * W[ci,co,a,b]; N index = (quadrant, co).
struct ConvTP {
    const float* in; long long in_fs;
    const float* w; const float* bias;
    float* out; long long out_fs;
    int h, w_, cin, cout;
    long long npix;          // n*h*w
    int nblocks_n;           // column blocks
    unsigned nblocks;
};

template <int CK, int MT, int NT, int WM, int WN, int UPS, int ACT>
__global__ __launch_bounds__(256) void convt2x2_mfma_kernel(ConvTP p) {
    constexpr int TM = 32 * MT * WM, PS = CK + 4;
    __shared__ __attribute__((aligned(16))) float tile[TM * PS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    unsigned L = vad_xcd_remap(blockIdx.x, p.nblocks);
    const int nb_n = L % p.nblocks_n;
    const long long m0 = (long long)(L / p.nblocks_n) * TM;
    const int hw = p.h * p.w_;
    const int ctiles = p.cout / 32;

    const float* wp[NT];
    float bv[NT];
    int qd[NT], cob[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int g = (nb_n * WN + wn) * NT + nt;
        qd[nt] = g / ctiles;
        cob[nt] = (g % ctiles) * 32 + li;
        wp[nt] = p.w + ((size_t)qd[nt] * (p.cin / 8) * p.cout + cob[nt]) * 8 + 4 * lh;
        bv[nt] = p.bias[cob[nt]];
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bv[nt];

    const size_t wstep = (size_t)p.cout * 8;
    for (int ch = 0; ch < p.cin / CK; ++ch) {
        __syncthreads();
        for (int idx = tid; idx < TM * (CK / 4); idx += 256) {
            const int c4 = idx % (CK / 4), pm = idx / (CK / 4);
            const long long m = m0 + pm;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < p.npix) {
                const long long nn = m / hw;
                const int rem = (int)(m - nn * hw);
                v = *(const f32x4*)(p.in + (size_t)nn * p.in_fs + (size_t)rem * p.cin + ch * CK + c4 * 4);
            }
            *(f32x4*)&tile[pm * PS + c4 * 4] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k8 = 0; k8 < CK / 8; ++k8) {
            f32x4 a[MT], b[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = *(const f32x4*)&tile[((wm * MT + mt) * 32 + li) * PS + k8 * 8 + 4 * lh];
            const size_t woff = (size_t)(ch * (CK / 8) + k8) * wstep;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = *(const f32x4*)(wp[nt] + woff);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = MFMA32(a[mt][j], b[nt][j], acc[mt][nt]);
        }
    }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const long long m = m0 + (wm * MT + mt) * 32 + row;
            if (m < p.npix) {
                const long long nn = m / hw;
                const int rem = (int)(m - nn * hw);
                const int y = rem / p.w_, x = rem - y * p.w_;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float v = vad_act(acc[mt][nt][r], ACT);
                    size_t o;
                    if (UPS) {
                        const int oy = 2 * y + (qd[nt] >> 1), ox = 2 * x + (qd[nt] & 1);
                        o = (size_t)nn * p.out_fs + ((size_t)oy * (2 * p.w_) + ox) * p.cout + cob[nt];
                    } else {
                        o = (size_t)nn * p.out_fs + (size_t)rem * p.cout + cob[nt];
                    }
                    p.out[o] = v;
                }
            }
        }
    }
}

template <int UPS>
static int launch_convt(ConvTP& p, int act, hipStream_t s) {
    // 128 pixels x 128 columns per work-group; the column count is (UPS ? 4 : 1) * cout
    const int ncols = (UPS ? 4 : 1) * p.cout;
    constexpr int TM = 128;
    const long long mblocks = (p.npix + TM - 1) / TM;
    dim3 b(256);
    if (ncols % 128 == 0) {
        p.nblocks_n = ncols / 128;
        const long long nb = mblocks * p.nblocks_n;
        VAD_REQUIRE(nb < (1ll << 31), "convt: grid too large");
        p.nblocks = (unsigned)nb;
        dim3 g((unsigned)nb);
        if (act == VAD_ACT_RELU) hipLaunchKernelGGL((convt2x2_mfma_kernel<32, 2, 2, 2, 2, UPS, VAD_ACT_RELU>), g, b, 0, s, p);
        else if (act == VAD_ACT_LEAKY) hipLaunchKernelGGL((convt2x2_mfma_kernel<32, 2, 2, 2, 2, UPS, VAD_ACT_LEAKY>), g, b, 0, s, p);
        else hipLaunchKernelGGL((convt2x2_mfma_kernel<32, 2, 2, 2, 2, UPS, VAD_ACT_NONE>), g, b, 0, s, p);
    } else {
        // narrow outputs (1x1 conv with cout 32/64/96): 128 pixels x 32 columns
        p.nblocks_n = ncols / 32;
        const long long nb = mblocks * p.nblocks_n;
        VAD_REQUIRE(nb < (1ll << 31), "convt: grid too large");
        p.nblocks = (unsigned)nb;
        dim3 g((unsigned)nb);
        if (act == VAD_ACT_RELU) hipLaunchKernelGGL((convt2x2_mfma_kernel<32, 1, 1, 4, 1, UPS, VAD_ACT_RELU>), g, b, 0, s, p);
        else if (act == VAD_ACT_LEAKY) hipLaunchKernelGGL((convt2x2_mfma_kernel<32, 1, 1, 4, 1, UPS, VAD_ACT_LEAKY>), g, b, 0, s, p);
        else hipLaunchKernelGGL((convt2x2_mfma_kernel<32, 1, 1, 4, 1, UPS, VAD_ACT_NONE>), g, b, 0, s, p);
    }
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_convt2x2(const float* in, long long in_fs, const float* w, const float* bias,
                            float* out, long long out_fs, int n, int h, int wd, int cin, int cout,
                            int act, int precision, void* stream) {
    return vad_convt2x2_stats(in, in_fs, w, bias, out, out_fs, n, h, wd, cin, cout, act, precision, nullptr, nullptr, stream);
}

// upper bound of the rows x 2 x cout floats a stats launch writes (one row per wave of a persistent grid, <= 4 work-groups per CU)
size_t vad_convt2x2_stats_floats(int cout) { return (size_t)vad_num_cus() * 4 * 4 * 2 * (size_t)cout; }

// stats / stats_rows as in vad_conv3x3_stats (un-activated launches, cout <= 128): [rows][2][cout], shifted by the bias
int vad_convt2x2_stats(const float* in, long long in_fs, const float* w, const float* bias,
                       float* out, long long out_fs, int n, int h, int wd, int cin, int cout,
                       int act, int precision, float* stats, int* stats_rows, void* stream) {
    if (stats_rows) *stats_rows = 0;
    VAD_REQUIRE((stats == nullptr) == (stats_rows == nullptr), "convt2x2: stats and stats_rows come together");
    VAD_REQUIRE(in && w && bias && out, "convt2x2: null pointer");
    VAD_REQUIRE_PREC("convt2x2");
    const ConvKnobs kn;
    VAD_REQUIRE(n > 0 && h > 0 && wd > 0, "convt2x2: bad shape");
    VAD_REQUIRE(cin % 32 == 0 && cout % 32 == 0 && cin > 0 && cout > 0,
                "convt2x2: cin=%d cout=%d must be positive multiples of 32", cin, cout);
    VAD_REQUIRE(act >= 0 && act <= 2, "convt2x2: bad act");
    VAD_REQUIRE(precision != VAD_PREC_BF16S || act == VAD_ACT_NONE, "convt2x2: VAD_PREC_BF16S (bf16 tensors) is the training form: no activation");
    ConvTP p{};
    p.in = in; p.in_fs = in_fs ? in_fs : (long long)h * wd * cin;
    p.w = w; p.bias = bias; p.out = out;
    p.out_fs = out_fs ? out_fs : (long long)4 * h * wd * cout;
    p.h = h; p.w_ = wd; p.cin = cin; p.cout = cout; p.npix = (long long)n * h * wd;
    if (kn.variant == 0 && precision == VAD_PREC_FP32) return launch_convt<1>(p, act, (hipStream_t)stream);
    // wave-persistent LDS-free kernel (convt_pkernel.h): 32-bit offsets inside one frame
    VAD_REQUIRE(4ll * h * wd * cout * 4 < (1ll << 31) && (long long)h * wd * cin * 4 < (1ll << 31) && 4ll * cin * cout * 4 < (1ll << 31),
                "convt2x2: frame %dx%d (cin %d, cout %d) too large for 32-bit offsets", h, wd, cin, cout);
    ConvTP2 q{};
    q.in = p.in; q.in_fs = p.in_fs; q.w = w; q.bias = bias; q.out = out; q.out_fs = p.out_fs;
    q.n = n; q.h = h; q.w_ = wd; q.cin = cin; q.cout = cout;
    const int prec = precision;                      // split-fp16 operands: two accumulator sets, so half the columns per wave (bf16 shares the tiling)
    // wave tile 2 x 4 (exact) measured best of {2x4, 1x4, 2x2, 1x2}: dec4.0 2.18 / 2.47 / 2.28 / 2.78 us per frame
    int mt = 2, nt = prec ? 2 : 4;
    // latency tiling (see vad_conv3x3): fewer wave items than the chip has SIMDs -> one 32-pixel x 32-column tile per item
    // (dec1.0 on one 256x256 frame: 16 items of 27 us each -> 128 items); exact fp32, activated launches = the scoring path
    const bool small_ct = prec == VAD_PREC_FP32 && act == VAD_ACT_RELU &&
                          (long long)n * ((wd + 15) / 16) * ((h + 3) / 4) * (4 * cout / 128) < 4ll * vad_num_cus();
    if (small_ct) { mt = 1; nt = 1; }
    q.tiles_x = (wd + 15) / 16; q.tiles_y = (h + 2 * mt - 1) / (2 * mt);
    q.ngroups = 4 * cout / (32 * nt);
    if (prec == VAD_PREC_BF16S) {          // bf16 tensors in, bf16 tensors out (un-activated: host-checked above)
        const long long items16 = (long long)n * q.tiles_x * q.tiles_y * q.ngroups;
        VAD_REQUIRE(items16 > 0 && items16 < (1ll << 31), "convt2x2: %lld work items out of range", items16);
        q.nitems = (unsigned)items16;
        const unsigned want16 = (unsigned)((items16 + 3) / 4);
        const bool st16 = stats && cout <= 128;
        q.stats = st16 ? stats : nullptr;
        if (st16) {
            static std::atomic<unsigned> cap_{0};
            unsigned cap = cap_.load(std::memory_order_relaxed);
            if (!cap) cap_ = cap = persistent_grid(convt2x2_pkernel<2, 2, VAD_ACT_NONE, 2, 1, 1, 1>, ~0u);
            hipLaunchKernelGGL((convt2x2_pkernel<2, 2, VAD_ACT_NONE, 2, 1, 1, 1>), dim3(want16 < cap ? want16 : cap), dim3(256), 0, (hipStream_t)stream, q);
            *stats_rows = (int)((want16 < cap ? want16 : cap) * 4);
        } else {
            static std::atomic<unsigned> cap_{0};
            unsigned cap = cap_.load(std::memory_order_relaxed);
            if (!cap) cap_ = cap = persistent_grid(convt2x2_pkernel<2, 2, VAD_ACT_NONE, 2, 0, 1, 1>, ~0u);
            hipLaunchKernelGGL((convt2x2_pkernel<2, 2, VAD_ACT_NONE, 2, 0, 1, 1>), dim3(want16 < cap ? want16 : cap), dim3(256), 0, (hipStream_t)stream, q);
        }
        VAD_LAUNCH_CHECK();
        return VAD_OK;
    }
    const long long items = (long long)n * q.tiles_x * q.tiles_y * q.ngroups;
    VAD_REQUIRE(items > 0 && items < (1ll << 31), "convt2x2: %lld work items out of range", items);
    q.nitems = (unsigned)items;
    const unsigned want = (unsigned)((items + 3) / 4);
    const bool with_stats = stats && act == VAD_ACT_NONE && cout <= 128;
    q.stats = with_stats ? stats : nullptr;
#define CT_LAUNCH_ST(A, MT_, NT_, P, ST)                                                                     \
    {                                                                                                        \
        static std::atomic<unsigned> cap_{0};                                                                \
        unsigned cap = cap_.load(std::memory_order_relaxed);                                                 \
        if (!cap) cap_ = cap = persistent_grid(convt2x2_pkernel<MT_, NT_, A, P, ST>, ~0u);                   \
        hipLaunchKernelGGL((convt2x2_pkernel<MT_, NT_, A, P, ST>), dim3(want < cap ? want : cap), dim3(256), 0, (hipStream_t)stream, q); \
        if (ST) *stats_rows = (int)((want < cap ? want : cap) * 4);                                          \
    }
#define CT_LAUNCH_(A, MT_, NT_, P) CT_LAUNCH_ST(A, MT_, NT_, P, 0)
#define CT_LAUNCH(A)                                                                                         \
    {                                                                                                        \
        if (prec == VAD_PREC_SPLIT) CT_LAUNCH_(A, 2, 2, 1)                                                   \
        else if (prec == VAD_PREC_BF16) CT_LAUNCH_(A, 2, 2, 2)                                               \
        else CT_LAUNCH_(A, 2, 4, 0)                                                                          \
    }
    if (small_ct) CT_LAUNCH_(VAD_ACT_RELU, 1, 1, 0)
    else if (act == VAD_ACT_RELU) CT_LAUNCH(VAD_ACT_RELU)
    else if (act == VAD_ACT_LEAKY) CT_LAUNCH(VAD_ACT_LEAKY)
    else if (with_stats) {       // the statistics epilogue is its own instantiation (un-activated launches only)
        if (prec == VAD_PREC_SPLIT) CT_LAUNCH_ST(VAD_ACT_NONE, 2, 2, 1, 1)
        else if (prec == VAD_PREC_BF16) CT_LAUNCH_ST(VAD_ACT_NONE, 2, 2, 2, 1)
        else CT_LAUNCH_ST(VAD_ACT_NONE, 2, 4, 0, 1)
    } else CT_LAUNCH(VAD_ACT_NONE)
#undef CT_LAUNCH_ST
#undef CT_LAUNCH_
#undef CT_LAUNCH
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

// 1x1 convolution on bf16 tensors with bf16 operands (VAD_PREC_BF16S; weights from vad_train_pack_conv1x1_p): the UPS 0 form
// of the wave-persistent transposed-convolution kernel.  The pixel list is presented as frames of h x w pixels (w a
// multiple of 16) so that its 2-row x 16-column M-tiles are full.  Any other precision: the fp32 kernel below.
int vad_conv1x1_p(const void* in, const float* w, const float* bias, void* out, long long npix, int cin, int cout, int precision, void* stream) {
    if (precision != VAD_PREC_BF16S) return vad_conv1x1((const float*)in, w, bias, (float*)out, npix, cin, cout, stream);
    VAD_REQUIRE(in && w && bias && out, "conv1x1: null pointer");
    VAD_REQUIRE(npix > 0 && cin % 32 == 0 && cout % 32 == 0 && cin > 0 && cout > 0, "conv1x1 (bf16): channels must be multiples of 32");
    int wd = 16, h = 1;
    long long n = 1;
    int frame_pix = 0;
    if (npix % 16 == 0) {
        for (int c : {64, 32}) if (npix % c == 0) { wd = c; break; }
        const long long rows = npix / wd;
        for (int c : {256, 128, 64, 32, 16, 8, 4, 2}) if (rows % c == 0) { h = c; break; }
        n = rows / h;
    } else {                                   // ragged: one frame of ceil(npix / 16) rows whose last row is partly valid
        const long long rows = (npix + 15) / 16;
        VAD_REQUIRE(rows < (1 << 24), "conv1x1 (bf16): %lld pixels (not a multiple of 16) are too many for one ragged frame", npix);
        h = (int)rows;
        frame_pix = (int)npix;
    }
    VAD_REQUIRE(n < (1ll << 31) && (long long)h * wd * (cin > cout ? cin : cout) * 2 < (1ll << 31), "conv1x1 (bf16): shape out of range");
    ConvTP2 q{};
    q.frame_pix = frame_pix;
    q.in = (const float*)in; q.in_fs = (long long)h * wd * cin; q.w = w; q.bias = bias; q.out = (float*)out; q.out_fs = (long long)h * wd * cout;
    q.n = (int)n; q.h = h; q.w_ = wd; q.cin = cin; q.cout = cout;
    q.tiles_x = wd / 16; q.tiles_y = (h + 3) / 4;
    q.ngroups = cout / 64;
    const bool narrow = cout % 64 != 0;        // 32 / 96 columns: one N-tile per item
    if (narrow) q.ngroups = cout / 32;
    const long long items = n * q.tiles_x * q.tiles_y * q.ngroups;
    VAD_REQUIRE(items > 0 && items < (1ll << 31), "conv1x1 (bf16): %lld work items out of range", items);
    q.nitems = (unsigned)items;
    const unsigned want = (unsigned)((items + 3) / 4);
    if (narrow) {
        static std::atomic<unsigned> cap_{0};
        unsigned cap = cap_.load(std::memory_order_relaxed);
        if (!cap) cap_ = cap = persistent_grid(convt2x2_pkernel<2, 1, VAD_ACT_NONE, 2, 0, 1, 0>, ~0u);
        hipLaunchKernelGGL((convt2x2_pkernel<2, 1, VAD_ACT_NONE, 2, 0, 1, 0>), dim3(want < cap ? want : cap), dim3(256), 0, (hipStream_t)stream, q);
    } else {
        static std::atomic<unsigned> cap_{0};
        unsigned cap = cap_.load(std::memory_order_relaxed);
        if (!cap) cap_ = cap = persistent_grid(convt2x2_pkernel<2, 2, VAD_ACT_NONE, 2, 0, 1, 0>, ~0u);
        hipLaunchKernelGGL((convt2x2_pkernel<2, 2, VAD_ACT_NONE, 2, 0, 1, 0>), dim3(want < cap ? want : cap), dim3(256), 0, (hipStream_t)stream, q);
    }
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_conv1x1(const float* in, const float* w, const float* bias, float* out,
                           long long npix, int cin, int cout, void* stream) {
    VAD_REQUIRE(in && w && bias && out, "conv1x1: null pointer");
    VAD_REQUIRE(npix > 0 && cin % 32 == 0 && cout % 32 == 0 && cin > 0 && cout > 0, "conv1x1: bad shape");
    ConvTP p{};
    // one "frame" of npix x 1 pixels
    p.in = in; p.in_fs = 0; p.w = w; p.bias = bias; p.out = out; p.out_fs = 0;
    p.h = 1; p.w_ = (int)((npix < (1ll << 30)) ? npix : 0); p.cin = cin; p.cout = cout; p.npix = npix;
    VAD_REQUIRE(p.w_ > 0, "conv1x1: npix too large");
    return launch_convt<0>(p, VAD_ACT_NONE, (hipStream_t)stream);
}
