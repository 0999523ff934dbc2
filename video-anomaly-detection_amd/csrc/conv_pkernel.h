// Persistent form of the 3x3 implicit-GEMM convolution (included by conv_mfma.hip; same operand
// layouts as conv3x3_mfma_kernel there).
//
// A work-group owns ONE spatial tile position and ONE output-channel block and walks the frames
// n = g, g + G, g + 2G, ... of the batch.  Everything that depends on geometry only (which pixels
// of the halo tile exist, their offsets inside a frame, the lane's weight rows and bias, the
// epilogue's store offsets) is computed once; per frame only a base pointer moves.
// (frame, 32-channel chunk) pairs form a flat sequence of stages:
//   * the NEXT stage's input tile is fetched global -> registers right after the barrier that
//     publishes the current one (issue early, ds_write late), so its HBM/L2 latency runs under
//     ~36 MFMA steps instead of stalling every wave of the CU at once;
//   * fragments for step s+1 are loaded before the MFMAs of step s (sched_barrier keeps hipcc
//     from sinking the loads back to their first use);
//   * the next frame's first B fragments are requested BEFORE this frame's stores, because vmcnt
//     retires in order and loads issued behind the stores would wait for every store to be acked.
// All addressing is 32-bit (24-bit multiplies, byte offsets from wave-uniform bases); the host
// checks that a frame of any tensor and the weight blob stay below 2^31 bytes.
#pragma once

__device__ __forceinline__ void pf_set(float& d, float v) { d = v; }
__device__ __forceinline__ void pf_set(float& d, f32x4 v) { d = v[0]; }
__device__ __forceinline__ void pf_set(f32x4& d, f32x4 v) { d = v; }
__device__ __forceinline__ void pf_set(f32x4& d, float v) { d = f32x4{v, v, v, v}; }
__device__ __forceinline__ float pf_get_f(float v) { return v; }
__device__ __forceinline__ float pf_get_f(f32x4 v) { return v[0]; }
__device__ __forceinline__ f32x4 pf_get_v(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 pf_get_v(float v) { return f32x4{v, v, v, v}; }
// (bf16 tensors: a staging slot is 4 channels = 8 bytes, kept as the raw bit patterns)
__device__ __forceinline__ void pf_set(u32x2& d, u32x2 v) { d = v; }
__device__ __forceinline__ void pf_set(u32x2& d, float) {}
__device__ __forceinline__ void pf_set(u32x2& d, f32x4) {}
__device__ __forceinline__ void pf_set(float& d, u32x2) {}
__device__ __forceinline__ void pf_set(f32x4& d, u32x2) {}
__device__ __forceinline__ float pf_get_f(u32x2 v) { return __uint_as_float(v[0]); }
__device__ __forceinline__ f32x4 pf_get_v(u32x2 v) { return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), 0.f, 0.f}; }
__device__ __forceinline__ u32x2 pf_get_u(u32x2 v) { return v; }
__device__ __forceinline__ u32x2 pf_get_u(float) { return u32x2{0u, 0u}; }
__device__ __forceinline__ u32x2 pf_get_u(f32x4) { return u32x2{0u, 0u}; }

#ifdef VAD_STAMPS
#define STAMP(k)                                                                                      \
    {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (st_prev) st_sum[k] += t_ - st_prev;                                                       \
        st_prev = t_;                                                                                 \
        if ((k) == 5) ++st_n;                                                                         \
    }
#else
#define STAMP(k)
#endif

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// fp32 value -> (hi, lo) fp16 pair with hi + lo * 2^-11 == v to 22 significant bits: hi = fp16(v),
// lo = fp16((v - hi) * 2^11).  fp16 x fp16 products are exact in fp32, so
//   a*b = ah*bh + (ah*bl + al*bh) * 2^-11 + O(2^-22 |a*b|)
// with fp32 accumulation: three fp16 MFMAs (16x the fp32 pipe rate each) replace one fp32 MFMA.  |v| must stay
// below 65504 (fp16 range); activations of these networks are O(1..100).
__device__ __forceinline__ void vad_split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * 2048.0f);
}

// PREC 2 (training only): bf16 operands, ONE v_mfma_f32_32x32x16_bf16 per step, fp32 accumulate.  It rides on the
// split-fp16 data path: the same [8 x hi | 8 x lo] 16-bit slots per 8-channel block, with hi = the bf16 bit pattern of v
// (round to nearest even) and the lo slots unused, so staging, fragment loads and the packed-weight layout are shared and
// only the conversion and the MFMA differ.  8 significant bits: no score-parity claim, gated by the loss-curve test.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMAB16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
template <int PREC>
__device__ __forceinline__ void vad_split_p(float v, _Float16& hi, _Float16& lo) {
    if constexpr (PREC == 2) {
        hi = __builtin_bit_cast(_Float16, (__bf16)v);
        lo = (_Float16)0.f;
    } else {
        vad_split(v, hi, lo);
    }
}
// one k-step of 16 channels for one (M-tile, N-tile) pair
template <int PREC>
__device__ __forceinline__ void vad_mma16(f32x16& acc, f32x16& corr, f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl) {
    if constexpr (PREC == 2) {
        acc = MFMAB16(ah, bh, acc);
    } else {
        acc = MFMA16(ah, bh, acc);
        corr = MFMA16(ah, bl, corr);
        corr = MFMA16(al, bh, corr);
    }
}

// store one activation (channel ch of the pixel whose tile float offset is pixoff): fp32, or its (hi, lo) 16-bit pair
template <int PREC>
__device__ __forceinline__ void tile_put_t(float* tile, int pixoff_plus_ch, int ch, float v) {
    if constexpr (PREC) {
        _Float16 hi, lo;
        vad_split_p<PREC>(v, hi, lo);
        _Float16* blk = (_Float16*)&tile[pixoff_plus_ch - ch + (ch >> 3) * 8];   // 8-channel block = 32 bytes = 8 floats
        blk[ch & 7] = hi;
        blk[8 + (ch & 7)] = lo;
    } else {
        tile[pixoff_plus_ch] = v;
    }
}

// IO16 (PREC 2, plain un-activated launches = the training convolutions of VAD_PREC_BF16S): input and output tensors are
// bf16 in memory.  Staging is then a COPY - a thread's 4 channels are 8 bytes, written to the slot the conversion path
// writes them to - and the epilogue rounds the fp32 accumulators to bf16 (the statistics still come from the accumulators).
template <int CK, int MT, int NT, int WM, int WN, int MODE, int ACT, int FUSE_C3 = 0, int PREC = 0, int WITH_STATS = 0, int IO16 = 0>
// Work-groups per CU: 3 for the exact-fp32 fused first layer (it fits 168 registers; its two barriers per tile need the
// third resident group), else 2 (three groups measured no gain on the cout-32 tilings, and the deeper weight prefetch below
// needs the registers).
__global__ __launch_bounds__(256, (FUSE_C3 && !PREC) ? 3 : 2) void conv3x3_mfma_pkernel(Conv3P p) {
    static_assert(WM * WN == 4, "4 waves per work-group");
    static_assert(MODE != MODE_LSTM || NT == 4, "LSTM mode: one N-tile per gate");
    static_assert(!FUSE_C3 || CK == 32, "fused first layer produces exactly one 32-channel chunk");
    static_assert(CK == 32, "staging slots assume 8 channel quads per pixel");
    static_assert(!IO16 || (PREC == 2 && MODE == MODE_PLAIN && !FUSE_C3), "bf16 tensors: bf16 operands, plain convolution");
    constexpr unsigned ES = IO16 ? 2u : 4u;         // bytes per activation element in memory
    constexpr int TH = 2 * MT * WM, LH = TH + 2, LW = 18, PS = CK + 4;
    // FUSE_C3: the staged data are the 3 NCHW input planes of the tile with a 2-pixel halo (scalar floats);
    // otherwise float4 channel quads of the NHWC tile with a 1-pixel halo.
    constexpr int XH = LH + 2, XW = 20, XS = 20;
    constexpr int NPIX = LH * LW, NT0 = (NPIX + 31) / 32;
    constexpr int TOT = FUSE_C3 ? 3 * XH * XW : NPIX * (CK / 4), NPF = (TOT + 255) / 256;
    // PREC 0: exact fp32 MFMA (32x32x2, 8 channels per step).  PREC 1: split-fp16 MFMA (32x32x16, 16 channels per
    // step, 3 MFMAs): the LDS tile then holds, per pixel and 8-channel block, [8 x hi fp16 | 8 x lo fp16] (the same
    // 32 bytes), converted from the fp32 activations while staging; HBM formats are unchanged.
    constexpr int KS = PREC ? 16 : 8;
    constexpr int NS = 9 * (CK / KS);
    // B (weight) fragments come from L2 and are requested PB steps ahead into a ring of NB register sets; the fp16
    // steps are 5x shorter than the fp32 ones, so they need the deeper prefetch to cover an L2 round trip.
    // Exact fp32: a step is 1,000+ cycles, one step ahead covers L2 - but vmcnt counts loads AND stores in order (gfx9), so
    // the weight fragments requested after a tile's epilogue wait behind all of its stores.  The fragments of the next tile's
    // first PB steps therefore go out BEFORE the stores (section "next frame's first B fragments" below): with PB = 1 only
    // step 0 ran while the stores drained and the un-pooled layers (4x the stores of the pooled ones) lost 10-20 % there.
    // (The same in-order rule couples the weight stream to the input-tile prefetch - weight loads issued after ISSUE() retire
    // behind it - but a depth of 8 on the cout-32 tiling, enough to cover an HBM round trip, measured no different from 3.)
    constexpr int PB32 = (MODE == MODE_POOL || FUSE_C3) ? 1 : 3;
    constexpr int PB = (PREC == 2) ? 5 : (PREC && NT <= 2) ? 2 : PREC ? 1 : PB32, NB = PB + 1;   // bf16 steps are a third of the split ones again
    static_assert(NS % 2 == 0 && NS % NB == 0, "fragment ring parity must be the same in every chunk");
    __shared__ __attribute__((aligned(16))) float tile[NPIX * PS];
    // raw input halo of the fused first layer.  Two barriers per frame, not three: the NEXT frame's halo is written right
    // behind the barrier that ends this frame's first stage (every wave is done reading the buffer) and is published by the
    // barrier at the top of the next frame.  (A second buffer would let it go out earlier still, but 4.8 KB more LDS per
    // work-group drops the kernel from three work-groups per CU to two.)
    __shared__ float xin[FUSE_C3 ? 3 * XH * XS : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int prow = (li >> 1) & 1, pcol = 2 * (li >> 2) + (li & 1);

    // ---- geometry of this work-group (fixed for its whole life)
    const unsigned per_frame = (unsigned)(p.tiles_x * p.tiles_y * p.cblocks);
    unsigned L = vad_xcd_remap(blockIdx.x % per_frame, per_frame);
    const int fg = blockIdx.x / per_frame, fgroups = gridDim.x / per_frame;
    const int cb = L % p.cblocks; L /= p.cblocks;
    // BatchNorm statistics of the training forward (un-pooled, un-activated launches only): taken from the accumulators, two
    // levels (per frame tile, then per work-group), shifted by the bias; row of this work-group in p.stats
    // (WITH_STATS: its own instantiation - the accumulators live across the frame loop and cost the training tilings, which
    // are at their register limit, another 30-70 bytes of scratch when they are compiled in everywhere)
    constexpr bool STATS = WITH_STATS && MODE == MODE_PLAIN && ACT == VAD_ACT_NONE && !FUSE_C3;
    const unsigned stats_row = (unsigned)(blockIdx.x / per_frame) * (unsigned)(p.tiles_x * p.tiles_y) + L;
    float st_s[STATS ? NT : 1], st_q[STATS ? NT : 1];
#pragma unroll
    for (int nt = 0; nt < (STATS ? NT : 1); ++nt) st_s[nt] = st_q[nt] = 0.f;
    const int x0 = (L % p.tiles_x) * 16, y0 = (L / p.tiles_x) * TH;
    const int H = p.h, W = p.w_;

    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        abase[mt] = ((2 * (wm * MT + mt) + prow) * LW + pcol) * PS + (PREC ? 8 : 4) * lh;

    const int nch_a = p.cin_a / CK;
    const int nch = (p.in2 ? p.cin : p.cin_a) / CK;
    const unsigned wstep = (unsigned)p.cout * (PREC ? 64u : 32u);  // bytes per (tap, k-step) slab
    const unsigned wtap = (unsigned)(p.cin / KS) * wstep;          // bytes per tap
    const __amdgpu_buffer_rsrc_t rw = vad_rsrc(p.w, (p.stagger & 1) ? 0u : 9u * wtap);   // stagger bit 0 (debug): price the weight traffic

    // staging slots: slot i of this thread is float4 (or float) number tid + 256 i of the staged tile.
    // svo[i] = byte offset of the slot inside one frame of the source (channel chunk 0), or VAD_OOB when the
    // slot is zero padding / past the end of the tile: the buffer load then returns 0 by itself.
    unsigned svo[NPF];
    int slds0;                                                     // LDS float offset of slot 0; slot i adds a constant
    const int c4 = tid & 7;
    if (FUSE_C3) {
        slds0 = 0;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + 256 * i;
            const int lx = idx % XW, t = idx / XW, ly = t % XH, c = t / XH;
            const int gy = y0 - 2 + ly, gx = x0 - 2 + lx;
            const bool ok = idx < TOT && gy >= 0 && gy < H && gx >= 0 && gx < W;
            svo[i] = !ok ? VAD_OOB
                         : p.xu8 ? (unsigned)(__mul24(__mul24(gy, W) + gx, 3) + c)                      // uint8 NHWC, bytes
                                 : (unsigned)(__mul24(c, __mul24(H, W)) + __mul24(gy, W) + gx) * 4u;   // float NCHW
        }
    } else {
        const int pix0 = tid >> 3;
        slds0 = pix0 * PS + c4 * 4;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int pix = pix0 + 32 * i;
            const int ly = pix / LW, lx = pix - ly * LW;
            const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
            const bool ok = pix < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
            svo[i] = ok ? (unsigned)(__mul24(__mul24(gy, W) + gx, p.cin_a) + c4 * 4) * ES : VAD_OOB;
        }
    }

    typedef typename std::conditional<FUSE_C3 != 0, float, typename std::conditional<IO16 != 0, u32x2, f32x4>::type>::type pf_t;
    pf_t pf[NPF];    // next stage's input, in flight
    // bf16 tensors (IO16): a chunk's 18 k-steps are ~1 us of matrix work, less than a trip to HBM - one chunk of prefetch left
    // ~2 us exposed per chunk (the ConvLSTM data gradient: 16 chunks, 55 us per launch for 15 us of MFMAs).  A second register
    // set puts the fetch TWO chunks ahead: chunk c+2 is requested when chunk c has been written to LDS.  (Chunk counts are
    // 1 or even; with one chunk per frame the single-set pipeline below is used.)
    pf_t pf2[IO16 ? NPF : 1];
    float cpv[MODE == MODE_LSTM ? MT : 1][16];   // ConvLSTM: previous cell state of this lane's outputs, in flight during the last chunk

    // both sources of a two-source (ConvLSTM) launch have the same channel count (host-checked), so svo serves both
    const unsigned in_bytes = FUSE_C3 ? (unsigned)(3 * H * W) * (p.xu8 ? 1u : 4u) : (unsigned)(H * W) * (unsigned)p.cin_a * ES;
#define ISSUE_TO(PF_, n_, ch_)   /* plain NHWC source (not the fused first layer) into the staging set PF_ */      \
    {                                                                                                    \
        const char* src_ = ((ch_) < nch_a) ? (const char*)p.in + ((size_t)(n_) * p.in_fs + (ch_) * CK) * ES            \
                                           : (const char*)p.in2 + ((size_t)(n_) * p.in2_fs + ((ch_) - nch_a) * CK) * ES; \
        const unsigned skip_ = (unsigned)(((ch_) < nch_a) ? (ch_) : (ch_) - nch_a) * CK * ES;            \
        const __amdgpu_buffer_rsrc_t r_ = vad_rsrc(src_, in_bytes - skip_);                              \
        _Pragma("unroll") for (int i_ = 0; i_ < NPF; ++i_) {                                             \
            if constexpr (IO16) pf_set(PF_[i_], __builtin_bit_cast(u32x2, vad_bload2(r_, svo[i_], 0)));  \
            else pf_set(PF_[i_], vad_bload4(r_, svo[i_], 0));                                            \
        }                                                                                                \
    }
#define ISSUE(n_, ch_)                                                                                   \
    {                                                                                                    \
        if constexpr (FUSE_C3) {                                                                         \
            if (p.xu8) {   /* uint8 frames: the RAW byte stays in flight; it is normalised where it is written to LDS  \
                              (arithmetic behind the load would wait for it here: five exposed round trips per tile) */ \
                const __amdgpu_buffer_rsrc_t r_ = vad_rsrc((const unsigned char*)p.in + (size_t)(n_) * p.in_fs, in_bytes); \
                _Pragma("unroll") for (int i_ = 0; i_ < NPF; ++i_)                                       \
                    pf_set(pf[i_], __uint_as_float((unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(r_, (int)svo[i_], 0, 0))); \
            } else {                                                                                     \
                const __amdgpu_buffer_rsrc_t r_ = vad_rsrc(p.in + (size_t)(n_) * p.in_fs, in_bytes);     \
                _Pragma("unroll") for (int i_ = 0; i_ < NPF; ++i_) pf_set(pf[i_], vad_bload1(r_, svo[i_], 0)); \
            }                                                                                            \
        } else {                                                                                         \
            const char* src_ = ((ch_) < nch_a) ? (const char*)p.in + ((size_t)(n_) * p.in_fs + (ch_) * CK) * ES            \
                                               : (const char*)p.in2 + ((size_t)(n_) * p.in2_fs + ((ch_) - nch_a) * CK) * ES; \
            const unsigned skip_ = (unsigned)(((ch_) < nch_a) ? (ch_) : (ch_) - nch_a) * CK * ES;        \
            const __amdgpu_buffer_rsrc_t r_ = vad_rsrc(src_, in_bytes - skip_);                          \
            _Pragma("unroll") for (int i_ = 0; i_ < NPF; ++i_) {                                         \
                if constexpr (IO16) pf_set(pf[i_], __builtin_bit_cast(u32x2, vad_bload2(r_, svo[i_], 0))); \
                else pf_set(pf[i_], vad_bload4(r_, svo[i_], 0));                                         \
            }                                                                                            \
        }                                                                                                \
    }

    // fused first layer: the staged raw values of pf go to the halo buffer `xb_` (uint8 frames are normalised here; padding
    // must be 0.0 AFTER normalisation)
#define XWRITE(xb_)                                                                                      \
    {                                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < NPF; ++i_) {                                             \
            const int idx_ = tid + 256 * i_;                                                             \
            const int lx_ = idx_ % XW, t_ = idx_ / XW;   /* t = c * XH + ly */                           \
            float xv_ = pf_get_f(pf[i_]);                                                                \
            if (p.xu8) xv_ = svo[i_] == VAD_OOB ? 0.f : vad_norm_u8(__float_as_uint(xv_));               \
            if (idx_ < TOT) (xb_)[t_ * XS + lx_] = xv_;                                                  \
        }                                                                                                \
    }

#ifdef VAD_STAMPS
    unsigned long long st_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0, st_n = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    int n = fg;
    if (n >= p.n) return;                                          // (grid never exceeds the work; defensive)
    ISSUE(n, 0);
    const bool deep = IO16 && nch >= 2 && (nch & 1) == 0;          // (uniform) two-deep staging, see pf2
    if constexpr (IO16) {
        if (deep) { ISSUE_TO(pf2, n, 1); }
    }
    if constexpr (FUSE_C3) {
        // first frame's halo (published by the barrier at the top of the frame loop), then the second frame's prefetch: from
        // here on pf always holds the frame AFTER the one being computed
        XWRITE(xin);
        if (n + fgroups < p.n) { ISSUE(n + fgroups, 0); }
    }

    // per-lane weight rows / bias of this cout block
    f32x4 a[2][MT], b[NB][NT];          // PREC 0 fragments
    f16x8 ah[PREC ? 1 : 2][MT], al[PREC ? 1 : 2][MT], bh[NB][NT], bl[NB][NT];   // PREC 1 fragments (A single-buffered, refilled in halves)
    unsigned wl[NT];
    float bv[NT];
    int cofs[NT];                                                  // output channel (element offset) of N-tile nt
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (MODE == MODE_LSTM) ? nt * p.hid + (cb * WN + wn) * 32 + li : ((cb * WN + wn) * NT + nt) * 32 + li;
        cofs[nt] = co;
        wl[nt] = PREC ? (unsigned)co * 64u + 32u * lh : (unsigned)co * 32u + 16u * lh;
        bv[nt] = p.bias[co];
    }
#define LOAD_B(buf, chunk, step)                                                                  \
    {                                                                                             \
        const unsigned woff_ = (unsigned)((step) / (CK / KS)) * wtap +                            \
                               (unsigned)((chunk) * (CK / KS) + (step) % (CK / KS)) * wstep;      \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                       \
            if constexpr (PREC) {                                                                 \
                bh[buf][nt] = __builtin_bit_cast(f16x8, vad_bload4(rw, wl[nt], woff_));           \
                bl[buf][nt] = __builtin_bit_cast(f16x8, vad_bload4(rw, wl[nt] + 16u, woff_));     \
            } else b[buf][nt] = vad_bload4(rw, wl[nt], woff_);                                    \
        }                                                                                         \
    }
#define LOAD_A_HALF(step, m0, m1)   /* PREC 1: refill M-tiles [m0, m1) of the single A buffer for `step` */ \
    {                                                                                             \
        const int toff_ = ((((step) / (CK / KS)) / 3) * LW + (((step) / (CK / KS)) % 3)) * PS +   \
                          ((step) % (CK / KS)) * 16;                                              \
        _Pragma("unroll") for (int mt = (m0); mt < (m1); ++mt) {                                  \
            ah[0][mt] = __builtin_bit_cast(f16x8, *(const f32x4*)&tile[abase[mt] + toff_]);       \
            al[0][mt] = __builtin_bit_cast(f16x8, *(const f32x4*)&tile[abase[mt] + toff_ + 4]);   \
        }                                                                                         \
    }
#define LOAD_A(buf, step)                                                                         \
    {                                                                                             \
        const int toff_ = ((((step) / (CK / KS)) / 3) * LW + (((step) / (CK / KS)) % 3)) * PS +   \
                          ((step) % (CK / KS)) * (PREC ? 16 : 8);                                 \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                       \
            if constexpr (PREC) {                                                                 \
                ah[buf][mt] = __builtin_bit_cast(f16x8, *(const f32x4*)&tile[abase[mt] + toff_]); \
                al[buf][mt] = __builtin_bit_cast(f16x8, *(const f32x4*)&tile[abase[mt] + toff_ + 4]); \
            } else a[buf][mt] = *(const f32x4*)&tile[abase[mt] + toff_];                          \
        }                                                                                         \
    }
#pragma unroll
    for (int s0 = 0; s0 < PB; ++s0) LOAD_B(s0, 0, s0);

    float b0w[(FUSE_C3 && !PREC) ? 14 : 1];      // first-stage B fragments: re-read from L1 for every tile (see the frame loop)
    f16x8 b0h[(FUSE_C3 && PREC) ? 2 : 1], b0l[(FUSE_C3 && PREC) ? 2 : 1];
    float bias0 = 0.f;
    bool interior = false;
    if constexpr (FUSE_C3) {
        if constexpr (PREC) {
            const f32x4* ws = (const f32x4*)(p.w0 + 28 * 32);          // split-fp16 copy of the first-layer weights
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                b0h[ks] = __builtin_bit_cast(f16x8, ws[((ks * 32 + li) * 2 + lh) * 2]);
                b0l[ks] = __builtin_bit_cast(f16x8, ws[((ks * 32 + li) * 2 + lh) * 2 + 1]);
            }
        } else {
            // (the 14 first-stage weights of a lane are loaded per tile, not held: with them resident the kernel needed 189
            // registers, and under the 168 that three work-groups per CU allow hipcc spilled 21 values to scratch and
            // reloaded them - staging offsets, store offsets - inside the frame loop, behind vmcnt(0) waits that also drain
            // the next tile's prefetch)
        }
        bias0 = p.b0[li];
        interior = y0 > 0 && x0 > 0 && y0 + TH < H && x0 + 16 < W;
    }
    // Touch the per-lane constants so hipcc retires their loads BEFORE the frame loop: otherwise it guards
    // MFMAs inside the loop with descending vmcnt waits that also drain the freshly issued prefetch.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(bv[nt]));
    if constexpr (FUSE_C3) {
        if constexpr (PREC) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { asm volatile("" ::"v"(b0h[ks])); asm volatile("" ::"v"(b0l[ks])); }
        }
        asm volatile("" ::"v"(bias0));
    }

    // epilogue geometry: per-lane byte offset of (mt 0, q 0, pos 0) for each N-tile, plus wave-uniform strides
    const bool full_tile = (y0 + TH <= H) && (x0 + 16 <= W);
    const int ech = (MODE == MODE_LSTM) ? p.hid : p.cout;
    const int ow = (MODE == MODE_POOL) ? (W >> 1) : W, oh = (MODE == MODE_POOL) ? (H >> 1) : H;
    const int ey0 = (MODE == MODE_POOL) ? (y0 >> 1) + wm * MT : y0 + 2 * wm * MT;      // first output row of this lane
    const int ex0 = (MODE == MODE_POOL) ? (x0 >> 1) + lh : x0 + 2 * lh;                // first output column
    const unsigned erow = (unsigned)__mul24(ow, ech) * ES, ecol = (unsigned)ech * ES;  // bytes per output row / pixel
    unsigned eoff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) eoff[nt] = (unsigned)(__mul24(__mul24(ey0, ow) + ex0, ech) + cofs[nt]) * ES;
    const unsigned out_bytes = (unsigned)__mul24(oh, ow) * (unsigned)ech * ES;

    while (true) {
        f32x16 acc[MT][NT];
        f32x16 corr[PREC == 1 ? MT : 1][PREC == 1 ? NT : 1];   // PREC 1: sum of the cross terms, scaled by 2^11
        if constexpr (PREC == 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) corr[mt][nt][r] = 0.f;
        }
        if constexpr (!FUSE_C3) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bv[nt];
        }

        const int nn = n + fgroups;
        const bool has_next = nn < p.n;

        if constexpr (IO16) {
            if (deep) {
                // one chunk: publish the staged set, request the chunk two ahead into the same set, run the 18 k-steps
                auto run_chunk = [&](int ch, pf_t (&cur)[NPF], bool more, int n2, int ch2) {
                    __syncthreads();                   // every wave is done reading the previous stage
#pragma unroll
                    for (int i = 0; i < NPF; ++i)
                        if ((tid >> 3) + 32 * i < NPIX)
                            *(u32x2*)&tile[(tid >> 3) * PS + i * 32 * PS + (c4 >> 1) * 8 + (c4 & 1) * 2] = pf_get_u(cur[i]);
                    __syncthreads();
                    if (more) { ISSUE_TO(cur, n2, ch2); }
                    LOAD_A_HALF(0, 0, MT);
                    constexpr int MH = (MT + 1) / 2;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int bcur = s % NB, bnxt = (s + PB) % NB;
                        if (s + PB < NS) { LOAD_B(bnxt, ch, s + PB); }
                        else if (ch + 1 < nch) { LOAD_B(bnxt, ch + 1, s + PB - NS); }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mt = 0; mt < MH; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                vad_mma16<PREC>(acc[mt][nt], corr[0][0], ah[0][mt], al[0][mt], bh[bcur][nt], bl[bcur][nt]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (s + 1 < NS) LOAD_A_HALF(s + 1, 0, MH);
#pragma unroll
                        for (int mt = MH; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                vad_mma16<PREC>(acc[mt][nt], corr[0][0], ah[0][mt], al[0][mt], bh[bcur][nt], bl[bcur][nt]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (s + 1 < NS) LOAD_A_HALF(s + 1, MH, MT);
                    }
                };
                for (int ch = 0; ch < nch; ch += 2) {      // the flat (frame, chunk) sequence, two sets alternating
                    const bool in2 = ch + 2 < nch, in3 = ch + 3 < nch;
                    run_chunk(ch, pf, in2 || has_next, in2 ? n : nn, in2 ? ch + 2 : 0);
                    run_chunk(ch + 1, pf2, in3 || has_next, in3 ? n : nn, in3 ? ch + 3 : 1);
                }
            }
        }
        for (int ch = (IO16 && deep) ? nch : 0; ch < nch; ++ch) {
            STAMP(0);
            __syncthreads();                       // every wave is done reading the previous stage
            STAMP(1);
            if constexpr (FUSE_C3) {
                if constexpr (!PREC) {   // wave-coalesced 128-B rows, L1 hits; nothing younger is in flight when the MFMAs wait for them
#pragma unroll
                    for (int s0 = 0; s0 < 14; ++s0) b0w[s0] = p.w0[(s0 * 2 + lh) * 32 + li];
                    __builtin_amdgcn_sched_barrier(0);
                }
                STAMP(2);
                STAMP(3);
                STAMP(7);
                // Fused first layer: Conv2d(3->32)+BN+LeakyReLU of the tile AND its halo, K = 27 padded to 28
                // (14 MFMAs per 32 pixels), written straight into the LDS tile the 32->32 convolution reads.
                // Each wave owns M-tiles wave, wave+4, wave+8 of the halo tile and runs their 14-step chains
                // INTERLEAVED (independent accumulators) at raised priority: a single dependent chain on a pipe
                // shared with another work-group's main loop advances one MFMA per two pipe slots.
                if constexpr (PREC) {
                    // split-fp16 first stage: K = 27 padded to 32 = two 16-deep steps, 3 MFMAs each (192 pipe cycles per
                    // 32 pixels instead of 896); the 16 gathered taps of a lane are split into hi / lo on the fly.
                    for (int t = wave; t < NT0; t += 4) {
                        const int q = t * 32 + li, qc = q < NPIX ? q : NPIX - 1;
                        const int abase0 = (qc / LW) * XS + (qc % LW);
                        f16x8 gh[2], gl[2];
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int k0 = 16 * ks + j, k1 = 16 * ks + 8 + j;     // lane half 0 / 1
                                const int o0 = (k0 < 27) ? ((k0 / 9) * XH + (k0 % 9) / 3) * XS + (k0 % 3) : 0;
                                const int o1 = (k1 < 27) ? ((k1 / 9) * XH + (k1 % 9) / 3) * XS + (k1 % 3) : 0;
                                const bool live = lh ? (k1 < 27) : (k0 < 27);
                                const float v = live ? xin[abase0 + (lh ? o1 : o0)] : 0.f;
                                _Float16 h_, l_;
                                vad_split(v, h_, l_);
                                gh[ks][j] = h_;
                                gl[ks][j] = l_;
                            }
                        f32x16 c0, cc;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { c0[r] = bias0; cc[r] = 0.f; }
                        __builtin_amdgcn_s_setprio(1);
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            c0 = MFMA16(gh[ks], b0h[ks], c0);
                            cc = MFMA16(gh[ks], b0l[ks], cc);
                            cc = MFMA16(gl[ks], b0h[ks], cc);
                        }
                        __builtin_amdgcn_s_setprio(0);
                        const int qb = (t * 32 + 4 * lh) * PS + li;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int q2 = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            const int ly2 = q2 / LW, lx2 = q2 - ly2 * LW;
                            const bool inside = interior || ((unsigned)(y0 - 1 + ly2) < (unsigned)H && (unsigned)(x0 - 1 + lx2) < (unsigned)W);
                            const float v = vad_act(fmaf(cc[r], 0x1p-11f, c0[r]), VAD_ACT_LEAKY);
                            if (q2 < NPIX) tile_put_t<PREC>(tile, qb + ((r & 3) + 8 * (r >> 2)) * PS, li, inside ? v : 0.f);
                        }
                    }
                } else {
                constexpr int TPW = (NT0 + 3) / 4;
                f32x16 c0[TPW];
                // bias0 in every register: the C operand of each chain's first k-step.  Built per tile (16 v_mov instead of the 48 that
                // initialising three accumulators took; kept across tiles it cost the 16 registers that three work-groups per CU do
                // not leave: 8 bytes of scratch inside the frame loop)
                f32x16 bias16;
#pragma unroll
                for (int r = 0; r < 16; ++r) bias16[r] = bias0;
                int abase0[TPW];
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const int q = (wave + 4 * u) * 32 + li, qc = q < NPIX ? q : NPIX - 1;
                    abase0[u] = (qc / LW) * XS + (qc % LW);
                }
                // the 14 k-steps in two halves: gather 7 x TPW A values, then their MFMAs (all 42 values at once cost 21 more
                // live registers than three work-groups per CU leave, and hipcc spilled loop-carried offsets instead)
#pragma unroll
                for (int h0 = 0; h0 < 14; h0 += 7) {
                    float av[TPW][7];
#pragma unroll
                    for (int u = 0; u < TPW; ++u)
#pragma unroll
                        for (int s0 = 0; s0 < 7; ++s0) {
                            const int k0 = 2 * (h0 + s0), k1 = 2 * (h0 + s0) + 1;
                            const int o0 = ((k0 / 9) * XH + (k0 % 9) / 3) * XS + (k0 % 3);
                            const int o1 = (k1 < 27) ? ((k1 / 9) * XH + (k1 % 9) / 3) * XS + (k1 % 3) : 0;
                            av[u][s0] = xin[abase0[u] + (lh ? o1 : o0)];
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int s0 = 0; s0 < 7; ++s0)
#pragma unroll
                        for (int u = 0; u < TPW; ++u) c0[u] = MFMA32(av[u][s0], b0w[h0 + s0], (h0 + s0 == 0) ? bias16 : c0[u]);   // first k-step: C = the bias splat (no accumulator initialisation)
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const int t = wave + 4 * u;
                    if (t < NT0) {   // wave-uniform
                        // straight-line stores (the tile is padded to NT0*32 pixels); only border tiles need the
                        // per-pixel inside test that provides conv #2's zero padding
                        const int qb = (t * 32 + 4 * lh) * PS + li;
                        if (interior && t < NT0 - 1) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                tile_put_t<PREC>(tile, qb + ((r & 3) + 8 * (r >> 2)) * PS, li, vad_act(c0[u][r], VAD_ACT_LEAKY));
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int q2 = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                const int ly2 = q2 / LW, lx2 = q2 - ly2 * LW;
                                const bool inside = (unsigned)(y0 - 1 + ly2) < (unsigned)H && (unsigned)(x0 - 1 + lx2) < (unsigned)W;
                                const float v = vad_act(c0[u][r], VAD_ACT_LEAKY);
                                if (q2 < NPIX) tile_put_t<PREC>(tile, qb + ((r & 3) + 8 * (r >> 2)) * PS, li, inside ? v : 0.f);
                            }
                        }
                    }
                }
                }
                STAMP(8);
                __syncthreads();
                STAMP(9);
                // NEXT frame's halo (prefetched a whole frame ago) into the buffer every wave has finished reading, then the
                // prefetch of the frame after it: in flight during the main loop below and the next first stage
                if (has_next) { XWRITE(xin); }
                if (nn + fgroups < p.n) { ISSUE(nn + fgroups, 0); }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)     // accumulators start after the fused stage: its registers are dead now
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = bv[nt];
            } else {
#pragma unroll
                for (int i = 0; i < NPF; ++i)
                    if ((tid >> 3) + 32 * i < NPIX) {
                        if constexpr (IO16) {   // already bf16: the 8 bytes go where the conversion path below puts them
                            float* blk = &tile[(tid >> 3) * PS + i * 32 * PS + (c4 >> 1) * 8 + (c4 & 1) * 2];
                            *(u32x2*)blk = pf_get_u(pf[i]);
                        } else if constexpr (PREC) {   // channels 4*c4..4*c4+3 = half (c4&1) of 8-channel block c4>>1
                            const f32x4 v = pf_get_v(pf[i]);
                            float* blk = &tile[(tid >> 3) * PS + i * 32 * PS + (c4 >> 1) * 8 + (c4 & 1) * 2];
                            if constexpr (PREC == 2) {
                                // bf16: two packed conversions and ONE 8-byte write; the lo half of the block is never read
                                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                                const bf16x2_t p0 = {(__bf16)v[0], (__bf16)v[1]}, p1 = {(__bf16)v[2], (__bf16)v[3]};
                                *(u32x2_t*)blk = u32x2_t{__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1)};
                            } else {
                                f16x4 hi, lo;
#pragma unroll
                                for (int e = 0; e < 4; ++e) { _Float16 h_, l_; vad_split_p<PREC>(v[e], h_, l_); hi[e] = h_; lo[e] = l_; }
                                *(f16x4*)blk = hi;
                                *(f16x4*)(blk + 4) = lo;
                            }
                        } else {
                            *(f32x4*)&tile[slds0 + i * 32 * PS] = pf_get_v(pf[i]);
                        }
                    }
                STAMP(2);
                __syncthreads();
                STAMP(3);
                if (ch + 1 < nch) { ISSUE(n, ch + 1); }
                else if (has_next) { ISSUE(nn, 0); }
                if constexpr (MODE == MODE_LSTM) {
                    // previous cell state of this tile: requested one chunk (36 k-steps) before the epilogue reads it.  Loaded in
                    // the epilogue each value cost a vmcnt(0) - its own round trip plus the acknowledgement of the two stores in
                    // front of it - 16 times per tile in a row.
                    if (ch + 1 == nch) {
                        const __amdgpu_buffer_rsrc_t rc_in = vad_rsrc(p.c_prev ? p.c_prev + (size_t)n * (out_bytes / 4) : p.c_out, p.c_prev ? out_bytes : 0u);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int dy = 2 * mt + ((r & 3) >> 1), dx = 4 * (r >> 2) + (r & 1);
                                const bool ok = full_tile || ((ey0 + dy) < oh && (ex0 + dx) < ow);
                                cpv[mt][r] = vad_bload1(rc_in, ok ? eoff[0] : VAD_OOB, dy * erow + dx * ecol);   // zero-sized descriptor -> 0 (initial state)
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if constexpr (PREC) { LOAD_A_HALF(0, 0, MT); } else { LOAD_A(0, 0); }
            STAMP(4);
            if constexpr (PREC) {
                // fp16 steps: B ring from L2 (PB ahead); A single-buffered in registers and refilled for step s+1 in two
                // halves, each right after the MFMAs that consumed it, so an LDS round trip hides under the other half.
                constexpr int MH = (MT + 1) / 2;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int bcur = s % NB, bnxt = (s + PB) % NB;
                    if (s + PB < NS) { LOAD_B(bnxt, ch, s + PB); }
                    else if (ch + 1 < nch) { LOAD_B(bnxt, ch + 1, s + PB - NS); }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < MH; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            vad_mma16<PREC>(acc[mt][nt], corr[PREC == 1 ? mt : 0][PREC == 1 ? nt : 0], ah[0][mt], al[0][mt], bh[bcur][nt], bl[bcur][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < NS) LOAD_A_HALF(s + 1, 0, MH);
#pragma unroll
                    for (int mt = MH; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            vad_mma16<PREC>(acc[mt][nt], corr[PREC == 1 ? mt : 0][PREC == 1 ? nt : 0], ah[0][mt], al[0][mt], bh[bcur][nt], bl[bcur][nt]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < NS) LOAD_A_HALF(s + 1, MH, MT);
                }
            } else {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int cur = s & 1, nxt = cur ^ 1;
                    const int bcur = s % NB, bnxt = (s + PB) % NB;
                    if (s + 1 < NS) LOAD_A(nxt, s + 1);
                    if (s + PB < NS) { LOAD_B(bnxt, ch, s + PB); }
                    else if (ch + 1 < nch) { LOAD_B(bnxt, ch + 1, s + PB - NS); }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = VAD_KORDER(jj);       // channels j and 4+j of the 8-group; see VAD_KORDER
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = MFMA32(a[cur][mt][j], b[bcur][nt][j], acc[mt][nt]);
                    }
                }
            }
            STAMP(5);
        }
        if constexpr (PREC == 1) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(corr[mt][nt][r], 0x1p-11f, acc[mt][nt][r]);
        }

        // next frame's first B fragments go out BEFORE this frame's stores (in-order vmcnt)
        if (has_next) {
#pragma unroll
            for (int s0 = 0; s0 < PB; ++s0) LOAD_B(s0, 0, s0);
        }

        // ------------------------------------------------------------ epilogue of this frame's tile
        // Stores go through a buffer descriptor of this frame's output: offset = lane part (VGPR) + wave-uniform
        // (row, column) part (SGPR); elements outside a partial tile get offset VAD_OOB and are dropped.
        if (MODE == MODE_LSTM) {
            const __amdgpu_buffer_rsrc_t rc_out = vad_rsrc(p.c_out + (size_t)n * (out_bytes / 4), out_bytes);
            const __amdgpu_buffer_rsrc_t rh_out = vad_rsrc(p.out + (size_t)n * p.out_fs, out_bytes);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dy = 2 * mt + ((r & 3) >> 1), dx = 4 * (r >> 2) + (r & 1);
                    const bool ok = full_tile || ((ey0 + dy) < oh && (ex0 + dx) < ow);
                    const unsigned vo = ok ? eoff[0] : VAD_OOB;                 // gate 0's channel == hidden channel
                    const unsigned so = dy * erow + dx * ecol;
                    float cn, hn;
                    vad_lstm_cell(acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r], cpv[mt][r], cn, hn);
                    vad_bstore1(cn, rc_out, vo, so);
                    vad_bstore1(hn, rh_out, vo, so);
                }
            }
        } else {
            const __amdgpu_buffer_rsrc_t ro = vad_rsrc((const char*)p.out + (size_t)n * p.out_fs * ES, (p.stagger & 2) ? 0u : out_bytes);   // stagger bit 1 (debug): zero-sized = every store dropped
            if (MODE == MODE_POOL) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool ok = full_tile || ((ey0 + mt) < oh && (ex0 + 2 * q) < ow);
                        const unsigned so = mt * erow + 2 * q * ecol;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            // MaxPool2d(act(.)) == act(MaxPool2d(.)) bit for bit (ReLU / LeakyReLU are non-decreasing): one
                            // activation per window instead of four (VALU work shares the pipe with the exact-fp32 MFMAs)
                            const float m = fmaxf(fmaxf(acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1]), fmaxf(acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]));
                            vad_bstore1(vad_act(m, ACT), ro, ok ? eoff[nt] : VAD_OOB, so);
                        }
                    }
                }
            } else {
                float ts[STATS ? NT : 1], tq[STATS ? NT : 1];
#pragma unroll
                for (int nt = 0; nt < (STATS ? NT : 1); ++nt) ts[nt] = tq[nt] = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
#pragma unroll
                        for (int pos = 0; pos < 4; ++pos) {
                            const int dy = 2 * mt + (pos >> 1), dx = 4 * q + (pos & 1);
                            const bool ok = full_tile || ((ey0 + dy) < oh && (ex0 + dx) < ow);
                            const unsigned so = dy * erow + dx * ecol;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                if constexpr (IO16) vad_bstore_h(vad_f_bf16(acc[mt][nt][4 * q + pos]), ro, ok ? eoff[nt] : VAD_OOB, so);
                                else vad_bstore1(vad_act(acc[mt][nt][4 * q + pos], ACT), ro, ok ? eoff[nt] : VAD_OOB, so);
                                if constexpr (STATS) {
                                    if (p.stats) {   // (uniform)
                                        const float d = ok ? acc[mt][nt][4 * q + pos] - bv[nt] : 0.f;
                                        ts[nt] += d;
                                        tq[nt] = fmaf(d, d, tq[nt]);
                                    }
                                }
                            }
                        }
                    }
                }
                if constexpr (STATS) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) { st_s[nt] += ts[nt]; st_q[nt] += tq[nt]; }
                }
            }
        }
        STAMP(6);
        if (!has_next) break;
        n = nn;
    }
#undef LOAD_A
#undef LOAD_A_HALF
#undef LOAD_B
#undef ISSUE
#undef ISSUE_TO
#undef XWRITE
    if constexpr (STATS) {
        if (p.stats) {           // (uniform)  partial sums of this work-group: lane halves, then the WM waves of a column block
            __syncthreads();     // every wave is done with the LDS tile: reuse it as [wave][sum | sum of squares][nt][32]
            float* red = tile;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float a = st_s[nt] + __shfl_xor(st_s[nt], 32), q = st_q[nt] + __shfl_xor(st_q[nt], 32);
                if (lh == 0) { red[((wave * 2 + 0) * NT + nt) * 32 + li] = a; red[((wave * 2 + 1) * NT + nt) * 32 + li] = q; }
            }
            __syncthreads();
            if (wm == 0 && lh == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float t = red[((wn * 2 + j) * NT + nt) * 32 + li];
#pragma unroll
                        for (int m = 1; m < WM; ++m) t += red[(((m * WN + wn) * 2 + j) * NT + nt) * 32 + li];
                        p.stats[((size_t)stats_row * 2 + j) * p.cout + cofs[nt]] = t;
                    }
            }
        }
    }
#ifdef VAD_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 12;
        for (int i = 0; i < 10; ++i) d[i] = st_sum[i];
        // shader clock in kHz over this wave's life: cycles / (100 MHz ticks) * 1e5
        d[10] = (__builtin_amdgcn_s_memtime() - st_t0) * 100000ull / (__builtin_amdgcn_s_memrealtime() - st_r0);
        d[11] = st_n;
    }
#endif
}
#undef STAMP
