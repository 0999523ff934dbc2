// Small-grid form of the ConvLSTM gate convolution (included by conv_mfma.hip after conv_pkernel.h).
//
// The reference's own callers use small batches: 4 clips per batch in evaluate_video.py:416 / train_video.py:306, ONE
// window per forward in generate_video_output (evaluate_video.py:344).  A ConvLSTM step on a 16x16 map is then a GEMM of
// M = B*256 pixels, N = 512 gate columns, K = 2304, and the time of a step is not throughput but the serial K loop of ONE
// wave: the 32x32x2 kernel gives a wave a 32-pixel x 128-column tile = 4,608 dependent-per-accumulator MFMAs of 64 cycles
// (123 us at 2.4 GHz) however few work-groups there are (8 per clip: 32 of 256 CUs busy at B = 4).
//
// This kernel cuts the wave tile to 16 pixels x 64 columns (4 gates x 16 hidden channels) on v_mfma_f32_16x16x4_f32
// (32 cycles, same FLOP/cycle): 2,304 MFMAs of 32 cycles per wave = 31 us per step, and 4x the waves (128 per clip).
//
// Bit-identical to conv3x3_mfma_pkernel's LSTM mode, not merely close: both MFMAs are k-ordered fp32 fmaf chains, and the
// k order is reproduced exactly - per 32-channel chunk: tap-major, then 8-channel groups, and inside a group channels
// 0,4,2,6,1,5,3,7 (VAD_KORDER in conv_mfma.hip): the 32x32x2 kernels run their steps j = 0,2,1,3 over channel pairs (j, 4+j);
// here k lane kq reads the ADJACENT channels (c, c+1), c = (0,4,2,6)[kq], and two 16x16x4 instructions take the first and the
// second of each pair.  Same bias-initialised accumulator, same gate functions: a clip scored alone equals the same clip
// inside a batch of 64 (tests assert torch.equal), which is what keeps sharded and chunked scoring exact.
//
// Operands: A = NHWC tile with 1-pixel halo in LDS (2 rows x 16 columns + halo, pixel stride 36 floats, next chunk
// prefetched into registers), one ds_read_b64 per lane and step; B = the packed weights [tap][cin/8][cout][8] straight from
// L2, one 8-byte load per lane, gate and step: a wave reads 512 contiguous bytes, every byte once (16-byte loads with a
// per-lane element pick moved 2x the bytes through the 64 B/clk vector L1 and made it, not the matrix pipe, the limit:
// 64 us per launch instead of ~35).  M-tile = 2 rows x 8 columns in pooling-window order, so a lane's 4 accumulator
// registers are a 2x2 pixel block.  Exact fp32 only.
#pragma once

#define MFMA16X4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// 4 waves = 2 M-tiles (left / right 8 columns of a 2 x 16 pixel tile) x 2 blocks of 16 hidden channels
template <int NB>
__global__ __launch_bounds__(256) void convlstm_small_kernel(Conv3P p) {
    constexpr int CK = 32, LH = 4, LW = 18, PS = CK + 4, NPIX = LH * LW, TOT = NPIX * (CK / 4), NPF = (TOT + 255) / 256;
    __shared__ __attribute__((aligned(16))) float tile[NPIX * PS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, kq = lane >> 4;
    const int H = p.h, W = p.w_, hid = p.hid;

    // column block slowest inside a frame: the work-groups an XCD runs (consecutive logical ids) then read the SAME quarter of
    // the 4.7 MB gate weights, which fits its 4 MB L2; the first reader of a line pays the Infinity-Cache trip, the rest hit
    unsigned L = vad_xcd_remap(blockIdx.x, p.nblocks);
    const int x0 = (L % p.tiles_x) * 16; L /= p.tiles_x;
    const int y0 = (L % p.tiles_y) * 2; L /= p.tiles_y;
    const int cb = L % p.cblocks;
    const int n = L / p.cblocks;

    // A operand: M index li -> pooling-window order inside the wave's 2 x 8 pixel M-tile
    const int arow = (li >> 1) & 1, acol = 8 * wm + 2 * (li >> 2) + (li & 1);
    const int kc = ((kq & 1) << 2) | (kq & 2);        // first channel of this k lane's pair: (0,4,2,6)[kq]
    const int abase = (arow * LW + acol) * PS + kc;

    // B operand / bias: gate g, hidden channel hc -> column g*hid + hc
    const int hc = (cb * 2 + wn) * 16 + li;
    const unsigned wstep = (unsigned)p.cout * 32u;                 // bytes per (tap, 8-channel group) slab
    const unsigned wtap = (unsigned)(p.cin / 8) * wstep;           // bytes per tap
    const __amdgpu_buffer_rsrc_t rw = vad_rsrc(p.w, 9u * wtap);
    unsigned wl[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) wl[g] = (unsigned)(g * hid + hc) * 32u + 4u * (unsigned)kc;

    const int nch_a = p.cin_a / CK;
    const int nch = (p.in2 ? p.cin : p.cin_a) / CK;
    // x half computed ahead (p.zx: bias + the chunks [0, nch_a) of every accumulator chain, stored as fp32): start behind them
    const int ch0 = p.zx ? nch_a : 0;

    // staging: slot i of this thread = float4 number tid + 256 i of the halo tile (pixel-major, 8 quads per pixel)
    f32x4 pf[NPF];
    auto issue = [&](int ch) {
        const bool a = ch < nch_a;
        const float* src = a ? p.in + (size_t)n * p.in_fs : p.in2 + (size_t)n * p.in2_fs;
        const int pstride = a ? p.cin_a : p.cin - p.cin_a;
        const int coff = (a ? ch : ch - nch_a) * CK;
        const __amdgpu_buffer_rsrc_t r = vad_rsrc(src, (unsigned)(H * W) * (unsigned)pstride * 4u);
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + 256 * i, pix = idx >> 3, c4 = idx & 7;
            const int ly = pix / LW, lx = pix - ly * LW;
            const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
            const bool ok = idx < TOT && gy >= 0 && gy < H && gx >= 0 && gx < W;
            pf[i] = vad_bload4(r, ok ? (unsigned)(__mul24(__mul24(gy, W) + gx, pstride) + coff + c4 * 4) * 4u : VAD_OOB, 0);
        }
    };
    if (ch0 < nch) issue(ch0);

    // B fragments run PB steps ahead in a ring of NB register sets: a step is only 8 MFMAs x 32 cycles, and a weight line that
    // misses L2 comes back from the Infinity Cache in ~550 cycles (one step ahead left ~300 cycles exposed per step: 73 us
    // per launch instead of the 31 us of its MFMAs)
    constexpr int PB = NB - 1;
    static_assert(36 % NB == 0, "ring position must be the same in every chunk");
    f32x2 b[NB][4];
#define SLOAD_B(buf, chunk, step)                                                                                     \
    {                                                                                                                 \
        const unsigned woff_ = (unsigned)((step) >> 2) * wtap + (unsigned)((chunk) * 4 + ((step) & 3)) * wstep;       \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) b[buf][g] = vad_bload2(rw, wl[g], woff_);                       \
    }
    if (ch0 < nch) {
#pragma unroll
        for (int s0 = 0; s0 < PB; ++s0) SLOAD_B(s0, ch0, s0);
    }

    // Every other load of the prologue goes out behind the first tile and the first weights, and nothing waits before all
    // of them are requested: ONE round trip in front of the first MFMA (bias -> accumulators -> tile was three in a row, ~3 us
    // of a ~35 us launch).
    const size_t cfs = (size_t)H * W * hid;
    float cpv[4];                            // previous cell state of this lane's 2x2 block (r = (dy, dx) of window kq)
    {   // branch-free: a zero-sized descriptor (initial state) or an out-of-range offset (partial tile) reads 0.0
        const __amdgpu_buffer_rsrc_t rc = vad_rsrc(p.c_prev ? p.c_prev + (size_t)n * cfs : p.c_out, p.c_prev ? (unsigned)cfs * 4u : 0u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y0 + (r >> 1), x = x0 + 8 * wm + 2 * kq + (r & 1);
            cpv[r] = vad_bload1(rc, (y < H && x < W) ? (unsigned)(__mul24(__mul24(y, W) + x, hid) + hc) * 4u : VAD_OOB, 0);
        }
    }
    f32x4 acc[4];
    if (p.zx) {
        const __amdgpu_buffer_rsrc_t rz = vad_rsrc(p.zx + (size_t)n * p.zx_fs, (unsigned)(H * W) * (unsigned)(4 * hid) * 4u);
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = y0 + (r >> 1), x = x0 + 8 * wm + 2 * kq + (r & 1);
                acc[g][r] = vad_bload1(rz, (y < H && x < W) ? (unsigned)(__mul24(__mul24(y, W) + x, 4 * hid) + g * hid + hc) * 4u : VAD_OOB, 0);
            }
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float bv = p.bias[g * hid + hc];
            acc[g] = f32x4{bv, bv, bv, bv};
        }
    }

    for (int ch = ch0; ch < nch; ++ch) {
        __syncthreads();                               // every wave is done reading the previous chunk
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + 256 * i;
            if (idx < TOT) *(f32x4*)&tile[(idx >> 3) * PS + (idx & 7) * 4] = pf[i];
        }
        __syncthreads();
        if (ch + 1 < nch) issue(ch + 1);               // in flight during the 36 steps below

        f32x2 a[2];
        a[0] = *(const f32x2*)&tile[abase];
#pragma unroll
        for (int s = 0; s < 36; ++s) {                 // (tap, 8-channel group) steps: same order as the 32x32x2 kernel
            const int cur = s & 1, nxt = cur ^ 1;
            const int bcur = s % NB, bnxt = (s + PB) % NB;
            if (s + 1 < 36) {
                const int tap = (s + 1) >> 2;
                a[nxt] = *(const f32x2*)&tile[abase + ((tap / 3) * LW + tap % 3) * PS + ((s + 1) & 3) * 8];
            }
            if (s + PB < 36) { SLOAD_B(bnxt, ch, s + PB); }
            else if (ch + 1 < nch) { SLOAD_B(bnxt, ch + 1, s + PB - 36); }
            __builtin_amdgcn_sched_barrier(0);         // keep the prefetch above this step's MFMAs
            // Two gates at a time, each chain's second MFMA two instructions behind its first: four chains taken round-robin
            // (0 1 2 3 0 1 2 3) issue a 16x16x4 MFMA every 48 clocks instead of every 32 (tools/ubench/mfma_chain.hip; two or
            // three chains, or this order, reach 32).  The order inside every chain - the k order - is unchanged.
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                acc[gp] = MFMA16X4(a[cur][0], b[bcur][gp][0], acc[gp]);                            // channels 0,4,2,6
                acc[gp + 1] = MFMA16X4(a[cur][0], b[bcur][gp + 1][0], acc[gp + 1]);
                __builtin_amdgcn_sched_barrier(0);                                                 // (hipcc re-sorts independent MFMAs)
                acc[gp] = MFMA16X4(a[cur][1], b[bcur][gp][1], acc[gp]);                            // channels 1,5,3,7
                acc[gp + 1] = MFMA16X4(a[cur][1], b[bcur][gp + 1][1], acc[gp + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#undef SLOAD_B

    // epilogue: registers r = 2x2 pixel block (dy = r>>1, dx = r&1) of window kq; gates of hidden channel hc in this lane
    float* cout_ = p.c_out + (size_t)n * cfs;
    float* hout = p.out + (size_t)n * p.out_fs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + (r >> 1), x = x0 + 8 * wm + 2 * kq + (r & 1);
        if (y < H && x < W) {
            const size_t o = ((size_t)y * W + x) * hid + hc;
            float cn, hn;
            vad_lstm_cell(acc[0][r], acc[1][r], acc[2][r], acc[3][r], cpv[r], cn, hn);
            cout_[o] = cn;
            hout[o] = hn;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Gate-split form, for the smallest grids (one dense window, up to a few clips: generate_video_output, evaluate_video.py:344).
// The kernel above is at its limit there: a wave's K loop is a fixed chain of 8 MFMAs per step at ~39 clocks each (4.6 us
// per 32-channel chunk, tools/gpu_lstm_slope.sh), and with one clip only 32 of 256 CUs hold a work-group.  Here a wave takes
// ONE gate of its 16 pixels x 16 hidden channels: 8 waves per work-group (2 M-tiles x 4 gates) share the same LDS tile, a
// step is 2 dependent MFMAs (44 clocks each as a single chain, tools/ubench/mfma_chain.hip) instead of 8, and the grid has
// twice the work-groups.  The four gates of a cell meet through 6 KB of LDS; the waves of gate 0 apply the cell update.
// CELL = 0 is the same K loop without the cell: bias + the convolution of channels [0, cin_a) written as pre-activations
// [n][h][w][cout] (the x halves computed ahead of the recurrence; cout / 4 takes the place of hid).
// Bit-identical to both other forms: every accumulator chain runs the same k order from the same start value.
// MTW = M-tiles (2 rows x 8 columns) per work-group: 4 MTW waves.  MTW = 1 doubles the work-groups once more - while they still
// fit one per CU a wave has its SIMD's matrix pipe to itself (two waves of dependent 2-MFMA steps on one SIMD are pipe-bound).
template <int CELL, int MTW, int ACT = VAD_ACT_NONE, int POOL = 0>
__global__ __launch_bounds__(256 * MTW) void convlstm_gate_kernel(Conv3P p) {
    static_assert(!CELL || (ACT == VAD_ACT_NONE && !POOL), "activation / pooling belong to the plain-convolution form");
    constexpr int NT = 256 * MTW, TW = 8 * MTW;
    constexpr int CK = 32, LH = 4, LW = TW + 2, PS = CK + 4, NPIX = LH * LW, TOT = NPIX * (CK / 4), NPF = (TOT + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float tile[NPIX * PS];
    __shared__ __attribute__((aligned(16))) float zbuf[3 * MTW * 64 * 4];        // gates 1..3 of every M-tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % MTW, g = wave / MTW;
    const int li = lane & 15, kq = lane >> 4;
    const int H = p.h, W = p.w_, hid = p.hid;                            // (CELL = 0: hid = cout / 4)

    unsigned L = vad_xcd_remap(blockIdx.x, p.nblocks);
    const int x0 = (L % p.tiles_x) * TW; L /= p.tiles_x;
    const int y0 = (L % p.tiles_y) * 2; L /= p.tiles_y;
    const int cb = L % p.cblocks;
    const int n = L / p.cblocks;

    const int arow = (li >> 1) & 1, acol = 8 * wm + 2 * (li >> 2) + (li & 1);
    const int kc = ((kq & 1) << 2) | (kq & 2);
    const int abase = (arow * LW + acol) * PS + kc;

    const int hc = cb * 16 + li;
    const unsigned wstep = (unsigned)p.cout * 32u;
    const unsigned wtap = (unsigned)(p.cin / 8) * wstep;
    const __amdgpu_buffer_rsrc_t rw = vad_rsrc(p.w, 9u * wtap);
    const unsigned wl = (unsigned)(g * hid + hc) * 32u + 4u * (unsigned)kc;

    const int nch_a = p.cin_a / CK;
    const int nch = CELL ? (p.in2 ? p.cin : p.cin_a) / CK : nch_a;
    const int ch0 = (CELL && p.zx) ? nch_a : 0;

    f32x4 pf[NPF];
    auto issue = [&](int ch) {
        const bool a = ch < nch_a;
        const float* src = a ? p.in + (size_t)n * p.in_fs : p.in2 + (size_t)n * p.in2_fs;
        const int pstride = a ? p.cin_a : p.cin - p.cin_a;
        const int coff = (a ? ch : ch - nch_a) * CK;
        const __amdgpu_buffer_rsrc_t r = vad_rsrc(src, (unsigned)(H * W) * (unsigned)pstride * 4u);
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + NT * i, pix = idx >> 3, c4 = idx & 7;
            const int ly = pix / LW, lx = pix - ly * LW;
            const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
            const bool ok = idx < TOT && gy >= 0 && gy < H && gx >= 0 && gx < W;
            pf[i] = vad_bload4(r, ok ? (unsigned)(__mul24(__mul24(gy, W) + gx, pstride) + coff + c4 * 4) * 4u : VAD_OOB, 0);
        }
    };
    if (ch0 < nch) issue(ch0);

    // B fragments PB steps ahead: a step is two MFMAs (~90 clocks), a weight line from the Infinity Cache ~550
    constexpr int NB = 12, PB = NB - 1;
    static_assert(36 % NB == 0, "ring position must be the same in every chunk");
    f32x2 b[NB];
#define GLOAD_B(buf, chunk, step) \
    b[buf] = vad_bload2(rw, wl, (unsigned)((step) >> 2) * wtap + (unsigned)((chunk) * 4 + ((step) & 3)) * wstep)
    if (ch0 < nch) {
#pragma unroll
        for (int s0 = 0; s0 < PB; ++s0) GLOAD_B(s0, ch0, s0);
    }

    const size_t cfs = (size_t)H * W * hid;
    float cpv[4] = {0.f, 0.f, 0.f, 0.f};
    if (CELL && g == 0) {   // (wave-uniform) branch-free inside: a zero-sized descriptor (initial state) or an out-of-range offset reads 0.0
        const __amdgpu_buffer_rsrc_t rc = vad_rsrc(p.c_prev ? p.c_prev + (size_t)n * cfs : p.c_out, p.c_prev ? (unsigned)cfs * 4u : 0u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y0 + (r >> 1), x = x0 + 8 * wm + 2 * kq + (r & 1);
            cpv[r] = vad_bload1(rc, (y < H && x < W) ? (unsigned)(__mul24(__mul24(y, W) + x, hid) + hc) * 4u : VAD_OOB, 0);
        }
    }
    f32x4 acc;
    if (CELL && p.zx) {
        const __amdgpu_buffer_rsrc_t rz = vad_rsrc(p.zx + (size_t)n * p.zx_fs, (unsigned)(H * W) * (unsigned)(4 * hid) * 4u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = y0 + (r >> 1), x = x0 + 8 * wm + 2 * kq + (r & 1);
            acc[r] = vad_bload1(rz, (y < H && x < W) ? (unsigned)(__mul24(__mul24(y, W) + x, 4 * hid) + g * hid + hc) * 4u : VAD_OOB, 0);
        }
    } else {
        const float bv = p.bias[g * hid + hc];
        acc = f32x4{bv, bv, bv, bv};
    }

    for (int ch = ch0; ch < nch; ++ch) {
        __syncthreads();                               // every wave is done reading the previous chunk
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + NT * i;
            if (idx < TOT) *(f32x4*)&tile[(idx >> 3) * PS + (idx & 7) * 4] = pf[i];
        }
        __syncthreads();
        if (ch + 1 < nch) issue(ch + 1);               // in flight during the 36 steps below

        f32x2 a[2];
        a[0] = *(const f32x2*)&tile[abase];
#pragma unroll
        for (int s = 0; s < 36; ++s) {                 // (tap, 8-channel group) steps: same order as the other two kernels
            const int cur = s & 1, nxt = cur ^ 1;
            const int bcur = s % NB, bnxt = (s + PB) % NB;
            if (s + 1 < 36) {
                const int tap = (s + 1) >> 2;
                a[nxt] = *(const f32x2*)&tile[abase + ((tap / 3) * LW + tap % 3) * PS + ((s + 1) & 3) * 8];
            }
            if (s + PB < 36) { GLOAD_B(bnxt, ch, s + PB); }
            else if (ch + 1 < nch) { GLOAD_B(bnxt, ch + 1, s + PB - 36); }
            __builtin_amdgcn_sched_barrier(0);         // keep the prefetch above this step's MFMAs
            acc = MFMA16X4(a[cur][0], b[bcur][0], acc);                                            // channels 0,4,2,6
            acc = MFMA16X4(a[cur][1], b[bcur][1], acc);                                            // channels 1,5,3,7
        }
    }
#undef GLOAD_B

    if (!CELL) {          // [n][h][w][cout] (POOL: [n][h/2][w/2][cout]); ACT / POOL as the 32x32x2 kernels apply them
        float* zo = p.out + (size_t)n * p.out_fs;
        if (POOL) {       // the lane's four registers are one 2x2 window (H, W even: host-checked); activation is monotonic: act(max) == max(act)
            const int y = y0, x = x0 + 8 * wm + 2 * kq;
            const float m = vad_act(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])), ACT);
            if (y < H && x < W) zo[((size_t)(y >> 1) * (W >> 1) + (x >> 1)) * p.cout + g * hid + hc] = m;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = y0 + (r >> 1), x = x0 + 8 * wm + 2 * kq + (r & 1);
                // plain fmaxf, not vad_act's inline-asm v_max: hipcc's hazard recogniser puts no wait states between an MFMA and an
                // asm statement that reads its result (same values: maxNum of two non-NaN floats)
                const float v = acc[r];
                const float a_ = ACT == VAD_ACT_LEAKY ? fmaxf(v, 0.2f * v) : (ACT == VAD_ACT_RELU ? fmaxf(v, 0.f) : v);
                if (y < H && x < W) zo[((size_t)y * W + x) * p.cout + g * hid + hc] = a_;
            }
        }
        return;
    }
    // the gates of a cell meet: gates 1..3 through LDS to the wave of gate 0 with the same M-tile (same lane = same pixels, same hc)
    if (g > 0) *(f32x4*)&zbuf[(((g - 1) * MTW + wm) * 64 + lane) * 4] = acc;
    __syncthreads();
    if (g > 0) return;
    const f32x4 zf = *(const f32x4*)&zbuf[((0 * MTW + wm) * 64 + lane) * 4];
    const f32x4 zg = *(const f32x4*)&zbuf[((1 * MTW + wm) * 64 + lane) * 4];
    const f32x4 zo_ = *(const f32x4*)&zbuf[((2 * MTW + wm) * 64 + lane) * 4];
    float* cout_ = p.c_out + (size_t)n * cfs;
    float* hout = p.out + (size_t)n * p.out_fs;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + (r >> 1), x = x0 + 8 * wm + 2 * kq + (r & 1);
        if (y < H && x < W) {
            const size_t o = ((size_t)y * W + x) * hid + hc;
            float cn, hn;
            vad_lstm_cell(acc[r], zf[r], zg[r], zo_[r], cpv[r], cn, hn);
            cout_[o] = cn;
            hout[o] = hn;
        }
    }
}
