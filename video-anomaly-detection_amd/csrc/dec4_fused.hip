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
//     M = 27 (tap, co) rows in 32 slots, K = 32 channels, N = pixels.  In that orientation the B operand of step r is
//     lane = pixel, k = (channel c_r, c_r + 4) - exactly register r of GEMM 1's result in its two lane halves: bias and ReLU
//     are applied in place and the 16 registers are fed straight back into the matrix pipe.  No LDS transpose.
//   The 3x3 neighbourhood is then 27 additions per output pixel: out[y][x][co] = sum_{dy,dx} P[y+dy-1][x+dx-1][(dy,dx),co].
//     A work-group owns a band of rows over the whole width (no x halo to recompute; the y halo is one input row per band
//     end) and walks it one OUTPUT row at a time ("phase"): the four waves put that row of P into LDS, one 36-float record
//     per column (slot order below: a lane's four accumulator registers are one 16-byte store, a reader's nine values per
//     neighbour column two 16-byte loads and one 4-byte load), and after one barrier thread x adds its nine row sums
//     s[dy][co] = sum_dx P[(dy,dx,co)][x+dx-1] into three running output rows held in registers: row y is complete when P
//     rows y-1, y, y+1 have passed.  Tanh, error, optional recon / error-map stores and the per-row partial sum follow.
//
// ONE PIPE.  An exact-fp32 MFMA and a VALU instruction never overlap (tools/ubench/mfma_valu_mix.hip: in one wave every
// v_fmac behind a 32x32x2_f32 MFMA adds 5.5-7.5 clocks to its 64 - an fp16 MFMA hides four of them - and a second wave's
// VALU stream next to a wave of back-to-back fp32 MFMAs makes no progress until they end).  So the second work-group of the
// CU cannot hide the combine, and every VALU instruction of the row loop is matrix time: the loop issues ~110 of them next
// to its 64 MFMAs (first form: ~330).  What that took: the column-record LDS layout above (16-byte accesses, no register
// shuffles), zero padding by READING an all-zero column (P rows / columns outside the image are computed like any other and
// never read: three address selects), the bias as the C operand of each chain's first MFMA, ReLU as one integer max per
// value, all 32 ahead of GEMM 2, packed adds, a DPP row sum, buffer loads with scalar bases, the next input row loaded
// straight into the registers GEMM 1 has just read.
// SOFTWARE PIPELINE.  The combine of phase t-1 is issued between the GEMM-1 MFMAs of phase t (scheduling fences keep its
// stages in place): its LDS latency and every wait pass while the matrix pipe works, one barrier per phase.  The block is
// branch-free (loads use out-of-range offsets or clamped addresses, the partial-sum store an out-of-range offset when
// there is nothing to store; only the optional map stores sit under a uniform condition) and the pipeline runs across
// work items: one fill and one drain per work-group.
//
// Per 256x256 frame: 2 x 0.134 GFLOP on the fp32 matrix pipe (GEMM 2 pads 27 -> 32 rows; the unfused tail ran 0.113 GFLOP on
// the VALU, the same pipe) and 2.1 MB + 0.79 MB read from HBM: bound by the exact-fp32 matrix pipe at ~1.8 us per frame.
// Sums are taken in a fixed order that depends on the pixel only (never on the band height, batch or rank).
#include <atomic>
#include "vad_common.h"

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {
// One P record per LDS column jl = 2 + j (j = output column - 2 * xs0 in [-2, 257]): 32 GEMM-2 rows + 4 floats of padding
// (36-float pitch: 16-byte accesses of 8 consecutive lanes, at a stride of one or two columns, touch 32 distinct banks).
// Slot of (dy, dx, co), idx = 3 dy + co: dx 0 -> idx, dx 1 -> 12 + idx, dx 2 -> 24 + idx (idx < 8) / slot 9 (idx 8);
// slots 10, 11, 21..23 carry zero weights.
constexpr int D4_CPITCH = 36;
constexpr int D4_COLS = 260;
constexpr int D4_BUF = D4_COLS * D4_CPITCH;   // floats per P buffer (37,440 B; two buffers: dynamic LDS)
constexpr int D4_LDS_BYTES = 2 * D4_BUF * 4;

struct Dec4P {
    const float* in;       // [n][H][W][32] NHWC: dec3.3 output
    const float* wt;       // transposed-conv weights, fp32 pack [q][cin/8][cout][8] (BatchNorm folded)
    const float* bt;       // [32] folded bias
    const float* w2;       // GEMM form of the 3x3 weights: [m 32][lane half 2][r 16], m = tap * 3 + co (pack.cpp)
    const float* b3;       // [3]
    const void* x;         // original frames (format per template argument)
    float* partials;       // [n][2H * nstrips * 4]
    float* recon;          // NCHW or NULL
    float* errmap;         // [n][2H][2W] or NULL
    int n, H, W;           // input map size; the output is 2H x 2W
    int R, nbands, nstrips;
    unsigned nitems;       // n * nbands * nstrips
    unsigned long long* dbg;   // VAD_D4_STAMPS diagnostic build only
};

struct D4Phase { int valid, n, strip, r0, r1, yp, row_ok; };   // one output row of P (all fields wave-uniform); row_ok: its input row is inside the image
struct D4Out { float rc[3], e, ws; };           // one thread's results of a combine

// Sum over the 64 lanes, valid in lane 63: six DPP adds on the VALU (a __shfl_xor butterfly is six dependent ds_bpermute,
// each a full LDS latency).  Fixed order: quads, 8s, rows of 16, rows 0+1 / 2+3, halves.
__device__ __forceinline__ float d4_wave_sum63(float v) {
#define D4_DPP(ctrl, rows) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), (ctrl), (rows), 0xF, false))
    v += D4_DPP(0xB1, 0xF);     // quad_perm [1,0,3,2]
    v += D4_DPP(0x4E, 0xF);     // quad_perm [2,3,0,1]
    v += D4_DPP(0x141, 0xF);    // row_half_mirror: the other quad of each 8 (every lane of a quad holds the quad's sum)
    v += D4_DPP(0x140, 0xF);    // row_mirror: the other 8 of each row
    v += D4_DPP(0x142, 0xA);    // row_bcast15 into rows 1 and 3
    v += D4_DPP(0x143, 0xC);    // row_bcast31 into rows 2 and 3
#undef D4_DPP
    return v;
}

__device__ __forceinline__ f32x4 pk_add4(f32x4 a, f32x4 b) {          // two v_pk_add_f32 (IEEE adds, same results as four v_add_f32)
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(f32x2{a[0], a[1]}), "v"(f32x2{b[0], b[1]}));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(f32x2{a[2], a[3]}), "v"(f32x2{b[2], b[3]}));
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

#ifdef VAD_D4_STAMPS
#define STAMP(k)                                                                                      \
    {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (st_prev) st_sum[k] += t_ - st_prev;                                                       \
        st_prev = t_;                                                                                 \
        if ((k) == 3) ++st_n;                                                                         \
    }
#else
#define STAMP(k)
#endif

template <bool XU8>
__global__ __launch_bounds__(256, 2) void dec4_score_kernel(Dec4P p) {
    extern __shared__ __attribute__((aligned(16))) float P[];            // 2 * D4_BUF
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
    // GEMM 2 A fragments: lane li is P slot li; its (tap, co) row of the packed weights, or a zero row
    int msrc = 27;
    {
        const int dx = li < 10 ? (li == 9 ? 2 : 0) : (li >= 12 && li < 21 ? 1 : (li >= 24 ? 2 : -1));
        const int idx = li == 9 ? 8 : (li < 9 ? li : (li < 21 ? li - 12 : li - 24));
        if (dx >= 0) msrc = ((idx / 3) * 3 + dx) * 3 + idx % 3;
    }
    f32x16 biasv;
    float w2f[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        biasv[r] = p.bt[(r & 3) + 8 * (r >> 2) + 4 * lh];
        w2f[r] = p.w2[(msrc * 2 + lh) * 16 + r];
    }
    const float c3b0 = p.b3[0], c3b1 = p.b3[1], c3b2 = p.b3[2];

    // zero both P buffers once: the columns no lane ever writes (jl 0, 1, 258, 259) are zero padding / never-owned halo
    for (int i = tid; i < 2 * D4_BUF / 4; i += 256) ((f32x4*)P)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const unsigned in_bytes = (unsigned)(H * W) * 32u * 4u;
    const size_t plane = (size_t)H2 * W2;
    const int j = tid;                                                   // this thread's output column within the strip (combine)
#ifdef VAD_D4_STAMPS
    unsigned long long st_sum[4] = {0, 0, 0, 0}, st_prev = 0, st_n = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    auto decode = [&](unsigned item, int& n, int& strip, int& r0, int& r1) {
        unsigned t_ = item;
        strip = t_ % p.nstrips; t_ /= p.nstrips;
        const int band = t_ % p.nbands;
        n = t_ / p.nbands;
        r0 = band * p.R;
        r1 = (r0 + p.R < H) ? r0 + p.R : H;
    };
    // Loads are issued unconditionally wherever possible (invalid rows / lanes get the out-of-range offset or a clamped
    // address): a load under a branch leaves hipcc unable to count what is in flight on the other path, and it then waits
    // for vmcnt(0).
    auto load_row = [&](f32x4 (&dst)[4], bool have, int n, int strip, int iy) {
        const int cx = 126 * strip + 32 * wave + li;                     // this lane's input column (GEMM phases)
        const bool ok = have && iy >= 0 && iy < H && cx < W;
        const __amdgpu_buffer_rsrc_t rin = vad_rsrc(p.in + (size_t)(have ? n : 0) * H * W * 32, in_bytes);
        const unsigned off = ok ? (unsigned)(__mul24(__mul24(iy, W) + cx, 32) + 4 * lh) * 4u : VAD_OOB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dst[ks] = vad_bload4(rin, off, (unsigned)ks * 32u);
    };
    // What thread j needs to know about its output column of one strip: computed once per work item (every VALU
    // instruction in the phase loop costs matrix-pipe time, see the header)
    struct LaneCols {
        int own;          // this thread stores / scores column ox
        int ox;           // output column
        unsigned xoff;    // byte offset of (own ? ox : 0) within one row of the original frame
        int c0, c1, c2;   // LDS float offsets of the three neighbour column records (column 0 - all zero - past the right edge)
    };
    auto lane_cols = [&](int strip) -> LaneCols {
        const int xs0 = 126 * strip, jlo = strip ? 2 : 0;
        int jhi = W2 - 2 * xs0;
        const int jcap = (strip == p.nstrips - 1) ? 256 : 254;
        if (jhi > jcap) jhi = jcap;
        LaneCols c;
        c.own = j >= jlo && j < jhi;
        c.ox = 2 * xs0 + j;
        c.xoff = (unsigned)(c.own ? c.ox : 0) * (XU8 ? 3u : 4u);
        c.c0 = (j + 1) * D4_CPITCH;
        c.c1 = (j + 2) * D4_CPITCH;
        c.c2 = c.ox + 1 < W2 ? (j + 3) * D4_CPITCH : 0;
        return c;
    };
    const unsigned x_frame_bytes = (unsigned)plane * (XU8 ? 3u : 12u);
    auto load_x = [&](unsigned (&dst)[3], const D4Phase& ph, const LaneCols& lc) {   // original-input values of ph's output row, raw
        int yo = ph.yp - 1;                                              // clamped into the band (rows outside are never used)
        yo = yo < 2 * ph.r0 ? 2 * ph.r0 : yo;
        yo = yo < 2 * ph.r1 ? yo : 2 * ph.r1 - 1;
        const __amdgpu_buffer_rsrc_t rx = vad_rsrc((const char*)p.x + (size_t)ph.n * x_frame_bytes, x_frame_bytes);
        if (XU8) {
            const unsigned so = (unsigned)(yo * W2) * 3u;
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[c] = (unsigned)(unsigned char)__builtin_amdgcn_raw_buffer_load_b8(rx, (int)lc.xoff, (int)(so + c), 0);
        } else {
            const unsigned so = (unsigned)(yo * W2) * 4u;
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[c] = __float_as_uint(vad_bload1(rx, lc.xoff, so + (unsigned)c * (unsigned)plane * 4u));
        }
    };
    // tanh as vad_tanh computes it, bit for bit, with the two constant factors of exp(2 v) = exp2(v * 2 log2(e)) folded
    // (a multiplication by 2 is exact, so (2 v) * c == v * (2 c))
    auto tanh_ = [](float v) {
        return __builtin_fmaf(-2.0f, vad_rcp(__builtin_amdgcn_exp2f(v * (2.0f * 1.44269504088896341f)) + 1.0f), 1.0f);
    };

    float Oa[3] = {0.f, 0.f, 0.f}, Ob[3] = {0.f, 0.f, 0.f};              // rows yp-1 (two of three terms) and yp (one term)
    // No reset between work items: an item's first output row 2 r0 starts from `Ob = s[0]` of its first phase, and what the
    // previous item left in Oa / Ob only reaches rows 2 r0 - 2 and 2 r0 - 1, which are not this item's to store.

    // ---- combine of phase ph (its P row is in rbuf), values only (the stores follow), cut into stages issued between the
    // GEMM-1 MFMAs of the next phase with a scheduling fence after each: the LDS reads go out first and their latency, like
    // every wait in the combine, passes while the matrix pipe works.
    f32x4 g0a, g0b, g1a, g1b, g2a, g2b;
    float g0c, g1c, g2c, cs[9], fin[3];
    D4Out o;
    auto combine_stage = [&](int k, const D4Phase& ph, const LaneCols& lc, const float* rbuf, const unsigned (&xr)[3]) {
        if (k == 0) {
            // outside the image P is zero padding: the whole row (uniform), or the right neighbour of the last column -
            // read the all-zero column 0 instead (three address selects, not 27 value selects)
            const float* q0 = rbuf + (ph.row_ok ? lc.c0 : 0);
            const float* q1 = rbuf + (ph.row_ok ? lc.c1 : 0);
            const float* q2 = rbuf + (ph.row_ok ? lc.c2 : 0);
            g0a = *(const f32x4*)(q0); g0b = *(const f32x4*)(q0 + 4); g0c = q0[8];
            g1a = *(const f32x4*)(q1 + 12); g1b = *(const f32x4*)(q1 + 16); g1c = q1[20];
            g2a = *(const f32x4*)(q2 + 24); g2b = *(const f32x4*)(q2 + 28); g2c = q2[9];
        } else if (k == 1) {                                             // cs[3 dy + co] = (dx 0 + dx 1) + dx 2, two sums per
            const f32x4 sa = pk_add4(pk_add4(g0a, g1a), g2a);            // instruction (hipcc scalarises a plain vector add here)
#pragma unroll
            for (int i = 0; i < 4; ++i) cs[i] = sa[i];
        } else if (k == 2) {
            const f32x4 sb = pk_add4(pk_add4(g0b, g1b), g2b);
#pragma unroll
            for (int i = 0; i < 4; ++i) cs[4 + i] = sb[i];
            cs[8] = (g0c + g1c) + g2c;
        } else if (k == 3) {
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                fin[co] = Oa[co] + cs[6 + co];
                Oa[co] = Ob[co] + cs[3 + co];
                Ob[co] = cs[co];
                asm volatile("" : "+v"(Oa[co]), "+v"(Ob[co]));           // computed HERE (loop-carried: hipcc would sink the adds to the trip's end)
            }
        } else if (k == 4) {
            o.rc[0] = tanh_(fin[0] + c3b0);
        } else if (k == 5) {
            o.rc[1] = tanh_(fin[1] + c3b1);
        } else if (k == 6) {
            o.rc[2] = tanh_(fin[2] + c3b2);
        } else {
            float d[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) d[c] = (XU8 ? vad_norm_u8(xr[c]) : __uint_as_float(xr[c])) - o.rc[c];
            // spelled out: the three copies of this code (a = 0 / a = 1 blocks, drain) must contract identically
            const float e = __builtin_fmaf(d[2], d[2], __builtin_fmaf(d[1], d[1], d[0] * d[0]));
            o.e = lc.own ? e : 0.f;
            o.ws = d4_wave_sum63(o.e);
        }
    };
    // the per-row partial sum: one predicated buffer store (out-of-range offset when there is nothing to store), no branch
    const unsigned part_frame = (unsigned)H2 * p.nstrips * 4u;           // floats per frame
    auto store_partial = [&](const D4Phase& ph, float ws) {
        const int yo = ph.yp - 1;
        const bool outrow = ph.valid && yo >= 2 * ph.r0 && yo < 2 * ph.r1;   // (uniform) row yo is complete and this band's
        const __amdgpu_buffer_rsrc_t rp = vad_rsrc(p.partials + (size_t)ph.n * part_frame, part_frame * 4u);
        const unsigned off = (outrow && lane == 63) ? (unsigned)((yo * p.nstrips + ph.strip) * 4 + wave) * 4u : VAD_OOB;
        vad_bstore1(ws, rp, off, 0u);
    };
    auto store_maps = [&](const D4Phase& ph, const LaneCols& lc, const D4Out& o) {   // optional outputs (not on the scoring path)
        const int yo = ph.yp - 1;
        if (!(ph.valid && yo >= 2 * ph.r0 && yo < 2 * ph.r1)) return;
        if (lc.own) {
            const size_t o0 = (size_t)ph.n * 3 * plane + (size_t)yo * W2 + lc.ox;
            if (p.recon) {
#pragma unroll
                for (int c = 0; c < 3; ++c) p.recon[o0 + c * plane] = o.rc[c];
            }
            if (p.errmap) p.errmap[(size_t)ph.n * plane + (size_t)yo * W2 + lc.ox] = o.e / 3.0f;
        }
    };
    const bool want_maps = p.recon || p.errmap;

    unsigned item = blockIdx.x;
    if (item >= p.nitems) return;                                        // (the grid never exceeds the work; whole group)
    int n, strip, r0, r1;
    decode(item, n, strip, r0, r1);
    f32x4 fr[4];                                                         // input fragments of the current input row
    load_row(fr, true, n, strip, r0 - 1);
    D4Phase prev = {0, n, strip, r0, r1, 2 * r0, 0};
    LaneCols lc_prev = lane_cols(strip);
    unsigned xc[3] = {0u, 0u, 0u}, xn[3];
    unsigned phase = 0;                                                  // P buffer parity
    const int jl_w = 2 + 2 * (32 * wave + li);                           // LDS column of this lane's b = 0 output

    for (;;) {
        const unsigned nitem = item + gridDim.x;
        const bool have_next = nitem < p.nitems;
        int nn, nstrip, nr0, nr1;
        decode(have_next ? nitem : item, nn, nstrip, nr0, nr1);
        const LaneCols lc = lane_cols(strip);
        bool first = true;                                               // (of this item: `prev` still belongs to the last one)
        for (int iy = r0 - 1; iy <= r1; ++iy) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (a == 0 ? iy < r0 : iy >= r1) continue;               // band ends: only a = 1 of row r0-1, only a = 0 of row r1
                const D4Phase cur = {1, n, strip, r0, r1, 2 * iy + a, iy >= 0 && iy < H};
                float* wbuf = P + (phase & 1u) * D4_BUF;
                const float* rbuf = P + ((phase & 1u) ^ 1u) * D4_BUF;
                STAMP(0);
                // ---- the 64 MFMAs of `cur` with the combine of `prev` between them
                // The bias is the C operand of each chain's first MFMA (no accumulator initialisation to issue); P rows / columns
                // outside the image are computed like any other and never read (combine stage 0).
                f32x16 acc1[2] = {biasv, biasv}, acc2[2];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[b][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc1[b] = MFMA32(wA[a][b][ks][jj], fr[ks][jj], acc1[b]);
                        if (jj & 1) {                                    // behind every fourth MFMA: one combine stage
                            const int k = 2 * ks + (jj >> 1);
                            combine_stage(k, prev, lc_prev, rbuf, xc);
                            if (k == 0) load_x(xn, cur, lc);             // consumed by the NEXT trip's combine
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                // GEMM 1 has issued: this phase's pixels are consumed, take the next phase's (inline asm: as a plain assignment
                // of a loop-carried value hipcc sinks the copy to the end of the trip) ...
#pragma unroll
                for (int c = 0; c < 3; ++c) asm volatile("v_mov_b32 %0, %1" : "=v"(xc[c]) : "v"(xn[c]));
                __builtin_amdgcn_sched_barrier(0);
                // ... and if it was this input row's last phase, the row's registers are free: the next input row in sequence
                // is loaded straight into them (32 MFMAs, the LDS stores and a barrier pass before its first use)
                if (a == 1) load_row(fr, true, n, strip, iy + 1);        // (a = 1 exists only below r1)
                else if (iy == r1) load_row(fr, have_next, nn, nstrip, nr0 - 1);   // band end: the next item's first row
                store_partial(prev, o.ws);
                __builtin_amdgcn_sched_barrier(0);
                // ReLU as an INTEGER max with 0 (negative floats are negative integers): one instruction, where fmaxf costs a
                // canonicalising v_max as well - and, unlike an inline-asm v_max_f32, one whose MFMA hazards hipcc handles
                // All 32 first, then the 32 MFMAs back to back: interleaved, every MFMA waits two states for the VALU result it reads.
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float t = acc1[b][r];                      // (not __builtin_bit_cast on the element: hipcc then reads element 0)
                        const int bits = __float_as_int(t);
                        acc1[b][r] = __int_as_float(bits > 0 ? bits : 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc2[b] = MFMA32(w2f[r], acc1[b][r], acc2[b]);
                STAMP(1);
                // P slots 8 g + 4 lh + {0..3} of this lane's two output pixels (b = 0, 1): one 16-byte store per g
                {
                    float* dst = wbuf + jl_w * D4_CPITCH + 4 * lh;
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            *(f32x4*)(dst + b * D4_CPITCH + 8 * g) =
                                f32x4{acc2[b][4 * g], acc2[b][4 * g + 1], acc2[b][4 * g + 2], acc2[b][4 * g + 3]};
                }
                if (want_maps) store_maps(prev, lc_prev, o);
                STAMP(2);
                __syncthreads();
                STAMP(3);
                prev = cur;
                if (first) { lc_prev = lc; first = false; }
                ++phase;
            }
        }
        if (!have_next) break;
        item = nitem; n = nn; strip = nstrip; r0 = nr0; r1 = nr1;
    }
    // drain: the last phase's combine
    {
        const float* rbuf = P + ((phase & 1u) ^ 1u) * D4_BUF;
#pragma unroll
        for (int k = 0; k < 8; ++k) combine_stage(k, prev, lc_prev, rbuf, xc);
        store_partial(prev, o.ws);
        if (want_maps) store_maps(prev, lc_prev, o);
    }
#ifdef VAD_D4_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 12;
        for (int i = 0; i < 4; ++i) d[i] = st_sum[i];
        d[8] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID
        d[9] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));     // XCC_ID
        d[10] = (__builtin_amdgcn_s_memtime() - st_t0) * 100000ull / (__builtin_amdgcn_s_memrealtime() - st_r0);
        d[11] = st_n;
    }
#endif
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

static unsigned long long* g_d4_dbg = nullptr;   // VAD_D4_STAMPS diagnostic build (tools/d4_stamps.py)
extern "C" int vad_debug_set_dec4_stamps(void* q) { g_d4_dbg = (unsigned long long*)q; return VAD_OK; }
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
    p.dbg = g_d4_dbg;
    // 73 KB of LDS per work-group (two P buffers): above the 64 KB a kernel gets without asking
    static std::atomic<int> per_cu_cached[64];                            // per device: the attribute belongs to the device's code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int per_cu = per_cu_cached[dev].load(std::memory_order_relaxed);
    if (!per_cu) {
        VAD_REQUIRE(hipFuncSetAttribute((const void*)dec4_score_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, D4_LDS_BYTES) == hipSuccess &&
                    hipFuncSetAttribute((const void*)dec4_score_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, D4_LDS_BYTES) == hipSuccess,
                    "dec4_score: this device does not give a work-group %d bytes of LDS", D4_LDS_BYTES);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dec4_score_kernel<false>, 256, D4_LDS_BYTES) != hipSuccess || per_cu < 1) per_cu = 1;
        per_cu_cached[dev] = per_cu;
    }
    // Band height R: a band costs 2 R + 2 phases (the y halo is one input row per band end), and the work-groups of a CU share
    // its matrix pipe - two of them only hide each other's waits (~10 %).  Pick the R with the least phases on the fullest CU.
    // The result does not depend on it (every output pixel's sums are ordered by pixel only).
    const long long ncu = d4_num_cus(), slots = ncu * per_cu;
    int R = g_vad_dec4_band.load(std::memory_order_relaxed);
    if (R <= 0) {
        double best = 0.0;
        for (int r = 1;; r <<= 1) {
            const int rr = r < h ? r : h;
            const long long items_r = (long long)n * p.nstrips * ((h + rr - 1) / rr);
            const long long grid_r = items_r < slots ? items_r : slots;
            const long long per_wg = (items_r + grid_r - 1) / grid_r, wpc = (grid_r + ncu - 1) / ncu;
            const double cost = (double)per_wg * (2 * rr + 2) * (double)wpc * (wpc > 1 ? 0.9 : 1.0);
            if (R <= 0 || cost <= best) { best = cost; R = rr; }       // (ties: the taller band, less halo traffic)
            if (r >= h) break;
        }
    }
    if (R > h) R = h;
    p.R = R;
    p.nbands = (h + R - 1) / R;
    const long long items = (long long)n * p.nbands * p.nstrips;
    VAD_REQUIRE(items < (1ll << 31), "dec4_score: %lld work items out of range", items);
    p.nitems = (unsigned)items;
    const unsigned grid = (unsigned)(items < slots ? items : slots);
    if (fmt == VAD_X_U8_NHWC) hipLaunchKernelGGL(dec4_score_kernel<true>, dim3(grid), dim3(256), D4_LDS_BYTES, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(dec4_score_kernel<false>, dim3(grid), dim3(256), D4_LDS_BYTES, (hipStream_t)stream, p);
    VAD_LAUNCH_CHECK();
    return VAD_OK;
}

extern "C" int vad_dec4_score(const float* in_nhwc, const float* wt_packed, const float* bt, const float* w2_gemm,
                              const float* bias3, const float* x_nchw, float* partials, float* recon_nchw, float* errmap,
                              int n, int h, int w, void* stream) {
    return vad_dec4_score_fmt(in_nhwc, wt_packed, bt, w2_gemm, bias3, x_nchw, VAD_X_F32_NCHW, partials, recon_nchw, errmap, n, h, w, stream);
}
