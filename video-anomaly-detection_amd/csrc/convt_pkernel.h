// ConvTranspose2d(k2, s2) + bias + act as a wave-persistent MFMA GEMM without LDS (included by
// conv_mfma.hip after conv_pkernel.h).
//
//   out[n, 2y+a, 2x+b, co] = bias[co] + sum_ci in[n,y,x,ci] * W[ci,co,a,b]        (reference
//   models/autoencoder.py:104-131, models/video_autoencoder.py:244-256; BatchNorm folded on the host)
//
// GEMM: M = input pixels, K = Cin, N = (quadrant q = 2a+b, co).  Every output element has exactly ONE
// contributing input pixel, so there is no halo and no reuse between waves worth an LDS tile: each
// wave walks work items on its own — an item is MT M-tiles (2 rows x 16 columns each, stacked
// vertically) x NT N-tiles (32 columns each) — with A fragments read straight from the NHWC input
// (a lane's 4 or 8 consecutive channels of its pixel) and B fragments from the L2-resident packed
// weights, both one step ahead in registers.  No barriers, no LDS, occupancy set by registers only.
// These layers are store-bound (dec4.0 writes 8.4 MB per 256x256 frame): the epilogue uses buffer
// stores with wave-uniform (SGPR) offsets per element; out-of-image elements are dropped by the
// descriptor's range check.  PREC 1 = split-fp16 arithmetic (see conv_pkernel.h).
#pragma once

struct ConvTP2 {
    const float* in; long long in_fs;
    const float* w; const float* bias;
    float* out; long long out_fs;
    int n, h, w_, cin, cout;
    int tiles_x, tiles_y;     // M-tile groups per frame: tiles_x = ceil(W/16), tiles_y = ceil(H/(2*MT))
    int ngroups;              // column groups of NT*32 columns: 4*cout / (NT*32)
    unsigned nitems;          // n * tiles_y * tiles_x * ngroups
    int frame_pix;            // UPS 0 only: valid pixels of a frame when its last row is ragged (0 = h*w); loads past them read 0,
                              // stores past them are dropped (buffer range check)
    float* stats;             // nullable (WITH_STATS instantiations, un-activated = training forward, cout <= 128): BatchNorm partial
                              // sums [wave of the grid][2][cout], shifted by the bias
};

// IO16 (PREC 2 only, VAD_PREC_BF16S): input and output tensors are bf16 in memory - a lane's 8 channels of a k-step are ONE
// 16-byte load that IS the MFMA fragment (no conversion), the epilogue rounds to bf16.  UPS 0: the same GEMM without the
// 2x2 scatter = a 1x1 convolution (N index = co; the data gradient of a transposed convolution over the space-to-depth
// gradient, and VideoAutoencoder.proj).
template <int MT, int NT, int ACT, int PREC, int WITH_STATS = 0, int IO16 = 0, int UPS = 1>
__global__ __launch_bounds__(256, (MT * NT * (PREC == 1 ? 2 : 1) <= 4) ? 4 : 2) void convt2x2_pkernel(ConvTP2 p) {
    static_assert(!IO16 || PREC == 2, "bf16 tensors go with bf16 operands");
    constexpr unsigned ES = IO16 ? 2u : 4u;
    constexpr int NQ = UPS ? 4 : 1;
    constexpr int KS = PREC ? 16 : 8;
    // BatchNorm statistics from the accumulators (as in conv_pkernel.h): a wave's items change their channel tile, so one
    // accumulator pair per channel tile (at most 4: cout <= 128), picked by a wave-uniform branch; two levels (item, wave)
    constexpr bool STATS = WITH_STATS && ACT == VAD_ACT_NONE;
    float st_s[STATS ? 4 : 1], st_q[STATS ? 4 : 1];
#pragma unroll
    for (int i = 0; i < (STATS ? 4 : 1); ++i) st_s[i] = st_q[i] = 0.f;
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const unsigned gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int H = p.h, W = p.w_, ctiles = p.cout / 32;
    const int nk = p.cin / KS;
    const unsigned wstep = (unsigned)p.cout * (PREC ? 64u : 32u);      // bytes per (q, k-step) slab
    const unsigned wq = (unsigned)nk * wstep;                          // bytes per quadrant
    const __amdgpu_buffer_rsrc_t rw = vad_rsrc(p.w, (unsigned)NQ * wq);
    const __amdgpu_buffer_rsrc_t rz = vad_rsrc(p.w, 0u);              // zero-sized: every load through it returns 0 at once
    const unsigned fpix = (!UPS && p.frame_pix) ? (unsigned)p.frame_pix : (unsigned)(H * W);
    const unsigned in_bytes = fpix * (unsigned)p.cin * ES;
    const int oh = UPS ? 2 * H : H, ow = UPS ? 2 * W : W;
    const unsigned out_bytes = (UPS ? (unsigned)(oh * ow) : fpix) * (unsigned)p.cout * ES;
    const int prow = li >> 4, pcol = li & 15;                         // A-operand pixel of this lane inside an M-tile

    for (unsigned item = __builtin_amdgcn_readfirstlane(gw); item < p.nitems; item += nw) {
        unsigned r0 = item;
        const int ng = r0 % p.ngroups; r0 /= p.ngroups;
        const int x0 = (r0 % p.tiles_x) * 16; r0 /= p.tiles_x;
        const int y0 = (r0 % p.tiles_y) * (2 * MT);
        const int n = r0 / p.tiles_y;
        const __amdgpu_buffer_rsrc_t ra = vad_rsrc((const char*)p.in + (size_t)n * p.in_fs * ES, in_bytes);
        const __amdgpu_buffer_rsrc_t ro = vad_rsrc((const char*)p.out + (size_t)n * p.out_fs * ES, out_bytes);

        // per-lane offsets: A row of each M-tile (VAD_OOB outside the image -> zeros), B row / bias of each N-tile
        unsigned ao[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int y = y0 + 2 * mt + prow, x = x0 + pcol;
            ao[mt] = (y < H && x < W) ? (unsigned)(__mul24(__mul24(y, W) + x, p.cin) + (PREC ? 8 : 4) * lh) * ES : VAD_OOB;
        }
        unsigned bo[NT];
        int qd[NT], co[NT];
        float bvv[NT];
        f32x16 acc[MT][NT], corr[PREC == 1 ? MT : 1][PREC == 1 ? NT : 1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int g = ng * NT + nt;
            qd[nt] = g / ctiles;
            co[nt] = (g % ctiles) * 32 + li;
            bo[nt] = (unsigned)qd[nt] * wq + (unsigned)co[nt] * (PREC ? 64u : 32u) + (PREC ? 32u : 16u) * lh;
            const float bv = p.bias[co[nt]];
            bvv[nt] = bv;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[mt][nt][r] = bv;
                    if constexpr (PREC == 1) corr[mt][nt][r] = 0.f;
                }
        }

        if constexpr (PREC) {
            f32x4 a0[2][MT], a1[2][MT];        // fp32 channels 8h..8h+3 / 8h+4..8h+7 of the k-step, double-buffered
            f16x8 bh[2][NT], bl[2][NT];
#define CT_LOAD(buf, ks)                                                                          \
    {                                                                                             \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                       \
            a0[buf][mt] = vad_bload4(ra_, ao[mt], (unsigned)(ks) * 16u * ES);                      \
            if constexpr (!IO16) a1[buf][mt] = vad_bload4(ra_, ao[mt], (unsigned)(ks) * 64u + 16u); \
        }                                                                                         \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                       \
            bh[buf][nt] = __builtin_bit_cast(f16x8, vad_bload4(rw_, bo[nt], (unsigned)(ks) * wstep)); \
            bl[buf][nt] = __builtin_bit_cast(f16x8, vad_bload4(rw_, bo[nt] + 16u, (unsigned)(ks) * wstep)); \
        }                                                                                         \
    }
            { const __amdgpu_buffer_rsrc_t ra_ = ra, rw_ = rw; CT_LOAD(0, 0); }
            for (int ks = 0; ks < nk; ks += 2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    {
                        // nk is even (cin % 32 == 0, host-checked).  The prefetch is UNCONDITIONAL - behind the last step it goes
                        // through a zero-sized descriptor (answered with zeros, no memory access): issued under
                        // `if (ks+u+1 < nk)`, hipcc merged the two paths into a vmcnt(0) in front of the MFMAs, which waits for the
                        // prefetch it has just issued - every k-step paid a full L2 round trip.
                        const bool more = ks + u + 1 < nk;
                        const __amdgpu_buffer_rsrc_t ra_ = more ? ra : rz, rw_ = more ? rw : rz;
                        if (u == 0) { CT_LOAD(1, ks + 1); } else { CT_LOAD(0, ks + 2); }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            f16x8 ah, al;
                            if constexpr (IO16) {
                                ah = __builtin_bit_cast(f16x8, a0[u][mt]);       // 8 bf16 channels as loaded
                                al = ah;                                          // (unused by the bf16 MFMA)
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    _Float16 h_, l_;
                                    vad_split_p<PREC>(a0[u][mt][e], h_, l_); ah[e] = h_; al[e] = l_;
                                    vad_split_p<PREC>(a1[u][mt][e], h_, l_); ah[4 + e] = h_; al[4 + e] = l_;
                                }
                            }
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                vad_mma16<PREC>(acc[mt][nt], corr[PREC == 1 ? mt : 0][PREC == 1 ? nt : 0], ah, al, bh[u][nt], bl[u][nt]);
                        }
                    }
                }
            }
#undef CT_LOAD
            if constexpr (PREC == 1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = fmaf(corr[mt][nt][r], 0x1p-11f, acc[mt][nt][r]);
            }
        } else {
            f32x4 a[2][MT], b[2][NT];
#define CT_LOAD(buf, ks)                                                                          \
    {                                                                                             \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) a[buf][mt] = vad_bload4(ra_, ao[mt], (unsigned)(ks) * 32u); \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) b[buf][nt] = vad_bload4(rw_, bo[nt], (unsigned)(ks) * wstep); \
    }
            { const __amdgpu_buffer_rsrc_t ra_ = ra, rw_ = rw; CT_LOAD(0, 0); }
            for (int ks = 0; ks < nk; ks += 2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    {
                        // nk is even (cin % 32 == 0, host-checked).  The prefetch is UNCONDITIONAL - behind the last step it goes
                        // through a zero-sized descriptor (answered with zeros, no memory access): issued under
                        // `if (ks+u+1 < nk)`, hipcc merged the two paths into a vmcnt(0) in front of the MFMAs, which waits for the
                        // prefetch it has just issued - every k-step paid a full L2 round trip.
                        const bool more = ks + u + 1 < nk;
                        const __amdgpu_buffer_rsrc_t ra_ = more ? ra : rz, rw_ = more ? rw : rz;
                        if (u == 0) { CT_LOAD(1, ks + 1); } else { CT_LOAD(0, ks + 2); }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    acc[mt][nt] = MFMA32(a[u][mt][j], b[u][nt][j], acc[mt][nt]);
                    }
                }
            }
#undef CT_LOAD
        }

        // epilogue: D row = (r&3) + 8*(r>>2) + 4*lh -> pixel (prow = row>>4, pcol = row&15) of the M-tile
        const bool full = (y0 + 2 * MT <= H) && (x0 + 16 <= W);
        const unsigned erow = (unsigned)__mul24(ow, p.cout) * ES, ecol = (unsigned)p.cout * ES;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int qa = UPS ? qd[nt] >> 1 : 0, qb = UPS ? qd[nt] & 1 : 0;
            // lane part: column 4*lh of the tile, this lane's channel; uniform part added per element below
            constexpr int sc = UPS ? 2 : 1;
            const unsigned vlane = (unsigned)(__mul24(sc * 4 * lh, p.cout) + co[nt]) * ES;
            const unsigned ubase = ((unsigned)__mul24(sc * y0 + qa, ow) + (unsigned)(sc * x0 + qb)) * ecol;
            float ts = 0.f, tq = 0.f;
            // ONE uniform branch per N-tile, not one per element: interior tiles (nearly all) store through the lane offset as it
            // is - a ReLU and a store with a scalar offset per element, nothing else on the pipe the fp32 MFMAs use (with the test
            // inside the loop hipcc emitted a branch, a compare and a select around every store)
            auto store_tile = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dy = 2 * mt + (r >> 3), dx = (r & 3) + 8 * ((r >> 2) & 1);      // + 4*lh in the lane part
                        const bool ok = FULL || ((y0 + dy) < H && (x0 + dx + 4 * lh) < W);
                        const unsigned so = ubase + (unsigned)(sc * dy) * erow + (unsigned)(sc * dx) * ecol;
                        if constexpr (IO16) vad_bstore_h(vad_f_bf16(vad_act(acc[mt][nt][r], ACT)), ro, ok ? vlane : VAD_OOB, so);
                        else vad_bstore1(vad_act(acc[mt][nt][r], ACT), ro, ok ? vlane : VAD_OOB, so);
                        if constexpr (STATS) {
                            if (p.stats) {   // (uniform)
                                const float d = ok ? acc[mt][nt][r] - bvv[nt] : 0.f;
                                ts += d;
                                tq = fmaf(d, d, tq);
                            }
                        }
                    }
                }
            };
            if (full) store_tile(std::true_type{}); else store_tile(std::false_type{});
            if constexpr (STATS) {
                const int ct = (ng * NT + nt) % ctiles;                                   // wave-uniform
                if (ct == 0) { st_s[0] += ts; st_q[0] += tq; }
                else if (ct == 1) { st_s[1] += ts; st_q[1] += tq; }
                else if (ct == 2) { st_s[2] += ts; st_q[2] += tq; }
                else { st_s[3] += ts; st_q[3] += tq; }
            }
        }
    }
    if constexpr (STATS) {
        if (p.stats) {           // one row per wave of the grid (waves without items write zeros); lane halves hold different pixels
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const float a = st_s[ct] + __shfl_xor(st_s[ct], 32), q = st_q[ct] + __shfl_xor(st_q[ct], 32);
                if (lh == 0 && ct < ctiles) {
                    p.stats[((size_t)gw * 2 + 0) * p.cout + ct * 32 + li] = a;
                    p.stats[((size_t)gw * 2 + 1) * p.cout + ct * 32 + li] = q;
                }
            }
        }
    }
}
