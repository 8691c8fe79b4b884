// Varlen conv1d as an MFMA implicit GEMM on 256 x 256 tiles: the persistent, ping-pong form of conv_gemm.hip for the
// big frame-level FFN convs (network/encoder.py:16-30: Conv1d(d -> U, k) -> ReLU -> Conv1d(U -> d, k), k = 9 at the
// BASELINE dims; same "same"-padding-inside-each-utterance semantics as fs2_conv_gemm).
//
//   out[m, n] = act( sum_{tap, c} X[m + tap - pad, c] * W[tap][n][c] + bias[n] ) (+ res[m, n])
//
// Structure (gfx950, one 512-thread workgroup per CU, 146 KB of LDS):
//  * 8 waves as 2 (tokens) x 4 (channels); a wave owns up to 128 tokens x 64 channels = 8 x 4 MFMA 16x16x32 tiles
//    (128 accumulator registers).  The tile height is a run-time value (<= 256 tokens, in units of 16): the launch is
//    PERSISTENT - <= 256 workgroups, each given one (channel tile, K range) and one contiguous range of token rows which
//    it walks in tiles of equal height - so the work per workgroup is balanced to one 16-token unit whatever M is
//    (a 256 x 256 tiling of 25.6 k tokens x 1024 channels is 404 tiles = 1.58 rounds of 256 CUs).
//  * K is walked in (64-channel chunk, tap) steps.  Per chunk the X rows of the tile plus the 8 halo rows are staged
//    ONCE (33 KB, double-buffered) and every tap reads them as a shifted row view; the W tile of a step (256 channels
//    x 64 k) streams through a ring of five 16-KB half-stages (k 0..31 / k 32..63 of the step), filled four
//    half-steps ahead by LDS-DMA.
//  * Weights come pre-tiled (fs2_pack_weight_tiled): every half-stage is one contiguous 16-KB block that already has
//    the LDS image's order (64-byte rows, 16-byte slots XOR-swizzled for conflict-free ds_read_b128), so a wave's DMA
//    instruction moves 1 KB of consecutive bytes and a workgroup's W stream is one linear, cyclic walk.
//  * A half-step is {ds_read the 4 W + 8 X fragments, counted vmcnt, lgkmcnt(0)} barrier {28 MFMAs with the 2 (+1) DMA
//    pieces of the half-stage four ahead issued between them} barrier {4 MFMAs}.  Waves 4-7 run one barrier behind waves 0-3 (ping-pong): on every SIMD one wave is in its
//    MFMA segment while its partner reads LDS and issues DMA, and a wave's fragments are in its registers before the
//    barrier that starts its MFMA segment.  DMA completion is waited for with COUNTED vmcnt (never 0 in the loop),
//    one half-step before the first read of the data (the counted-vmcnt / raw-barrier scheme of the cdna guide's
//    8-phase template, with longer segments: measured with s_memtime stamps, a barrier costs ~100 cycles of matrix
//    pipe however short the segment between two of them is).
//  * Utterance edges: per-lane validity window of tap shifts per token row, applied to the X fragments only in tiles
//    that contain an edge.
//  * Epilogue as conv_gemm.hip: W rows are fed in a permuted order so that a lane ends with 16 consecutive channels
//    of one token.
#include "common.h"
#include <type_traits>

// Ablation switches for timing experiments (tools/build_variant.sh; results are wrong with any of them set):
//   B2_DBG_FULL: every tile taken as full height, no branches around the MFMA groups; B2_DBG_NOVMWAIT: no counted
//   vmcnt wait; B2_DBG_NODMA: no W DMA; B2_DBG_NOREAD: fragments read once per chunk only
#ifndef B2_DBG_FULL
#define B2_DBG_FULL 0
#endif
#ifndef B2_DBG_NOVMWAIT
#define B2_DBG_NOVMWAIT 0
#endif
#ifndef B2_DBG_NODMA
#define B2_DBG_NODMA 0
#endif
#ifndef B2_DBG_PARITY
#define B2_DBG_PARITY 0
#endif
#ifndef B2_DBG_STAMP
#define B2_DBG_STAMP 0     // s_memtime stamps of one chunk (block 0, waves 0 and 4) into LDS, dumped over the first bytes of `out`
#endif
#ifndef B2_EVEN_TILES
#define B2_EVEN_TILES 0
#endif
#ifndef B2_DBG_NOREAD
#define B2_DBG_NOREAD 0
#endif

namespace fs2 {

constexpr int B2_XROWS = 256 + 8;                 // tile rows + halo of a 9-tap conv
constexpr int B2_XPIECES = B2_XROWS / 8;          // 1-KB DMA pieces (8 rows x 128 B) of an X image
constexpr int B2_XBYTES = B2_XROWS * 128;
constexpr int B2_HS = 256 * 64;                   // one W half-stage: 256 channels x 32 k (bf16)
constexpr int B2_RING = 5;
constexpr int B2_XBASE = B2_RING * B2_HS;
constexpr int B2_LDS = B2_XBASE + 2 * B2_XBYTES + (B2_DBG_STAMP ? 8192 : 0);  // 149504 B
#if B2_DBG_STAMP
#define B2_STAMP(k)                                                                                  \
    do {                                                                                             \
        if (stamp_on) {                                                                              \
            const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                              \
            if (lane == 0) ((unsigned*)(smem + B2_XBASE + 2 * B2_XBYTES + wm * 4096))[HH * 6 + (k)] = t_; \
        }                                                                                            \
    } while (0)
#define B2_GSTAMP(slot)                                                                              \
    do {                                                                                             \
        if (blockIdx.x == 0 && wn == 0) {                                                            \
            const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                              \
            if (lane == 0) ((unsigned*)(smem + B2_XBASE + 2 * B2_XBYTES + wm * 4096))[900 + (slot)] = t_; \
        }                                                                                            \
    } while (0)
#define B2_RSTAMP(slot)                                                                              \
    do {                                                                                             \
        if (blockIdx.x == 0 && wn == 0) {                                                            \
            const unsigned t_ = (unsigned)__builtin_amdgcn_s_memrealtime();                          \
            if (lane == 0) ((unsigned*)(smem + B2_XBASE + 2 * B2_XBYTES + wm * 4096))[900 + (slot)] = t_; \
        }                                                                                            \
    } while (0)
#else
#define B2_STAMP(k) do { } while (0)
#define B2_GSTAMP(slot) do { } while (0)
#define B2_RSTAMP(slot) do { } while (0)
#endif

struct Big2Args {
    const bf16_t* X;
    const unsigned char* Wt;      // tiled weights (fs2_pack_weight_tiled)
    const float* bias;
    const float* res;
    void* out;
    const int2* seg;
    int M, N, Cin, ldx, ldr, ldo, relu, out_f32;
    int nchunks, n_tiles, ksplit, m_groups, units_total, xcd_per_unit, krot;
    long long split_stride;
};

typedef __attribute__((address_space(3))) void b2_lds_void;
typedef const __attribute__((address_space(1))) void b2_gbl_void;
__device__ __forceinline__ void b2_glds16(const void* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((b2_gbl_void*)gsrc, (b2_lds_void*)lds_wave_base, 16, 0, 0);
}

template <int I, int N, class F> __device__ __forceinline__ void b2_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        b2_static_for<I + 1, N>(f);
    }
}

#define B2_BARRIER()                            \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)

// every LDS read of the wave has returned (s_waitcnt with vmcnt / expcnt at their maxima, lgkmcnt 0): written as the
// builtin, so that hipcc's own wait insertion knows about it and adds none behind the branches that follow
#define B2_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)

// X pieces a wave issues in half-step hh of a chunk (for the NEXT chunk's image): 5 per wave and chunk (8 waves x 5
// >= 33 pieces), no earlier than half-step 1 (the target buffer was read until the last phase of the previous chunk)
// and no later than half-step HSPC - 4 (the counted wait of the chunk's last half-step covers what was issued before
// its last three half-steps).
template <int KT> __host__ __device__ constexpr int b2_xcount(int hh) {
    constexpr int HSPC = 2 * KT;
    constexpr int last = HSPC - 4 < 5 ? HSPC - 4 : 5;      // issue half-steps 1 .. last
    if (hh < 1 || hh > last) return 0;
    const int base = 5 / last, extra = 5 % last;
    return base + (hh <= extra ? 1 : 0);
}
template <int KT> __host__ __device__ constexpr int b2_xfirst(int hh) {     // index of the first piece slot of half-step hh
    int n = 0;
    for (int h = 0; h < hh; ++h) n += b2_xcount<KT>(h);
    return n;
}
// DMA instructions a wave issues in the MFMA segments of the two half-steps before hh (cyclic over the chunk): the
// count its wait in the load segment of hh leaves in flight
template <int KT> __host__ __device__ constexpr int b2_inflight(int hh) {
    constexpr int HSPC = 2 * KT;
    int n = 4;
    for (int d = 1; d < 3; ++d) n += b2_xcount<KT>((hh - d + HSPC) % HSPC);
    return n;
}

template <int KT>
__global__ __launch_bounds__(512, 2) void conv_gemm256_kernel(Big2Args a) {
    static_assert(KT >= 3 && KT <= 9 && (KT & 1), "taps");
    constexpr int HSPC = 2 * KT;
    constexpr int pad = (KT - 1) / 2;
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if B2_DBG_PARITY
    const int wm = wave & 1, wn = wave >> 1;
#else
    const int wm = wave >> 2, wn = wave & 3;
#endif
    const int g = lane >> 4, i = lane & 15;

    // ---- which (channel tile, K range, row group) this workgroup owns
    const int U = a.n_tiles * a.ksplit;
    int unit, mg;
    if (a.xcd_per_unit > 0) {      // workgroup ids are dealt round-robin to the 8 XCDs: a unit's W stream stays in few L2s
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        unit = xcd / a.xcd_per_unit;
        mg = slot * a.xcd_per_unit + xcd % a.xcd_per_unit;
    } else {
        unit = blockIdx.x % U;
        mg = blockIdx.x / U;
    }
    if (mg >= a.m_groups) return;
    const int n_tile = unit % a.n_tiles, kpart = unit / a.n_tiles;
    const int n0 = n_tile * 256;
    const int cps = (a.nchunks + a.ksplit - 1) / a.ksplit;
    const int c_lo = kpart * cps;
    const int nch = min(a.nchunks, c_lo + cps) - c_lo;
    if (nch <= 0) return;
    const int u_lo = (int)((long long)a.units_total * mg / a.m_groups);
    const int u_hi = (int)((long long)a.units_total * (mg + 1) / a.m_groups);
    const int nu_wg = u_hi - u_lo;
    if (nu_wg <= 0) return;
    const int ntile_wg = (nu_wg + 15) >> 4;
    const int rot = a.krot ? mg % nch : 0;            // phase of the cyclic chunk walk (spreads the L2 requests for W)

    const bf16_t* __restrict__ X = a.X;
    const unsigned char* wbase = a.Wt + (size_t)(n_tile * a.nchunks + c_lo) * (HSPC * B2_HS);
    const int range_hs = nch * HSPC;

    // ---- per-lane constants
    // W fragment of channel tile nt: row wn * 64 + (i >> 2) * 16 + 4 nt + (i & 3) (the permuted order that leaves a lane
    // with 16 consecutive channels), 64-byte rows, slot g ^ swz with swz = 3 * bit 5 of the row
    const unsigned wlane = (unsigned)((wn * 64 + (i >> 2) * 16 + (i & 3)) * 64 + ((g ^ (((i >> 3) & 1) * 3)) << 4));
    const int prow = lane >> 3, pslot = lane & 7;     // X piece: lane -> row 8 p + prow, 16-byte slot pslot

    // ---- DMA issue helpers
    int pf = rot * HSPC;                              // next half-stage of the W stream to prefetch (index in the range)
    auto issue_w = [&](int ring_slot, int sub) __attribute__((always_inline)) {
        const int piece = sub * 8 + wave;
        b2_glds16(wbase + (size_t)pf * B2_HS + piece * 1024 + lane * 16, smem + ring_slot * B2_HS + piece * 1024);
    };
    auto issue_x = [&](int p, int row0, int c0, int buf) __attribute__((always_inline)) {
        p = p < B2_XPIECES ? p : B2_XPIECES - 1;      // spare slots repeat the last piece (same bytes, same place)
        const int r = 8 * p + prow;
        int gm = row0 + r;
        gm = gm < 0 ? 0 : (gm >= a.M ? a.M - 1 : gm); // rows outside the matrix are always masked
        const int c = c0 + ((pslot ^ (r & 7)) << 3);
        b2_glds16(X + (size_t)gm * a.ldx + c, smem + B2_XBASE + buf * B2_XBYTES + p * 1024);
    };

    // tile j of this workgroup: units [u_lo + nu_wg * j / ntile_wg, u_lo + nu_wg * (j + 1) / ntile_wg)
    auto tile_units = [&](int j, int& t_u0, int& t_nu) __attribute__((always_inline)) {
#if B2_EVEN_TILES
        t_u0 = u_lo + nu_wg * j / ntile_wg;
        t_nu = u_lo + nu_wg * (j + 1) / ntile_wg - t_u0;
#else
        // full tiles first, the remainder last: a half-step costs its LDS reads and barriers whatever the tile height,
        // so 256 + 144 rows beat 200 + 200
        t_u0 = u_lo + 16 * j;
        t_nu = min(16, u_hi - t_u0);
#endif
    };

    // bias of the lane's 16 channels, loaded once (an ordinary load inside the tile loop would make hipcc wait vmcnt(0)
    // there: every DMA in flight)
    const int nbase = n0 + wn * 64 + g * 16;
    const bool first = kpart == 0;                    // split-K: bias and residual go into partial 0
    f32x4 bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.bias && first) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bv[nt] = *(const f32x4*)(a.bias + nbase + 4 * nt);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0), as the builtin: hipcc then knows the bias has arrived
    B2_GSTAMP(0);
    B2_RSTAMP(30);
    // ---- prologue: X image of (tile 0, first chunk) and half-stages 0..3
    {
        int t_u0, t_nu;
        tile_units(0, t_u0, t_nu);
        for (int p = wave; p < 40; p += 8) issue_x(p, t_u0 * 16 - pad, (c_lo + rot) * 64, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            issue_w(k, 0);
            issue_w(k, 1);
            pf = pf + 1 == range_hs ? 0 : pf + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // X image and half-stage 0 landed; 1..3 (2 instructions each) may still fly
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();        // waves 4-7 run one barrier behind (ping-pong)

    B2_GSTAMP(1);
    int rslot = 0, wslot = 4;                         // ring slot read in this half-step / filled in this half-step
    int xbuf = 0;

    f32x4 acc[8][4];
    const int total_chunks = ntile_wg * nch;
    int qglobal = 0;

    for (int tj = 0; tj < ntile_wg; ++tj) {
        int t_u0, t_nu;
        tile_units(tj, t_u0, t_nu);
        const int t_lo = t_u0 * 16;
        // units of this wave: waves 0-3 take the first ceil(t_nu / 2) units of the tile, waves 4-7 the rest; each half
        // of the wave's units goes to one of the two phases of a half-step
        const int u0w = (t_nu + 1) >> 1;
        const int nuw = wm == 0 ? u0w : t_nu - u0w;
        const int rowoff = wm == 0 ? 0 : u0w * 16;
        const int na = __builtin_amdgcn_readfirstlane((nuw + 1) >> 1);
        const int nb = __builtin_amdgcn_readfirstlane(nuw - na);
        // per lane and accumulator slot s: the set of taps whose shifted input row lies outside the utterance of the lane's
        // token row (bit `tap` of a 9-bit field, three slots per register), and per wave the set of slots in which any
        // lane has such a tap: only those slots' fragments are masked
        unsigned kb[3] = {0u, 0u, 0u};
        unsigned mflags = 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int act = s < 4 ? (s < na) : ((s - 4) < nb);
            const int m = t_lo + rowoff + ((s < 4 ? s : na + s - 4) * 16) + i;
            // (loaded unconditionally from a clamped row: a guarded load makes hipcc branch around it and wait for it alone)
            const int2 sg = a.seg[m < a.M ? m : a.M - 1];
            const bool ok = act && m < a.M;
            const int l = ok ? sg.x - m : 1, h = ok ? sg.y - 1 - m : 0;      // valid shifts: l .. h
            unsigned bits = 0;
#pragma unroll
            for (int t = 0; t < KT; ++t) bits |= ((t - pad) < l || (t - pad) > h) ? (1u << t) : 0u;
            if (!act) bits = 0;
            if (__ballot(bits != 0u) != 0ull) mflags |= 1u << s;
            kb[s / 3] |= bits << (9 * (s % 3));
        }
        mflags = __builtin_amdgcn_readfirstlane(mflags);
        B2_GSTAMP(4 + 3 * (tj < 4 ? tj : 4) - 3);    // (tile set-up done: slot 1 + 3 tj)
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[s][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int q = 0; q < nch; ++q, ++qglobal) {
            // X image to prefetch during this chunk: the next (tile, chunk) of the walk; past the end: this one again
            int nx_row0, nx_c0;
            {
                int ntj = tj, nq = q + 1;
                if (nq == nch) { nq = 0; ntj = tj + 1; }
                if (qglobal + 1 >= total_chunks) { ntj = tj; nq = q; }
                int n_u0, n_nu;
                tile_units(ntj, n_u0, n_nu);
                nx_row0 = n_u0 * 16 - pad;
                int cc = rot + nq;
                cc = cc >= nch ? cc - nch : cc;
                nx_c0 = (c_lo + cc) * 64;
            }
            const unsigned xb = (unsigned)(B2_XBASE + xbuf * B2_XBYTES);
#if B2_DBG_STAMP
            const bool stamp_on = blockIdx.x == 0 && wn == 0 && tj == 0 && q == 1;
#endif
#if B2_DBG_NOREAD
            bf16x8 dbg_w[4], dbg_x[8];
#endif

            b2_static_for<0, HSPC>([&](auto hh_c) __attribute__((always_inline)) {
                constexpr int HH = decltype(hh_c)::value;
                constexpr int tap = HH >> 1, ks = HH & 1;
                constexpr int shift = tap - pad;
                constexpr int NXI = b2_xcount<KT>(HH);
                constexpr int XF = b2_xfirst<KT>(HH);
                bf16x8 wf[4], xf[8];
#if B2_DBG_NOREAD
                if (HH > 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wf[j] = dbg_w[j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[j] = dbg_x[j];
                }
#endif
#if B2_DBG_FULL
                constexpr int na_l = 4, nb_l = 4;
#else
                int na_l = na, nb_l = nb;             // (opaque: keeps the compares scalar and inside the loop)
                asm volatile("" : "+s"(na_l), "+s"(nb_l));
#endif
                int i_l = i;
                asm volatile("" : "+v"(i_l));         // (opaque: else the 18 per-lane X addresses of a chunk are hoisted into registers)
                const int row = rowoff + i_l + tap;
                const unsigned xaddr = xb + (unsigned)(row * 128 + ((((ks << 2) | g) ^ (row & 7)) << 4));
                // ---------------- load segment: the fragments of this half-step, the DMA of the one four ahead
                if (!B2_DBG_NOREAD || HH == 0) {
                    const unsigned char* ws = smem + rslot * B2_HS + wlane;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const bf16x8*)(ws + nt * 256);
                    const unsigned char* xs = smem + xaddr;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xf[j] = *(const bf16x8*)(xs + j * 2048);
                    const unsigned char* xs1 = xs + na_l * 2048;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xf[4 + j] = *(const bf16x8*)(xs1 + j * 2048);
                }
#if B2_DBG_NOREAD
                if (HH == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dbg_w[j] = wf[j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) dbg_x[j] = xf[j];
                }
#endif
                // the next half-stage and, in the last half-step of a chunk, the next X image have landed once all but
                // the DMAs issued in the last two MFMA segments are done
                B2_STAMP(4);
#if !B2_DBG_NOVMWAIT
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(b2_inflight<KT>(HH)) : "memory");
#endif
                B2_STAMP(5);
                B2_LGKM0();          // fragments in registers BEFORE the barrier: the MFMAs start the moment it opens
                // utterance edges: zero the fragments of (flagged slot, tap) pairs that would read across an edge - here,
                // under the partner wave's MFMA segment, not behind the barrier
                unsigned mf = mflags;
                asm volatile("" : "+s"(mf));          // (opaque: a scalar test inside the loop, not a hoisted lane mask)
                if (mf != 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (mf & (1u << j)) {
                            unsigned w = kb[j / 3];
                            asm volatile("" : "+v"(w));               // (keeps the bit test inside the branch)
                            if ((w >> (9 * (j % 3) + tap)) & 1u) xf[j] = frag_zero<bf16_t>();
                        }
                }
                B2_STAMP(0);
                B2_BARRIER();
                B2_STAMP(1);
                // ---------------- MFMA segment: all slots but slot 4 ...
                __builtin_amdgcn_s_setprio(1);
                // (the DMA of the half-stage four ahead - and of the next X image - is issued between the MFMAs: an LDS-DMA
                // instruction costs its wave 60-180 cycles of issue, which here run under the matrix pipe's work instead of
                // in the load segment, where they made that segment longer than the partner's MFMA segment)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j != 4 && __builtin_expect(j < 4 ? j < na_l : j - 4 < nb_l, 1)) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[j][nt] = mma16(wf[nt], xf[j], acc[j][nt]);
                    }
#if !B2_DBG_NODMA
                    if (j == 0) issue_w(wslot, 0);
                    if (j == 2) issue_w(wslot, 1);
#endif
                    if (j == 5) {
#pragma unroll
                        for (int k = 0; k < NXI; ++k) issue_x((XF + k) * 8 + wave, nx_row0, nx_c0, xbuf ^ 1);
                    }
                }
                B2_STAMP(2);
                B2_BARRIER();
                // ... and slot 4 BEHIND the barrier that starts the partner's MFMA segment: these four keep the matrix
                // pipe fed while the partner wakes up (MFMAs are register-only: no memory ordering depends on their place)
                if (__builtin_expect(nb_l > 0, 1)) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[4][nt] = mma16(wf[nt], xf[4], acc[4][nt]);
                }
                __builtin_amdgcn_s_setprio(0);
                B2_STAMP(3);
                rslot = rslot + 1 == B2_RING ? 0 : rslot + 1;
                wslot = wslot + 1 == B2_RING ? 0 : wslot + 1;
                pf = pf + 1 == range_hs ? 0 : pf + 1;
            });
            xbuf ^= 1;
        }

        B2_GSTAMP(2 + 3 * (tj < 4 ? tj : 4));
        // ---- epilogue of the tile: lane (g, i) holds token m of slot s, channels nb .. nb + 15 (acc[s][nt][r] <-> nb + 4 nt + r)
        {
            const int t_end = t_lo + t_nu * 16;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bool act = s < 4 ? (s < na) : ((s - 4) < nb);
                if (!act) continue;
                const int m = t_lo + rowoff + ((s < 4 ? s : na + s - 4) * 16) + i;
                const bool mok = m < a.M && m < t_end;
                const size_t mc = (size_t)(mok ? m : a.M - 1);
                f32x4 v[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) v[nt] = acc[s][nt] + bv[nt];
                if (a.relu) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[nt][r] = fmaxf(v[nt][r], 0.f);
                }
                if (a.res && first) {
                    f32x4 rv[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) rv[nt] = *(const f32x4*)(a.res + mc * a.ldr + nbase + 4 * nt);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) v[nt] += rv[nt];
                }
                if (a.out_f32) {
                    float* o = (float*)a.out + (size_t)kpart * a.split_stride + mc * a.ldo + nbase;
                    if (mok) {
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) *(f32x4*)(o + 4 * nt) = v[nt];
                    }
                } else {
                    bf16_t* o = (bf16_t*)a.out + mc * a.ldo + nbase;
                    bf16_t t16[16];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) t16[4 * nt + r] = from_f32<bf16_t>(v[nt][r]);
                    if (mok) {
                        *(vec16*)o = *(const vec16*)t16;
                        *(vec16*)(o + 8) = *(const vec16*)(t16 + 8);
                    }
                }
            }
        }
    }
    B2_GSTAMP(20);
    B2_RSTAMP(31);
    // every LDS-DMA (the walk prefetches past its end) has to land before the workgroup gives up its LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wm == 0) __builtin_amdgcn_s_barrier();
#if B2_DBG_STAMP
    if (blockIdx.x == 0 && wn == 0 && lane == 0) {
        const unsigned* st = (const unsigned*)(smem + B2_XBASE + 2 * B2_XBYTES + wm * 4096);
        for (int k = 0; k < 1024; ++k) ((unsigned*)a.out)[wm * 1024 + k] = st[k];
    }
#endif
}

__global__ void pack_weight_tiled_kernel(const bf16_t* __restrict__ src, uint4* __restrict__ dst, int N, int Cin, int taps,
                                         long long total16) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total16) return;
    const int slot = (int)(idx & 3), row = (int)((idx >> 2) & 255), half = (int)((idx >> 10) & 1);
    long long rest = idx >> 11;
    const int nchunks = Cin / 64;
    const int tap = (int)(rest % taps);
    rest /= taps;
    const int chunk = (int)(rest % nchunks);
    const int n_tile = (int)(rest / nchunks);
    const int n = n_tile * 256 + row;
    const int c = chunk * 64 + half * 32 + ((slot ^ (((row >> 5) & 1) * 3)) << 3);
    dst[idx] = *(const uint4*)(src + ((size_t)tap * N + n) * Cin + c);
}

}  // namespace fs2

// ---- host side ------------------------------------------------------------------------------------------------

// 0: the 256 x 256 kernel does not serve the shape, or fs2_conv_gemm is the faster one for it; else the number of K ranges
// (partial outputs) it will use when the caller allows split-K (`want_split` != 0: an fs2_add_layernorm call sums the
// fp32 partials), 1 otherwise.
// A half-step of this kernel costs its LDS reads and two barriers whatever the tile height, so it wins only where the
// rows of a workgroup fill whole 256-row tiles: measured against fs2_conv_gemm over M = 12 k .. 51 k tokens
// (tools/conv256_sweep.py, both FFN conv shapes): 0.86-0.94 of its time where the last tile of a workgroup is >= 14 of
// 16 units high (one tile per workgroup for 256 -> 1024 at M ~ 14-16 k; 1024 -> 256 at M ~ 28-33 k), 1.05-1.7 x
// elsewhere.  M == 1 << 20 asks whether the shape is served at all (weight packing).
extern "C" int fs2_conv_gemm256_plan(int M, int N, int Cin, int taps, int dtype, int want_split) {
    if (dtype != 1 || (taps != 9 && taps != 3)) return 0;
    if (N <= 0 || (N % 256) != 0 || Cin <= 0 || (Cin % 64) != 0) return 0;
    if (M < 4096) return 0;
    const int n_tiles = N / 256;
    if (n_tiles > 8 || (8 % n_tiles) != 0) return 0;
    int ks = 1;
    if (want_split) {
        // the fewest K ranges (at most 4, at least two chunks each) that give a workgroup >= 192 token rows when the 256
        // workgroups are dealt to (channel tile, K range, row group): taller tiles re-use each streamed W byte more
        const int nchunks = Cin / 64;
        while (ks < 4 && n_tiles * ks * 2 <= 8 && nchunks / (ks * 2) >= 2 && (long long)M * n_tiles * ks < 192LL * 256) ks *= 2;
    }
    if (M == (1 << 20)) return ks;
    if (ks > 2) return 0;                                           // (four K ranges: not measured against fs2_conv_gemm)
    const int groups = 256 / (n_tiles * ks);
    const int units = ((M + 15) / 16 + groups - 1) / groups;        // 16-row units of the largest row group
    const int last = units % 16;
    if (units > 32 || (last != 0 && last < 14)) return 0;
    return ks;
}

extern "C" int fs2_pack_weight_tiled(const void* W, void* dst, int N, int Cin, int taps, int dtype, hipStream_t stream) {
    FS2_CHECK_ARG(W && dst, "fs2_pack_weight_tiled: null pointer");
    FS2_CHECK_ARG(dtype == 1, "fs2_pack_weight_tiled: dtype=%d (bf16 only)", dtype);
    FS2_CHECK_ARG(N > 0 && N % 256 == 0 && Cin > 0 && Cin % 64 == 0 && taps >= 1,
                  "fs2_pack_weight_tiled: N=%d must be a multiple of 256, Cin=%d of 64", N, Cin);
    FS2_CHECK_ARG(((uintptr_t)W & 15) == 0 && ((uintptr_t)dst & 15) == 0, "fs2_pack_weight_tiled: 16-byte alignment");
    const long long total16 = (long long)taps * N * Cin / 8;
    hipLaunchKernelGGL(fs2::pack_weight_tiled_kernel, dim3((unsigned)((total16 + 255) / 256)), dim3(256), 0, stream,
                       (const fs2::bf16_t*)W, (uint4*)dst, N, Cin, taps, total16);
    FS2_CHECK_LAUNCH("fs2_pack_weight_tiled");
    return FS2_OK;
}

extern "C" int fs2_conv_gemm256(const void* X, const void* Wt, const float* bias, const float* res, void* out,
                                const int* seg, int M, int N, int Cin, int taps, int ldx, int ldr, int ldo, int relu,
                                int out_f32, int dtype, int ksplit, long long split_stride, hipStream_t stream) {
    using namespace fs2;
    FS2_CHECK_ARG(X && Wt && out && seg, "fs2_conv_gemm256: null X/W/out/seg");
    FS2_CHECK_ARG(dtype == 1 && (taps == 9 || taps == 3), "fs2_conv_gemm256: serves bf16 convs with 3 or 9 taps (dtype=%d taps=%d)", dtype, taps);
    FS2_CHECK_ARG(M > 0 && N > 0 && N % 256 == 0 && Cin > 0 && Cin % 64 == 0 && (8 % (N / 256)) == 0 && N / 256 <= 8,
                  "fs2_conv_gemm256: M=%d N=%d (multiple of 256, N/256 in {1,2,4,8}) Cin=%d (multiple of 64)", M, N, Cin);
    FS2_CHECK_ARG(ldx >= Cin && ldx % 8 == 0 && ldo >= N && ldo % 8 == 0 && (!res || (ldr >= N && ldr % 4 == 0)),
                  "fs2_conv_gemm256: leading dimensions");
    FS2_CHECK_ARG(relu == 0 || relu == 1, "fs2_conv_gemm256: activation code %d (0 none, 1 relu)", relu);
    FS2_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)Wt & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                      ((uintptr_t)bias & 15) == 0 && ((uintptr_t)res & 15) == 0,
                  "fs2_conv_gemm256: pointers must be 16-byte aligned");
    const int nchunks = Cin / 64, n_tiles = N / 256;
    FS2_CHECK_ARG(ksplit >= 1 && ksplit <= nchunks && n_tiles * ksplit <= 8 && (8 % (n_tiles * ksplit)) == 0,
                  "fs2_conv_gemm256: ksplit=%d (N/256 * ksplit must divide 8)", ksplit);
    FS2_CHECK_ARG(ksplit == 1 || (!relu && out_f32 && split_stride >= (long long)M * ldo && split_stride % 4 == 0),
                  "fs2_conv_gemm256: split-K partial outputs are fp32, take no activation and need split_stride >= M*ldo");
    Big2Args a;
    a.X = (const bf16_t*)X; a.Wt = (const unsigned char*)Wt; a.bias = bias; a.res = res; a.out = out; a.seg = (const int2*)seg;
    a.M = M; a.N = N; a.Cin = Cin; a.ldx = ldx; a.ldr = ldr; a.ldo = ldo; a.relu = relu; a.out_f32 = out_f32;
    a.nchunks = nchunks; a.n_tiles = n_tiles; a.ksplit = ksplit;
    a.units_total = (M + 15) / 16;
    const int U = n_tiles * ksplit;
    a.xcd_per_unit = 8 / U;
    int groups = 256 / U;                               // one workgroup per CU
    if (groups > a.units_total) groups = a.units_total;
    a.m_groups = groups;
    a.krot = 1;
    a.split_stride = ksplit > 1 ? split_stride : 0;
    const int grid = 8 * ((groups + a.xcd_per_unit - 1) / a.xcd_per_unit);
    static unsigned long long lds_ok9 = 0, lds_ok3 = 0;
    if (taps == 9) {
        fs2_allow_lds((const void*)conv_gemm256_kernel<9>, B2_LDS, &lds_ok9);
        hipLaunchKernelGGL((conv_gemm256_kernel<9>), dim3(grid), dim3(512), B2_LDS, stream, a);
    } else {
        fs2_allow_lds((const void*)conv_gemm256_kernel<3>, B2_LDS, &lds_ok3);
        hipLaunchKernelGGL((conv_gemm256_kernel<3>), dim3(grid), dim3(512), B2_LDS, stream, a);
    }
    FS2_CHECK_LAUNCH("fs2_conv_gemm256");
    return FS2_OK;
}
