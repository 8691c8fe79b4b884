// k = 1 GEMM with the WEIGHTS IN REGISTERS, for the short-K projections of the frame-level half: the q | k | v
// projection of MultiHeadedAttention (network/encoder.py:57-59, :73-76: three nn.Linear(d, d) fused into one [3d, d]
// matrix), K = d = 256.
//
//   out[m, n] = sum_c X[m, c] * W[n][c] + bias[n]            X [M, 256] bf16, W [N][256] bf16, out bf16 or fp32
//
// Why not the tiled kernel (conv_gemm.hip): with K = 256 a 128 x 128 tile is a prologue, four K steps and an epilogue -
// its waves wait 46 % of their life (rocprofv3 PMC, profiles/r03_b_gemm_k1_qkv_pmc.txt), MFMA busy 19 %, 23 us for 53 MB
// of algorithmic traffic.  Here nothing is tiled over N x K: a wave keeps its 64 output channels x all 256 k of W in
// REGISTERS for the whole launch (32 fragments = 128 VGPRs, loaded once), and the launch is a stream over token rows:
//  * persistent: ONE four-wave workgroup per CU; a workgroup owns 256 output channels (4 waves x 64) and one contiguous,
//    balanced range of token rows, which it walks in pieces of up to 64 rows;
//  * a piece of X (64 rows x 512 B = 32 KB) comes by LDS-DMA into a ring of three, two pieces ahead, behind a COUNTED
//    vmcnt (the epilogue stores of the pieces in between are part of the count) and one barrier per piece;
//  * per 16-row tile of a piece a wave reads 8 X fragments from LDS (rows of 512 B, 16-byte slots XOR-swizzled by the
//    row: conflict-free ds_read_b128; requested one tile ahead) and issues 32 MFMAs against its register-resident W;
//  * epilogue per tile straight from the accumulators: W rows are held in a permuted order such that the four lanes of
//    a token hold 32 consecutive channels per store instruction: each instruction writes 64 contiguous bytes per row
//    (with 16-byte pieces 16 bytes apart - a lane's 16 consecutive channels as two stores - the same launch took 27.9
//    instead of 21.3 us: the write path, not HBM, priced the gaps);
//  * workgroup ids are dealt to the XCDs so that the N / 256 workgroups streaming the same rows share one L2.
// History and measurements: DESIGN.md section 4.6.
#include "common.h"

namespace fs2 {

constexpr int RW_K = 256;                 // channels in (fixed: the W fragments are a compile-time register array)
constexpr int RW_RING = 3;                // X pieces in LDS: one being read, two on their way

struct RegwArgs {
    const bf16_t* X;
    const bf16_t* W;
    const float* bias;
    void* out;
    int M, N, ldx, ldo;
    int n_tiles, groups, units_total;
};

typedef __attribute__((address_space(3))) void rw_lds_void;
typedef const __attribute__((address_space(1))) void rw_gbl_void;

// PR: token rows per piece (16 / 32 / 64).  One barrier, one counted wait and one DMA batch per PIECE, so the taller the
// piece the less of a wave's time goes to them: with 16-row pieces and two workgroups per CU (round 3's first form) a
// piece was 32 MFMAs behind a barrier and an exposed LDS round trip, the phases added up instead of overlapping
// (21 us at the QKV shape, 16.8 without the stores, 13.6 without the MFMAs).  Here ONE workgroup per CU (the register
// file is W's: 128 VGPRs, plus a double-buffered X fragment set) walks pieces of up to 64 rows = four 16-row tiles
// whose fragment reads are requested one tile ahead of the MFMAs.
#ifndef RW_WGS
#define RW_WGS 1          // workgroups per CU
#endif
template <bool OUT_F32, int PR>
__global__ __launch_bounds__(256, RW_WGS) void gemm_regw_kernel(RegwArgs a) {
    constexpr int S = OUT_F32 ? 4 : 2;                 // store instructions per wave and 16-row tile
    constexpr int RT = PR / 16;                        // 16-row tiles per piece
    constexpr int PIECE_BYTES = PR * RW_K * 2;
    constexpr int DMA = PIECE_BYTES / 1024 / 4;        // DMA instructions per wave and piece (1 KB = two rows each)
    constexpr int STORES = RT * S;                     // store instructions per wave and piece
    // vector-memory instructions a wave issues between the DMA of piece p and the wait for it (ring of three: stores of
    // p - 2, DMA of p + 1, stores of p - 1)
    constexpr int INFLIGHT = DMA + 2 * STORES;
    static_assert(INFLIGHT < 64 && DMA < 16, "vmcnt field");
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;

    // workgroup ids are dealt round-robin to the 8 XCDs: the n_tiles workgroups that stream the SAME token rows (one per
    // 256-channel tile) are given consecutive ids of ONE XCD, so the rows come from HBM once and from that L2 afterwards
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int n_tile = j % a.n_tiles, grp = (j / a.n_tiles) * 8 + xcd;
    if (grp >= a.groups) return;
    const int u_lo = (int)((long long)a.units_total * grp / a.groups);
    const int u_hi = (int)((long long)a.units_total * (grp + 1) / a.groups);
    const int npieces = u_hi - u_lo;
    if (npieces <= 0) return;
    const int nbase = n_tile * 256 + wave * 64;        // first output channel of this wave

    // ---- W in registers: fragment (nt, ks) = row nbase + (nt >> 1) * 32 + (i >> 2) * 8 + (nt & 1) * 4 + (i & 3),
    //      k = 32 ks + 8 g .. + 7: the accumulators of lane (g, i) are then channels cbase .. + 7 and cbase + 32 .. + 39
    bf16x8 wf[4][8];
    {
        const int wrow0 = nbase + (i >> 2) * 8 + (i & 3);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int r = wrow0 + (nt >> 1) * 32 + (nt & 1) * 4;
            r = r < a.N ? r : a.N - 1;                 // channels past N are never stored
            const bf16_t* wp = a.W + (size_t)r * RW_K + g * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) wf[nt][ks] = *(const bf16x8*)(wp + ks * 32);
        }
    }
    f32x4 bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int cbase = nbase + g * 8;                   // the lane's channels: cbase + 0..7 and cbase + 32 + 0..7
    const bool cols_ok = nbase + 64 <= a.N;
    if (a.bias && cols_ok) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bv[nt] = *(const f32x4*)(a.bias + cbase + (nt >> 1) * 32 + (nt & 1) * 4);
    }

    // ---- X pieces by LDS-DMA: a piece is PR rows x 512 B; one instruction moves two rows (lane l -> row + (l >> 5),
    //      16-byte slot l & 31 holding the row's chunk slot ^ (row & 15)); wave w issues rows 2 (w + 4 q), q < DMA
    auto issue_piece = [&](int piece, int slot) __attribute__((always_inline)) {
        const int p = piece < npieces ? piece : npieces - 1;      // past the end: the last piece again (uniform counts)
        const int row_base = (u_lo + p) * PR;
#pragma unroll
        for (int q = 0; q < DMA; ++q) {
            const int r = 2 * (wave + 4 * q) + (lane >> 5);    // row inside the piece
            int gm = row_base + r;
            gm = gm < a.M ? gm : a.M - 1;
            const int c = ((lane & 31) ^ (r & 15)) * 8;
            __builtin_amdgcn_global_load_lds((rw_gbl_void*)(a.X + (size_t)gm * a.ldx + c),
                                             (rw_lds_void*)(smem + slot * PIECE_BYTES + 2 * (wave + 4 * q) * 512), 16, 0, 0);
        }
    };
    // the first two pieces go out BEHIND the loads of W and the bias: the in-order vmcnt lets the loop start on piece 0 while
    // piece 1 is on its way (as the builtin, so that hipcc's own wait insertion knows W has arrived)
    issue_piece(0, 0);
    issue_piece(1, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70 | DMA);
    __builtin_amdgcn_sched_barrier(0);

    const unsigned xlane = (unsigned)(i * 512);
    for (int p = 0; p < npieces; ++p) {
        const int slot = p % RW_RING;
        // piece p landed (this wave's part), then every wave's.  Trip 0 waited above; trip 1 has DMA(2) and the stores of trip 0
        // behind DMA(1); from trip 2 on the count is INFLIGHT
        if (p >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        else if (p == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA + STORES) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // the slot of piece p - 1 is free (every wave read it before this barrier): piece p + 2 goes there
        issue_piece(p + 2, (p + 2) % RW_RING);
        const unsigned char* xs = smem + slot * PIECE_BYTES + xlane;
        bf16x8 xf[2][8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xf[0][ks] = *(const bf16x8*)(xs + ((((ks << 2) | g) ^ i) << 4));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (rt + 1 < RT) {          // the next tile's fragments are requested before this tile's MFMAs
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    xf[(rt + 1) & 1][ks] = *(const bf16x8*)(xs + (rt + 1) * 16 * 512 + ((((ks << 2) | g) ^ i) << 4));
            }
            f32x4 acc[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = mma16(wf[nt][ks], xf[rt & 1][ks], acc[nt]);
            // epilogue: lane (g, i) holds token i of the tile, channels cbase + 4 nt + r
            const int m = (u_lo + p) * PR + rt * 16 + i;
            const bool ok = m < a.M && cols_ok;
            const size_t mc = (size_t)(m < a.M ? m : a.M - 1);
            f32x4 v[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) v[nt] = acc[nt] + bv[nt];
            // (every wave issues the same number of store instructions per piece: a store counts once per wave whatever its
            //  EXEC mask, and no tile of a piece lies wholly beyond M unless the piece is the matrix's last - where nothing
            //  waits on the count any more)
            if constexpr (OUT_F32) {
                float* o = (float*)a.out + mc * a.ldo + (cols_ok ? cbase : 0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (ok) *(f32x4*)(o + (nt >> 1) * 32 + (nt & 1) * 4) = v[nt];
                }
            } else {
                bf16_t t16[16];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t16[4 * nt + r] = from_f32<bf16_t>(v[nt][r]);
                bf16_t* o = (bf16_t*)a.out + mc * a.ldo + (cols_ok ? cbase : 0);
                if (ok) {
                    *(vec16*)o = *(const vec16*)t16;
                    *(vec16*)(o + 32) = *(const vec16*)(t16 + 8);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every DMA (the walk prefetches past its end) lands before the LDS is given up
}

}  // namespace fs2

// 1 where fs2_gemm_regw serves the shape and is the kernel to use, else 0 (no launch; part of the kernel dispatch)
extern "C" int fs2_gemm_regw_plan(int M, int N, int Cin, int dtype, int has_res, int relu, int out_f32) {
    if (dtype != 1 || Cin != fs2::RW_K || N < 256 || (N % 256) != 0) return 0;
    if (has_res || relu) return 0;
    if (M < 8192) return 0;
    (void)out_f32;
    return 1;
}

extern "C" int fs2_gemm_regw(const void* X, const void* W, const float* bias, void* out, int M, int N, int Cin, int ldx,
                             int ldo, int out_f32, int dtype, hipStream_t stream) {
    using namespace fs2;
    FS2_CHECK_ARG(X && W && out, "fs2_gemm_regw: null X/W/out");
    FS2_CHECK_ARG(dtype == 1 && Cin == RW_K, "fs2_gemm_regw: bf16 with Cin = %d only (dtype=%d Cin=%d)", RW_K, dtype, Cin);
    // (whole 256-channel tiles: every wave of every workgroup then issues the same number of store instructions per piece,
    //  which the counted vmcnt relies on)
    FS2_CHECK_ARG(M > 0 && N >= 256 && N % 256 == 0, "fs2_gemm_regw: M=%d N=%d (N a multiple of 256)", M, N);
    FS2_CHECK_ARG(ldx >= Cin && ldx % 8 == 0 && ldo >= N && ldo % 8 == 0, "fs2_gemm_regw: leading dimensions");
    FS2_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)bias & 15) == 0,
                  "fs2_gemm_regw: pointers must be 16-byte aligned");
    RegwArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.bias = bias; a.out = out;
    a.M = M; a.N = N; a.ldx = ldx; a.ldo = ldo;
    a.n_tiles = (N + 255) / 256;
    int groups = 256 * RW_WGS / a.n_tiles;                 // RW_WGS workgroup(s) per CU ...
    if (groups > 8) groups &= ~7;                          // ... and whole rounds of the 8 XCDs, so that the grid stays one round
    // rows per piece: 64 while a row group still gets about three pieces of them (one barrier and one wait per piece)
    const long rows_per_group = (M + groups - 1) / groups;
#ifdef RW_PR
    const int pr = RW_PR;
#else
    const int pr = rows_per_group >= 192 ? 64 : (rows_per_group >= 96 ? 32 : 16);
#endif
    a.units_total = (M + pr - 1) / pr;
    if (groups > a.units_total) groups = a.units_total;
    a.groups = groups;
    const int grid = 8 * ((groups + 7) / 8) * a.n_tiles;
    const int lds = RW_RING * pr * RW_K * 2;
    static unsigned long long ok[6] = {0, 0, 0, 0, 0, 0};
#define FS2_REGW_LAUNCH(F32, PRR, SLOT)                                                                        \
    do {                                                                                                       \
        fs2_allow_lds((const void*)gemm_regw_kernel<F32, PRR>, lds, &ok[SLOT]);                                \
        hipLaunchKernelGGL((gemm_regw_kernel<F32, PRR>), dim3(grid), dim3(256), lds, stream, a);               \
    } while (0)
    if (out_f32) {
        if (pr == 64) FS2_REGW_LAUNCH(true, 64, 0); else if (pr == 32) FS2_REGW_LAUNCH(true, 32, 1); else FS2_REGW_LAUNCH(true, 16, 2);
    } else {
        if (pr == 64) FS2_REGW_LAUNCH(false, 64, 3); else if (pr == 32) FS2_REGW_LAUNCH(false, 32, 4); else FS2_REGW_LAUNCH(false, 16, 5);
    }
#undef FS2_REGW_LAUNCH
    FS2_CHECK_LAUNCH("fs2_gemm_regw");
    return FS2_OK;
}
