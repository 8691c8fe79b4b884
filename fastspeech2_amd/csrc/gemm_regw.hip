// k = 1 GEMM with the WEIGHTS IN REGISTERS, for the short-K projections of the frame-level half: the q | k | v
// projection of MultiHeadedAttention (network/encoder.py:57-59, :73-76: three nn.Linear(d, d) fused into one [3d, d]
// matrix), K = d = 256.
//
//   out[m, n] = sum_c X[m, c] * W[n][c] + bias[n]            X [M, 256] bf16, W [N][256] bf16, out bf16 or fp32
//
// Why not the tiled kernel (conv_gemm.hip): with K = 256 a 128 x 128 tile is a prologue, four K steps and an epilogue -
// its waves wait 46 % of their life (rocprofv3 PMC, profiles/r03_k1_pmc_qkv.txt), MFMA busy 20 %, 23 us for 53 MB of
// algorithmic traffic.  Here nothing is tiled over N x K: a wave keeps its 64 output channels x all 256 k of W in
// REGISTERS for the whole launch (32 fragments = 128 VGPRs, loaded once), and the launch is a stream over token rows:
//  * persistent: 2 four-wave workgroups per CU; a workgroup owns 256 output channels (4 waves x 64) and one contiguous,
//    balanced range of token rows, which it walks in pieces of 16 rows;
//  * a piece of X (16 rows x 512 B = 8 KB) comes by LDS-DMA into a ring of four, three pieces ahead, behind a COUNTED
//    vmcnt (the epilogue stores of the pieces in between are part of the count) and one barrier per piece;
//  * per piece a wave reads 8 X fragments from LDS (0.25 reads per MFMA; rows of 512 B, 16-byte slots XOR-swizzled by
//    the row: conflict-free ds_read_b128) and issues 32 MFMAs against its register-resident W;
//  * epilogue per piece straight from the accumulators: W rows are held in a permuted order such that the four lanes of
//    a token hold 32 consecutive channels per store instruction: each instruction writes 64 contiguous bytes per row
//    (with 16-byte pieces 16 bytes apart - a lane's 16 consecutive channels as two stores - the same launch took 27.9
//    instead of 21.3 us: the write path, not HBM, priced the gaps).
//  * workgroup ids are dealt to the XCDs so that the N / 256 workgroups streaming the same rows share one L2.
// Measured (QKV of the bench, M = 25 611, N = 768): 21.3 us against 22.8 - 23.1 us of the tiled kernel.  Ablations on
// the same box: without the stores 16.8 us, without the MFMAs 13.6 us, i.e. the phases add instead of overlapping; and
// loading W into registers is 64 MB of L2 -> CU traffic (512 workgroups x 128 KB), about 4 us of the launch.  Whole-line
// stores through LDS (22.2 us) and ids dealt n-tile-minor (22.0 us) are no better.  See DESIGN.md section 4.6.
#include "common.h"

namespace fs2 {

constexpr int RW_K = 256;                 // channels in (fixed: the W fragments are a compile-time register array)
constexpr int RW_PIECE_ROWS = 16;
constexpr int RW_PIECE_BYTES = RW_PIECE_ROWS * RW_K * 2;      // 8 KB
constexpr int RW_RING = 4;
constexpr int RW_LDS = RW_RING * RW_PIECE_BYTES;              // 32 KB

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

template <bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_regw_kernel(RegwArgs a) {
    constexpr int S = OUT_F32 ? 4 : 2;                 // store instructions per wave and piece
    constexpr int INFLIGHT = 4 + 3 * S;                // vector-memory instructions issued after the DMA of piece p, up to its wait
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;

    // workgroup ids are dealt round-robin to the 8 XCDs: the n_tiles workgroups that stream the SAME token rows (one per
    // 256-channel tile) are given consecutive ids of ONE XCD, so the rows come from HBM once and from that L2 afterwards
    // (with ids n_tile-minor they sat on different XCDs: X fetched n_tiles times, 78 MB instead of 52 MB at the QKV shape)
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
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) as the builtin: hipcc knows W and bias have arrived

    // ---- X pieces by LDS-DMA: piece = 16 rows x 512 B; wave w issues rows 4 w .. 4 w + 3 as two instructions of two rows;
    //      lane l -> row + (l >> 5), 16-byte slot (l & 31) holding the row's chunk slot ^ (row & 15)
    auto issue_piece = [&](int piece, int slot) __attribute__((always_inline)) {
        int p = piece < npieces ? piece : npieces - 1;            // past the end: the last piece again (keeps the counts uniform)
        const int row_base = (u_lo + p) * RW_PIECE_ROWS;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = wave * 4 + q * 2 + (lane >> 5);          // row inside the piece
            int gm = row_base + r;
            gm = gm < a.M ? gm : a.M - 1;
            const int c = ((lane & 31) ^ (r & 15)) * 8;
            __builtin_amdgcn_global_load_lds((rw_gbl_void*)(a.X + (size_t)gm * a.ldx + c),
                                             (rw_lds_void*)(smem + slot * RW_PIECE_BYTES + (wave * 4 + q * 2) * 512), 16, 0, 0);
        }
    };
    issue_piece(0, 0);
    issue_piece(1, 1);
    issue_piece(2, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const unsigned xlane = (unsigned)(i * 512);
    for (int p = 0; p < npieces; ++p) {
        const int slot = p & 3;
        // piece p landed (this wave's part: all but the INFLIGHT youngest vector-memory instructions are done), then every wave's
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 xf[8];
        {
            const unsigned char* xs = smem + slot * RW_PIECE_BYTES + xlane;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) xf[ks] = *(const bf16x8*)(xs + ((((ks << 2) | g) ^ i) << 4));
        }
        // the slot of piece p - 1 is free (its fragments were in registers before this barrier): piece p + 3 goes there
        issue_piece(p + 3, (p + 3) & 3);
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = mma16(wf[nt][ks], xf[ks], acc[nt]);
        // epilogue: lane (g, i) holds token i of the piece, channels cbase + 4 nt + r
        const int m = (u_lo + p) * RW_PIECE_ROWS + i;
        const bool ok = m < a.M && cols_ok;
        const size_t mc = (size_t)(m < a.M ? m : a.M - 1);
        f32x4 v[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) v[nt] = acc[nt] + bv[nt];
        // (the stores are issued by every lane of every wave - predicated ones to a clamped, never-read address pattern would
        //  change nothing for the count: a store instruction counts once per wave whatever its EXEC mask)
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
    a.units_total = (M + RW_PIECE_ROWS - 1) / RW_PIECE_ROWS;
    int groups = 512 / a.n_tiles;                          // two workgroups per CU ...
    if (groups > 8) groups &= ~7;                          // ... and whole rounds of the 8 XCDs, so that the grid stays one round
    if (groups > a.units_total) groups = a.units_total;
    a.groups = groups;
    const int grid = 8 * ((groups + 7) / 8) * a.n_tiles;
    if (out_f32)
        hipLaunchKernelGGL(gemm_regw_kernel<true>, dim3(grid), dim3(256), RW_LDS, stream, a);
    else
        hipLaunchKernelGGL(gemm_regw_kernel<false>, dim3(grid), dim3(256), RW_LDS, stream, a);
    FS2_CHECK_LAUNCH("fs2_gemm_regw");
    return FS2_OK;
}
