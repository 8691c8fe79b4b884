// Output projection + residual + LayerNorm in one pass over the token rows (post-LN FFT block, frame level):
//
//   y[m, :] = LayerNorm( res[m, :] + X[m, :] . W^T + bias )        (eps 1e-12, network/utils.py:39-47)
//
// i.e. `x = norm1(residual + linear_out(context))` of EncoderLayer.forward (network/encoder.py:89 and :171-180 with
// normalize_before = False), for d = 256: X [M, 256] bf16 (the attention context), W [256][256] bf16, res [M, 256]
// fp32 (the residual stream), outputs fp32 [M, 256] (the new residual stream) and bf16 [M, 256] (the next GEMM's operand).
//
// As two launches (fs2_conv_gemm with the residual in its epilogue, then fs2_add_layernorm) the pair moves the fp32
// projection output through HBM twice and takes 20.7 + ~11 us at the bench batch (25.6 k rows), the GEMM at a quarter
// of the HBM rate (a 128 x 128 tile with K = 256 is all prologue and epilogue).  Here the launch
// is one stream over the rows, bounded by its 3 KB of HBM traffic per row:
//  * the whole W stays in REGISTERS: a workgroup is 4 waves, wave w holds channels 64 w .. 64 w + 63 x all 256 k
//    (32 fragments = 128 VGPRs, loaded once), so a workgroup owns whole rows and the LayerNorm statistics are two
//    exchanges of 16 floats per wave through LDS;
//  * persistent: two workgroups per CU and a contiguous, balanced range of 16-row pieces each; a piece (X: 8 KB, res:
//    16 KB) comes by LDS-DMA into a ring of three, two pieces ahead, behind a COUNTED vmcnt (the stores of the pieces in
//    between are part of the count; no ordinary global load inside the loop - hipcc would wait vmcnt(0) for its first
//    use and drain the ring);
//  * LDS rows are XOR-swizzled by the row (16-byte slots): the X fragment reads and the residual's 16-byte reads are
//    conflict-free; the swizzle is applied to the DMA's per-lane source address;
//  * A = W rows in natural order, B = X rows: lane (g, i) ends with token i, channels 64 w + 16 nt + 4 g + r - the four
//    lanes of a token write 64 contiguous bytes of fp32 (32 of bf16) per instruction.
#include "common.h"

namespace fs2 {

constexpr int PL_K = 256;                           // channels in = channels out (fixed: W is a compile-time register array)
constexpr int PL_ROWS = 16;                         // rows per piece
constexpr int PL_XB = PL_ROWS * PL_K * 2;           // 8 KB of X per piece
constexpr int PL_RB = PL_ROWS * PL_K * 4;           // 16 KB of residual per piece
constexpr int PL_SLOT = PL_XB + PL_RB;              // 24 KB
// Timing ablations (tools/build_variant.sh TAG -DPL_DBG_NOSTORE | -DPL_DBG_NOLN | -DPL_WGS=2; results are wrong with the
// first two): without the output stores (16 us instead of 22), without the two LayerNorm exchanges (-1 us), two
// workgroups per CU on a ring of three (24.1 us).
#ifndef PL_WGS
#define PL_WGS 1                                    // workgroups per CU (2: ring of three, 75 KB of LDS each; measured 24.1 vs 22.8 us)
#endif
constexpr int PL_RING = PL_WGS == 1 ? 4 : 3;
constexpr int PL_EX = PL_RING * PL_SLOT;            // statistics exchange: [2][4 waves][16 tokens] floats behind the ring
constexpr int PL_VEC = PL_EX + 2 * 4 * 16 * 4;      // bias | gamma | beta, 256 floats each (registers are W's)
constexpr int PL_LDS = PL_VEC + 3 * PL_K * 4;       // 75.5 KB: two workgroups per CU
constexpr int PL_DMA = 6;                           // DMA instructions per wave and piece (2 X + 4 residual)
constexpr int PL_STORES = 8;                        // store instructions per wave and piece (4 fp32 + 4 bf16)
constexpr int PL_INFLIGHT = (PL_RING - 2) * PL_DMA + (PL_RING - 1) * PL_STORES;   // issued between the DMA of piece p and its wait (see the loop)

struct ProjLnArgs {
    const bf16_t* X;
    const bf16_t* W;
    const float* bias;
    const float* res;
    const float* gamma;
    const float* beta;
    float* out_f32;
    bf16_t* out_t;
    int M, ldx, ldr, ldo;
    int groups, units_total;
    float eps;
};

typedef __attribute__((address_space(3))) void pl_lds_void;
typedef const __attribute__((address_space(1))) void pl_gbl_void;

// barrier for an LDS exchange: the wave's LDS writes are done (lgkmcnt(0)), nothing is said about vector memory - a
// __syncthreads() would wait vmcnt(0) and drain the DMA ring once per piece
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ __launch_bounds__(256, PL_WGS) void proj_ln_kernel(ProjLnArgs a) {
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;

    const int grp = blockIdx.x;
    if (grp >= a.groups) return;
    const int u_lo = (int)((long long)a.units_total * grp / a.groups);
    const int u_hi = (int)((long long)a.units_total * (grp + 1) / a.groups);
    const int npieces = u_hi - u_lo;
    if (npieces <= 0) return;

    // ---- X and residual pieces by LDS-DMA.  X: wave w issues rows 4 w .. 4 w + 3 as two instructions of two 512-byte
    //      rows (lane l -> row + (l >> 5), physical slot l & 31 <- logical slot (l & 31) ^ (row & 15)); residual: rows
    //      4 w .. 4 w + 3 as four instructions of one 1-KB row (physical slot l <- logical slot l ^ (row & 15)).
    auto issue_piece = [&](int piece, int slot) __attribute__((always_inline)) {
        const int p = piece < npieces ? piece : npieces - 1;      // past the end: the last piece again (uniform counts)
        const int row_base = (u_lo + p) * PL_ROWS;
        unsigned char* xs = smem + slot * PL_SLOT;
        unsigned char* rs = xs + PL_XB;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = wave * 4 + q * 2 + (lane >> 5);
            int gm = row_base + r;
            gm = gm < a.M ? gm : a.M - 1;
            const int c = ((lane & 31) ^ (r & 15)) * 8;
            __builtin_amdgcn_global_load_lds((pl_gbl_void*)(a.X + (size_t)gm * a.ldx + c),
                                             (pl_lds_void*)(xs + (wave * 4 + q * 2) * 512), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave * 4 + q;
            int gm = row_base + r;
            gm = gm < a.M ? gm : a.M - 1;
            const int c = (lane ^ (r & 15)) * 4;
            __builtin_amdgcn_global_load_lds((pl_gbl_void*)(a.res + (size_t)gm * a.ldr + c),
                                             (pl_lds_void*)(rs + r * 1024), 16, 0, 0);
        }
    };

    {   // bias | gamma | beta into LDS (read back per piece: the registers hold W); first, so that the wait for these
        // three loads does not take the loads of W with it
        float* vec = (float*)(smem + PL_VEC);
        vec[tid] = a.bias ? a.bias[tid] : 0.f;
        vec[PL_K + tid] = a.gamma[tid];
        vec[2 * PL_K + tid] = a.beta[tid];
    }
    // ---- W in registers: fragment (nt, ks) = row 64 w + 16 nt + i, k = 32 ks + 8 g .. + 7; per-channel vectors of the
    //      lane's channels 64 w + 16 nt + 4 g + r
    bf16x8 wf[4][8];
    {
        const bf16_t* wp = a.W + (size_t)(wave * 64 + i) * PL_K + g * 8;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) wf[nt][ks] = *(const bf16x8*)(wp + (size_t)nt * 16 * PL_K + ks * 32);
    }
    const int cbase = wave * 64 + 4 * g;               // channel of (nt, r): cbase + 16 nt + r
    // the first pieces go out BEHIND the loads of W and the vectors: the in-order vmcnt then lets the loop start on piece 0
    // while pieces 1 .. are still on their way
#pragma unroll
    for (int k = 0; k < PL_RING - 1; ++k) issue_piece(k, k);
    // W, the vectors and piece 0 have arrived (all but the DMA of the later pieces), the vectors are in LDS
    // (as the builtin, so that hipcc's own wait insertion knows W has arrived and adds no vmcnt(0) in front of the MFMAs;
    //  encoding: vmcnt in bits 3:0 (< 16 here), expcnt 6:4 at its maximum, lgkmcnt 11:8 = 0)
    static_assert((PL_RING - 2) * PL_DMA < 16, "vmcnt field");
    __builtin_amdgcn_s_waitcnt(0x0070 | ((PL_RING - 2) * PL_DMA));
    __builtin_amdgcn_sched_barrier(0);

    float* ex = (float*)(smem + PL_EX);                // [2][4][16]
    const unsigned xlane = (unsigned)(i * 512);
    const unsigned rlane = (unsigned)(i * 1024);
    for (int p = 0; p < npieces; ++p) {
        const int slot = p % PL_RING;
        // piece p landed: all but the PL_INFLIGHT youngest vector-memory instructions of this wave are done (issued since
        // its DMA, with a ring of three: stores of p - 2, DMA of p + 1, stores of p - 1), then every wave's
        // (the first trips have fewer stores behind them: piece p < PL_RING - 1 was issued in the prologue, in front of
        // the DMA of pieces p + 1 .. PL_RING - 2, and trips 0 .. p - 1 issued PL_DMA + PL_STORES instructions each)
        if (p >= PL_RING - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PL_INFLIGHT) : "memory");
        else if (p == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PL_RING - 2) * PL_DMA) : "memory");
        else if (p == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PL_RING - 3 > 0 ? PL_RING - 3 : 0) * PL_DMA + PL_DMA + PL_STORES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PL_DMA + PL_STORES)) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* xs = smem + slot * PL_SLOT;
        const unsigned char* rs = xs + PL_XB;
        bf16x8 xf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xf[ks] = *(const bf16x8*)(xs + xlane + ((((ks << 2) | g) ^ i) << 4));
        f32x4 rv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) rv[nt] = *(const f32x4*)(rs + rlane + (((wave * 16 + nt * 4 + g) ^ i) << 4));
        // the slot of piece p - 1 is free (every wave passed this barrier after its last read of it): piece p + 2 goes there
        issue_piece(p + PL_RING - 1, (p + PL_RING - 1) % PL_RING);
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = mma16(wf[nt][ks], xf[ks], acc[nt]);
        // lane (g, i): token i of the piece, channels cbase + 16 nt + r
        f32x4 v[4];
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            v[nt] = acc[nt] + *(const f32x4*)(smem + PL_VEC + (cbase + 16 * nt) * 4) + rv[nt];
#pragma unroll
            for (int r = 0; r < 4; ++r) s += v[nt][r];
        }
        s = rows_sum(s);                                // over the wave's 64 channels (the four lanes of token i)
#ifndef PL_DBG_NOLN
        if (g == 0) ex[wave * 16 + i] = s;
        lds_barrier();
        const float mean = (ex[i] + ex[16 + i] + ex[32 + i] + ex[48 + i]) * (1.0f / PL_K);
#else
        const float mean = s * (1.0f / PL_K);
#endif
        float q = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[nt][r] -= mean;
                q += v[nt][r] * v[nt][r];
            }
        q = rows_sum(q);
#ifndef PL_DBG_NOLN
        if (g == 0) ex[64 + wave * 16 + i] = q;
        lds_barrier();
        const float var = (ex[64 + i] + ex[80 + i] + ex[96 + i] + ex[112 + i]) * (1.0f / PL_K);
#else
        const float var = q * (1.0f / PL_K);
#endif
        const float rstd = 1.0f / sqrtf(var + a.eps);
        const int m = (u_lo + p) * PL_ROWS + i;
        const bool ok = m < a.M;
        const size_t mc = (size_t)(ok ? m : a.M - 1);
        float* o32 = a.out_f32 + mc * a.ldo + cbase;
        bf16_t* ot = a.out_t + mc * a.ldo + cbase;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 y;
            const f32x4 gv = *(const f32x4*)(smem + PL_VEC + (PL_K + cbase + 16 * nt) * 4);
            const f32x4 ev = *(const f32x4*)(smem + PL_VEC + (2 * PL_K + cbase + 16 * nt) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = v[nt][r] * rstd * gv[r] + ev[r];
            v[nt] = y;
#ifndef PL_DBG_NOSTORE
            if (ok) *(f32x4*)(o32 + 16 * nt) = y;
#else
            if (y[0] == 1234.5f) *(f32x4*)(o32 + 16 * nt) = y;
#endif
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf16_t t4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) t4[r] = from_f32<bf16_t>(v[nt][r]);
#ifndef PL_DBG_NOSTORE
            if (ok) *(uint2*)(ot + 16 * nt) = *(const uint2*)t4;
#endif
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every DMA (the walk prefetches past its end) lands before the LDS is given up
}

}  // namespace fs2

// 1 where fs2_proj_add_layernorm serves the shape and is the path to take, else 0 (no launch; part of the kernel dispatch)
extern "C" int fs2_proj_add_layernorm_plan(int M, int N, int Cin, int dtype) {
    if (dtype != 1 || Cin != fs2::PL_K || N != fs2::PL_K) return 0;
    return M >= 2048 ? 1 : 0;          // below: fewer pieces than CUs, the tiled GEMM + LayerNorm pair is as fast
}

extern "C" int fs2_proj_add_layernorm(const void* X, const void* W, const float* bias, const float* res, const float* gamma,
                                      const float* beta, float* out_f32, void* out_t, int M, int N, int Cin, int ldx, int ldr,
                                      int ldo, float eps, int dtype, hipStream_t stream) {
    using namespace fs2;
    FS2_CHECK_ARG(X && W && res && gamma && beta && out_f32 && out_t, "fs2_proj_add_layernorm: null pointer");
    FS2_CHECK_ARG(dtype == 1 && Cin == PL_K && N == PL_K, "fs2_proj_add_layernorm: bf16 with N = Cin = %d only (dtype=%d N=%d Cin=%d)",
                  PL_K, dtype, N, Cin);
    FS2_CHECK_ARG(M > 0, "fs2_proj_add_layernorm: M=%d", M);
    FS2_CHECK_ARG(ldx >= Cin && ldx % 8 == 0 && ldr >= N && ldr % 4 == 0 && ldo >= N && ldo % 8 == 0,
                  "fs2_proj_add_layernorm: leading dimensions");
    FS2_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)res & 15) == 0 && ((uintptr_t)out_f32 & 15) == 0 &&
                      ((uintptr_t)out_t & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)gamma & 15) == 0 &&
                      ((uintptr_t)beta & 15) == 0,
                  "fs2_proj_add_layernorm: pointers must be 16-byte aligned");
    ProjLnArgs a;
    a.X = (const bf16_t*)X; a.W = (const bf16_t*)W; a.bias = bias; a.res = res; a.gamma = gamma; a.beta = beta;
    a.out_f32 = out_f32; a.out_t = (bf16_t*)out_t; a.M = M; a.ldx = ldx; a.ldr = ldr; a.ldo = ldo; a.eps = eps;
    a.units_total = (M + PL_ROWS - 1) / PL_ROWS;
    int groups = 256 * PL_WGS;
    if (groups > a.units_total) groups = a.units_total;
    a.groups = groups;
    static unsigned long long lds_ok = 0;
    fs2_allow_lds((const void*)proj_ln_kernel, PL_LDS, &lds_ok);
    hipLaunchKernelGGL(proj_ln_kernel, dim3(groups), dim3(256), PL_LDS, stream, a);
    FS2_CHECK_LAUNCH("fs2_proj_add_layernorm");
    return FS2_OK;
}
