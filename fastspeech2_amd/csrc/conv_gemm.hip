// Varlen conv1d / linear as an MFMA implicit GEMM over channel-last packed activations.
//
//   out[m, n] = epi( sum_{tap, c} X[m + tap - pad, c] * W[tap][n][c] + bias[n] ) (+ res[m, n])
//
// X is the packed (ragged, no padding) token matrix [M, Cin]; every row m belongs to one
// utterance [seg.x, seg.y) and taps that would read across an utterance edge contribute zero —
// the batch-1 "same"-padding semantics of the reference (network/encoder.py:16-30 Conv1d
// padding=(k-1)//2; SURVEY.md §7 "varlen halos").  taps == 1 is a plain Linear
// (network/encoder.py:39-44, :57-60; network/fastspeech2.py:158).
//
// Tiling (gfx950): 256 threads = 4 waves as 2(M) x 2(N); block tile 128(M) x 128(N); each wave
// 64 x 64 = 4 x 4 MFMA 16x16 tiles.  Per channel-chunk the X tile plus its (taps-1)-row halo is
// staged ONCE in LDS and re-used by all taps as shifted row views (no im2col); the W tile of
// (tap, chunk) is streamed.  Both are double-buffered with register prefetch: global loads for
// step s+1 are issued before the MFMAs of step s and written to the other LDS buffer after them
// (one barrier per step).  MFMA operand roles are swapped (A = W rows, B = X rows) so a lane
// ends up with 4 consecutive output channels of one token -> 16-byte epilogue accesses.
#include "common.h"

namespace fs2 {

constexpr int CG_BM = 128;
constexpr int CG_BN = 128;
constexpr int CG_THREADS = 256;
constexpr int CG_MAX_TAPS = 9;
constexpr int CG_ROW_BYTES = 128;              // data bytes per LDS row (one channel chunk)
constexpr int CG_ROW_STRIDE = CG_ROW_BYTES + 16;  // +16 B pad: conflict-free ds_read_b128
constexpr int CG_XROWS = CG_BM + CG_MAX_TAPS - 1;
constexpr int CG_X_BYTES = CG_XROWS * CG_ROW_STRIDE;
constexpr int CG_W_BYTES = CG_BN * CG_ROW_STRIDE;
constexpr int CG_LDS_BYTES = 2 * CG_X_BYTES + 2 * CG_W_BYTES;

struct ConvGemmArgs {
    const void* X;       // [M, ldx] T
    const void* W;       // [taps][N][Cin] T
    const float* bias;   // [N] or null
    const float* res;    // [M, ldr] f32 or null
    void* out;           // [M, ldo]  f32 (out_f32 != 0) or T
    const int2* seg;     // [M] (start,end) row range of the row's utterance; null when taps == 1
    int M, N, Cin, taps;
    int ldx, ldr, ldo;
    int relu, out_f32;
    int n_tiles;
};

template <typename T>
__global__ __launch_bounds__(CG_THREADS, 2) void conv_gemm_kernel(ConvGemmArgs a) {
    constexpr int EPV = 16 / sizeof(T);           // elements per 16-byte vector
    constexpr int BKE = CG_ROW_BYTES / sizeof(T);  // channels per chunk (64 bf16 / 32 f32)
    constexpr int KSTEPS = BKE / 32;               // fragment steps per chunk
    constexpr int VPR = CG_ROW_BYTES / 16;         // 16-byte vectors per LDS row (8)
    constexpr int FRAG_BYTES = 8 * sizeof(T);
    typedef typename Frag<T>::type frag_t;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto xs_buf = [&](int b) -> unsigned char* { return smem + b * CG_X_BYTES; };
    auto ws_buf = [&](int b) -> unsigned char* { return smem + 2 * CG_X_BYTES + b * CG_W_BYTES; };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, i = lane & 15;

    // blocks that share W columns share `blockIdx % n_tiles` -> same XCD L2 under the round-robin
    // dispatch when n_tiles divides 8 or is a multiple of it (speed only)
    const int n_tile = blockIdx.x % a.n_tiles;
    const int m_tile = blockIdx.x / a.n_tiles;
    const int m0 = m_tile * CG_BM;
    const int n0 = n_tile * CG_BN;
    const int pad = (a.taps - 1) >> 1;
    const int xrows = CG_BM + a.taps - 1;
    const int nchunks = (a.Cin + BKE - 1) / BKE;
    const int nsteps = nchunks * a.taps;

    const T* __restrict__ X = (const T*)a.X;
    const T* __restrict__ W = (const T*)a.W;

    // per-lane validity window of tap shifts for each of its 4 token rows
    int lo[4], hi[4];
    bool edge = false;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + i;
        if (m < a.M) {
            if (a.seg) {
                const int2 s = a.seg[m];
                lo[mt] = s.x - m;
                hi[mt] = s.y - 1 - m;
            } else {
                lo[mt] = 0;
                hi[mt] = 0;
            }
        } else {
            lo[mt] = 1;
            hi[mt] = 0;
        }
        edge |= (lo[mt] > -pad) | (hi[mt] < pad);
    }
    const bool need_mask = __syncthreads_or(edge ? 1 : 0) != 0;

    // ---- register-staged prefetch state ---------------------------------------------------
    constexpr int XV = (CG_XROWS * VPR + CG_THREADS - 1) / CG_THREADS;  // 5
    constexpr int WV = (CG_BN * VPR) / CG_THREADS;                      // 4
    vec16 xreg[XV], wreg[WV];
    const vec16 zero16 = {0u, 0u, 0u, 0u};

    auto load_x = [&](int chunk) {
        const int c0 = chunk * BKE;
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            const int idx = tid + v * CG_THREADS;
            const int r = idx / VPR, cv = idx % VPR;
            int gm = m0 - pad + r;
            const int c = c0 + cv * EPV;
            // rows outside [0, M) are never a valid tap source (masked), clamp the address only
            gm = gm < 0 ? 0 : (gm >= a.M ? a.M - 1 : gm);
            if (r < xrows && c < a.Cin)
                xreg[v] = *(const vec16*)(X + (size_t)gm * a.ldx + c);
            else
                xreg[v] = zero16;
        }
    };
    auto store_x = [&](unsigned char* dst) {
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            const int idx = tid + v * CG_THREADS;
            const int r = idx / VPR, cv = idx % VPR;
            if (r < xrows) *(vec16*)(dst + r * CG_ROW_STRIDE + cv * 16) = xreg[v];
        }
    };
    auto load_w = [&](int tap, int chunk) {
        const int c0 = chunk * BKE;
        const T* wt = W + (size_t)tap * a.N * a.Cin;
#pragma unroll
        for (int v = 0; v < WV; ++v) {
            const int idx = tid + v * CG_THREADS;
            const int r = idx / VPR, cv = idx % VPR;
            const int n = n0 + r;
            const int c = c0 + cv * EPV;
            if (n < a.N && c < a.Cin)
                wreg[v] = *(const vec16*)(wt + (size_t)n * a.Cin + c);
            else
                wreg[v] = zero16;
        }
    };
    auto store_w = [&](unsigned char* dst) {
#pragma unroll
        for (int v = 0; v < WV; ++v) {
            const int idx = tid + v * CG_THREADS;
            const int r = idx / VPR, cv = idx % VPR;
            *(vec16*)(dst + r * CG_ROW_STRIDE + cv * 16) = wreg[v];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue
    load_x(0);
    load_w(0, 0);
    store_x(xs_buf(0));
    store_w(ws_buf(0));
    __syncthreads();

    int xcur = 0, wcur = 0;
    int tap = 0, chunk = 0;
    for (int step = 0; step < nsteps; ++step) {
        int ntap = tap + 1, nchunk = chunk;
        if (ntap == a.taps) { ntap = 0; nchunk = chunk + 1; }
        const bool has_next = (step + 1) < nsteps;
        const bool next_x = has_next && (ntap == 0);
        if (has_next) load_w(ntap, nchunk);
        if (next_x) load_x(nchunk);

        const unsigned char* xs = xs_buf(xcur);
        const unsigned char* ws = ws_buf(wcur);
        const int shift = tap - pad;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            frag_t wf[4], xf[4];
            const int koff = ks * 32 * (int)sizeof(T) + g * FRAG_BYTES;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                wf[nt] = *(const frag_t*)(ws + (wn * 64 + nt * 16 + i) * CG_ROW_STRIDE + koff);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                xf[mt] = *(const frag_t*)(xs + (wm * 64 + mt * 16 + i + tap) * CG_ROW_STRIDE + koff);
            if (need_mask) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    if (shift < lo[mt] || shift > hi[mt]) xf[mt] = frag_zero<T>();
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mma16(wf[nt], xf[mt], acc[mt][nt]);
        }

        if (has_next) store_w(ws_buf(wcur ^ 1));
        if (next_x) store_x(xs_buf(xcur ^ 1));
        __syncthreads();
        wcur ^= 1;
        if (next_x) xcur ^= 1;
        tap = ntap;
        chunk = nchunk;
    }

    // ---- epilogue: lane (g, i) holds out[m = .. + i][n = .. + 4g + r], r = 0..3 ----------------
    const bool vec_ok = ((a.N & 3) == 0) && ((a.ldo & 3) == 0) && (!a.res || (a.ldr & 3) == 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + i;
        if (m >= a.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + 4 * g;
            if (n >= a.N) continue;
            f32x4 v = acc[mt][nt];
            if (vec_ok) {
                if (a.bias) {
                    const f32x4 b = *(const f32x4*)(a.bias + n);
                    v += b;
                }
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                if (a.res) {
                    const f32x4 rr = *(const f32x4*)(a.res + (size_t)m * a.ldr + n);
                    v += rr;
                }
                if (a.out_f32) {
                    *(f32x4*)((float*)a.out + (size_t)m * a.ldo + n) = v;
                } else {
                    T* o = (T*)a.out + (size_t)m * a.ldo + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(v[r]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r >= a.N) continue;
                    float s = v[r];
                    if (a.bias) s += a.bias[n + r];
                    if (a.relu) s = fmaxf(s, 0.f);
                    if (a.res) s += a.res[(size_t)m * a.ldr + n + r];
                    if (a.out_f32)
                        ((float*)a.out)[(size_t)m * a.ldo + n + r] = s;
                    else
                        ((T*)a.out)[(size_t)m * a.ldo + n + r] = from_f32<T>(s);
                }
            }
        }
    }
}

}  // namespace fs2

// dtype: 0 = f32, 1 = bf16 (include/fs2_hip.h)
extern "C" int fs2_conv_gemm(const void* X, const void* W, const float* bias, const float* res, void* out,
                             const int* seg, int M, int N, int Cin, int taps, int ldx, int ldr, int ldo,
                             int relu, int out_f32, int dtype, hipStream_t stream) {
    using namespace fs2;
    FS2_CHECK_ARG(X && W && out, "fs2_conv_gemm: null X/W/out");
    FS2_CHECK_ARG(M >= 0 && N > 0 && Cin > 0, "fs2_conv_gemm: bad dims M=%d N=%d Cin=%d", M, N, Cin);
    FS2_CHECK_ARG(taps >= 1 && taps <= CG_MAX_TAPS && (taps & 1), "fs2_conv_gemm: taps=%d (odd, 1..%d)", taps,
                  CG_MAX_TAPS);
    FS2_CHECK_ARG(taps == 1 || seg, "fs2_conv_gemm: taps>1 needs the per-row segment table");
    FS2_CHECK_ARG(dtype == 0 || dtype == 1, "fs2_conv_gemm: dtype=%d", dtype);
    const int epv = dtype == 0 ? 4 : 8;
    FS2_CHECK_ARG(Cin % epv == 0 && ldx % epv == 0 && ldx >= Cin,
                  "fs2_conv_gemm: Cin=%d / ldx=%d must be multiples of %d (16-byte rows)", Cin, ldx, epv);
    FS2_CHECK_ARG(ldo >= N && (!res || ldr >= N), "fs2_conv_gemm: ldo/ldr smaller than N");
    FS2_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                      ((uintptr_t)bias & 15) == 0 && ((uintptr_t)res & 15) == 0,
                  "fs2_conv_gemm: pointers must be 16-byte aligned");
    if (M == 0) return FS2_OK;
    ConvGemmArgs a;
    a.X = X; a.W = W; a.bias = bias; a.res = res; a.out = out; a.seg = (const int2*)seg;
    a.M = M; a.N = N; a.Cin = Cin; a.taps = taps; a.ldx = ldx; a.ldr = ldr; a.ldo = ldo;
    a.relu = relu; a.out_f32 = out_f32;
    a.n_tiles = (N + CG_BN - 1) / CG_BN;
    const int m_tiles = (M + CG_BM - 1) / CG_BM;
    dim3 grid(a.n_tiles * m_tiles), block(CG_THREADS);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_gemm_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, CG_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)conv_gemm_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, CG_LDS_BYTES);
        attr_set = true;
    }
    if (dtype == 0)
        hipLaunchKernelGGL(conv_gemm_kernel<float>, grid, block, CG_LDS_BYTES, stream, a);
    else
        hipLaunchKernelGGL(conv_gemm_kernel<bf16_t>, grid, block, CG_LDS_BYTES, stream, a);
    FS2_CHECK_LAUNCH("fs2_conv_gemm");
    return FS2_OK;
}
