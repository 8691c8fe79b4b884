// Shared device helpers for the gfx950 (CDNA4 / MI355X) FastSpeech2 kernels.
// wave = 64 lanes; MFMA tiles are 16x16 with K = 32 elements per "fragment step".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fs2 {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;

constexpr int WAVE = 64;

// ---- element <-> float ---------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// ---- MFMA fragment abstraction --------------------------------------------------------
// A fragment is 8 consecutive K-elements held by one lane: lane l = (g = l >> 4, i = l & 15)
// holds elements k = 8g .. 8g+7 of row/column i, so one fragment step covers K = 32.
//   bf16: ONE v_mfma_f32_16x16x32_bf16 (A[i][8g+j], B[8g+j][i] — cdna guide §3).
//   f32 : EIGHT v_mfma_f32_16x16x4_f32; MFMA j consumes element j of every lane, i.e. its
//         k-slot g is the real k = 8g+j.  A and B use the same slot->k map, and the sum over
//         k is order-free, so the result is the exact fp32 product (fma chain).
// Both give C[row = 4g + r][col = i] in acc[r]  (rows from the A operand, cols from B).
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x8 type; };

__device__ __forceinline__ f32x4 mma16(const bf16x8& a, const bf16x8& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(const f32x8& a, const f32x8& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
    return c;
}

// N independent accumulators against one B fragment.  fp32: the 8 k-steps of 4 are walked in the outer loop, so that
// consecutive MFMAs write different accumulators (a single mma16 is a chain of 8 dependent 16x16x4 MFMAs, each waiting
// for the one before it).
template <int N> __device__ __forceinline__ void mma16_multi(const f32x8 (&a)[N], const f32x8& b, f32x4* c) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int n = 0; n < N; ++n) c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[n][j], b[j], c[n], 0, 0, 0);
}
template <int N> __device__ __forceinline__ void mma16_multi(const bf16x8 (&a)[N], const bf16x8& b, f32x4* c) {
#pragma unroll
    for (int n = 0; n < N; ++n) c[n] = mma16(a[n], b, c[n]);
}

template <typename T> __device__ __forceinline__ typename Frag<T>::type frag_zero() {
    typename Frag<T>::type z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (T)0.0f;
    return z;
}

// store 4 consecutive elements (8-byte bf16 / 16-byte f32 access)
template <typename T> __device__ __forceinline__ void store4(T* dst, const f32x4& v) {
    T t4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) t4[r] = from_f32<T>(v[r]);
    if constexpr (sizeof(T) == 2)
        *(uint2*)dst = *(const uint2*)t4;
    else
        *(f32x4*)dst = *(const f32x4*)t4;
}

// ---- split-fp16 storage (dtype FS2_F16X3) ----------------------------------------------------------------
// A value e is kept as two fp16 numbers, hi = fp16(e) and lo = fp16(e - hi): 22 of fp32's 24 significant bits on
// operands the fp16 matrix pipe can take (csrc/conv_gemm.hip).  Layout: 4 bytes per element like fp32; every aligned
// run of 32 consecutive elements (128 bytes) holds its 32 hi halves, then its 32 lo halves.  Buffers are 128-byte
// aligned, so the run and the position inside it follow from the address of the element's 4-byte slot alone.
struct f16pair_t { _Float16 h, l; };          // element tag, sizeof == 4 (never read through these members)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <> __device__ __forceinline__ void store4<f16pair_t>(f16pair_t* dst, const f32x4& v) {
    const uintptr_t a = (uintptr_t)dst;
    unsigned char* run = (unsigned char*)(a & ~(uintptr_t)127);
    const int off = (int)(a & 127) >> 1;           // 2 bytes per hi half; `dst` is 16-byte aligned -> off in {0, 8, .., 56}
    f16x4 hi, lo;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const _Float16 h = (_Float16)v[r];
        hi[r] = h;
        lo[r] = (_Float16)(v[r] - (float)h);       // exact difference, then rounded to fp16
    }
    *(f16x4*)(run + off) = hi;
    *(f16x4*)(run + 64 + off) = lo;
}

// 16-byte vector copy helpers
struct __attribute__((aligned(16))) vec16 { uint32_t x, y, z, w; };

// ---- wave reductions -------------------------------------------------------------------
// DPP forms (VALU only): a Hillis-Steele scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the row totals
// carried across the four rows (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3): lane 63 ends with
// the reduction over all 64 lanes, read back as a wave-uniform value.  The __shfl_xor butterfly they replace is six
// ds_bpermute round trips through the CU's LDS pipe per reduction, which is what bounded the one-wave-per-row
// LayerNorm kernels (layernorm_rowdot with a 4-wide head: 21.8 -> us per 24.7 k rows, tools/bench_hbm_kernels.py).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_src(float v, float fill) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), CTRL,
                                                                  ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_src<0x111, 0xF>(v, 0.f);     // row_shr:1 (lanes without a source take the fill value)
    v += dpp_src<0x112, 0xF>(v, 0.f);     // row_shr:2
    v += dpp_src<0x114, 0xF>(v, 0.f);     // row_shr:4
    v += dpp_src<0x118, 0xF>(v, 0.f);     // row_shr:8   -> lane 15 of each row: the row's sum
    v += dpp_src<0x142, 0xA>(v, 0.f);     // row_bcast:15 into rows 1, 3
    v += dpp_src<0x143, 0xC>(v, 0.f);     // row_bcast:31 into rows 2, 3
#if defined(FS2_WAVE_SUM_VGPR)            // experiment: the total handed on in a VGPR, no SGPR operand downstream
    return __shfl(v, 63, 64);
#elif defined(FS2_WAVE_SUM_NOP)           // experiment: wait states between the v_readlane and its first reader
    const float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    asm volatile("s_nop 7" ::: "memory");
    return r;
#else
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}
// sum over the 16 lanes of a DPP row, result in all 16 (xor butterfly: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror): four VALU steps for four independent rows of lanes at once
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_src<0xB1, 0xF>(v, 0.f);
    v += dpp_src<0x4E, 0xF>(v, 0.f);
    v += dpp_src<0x141, 0xF>(v, 0.f);
    v += dpp_src<0x140, 0xF>(v, 0.f);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    const float ninf = -__builtin_inff();
    v = fmaxf(v, dpp_src<0x111, 0xF>(v, ninf));
    v = fmaxf(v, dpp_src<0x112, 0xF>(v, ninf));
    v = fmaxf(v, dpp_src<0x114, 0xF>(v, ninf));
    v = fmaxf(v, dpp_src<0x118, 0xF>(v, ninf));
    v = fmaxf(v, dpp_src<0x142, 0xA>(v, ninf));
    v = fmaxf(v, dpp_src<0x143, 0xC>(v, ninf));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// reductions over the four lanes {i, i + 16, i + 32, i + 48} (the four 16-lane rows), result in all of them: two
// v_permlane swaps (VALU) instead of two ds_bpermute round trips, whose s_waitcnt lgkmcnt would also wait for every
// LDS read the wave has in flight.  swap32(a, a) = {lo, lo} / {hi, hi}; swap16(a, a) = {r0, r0, r2, r2} / {r1, r1, r3, r3}.
// Issued from inline asm: with the builtins (__builtin_amdgcn_permlane32_swap / 16_swap) hipcc of ROCm 7.2 used the
// FIRST result register for both results whenever the pair was consumed by one add / max (it emitted
// `v_permlane32_swap v20, v16; v_add_f32 v20, v20, v20`; tools/microbench/rows_probe.hip reproduces it).  The s_nops
// are the wait states hipcc would have inserted around a VALU-written operand of these instructions.
__device__ __forceinline__ void lane_swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_swap16(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float rows_max(float v) {
    float a = v, b = v;
    lane_swap32(a, b);
    a = b = fmaxf(a, b);
    lane_swap16(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float rows_sum(float v) {
    float a = v, b = v;
    lane_swap32(a, b);
    a = b = a + b;
    lane_swap16(a, b);
    return a + b;
}

// ---- LDS-DMA pipelines that span a barrier ----------------------------------------------------
// `s_waitcnt vmcnt(N)` retires all but this wave's N youngest vector-memory instructions, then a raw s_barrier
// (a __syncthreads() would wait vmcnt(0) and drain every LDS-DMA still in flight).
template <int N> __device__ __forceinline__ void vm_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---- dropout (training only) --------------------------------------------------------------
// Counter-based: element (row, col) of dropout site `stream` is kept iff hash(seed, stream, row, col) >= thresh,
// thresh = rate * 2^32; kept values are scaled by 1 / (1 - rate) (nn.Dropout).  The same function is
// evaluated again in the backward kernels, so no mask is stored.  thresh == 0 disables the site.
struct Dropout {
    uint32_t seed, stream, thresh;
    float inv_keep;
};

__device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t stream, uint32_t row, uint32_t col) {
    uint32_t x = (row * 0x9E3779B1u) ^ ((col + 0x7F4A7C15u) * 0x85EBCA77u) ^ seed ^ (stream * 0xC2B2AE3Du);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
// multiplier of element (row, col): 0 or 1 / (1 - rate)
__device__ __forceinline__ float drop_mul(const Dropout& d, uint32_t row, uint32_t col) {
    return drop_hash(d.seed, d.stream, row, col) >= d.thresh ? d.inv_keep : 0.f;
}

}  // namespace fs2

// host side: Dropout from (seed, stream, rate)
static inline fs2::Dropout fs2_make_dropout(unsigned seed, unsigned stream, float rate) {
    fs2::Dropout d;
    d.seed = seed; d.stream = stream;
    if (!(rate > 0.f)) { d.thresh = 0u; d.inv_keep = 1.f; return d; }
    double t = (double)rate * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    d.inv_keep = 1.0f / (1.0f - rate);
    return d;
}

// ---- dynamic-LDS opt-in, once per (kernel instantiation, device) -----------------------------------
// hipFuncSetAttribute acts on the current device's copy of the kernel, so a process that drives several GPUs has
// to repeat it on each; `done` is the caller's static bit mask (one bit per device ordinal < 64).
static inline void fs2_allow_lds(const void* kernel, int bytes, unsigned long long* done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(__atomic_load_n(done, __ATOMIC_RELAXED) & bit)) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        __atomic_fetch_or(done, bit, __ATOMIC_RELAXED);
    }
}

// ---- XCD-aware order of (block, head, utterance) grids ----------------------------------------------
// Workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  All blocks of one (utterance, head) pair
// read the same K/V (or Q/dO) rows; with the plain (block, head, utterance) grid they land on as many XCDs as the pair
// has blocks and every one of those L2s fetches the rows.  In this order pair p lives on XCD p % 8: the grid is 1-D,
// id -> (xcd = id & 7, slot = id >> 3), pair = xcd + 8 * (slot / nblk), block = slot % nblk.
struct PairOrder {
    int nblk, pairs, H;      // pairs == 0: plain grid
};
static inline dim3 fs2_pair_order_grid(PairOrder& o, int nblk, int H, int B, bool on) {
    o.nblk = nblk; o.H = H; o.pairs = 0;
    if (!on || nblk < 2 || B * H < 8) return dim3(nblk, H, B);
    o.pairs = B * H;
    return dim3(8 * nblk * ((B * H + 7) / 8), 1, 1);
}
// false: this workgroup has no work (padding of the 1-D grid)
__device__ __forceinline__ bool fs2_pair_order_decode(const PairOrder& o, int& b, int& h, int& blk) {
    b = blockIdx.z; h = blockIdx.y; blk = blockIdx.x;
    if (o.pairs > 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int pair = xcd + 8 * (slot / o.nblk);
        if (pair >= o.pairs) return false;
        blk = slot % o.nblk;
        b = pair / o.H;
        h = pair % o.H;
    }
    return true;
}

// ---- experiment switches ------------------------------------------------------------------------------
// The shipped library reads NO environment variables: FS2_ENV_INT(name, default) is the default.  A build made with
// -DFS2_EXPERIMENT (tools/build_variant.sh TAG -DFS2_EXPERIMENT, used through FS2_LIB=variants/libfs2_TAG.so) reads the
// named variable once per process instead - the A/B knobs the measurements in DESIGN.md were taken with.
#ifdef FS2_EXPERIMENT
#include <stdlib.h>
#define FS2_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define FS2_ENV_INT(name, dflt) (dflt)
#endif

// ---- error codes of the C ABI (include/fs2_hip.h) -----------------------------------------
#define FS2_OK 0
#define FS2_ERR_ARG (-1)
#define FS2_ERR_UNSUPPORTED (-2)
#define FS2_ERR_HIP (-3)

extern "C" void fs2_set_error(const char* fmt, ...);

#define FS2_CHECK_ARG(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            fs2_set_error(__VA_ARGS__);       \
            return FS2_ERR_ARG;               \
        }                                     \
    } while (0)

#define FS2_CHECK_LAUNCH(name)                                              \
    do {                                                                    \
        hipError_t e_ = hipGetLastError();                                  \
        if (e_ != hipSuccess) {                                             \
            fs2_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return FS2_ERR_HIP;                                             \
        }                                                                   \
    } while (0)
