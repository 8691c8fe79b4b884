// LDS-DMA fill-rate microbenchmark (gfx950): how many bytes per clock can a CU pull from L2 into LDS with
// global_load_lds_dwordx4, as a function of (a) the shape of one instruction's 1 KiB (contiguous, or 8 row segments of
// 128 B at a row stride, or 4 segments of 256 B), (b) the number of stages kept in flight, (c) waves per workgroup,
// (d) workgroups per CU.  No compute: wait (counted vmcnt) -> barrier -> issue the next stage.
//   hipcc --offload-arch=gfx950 -O3 -o dma_fill dma_fill.hip && ./dma_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int N> __device__ __forceinline__ void wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// SHAPE 0: 1 KiB contiguous per instruction; 1: 8 rows x 128 B, row stride `stride` bytes; 2: 4 rows x 256 B
// PW: instructions per wave per stage; AHEAD: stages in flight; a stage = NWAVES * PW KiB
template <int SHAPE, int PW, int AHEAD>
__global__ __launch_bounds__(1024) void dma_fill_kernel(const unsigned char* __restrict__ src, size_t region, int stride,
                                                       int iters, int* sink, int share, int phase) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int stage_bytes = nwaves * PW * 1024;
    constexpr int NST = AHEAD + 1;
    // share == 1: every workgroup streams its own region.  share > 1: `share` workgroups of the SAME XCD (ids are dealt
    // round-robin to the 8 XCDs) walk one region together; phase != 0 starts each of them `phase` iterations apart.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const unsigned char* base = src + (size_t)((slot / share) * 8 + xcd) * region;
    const int it0 = (slot % share) * phase;
    auto issue = [&](int it, int slot) __attribute__((always_inline)) {
        // the tile of iteration `it`: rows of `stride` bytes; piece p covers 1 KiB of it
        const size_t tile_off = ((size_t)(it + it0) * stage_bytes) % region;
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int p = wave * PW + q;
            size_t off;
            if (SHAPE == 0) off = tile_off + (size_t)p * 1024 + lane * 16;
            else if (SHAPE == 1) off = (tile_off + ((size_t)(8 * p + (lane >> 3)) * stride + (lane & 7) * 16)) % region;
            else off = (tile_off + ((size_t)(4 * p + (lane >> 4)) * stride + (lane & 15) * 16)) % region;
            __builtin_amdgcn_global_load_lds((gbl_void*)(base + off), (lds_void*)(smem + slot * stage_bytes + p * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) issue(s, s);
    int cur = 0, fill = AHEAD;
    for (int it = 0; it < iters; ++it) {
        wait_barrier<(AHEAD - 1) * PW>();
        issue(it + AHEAD, fill);
        cur = cur + 1 == NST ? 0 : cur + 1;
        fill = fill + 1 == NST ? 0 : fill + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && smem[blockIdx.x & 1023] == 255) sink[0] = 1;
}

template <int SHAPE, int PW, int AHEAD>
static void run(const char* name, const unsigned char* src, size_t total, int wgs, int threads, int stride, int* sink,
                double clock_ghz, int share = 1, int phase = 0) {
    const int nwaves = threads / 64;
    const int stage = nwaves * PW * 1024;
    const int lds = (AHEAD + 1) * stage;
    if (lds > 160 * 1024) return;
    const size_t region = (total / (wgs / share)) & ~(size_t)4095;
    const int iters = 400;
    hipFuncSetAttribute((const void*)dma_fill_kernel<SHAPE, PW, AHEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((dma_fill_kernel<SHAPE, PW, AHEAD>), dim3(wgs), dim3(threads), lds, 0, src, region, stride, iters, sink, share, phase);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)wgs * iters * stage;
    const double tbs = bytes / (best * 1e-3) / 1e12;
    const int wg_per_cu = (wgs + 255) / 256;
    if (share > 1) printf("[share %2d phase %d] ", share, phase);
    printf("%-34s wgs %4d x %4d thr, stage %3d KB, in flight/WG %3d KB, LDS %3d KB: %6.2f TB/s = %5.1f B/clk/CU (@%.1f GHz)\n", name,
           wgs, threads, stage / 1024, AHEAD * stage / 1024, lds / 1024, tbs, tbs * 1e12 / 256 / (clock_ghz * 1e9), clock_ghz);
    (void)wg_per_cu;
}

int main(int argc, char** argv) {
    const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 24) << 20;      // MiB streamed over by all workgroups together
    unsigned char* src; int* sink;
    hipMalloc(&src, total + (1 << 20)); hipMemset(src, 1, total + (1 << 20)); hipMalloc(&sink, 4);
    const double ghz = 2.4;
    printf("working set %zu MiB (fits L2+MALL when small): every workgroup loops over its own slice\n", total >> 20);
    for (int wgs : {256, 512, 768}) {
        run<0, 4, 1>("contiguous 1KiB, 1 stage ahead", src, total, wgs, 256, 0, sink, ghz);
        run<0, 4, 2>("contiguous 1KiB, 2 stages ahead", src, total, wgs, 256, 0, sink, ghz);
        run<1, 4, 1>("8x128B rows @512B, 1 ahead", src, total, wgs, 256, 512, sink, ghz);
        run<1, 4, 2>("8x128B rows @512B, 2 ahead", src, total, wgs, 256, 512, sink, ghz);
        run<2, 4, 2>("4x256B rows @2048B, 2 ahead", src, total, wgs, 256, 2048, sink, ghz);
    }
    run<0, 3, 2>("contiguous, 12 waves, 2 ahead", src, total, 256, 768, 0, sink, ghz);
    run<2, 3, 2>("4x256B @2048B, 12 waves, 2 ahead", src, total, 256, 768, 2048, sink, ghz);
    run<2, 3, 3>("4x256B @2048B, 12 waves, 3 ahead", src, total, 256, 768, 2048, sink, ghz);
    // the same bytes requested by several workgroups of one XCD at the same time (what the W tiles of a GEMM see)
    for (int share : {1, 2, 4, 8, 32}) {
        run<0, 4, 1>("contiguous, 3 WG/CU, 1 ahead", src, total, 768, 256, 0, sink, ghz, share, 0);
        if (share > 1) run<0, 4, 1>("contiguous, 3 WG/CU, 1 ahead", src, total, 768, 256, 0, sink, ghz, share, 1);
        if (share > 1) run<0, 4, 1>("contiguous, 3 WG/CU, 1 ahead", src, total, 768, 256, 0, sink, ghz, share, 3);
        run<0, 3, 2>("contiguous, 12 waves, 2 ahead", src, total, 256, 768, 0, sink, ghz, share, 0);
    }
    run<0, 8, 2>("contiguous, 4 waves x 8, 2 ahead", src, total, 256, 256, 0, sink, ghz);
    run<0, 8, 3>("contiguous, 4 waves x 8, 3 ahead", src, total, 256, 256, 0, sink, ghz);
    hipFree(src); hipFree(sink);
    return 0;
}
