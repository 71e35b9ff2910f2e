// Micro-benchmark for DESIGN.md 11.8 (what bounds pk_gemm_bf16's 256-tile loop at ~9.6 bytes / clock / CU?): per-CU and
// whole-chip throughput of the two ways a GEMM workgroup can fill its LDS staging slots -
//   DMA   global_load_lds_dwordx4 (16 bytes per lane straight into LDS, no registers)
//   VGPR  global_load_dwordx4 into registers + ds_write_b128
// with 512 threads per workgroup and one workgroup per CU (the 256-tile kernel's shape), DEPTH k-tiles of 64 KB in
// flight, over a working set that is either L2-resident per XCD (2 MB) or streams from HBM (2 GB).
// Prints bytes / clock / CU (s_memtime runs at 100 MHz: converted with the wall time) and TB/s for 1 and 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/load_path_bw.hip -o tools/ubench/load_path_bw.bin && tools/ubench/load_path_bw.bin
// Not on the product path.  Written at the end of round 4 without GPU time left: first run is round 5's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 512, TILE_BYTES = 65536;  // one k-tile of the 256-tile kernel: A 256 x 64 + B 256 x 64 bf16

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// every lane moves TILE_BYTES / THREADS / 16 = 8 pieces of 16 bytes per k-tile; DEPTH k-tiles are kept in flight
template <bool DMA, int DEPTH>
__global__ __launch_bounds__(THREADS, 1) void k(const unsigned char* __restrict__ src, size_t span, int tiles, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // [DEPTH][TILE_BYTES]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // span == 0: the L2-resident case - a 4-tile (256 KB) window shared by the workgroups of one XCD (block b runs on XCD
    // b % 8), 2 MB in all; otherwise every workgroup streams its own window of span / gridDim.x bytes, 1 KB per wave instruction
    size_t win = span == 0 ? 4 * (size_t)TILE_BYTES : span / gridDim.x;
    const unsigned char* base = src + (span == 0 ? (size_t)(blockIdx.x & 7) * win : (size_t)blockIdx.x * win);
    const size_t wtiles = win / TILE_BYTES;
    u32x4 acc = u32x4{0u, 0u, 0u, 0u};
    auto issue = [&](int t) {
        const unsigned char* g = base + (size_t)(t % wtiles) * TILE_BYTES;
        unsigned char* slot = smem + (size_t)(t % DEPTH) * TILE_BYTES;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const size_t off = (size_t)(p * 8 + wave) * 1024;  // 64 lanes x 16 bytes per instruction
            if (DMA) {
                glds16(g + off + lane * 16, slot + off);
            } else {
                const u32x4 v = *reinterpret_cast<const u32x4*>(g + off + lane * 16);
                *reinterpret_cast<u32x4*>(slot + off + lane * 16) = v;
            }
        }
    };
    for (int t = 0; t < DEPTH - 1 && t < tiles; ++t) issue(t);
    for (int t = 0; t < tiles; ++t) {
        if (t + DEPTH - 1 < tiles) issue(t + DEPTH - 1);
        // the k-tile t must have landed: everything but the newest (DEPTH - 1) x 8 loads of this lane
        if (DMA) {
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        __syncthreads();
        // touch the tile like a fragment read would (one 16-byte LDS read per lane), so that nothing is optimised away
        const u32x4 r = *reinterpret_cast<const u32x4*>(smem + (size_t)(t % DEPTH) * TILE_BYTES + tid * 16);
        acc[0] ^= r[0]; acc[1] ^= r[1]; acc[2] ^= r[2]; acc[3] ^= r[3];
        __syncthreads();
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <bool DMA, int DEPTH>
void run(const char* what, const unsigned char* src, size_t span, int blocks, unsigned* sink) {
    const int tiles = 400;
    (void)hipFuncSetAttribute((const void*)k<DMA, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * TILE_BYTES);
    hipLaunchKernelGGL((k<DMA, DEPTH>), dim3(blocks), dim3(THREADS), DEPTH * TILE_BYTES, 0, src, span, tiles, sink);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<DMA, DEPTH>), dim3(blocks), dim3(THREADS), DEPTH * TILE_BYTES, 0, src, span, tiles, sink);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_wg = (double)tiles * TILE_BYTES, clk = 2.4e9;  // 2.4 GHz engine clock (MI355X_MICROARCH.md)
    printf("%-5s depth %d  %-10s %3d workgroups: %7.3f ms  %6.1f bytes/clock/CU  %6.2f TB/s\n", DMA ? "DMA" : "VGPR", DEPTH, what, blocks, ms,
           bytes_per_wg / (ms * 1e-3 * clk), bytes_per_wg * blocks / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t big = (size_t)2 << 30;
    unsigned char* buf;
    unsigned* sink;
    if (hipMalloc(&buf, big) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("allocation failed\n"); return 1; }
    (void)hipMemset(buf, 1, big);
    (void)hipMemset(sink, 0, 4);
    for (int blocks : {1, 256}) {
        // L2-resident: the workgroups of an XCD cycle over the same 4 tiles (256 KB) -> after the first pass they come from L2
        run<true, 1>("L2", buf, 0, blocks, sink);
        run<true, 2>("L2", buf, 0, blocks, sink);
        run<false, 2>("L2", buf, 0, blocks, sink);
        // streaming: a private 8 MB (256 workgroups) / 2 GB (1 workgroup) window each (400 tiles = 26 MB per pass: the
        // 256-workgroup case wraps three times inside its window - 2 GB in all, far beyond L2 and the 256 MB MALL)
        run<true, 2>("HBM", buf, big, blocks, sink);
        run<false, 2>("HBM", buf, big, blocks, sink);
    }
    (void)hipFree(buf);
    (void)hipFree(sink);
    return 0;
}
