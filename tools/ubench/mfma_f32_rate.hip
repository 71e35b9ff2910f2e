// Micro-benchmark: what the fp32 matrix pipe sustains.  Every wave issues v_mfma_f32_16x16x4_f32 (the recurrences' form) or
// v_mfma_f32_32x32x2_f32 (pk_gemm's) on CHAINS independent accumulators, operands in registers, nothing else; grids of 1, 144
// (what a fourth-generation fp32 recurrence launch occupies) and 256 workgroups of 256 or 1024 threads (1 or 4 waves per SIMD).
// Prints TFLOP/s from the wall time (HIP events) and the nominal-clock cycles per MFMA that corresponds to (2.4 GHz): the
// data sheet's 157.3 TFLOP/s is 32 / 64 clocks per instruction at 2.4 GHz on 256 CUs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_rate.hip -o /tmp/mfma_f32_rate && /tmp/mfma_f32_rate
// Not on the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS, bool BIG>
__global__ void k(float* out, int iters) {
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-3f;
    f32x4 acc4[CHAINS];
    f32x16 acc16[BIG ? CHAINS : 1];
    for (int c = 0; c < CHAINS; ++c) acc4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < (BIG ? CHAINS : 1); ++c)
        for (int e = 0; e < 16; ++e) acc16[c][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (BIG) acc16[i % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16[i % CHAINS], 0, 0, 0);
            else acc4[i % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i % CHAINS], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc4[c][0];
    for (int c = 0; c < (BIG ? CHAINS : 1); ++c) s += acc16[c][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, bool BIG>
void run(int blocks, int threads) {
    float* out;
    (void)hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = BIG ? 4000 : 8000;
    hipLaunchKernelGGL((k<CHAINS, BIG>), dim3(blocks), dim3(threads), 0, 0, out, 200);  // warm-up
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<CHAINS, BIG>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64, n = (double)iters * 32;          // MFMAs per wave
    const double flop = waves * n * 4096.0 / (BIG ? 1.0 : 2.0);                          // 32x32x2: 4096, 16x16x4: 2048
    const double per_simd = (double)threads / 256;                                        // waves per SIMD
    const double clk = ms * 1e-3 * 2.4e9 / (n * per_simd);                                // nominal clocks per MFMA of a SIMD
    printf("%-14s chains %d  %4d WGs x %4d threads: %7.3f ms  %7.1f TFLOP/s  %5.1f nominal clocks per MFMA and SIMD\n",
           BIG ? "32x32x2_f32" : "16x16x4_f32", CHAINS, blocks, threads, ms, flop / (ms * 1e-3) / 1e12, clk);
    (void)hipFree(out);
}

int main() {
    for (int blocks : {1, 144, 256}) {
        run<8, false>(blocks, 256);
        run<8, false>(blocks, 1024);
        run<4, true>(blocks, 256);
        run<4, true>(blocks, 1024);
    }
    return 0;
}
