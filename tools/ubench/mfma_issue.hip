// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD (the persistent recurrences'
// situation) with 1, 2, 4 accumulation chains and with B operands in VGPRs; s_memtime ticks per MFMA.
// THREADS = 512 puts TWO waves on every SIMD (the eight-wave LSTM kernels; a K split over the waves of a SIMD): the
// printed figure is then per MFMA of ONE wave - half of it is the SIMD's aggregate issue interval.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int NB, int THREADS = 256>
__global__ __launch_bounds__(THREADS, 1) void k(float* out, unsigned long long* ticks, int iters) {
    bf16x8 a, b[NB];
    for (int e = 0; e < 8; ++e) a[e] = (short)(0x3f80 + threadIdx.x + e);
    for (int i = 0; i < NB; ++i)
        for (int e = 0; e < 8; ++e) b[i][e] = (short)(0x3f80 + i + e);
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 36; ++i) acc[i % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[i % NB], acc[i % CHAINS], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int CHAINS, int NB, int THREADS = 256>
void run(const char* name, int blocks) {
    float* out;
    unsigned long long* ticks;
    (void)hipMalloc(&out, sizeof(float) * THREADS * blocks);
    (void)hipMalloc(&ticks, 8);
    const int iters = 200;
    hipLaunchKernelGGL((k<CHAINS, NB, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, ticks, iters);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, NB, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, ticks, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-28s blocks=%3d  %.1f s_memtime ticks / MFMA   (%.3f ms wall -> %.1f ns / MFMA)\n", name, blocks,
           (double)t / (36.0 * iters), ms, ms * 1e6 / (36.0 * iters));
    (void)hipFree(out);
    (void)hipFree(ticks);
}

int main() {
    for (int blocks : {1, 216, 256}) {
        run<1, 36>("1 chain", blocks);
        run<2, 36>("2 chains", blocks);
        run<4, 36>("4 chains", blocks);
        run<4, 1>("4 chains, one B operand", blocks);
        run<2, 18, 512>("2 chains, 2 waves / SIMD", blocks);
        run<4, 18, 512>("4 chains, 2 waves / SIMD", blocks);
    }
    return 0;
}
