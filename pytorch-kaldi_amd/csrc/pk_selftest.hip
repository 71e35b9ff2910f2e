// pk_selftest.hip - on-device check of the MFMA fragment layouts the kernels
// assume (tests/ only).  One wave computes a 16x16 / 32x32 product with
// asymmetric operands; the host compares against a scalar reference.
#include <math.h>
#include <string.h>

#include "pk_common.h"
#define PK_REC2_PRECISE 1  // (no fast-math gate functions: only dpp_row_sum is used from this header)
#include "pk_rec2_common.h"

namespace {

// out layout: [0..255] 16x16x32 bf16, [256..511] 16x16x4 f32, [512..1535] 32x32x2 f32, [1536..2559] 32x32x16 bf16
__global__ void mfma_layout_kernel(const float* __restrict__ A16, const float* __restrict__ B16,  // [16][32], [32][16]
                                   const float* __restrict__ A32, const float* __restrict__ B32,  // [32][16], [16][32]
                                   float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    {   // v_mfma_f32_16x16x32_bf16: lane supplies A[row=lane&15][k=(lane>>4)*8+e], B[k][col=lane&15]
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            const int k = (lane >> 4) * 8 + e;
            a[e] = (short)pk_f2bf(A16[(lane & 15) * 32 + k]);
            b[e] = (short)pk_f2bf(B16[k * 16 + (lane & 15)]);
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) out[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = c[r];
    }
    {   // v_mfma_f32_16x16x4_f32: A[row=lane&15][k=lane>>4], B[k=lane>>4][col=lane&15]; use k 0..3 of the 16x32 operands
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < 8; ++kk) {
            const int k = kk * 4 + (lane >> 4);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(A16[(lane & 15) * 32 + k], B16[k * 16 + (lane & 15)], c, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) out[256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = c[r];
    }
    {   // v_mfma_f32_32x32x2_f32: A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31]
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        for (int kk = 0; kk < 8; ++kk) {
            const int k = kk * 2 + (lane >> 5);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(A32[(lane & 31) * 16 + k], B32[k * 32 + (lane & 31)], c, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r)
            out[512 + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];
    }
    {   // v_mfma_f32_32x32x16_bf16: A[row=lane&31][k=(lane>>5)*8+e], B[k][col=lane&31]
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            const int k = (lane >> 5) * 8 + e;
            a[e] = (short)pk_f2bf(A32[(lane & 31) * 16 + k]);
            b[e] = (short)pk_f2bf(B32[k * 32 + (lane & 31)]);
        }
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r)
            out[1536 + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];
    }
}

float bf(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    float y;
    memcpy(&y, &u, 4);
    return y;
}

}  // namespace

extern "C" int pk_selftest_mfma(void* stream, int* h_bad_count) {
    hipStream_t st = pk_stream(stream);
    float hA16[16 * 32], hB16[32 * 16], hA32[32 * 16], hB32[16 * 32], hout[2560];
    unsigned s = 12345u;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return ((int)(s >> 20) % 17 - 8) / 8.0f;  // exactly representable in bf16
    };
    for (float& v : hA16) v = rnd();
    for (float& v : hB16) v = rnd();
    for (float& v : hA32) v = rnd();
    for (float& v : hB32) v = rnd();
    float *dA16, *dB16, *dA32, *dB32, *dout;
    PK_CHECK_HIP(hipMalloc((void**)&dA16, sizeof(hA16)));
    PK_CHECK_HIP(hipMalloc((void**)&dB16, sizeof(hB16)));
    PK_CHECK_HIP(hipMalloc((void**)&dA32, sizeof(hA32)));
    PK_CHECK_HIP(hipMalloc((void**)&dB32, sizeof(hB32)));
    PK_CHECK_HIP(hipMalloc((void**)&dout, sizeof(hout)));
    PK_CHECK_HIP(hipMemcpyAsync(dA16, hA16, sizeof(hA16), hipMemcpyHostToDevice, st));
    PK_CHECK_HIP(hipMemcpyAsync(dB16, hB16, sizeof(hB16), hipMemcpyHostToDevice, st));
    PK_CHECK_HIP(hipMemcpyAsync(dA32, hA32, sizeof(hA32), hipMemcpyHostToDevice, st));
    PK_CHECK_HIP(hipMemcpyAsync(dB32, hB32, sizeof(hB32), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, st, dA16, dB16, dA32, dB32, dout);
    PK_LAUNCH_CHECK();
    PK_CHECK_HIP(hipMemcpyAsync(hout, dout, sizeof(hout), hipMemcpyDeviceToHost, st));
    PK_CHECK_HIP(hipStreamSynchronize(st));
    (void)hipFree(dA16); (void)hipFree(dB16); (void)hipFree(dA32); (void)hipFree(dB32); (void)hipFree(dout);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float r = 0.f;
            for (int k = 0; k < 32; ++k) r += bf(hA16[i * 32 + k]) * bf(hB16[k * 16 + j]);
            if (fabsf(hout[i * 16 + j] - r) > 1e-4f) ++bad;
            if (fabsf(hout[256 + i * 16 + j] - r) > 1e-4f) ++bad;
        }
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float r = 0.f;
            for (int k = 0; k < 16; ++k) r += hA32[i * 16 + k] * hB32[k * 32 + j];
            if (fabsf(hout[512 + i * 32 + j] - r) > 1e-4f) ++bad;
            if (fabsf(hout[1536 + i * 32 + j] - r) > 1e-4f) ++bad;
        }
    *h_bad_count = bad;
    return 0;
}

// v_permlane16_swap_b32 as the third-generation recurrences use it (pk_rec_persist3.hip::pack_chunk): with both
// operands = x, the lanes of the even 16-lane rows read (x of lane l, x of lane l + 16).
namespace {
__global__ void permlane_swap_kernel(unsigned* out) {
    const unsigned x = 1000u + threadIdx.x;
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    out[threadIdx.x * 2] = r[0];
    out[threadIdx.x * 2 + 1] = r[1];
}
}  // namespace

extern "C" int pk_selftest_permlane(void* stream, int* h_bad_count) {
    hipStream_t st = pk_stream(stream);
    unsigned h[128], *d;
    PK_CHECK_HIP(hipMalloc((void**)&d, sizeof(h)));
    hipLaunchKernelGGL(permlane_swap_kernel, dim3(1), dim3(64), 0, st, d);
    PK_LAUNCH_CHECK();
    PK_CHECK_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st));
    PK_CHECK_HIP(hipStreamSynchronize(st));
    (void)hipFree(d);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (((l >> 4) & 1) != 0) continue;  // the odd rows hold the same pairs again; only the even rows publish
        if (h[l * 2] != 1000u + l || h[l * 2 + 1] != 1000u + l + 16) ++bad;
    }
    *h_bad_count = bad;
    return 0;
}

// The DPP row all-reduce of the per-step LayerNorm exchange (pk_rec2_common.h::dpp_row_sum): every lane of a 16-lane
// row ends with the exact sum of the row's 16 integers-as-floats, bit-identical on all 16 lanes.
namespace {
__global__ void dpp_row_sum_kernel(float* out) {
    const float x = (float)(1 + (threadIdx.x & 15)) + 100.f * (float)(threadIdx.x >> 4);
    out[threadIdx.x] = dpp_row_sum(x);
}
}  // namespace

extern "C" int pk_selftest_dpp_row_sum(void* stream, int* h_bad_count) {
    hipStream_t st = pk_stream(stream);
    float h[64], *d;
    PK_CHECK_HIP(hipMalloc((void**)&d, sizeof(h)));
    hipLaunchKernelGGL(dpp_row_sum_kernel, dim3(1), dim3(64), 0, st, d);
    PK_LAUNCH_CHECK();
    PK_CHECK_HIP(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st));
    PK_CHECK_HIP(hipStreamSynchronize(st));
    (void)hipFree(d);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        if (h[l] != 136.f + 1600.f * (float)(l >> 4)) ++bad;
    *h_bad_count = bad;
    return 0;
}

// A stand-in for a collective's ring kernel running next to the persistent recurrences (tests/test_gpu_dp_two_ranks.py:
// the one multi-GPU hazard a single-GPU box can reproduce - somebody else's long-running workgroups on CUs the clusters
// of a persistent launch would like to have): `blocks` workgroups, one per CU (64 KB of LDS each), copy their private
// 64 KB slice of `buf` through LDS round and round until `usec` microseconds have passed (constant 100 MHz clock).
namespace {
__global__ __launch_bounds__(256) void cu_hog_kernel(float* buf, long ticks) {
    __shared__ float lds[16384];
    float* mine = buf + (long)blockIdx.x * 16384;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned rounds = 0;
    while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {
        for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = mine[i] + 1.0f;
        __syncthreads();
        for (int i = threadIdx.x; i < 16384; i += 256) mine[i] = lds[(i + 64) & 16383];
        __syncthreads();
        ++rounds;
    }
    if (threadIdx.x == 0) mine[0] = (float)rounds;
}
}  // namespace

extern "C" int pk_selftest_cu_hog(void* stream, int blocks, int usec, float* buf) {
    PK_REQUIRE(blocks >= 1 && blocks <= 256 && usec >= 1 && usec <= 2000000 && buf != nullptr,
               "pk_selftest_cu_hog: 1..256 workgroups, up to 2 s, a buffer of blocks x 16384 floats");
    hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(256), 0, pk_stream(stream), buf, (long)usec * 100);
    PK_LAUNCH_CHECK();
    return 0;
}
