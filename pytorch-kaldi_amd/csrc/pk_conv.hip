// pk_conv.hip - valid 1-D convolution (stride 1) fused with max_pool1d
// (kernel = stride = pool, floor), replacing F.conv1d + F.max_pool1d of the
// reference CNN / SincNet stacks (neural_networks.py:1546-1552, :1655-1661,
// :1805-1813).  The un-pooled conv output (201 MB for SincNet layer 1 at
// batch 128) is never written: each block computes a tile of it from an
// LDS-staged input window, reduces the pooling windows in registers and
// stores the pooled value plus the arg-max position for backward.
//
//   x [B,Cin,L]  w [Cout,Cin,K]  y [B,Cout,Lp]  Lp = (L-K+1)/pool
//
// Forward tile: one block = (b, 16 output channels, 256 pooled positions...
// capped so that the staged x window fits LDS).  Each thread owns one pooled
// position and loops over the 16 channels of the tile, so every x value read
// from LDS is reused 16 x and every weight is a broadcast read.
#include "pk_common.h"

namespace {

constexpr int CO_TILE = 16;
constexpr int LP_TILE = 128;   // pooled outputs per block (= threads)
constexpr int CI_CHUNK = 8;    // input channels staged per pass

__global__ __launch_bounds__(LP_TILE) void conv_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, int B, int Cin, int L,
                                                                int Cout, int K, int pool, int Lp,
                                                                float* __restrict__ y, int* __restrict__ argmax) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int span = LP_TILE * pool + K - 1;           // conv-input window of this tile
    float* xs = smem;                                  // [CI_CHUNK][span]
    float* ws = smem + CI_CHUNK * span;                // [CO_TILE][CI_CHUNK][K]
    const int b = blockIdx.z, co0 = blockIdx.y * CO_TILE, lp0 = blockIdx.x * LP_TILE;
    const int tid = threadIdx.x;
    const int lp = lp0 + tid;
    const int l0 = lp0 * pool;                         // first conv-input sample of the tile
    // running conv sums for the `pool` positions of my window, for CO_TILE channels
    // (pool <= 4 in every shipped recipe; larger pools loop in chunks of 4)
    for (int q0 = 0; q0 < pool; q0 += 4) {
        float acc[CO_TILE][4];
#pragma unroll
        for (int c = 0; c < CO_TILE; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[c][q] = 0.f;
        for (int ci0 = 0; ci0 < Cin; ci0 += CI_CHUNK) {
            const int nci = min(CI_CHUNK, Cin - ci0);
            __syncthreads();
            for (int i = tid; i < nci * span; i += LP_TILE) {
                const int ci = i / span, s = i - ci * span;
                const int l = l0 + s;
                xs[ci * span + s] = (l < L) ? x[((long)b * Cin + ci0 + ci) * L + l] : 0.f;
            }
            for (int i = tid; i < CO_TILE * nci * K; i += LP_TILE) {
                const int c = i / (nci * K), r = i - c * (nci * K);
                const int ci = r / K, k = r - ci * K;
                ws[(c * CI_CHUNK + ci) * K + k] = (co0 + c < Cout) ? w[((long)(co0 + c) * Cin + ci0 + ci) * K + k] : 0.f;
            }
            __syncthreads();
            for (int ci = 0; ci < nci; ++ci) {
                const float* xr = xs + ci * span + tid * pool + q0;
                for (int k = 0; k < K; ++k) {
                    float xv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) xv[q] = (q0 + q < pool) ? xr[k + q] : 0.f;
#pragma unroll
                    for (int c = 0; c < CO_TILE; ++c) {
                        const float wv = ws[(c * CI_CHUNK + ci) * K + k];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[c][q] = fmaf(wv, xv[q], acc[c][q]);
                    }
                }
            }
        }
        if (lp < Lp) {
#pragma unroll
            for (int c = 0; c < CO_TILE; ++c) {
                if (co0 + c >= Cout) continue;
                const long o = ((long)b * Cout + co0 + c) * Lp + lp;
                float best = (q0 == 0) ? -INFINITY : y[o];
                int bi = (q0 == 0) ? 0 : argmax[o];
                const float bv = bias ? bias[co0 + c] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q0 + q < pool) {
                        const float v = acc[c][q] + bv;
                        if (v > best) {  // first maximum wins, as torch's max_pool1d
                            best = v;
                            bi = lp * pool + q0 + q;
                        }
                    }
                }
                y[o] = best;
                argmax[o] = bi;
            }
        }
    }
}

// dz[b,co,lc] = dy routed to the arg-max position (dense, zero elsewhere)
__global__ void unpool_kernel(const float* __restrict__ dy, const int* __restrict__ argmax, long n, int Lp, int Lc,
                              float* __restrict__ dz) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long bc = i / Lp;
        dz[bc * Lc + argmax[i]] = dy[i];
    }
}

// dx[b,ci,l] = sum_co sum_k w[co,ci,k] * dz[b,co,l-k]
__global__ __launch_bounds__(256) void conv_bwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                             int B, int Cin, int L, int Cout, int K, int Lc,
                                                             float* __restrict__ dx) {
    extern __shared__ float wsm[];  // [Cout][K] for this ci
    const int ci = blockIdx.y, b = blockIdx.z;
    for (int i = threadIdx.x; i < Cout * K; i += blockDim.x) {
        const int co = i / K, k = i - co * K;
        wsm[i] = w[((long)co * Cin + ci) * K + k];
    }
    __syncthreads();
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    float s = 0.f;
    for (int co = 0; co < Cout; ++co) {
        const float* dzr = dz + ((long)b * Cout + co) * Lc;
        for (int k = 0; k < K; ++k) {
            const int lc = l - k;
            if (lc >= 0 && lc < Lc) s = fmaf(wsm[co * K + k], dzr[lc], s);
        }
    }
    dx[((long)b * Cin + ci) * L + l] = s;
}

// dw[co,ci,k] = sum_b sum_lp dy[b,co,lp] * x[b,ci,argmax+k]; one thread per (ci,k) of a chunk, one block row per co
__global__ __launch_bounds__(256) void conv_bwd_filter_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const int* __restrict__ argmax, int B, int Cin, int L,
                                                               int Cout, int K, int Lp, float* __restrict__ dw,
                                                               float* __restrict__ dbias) {
    const int co = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (ci,k) pair
    const bool active = idx < Cin * K;
    const int ci = active ? idx / K : 0, k = active ? idx - ci * K : 0;
    __shared__ float sdy[256];
    __shared__ int spos[256];
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* xr = x + ((long)b * Cin + ci) * L + k;
        const long o = ((long)b * Cout + co) * Lp;
        for (int lp0 = 0; lp0 < Lp; lp0 += 256) {
            __syncthreads();
            const int n = min(256, Lp - lp0);
            if ((int)threadIdx.x < n) {
                sdy[threadIdx.x] = dy[o + lp0 + threadIdx.x];
                spos[threadIdx.x] = argmax[o + lp0 + threadIdx.x];
            }
            __syncthreads();
            if (active)
                for (int j = 0; j < n; ++j) s = fmaf(sdy[j], xr[spos[j]], s);
            if (dbias && blockIdx.x == 0 && threadIdx.x == 0)
                for (int j = 0; j < n; ++j) sb += sdy[j];
        }
    }
    if (active) dw[((long)co * Cin + ci) * K + k] = s;
    if (dbias && blockIdx.x == 0 && threadIdx.x == 0) dbias[co] = sb;
}

}  // namespace

extern "C" int pk_conv1d_pool_fwd(void* stream, const float* x, const float* w, const float* bias, int B, int Cin, int L,
                                  int Cout, int K, int pool, float* y, int32_t* argmax) {
    PK_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && K > 0 && pool > 0 && L >= K, "pk_conv1d_pool_fwd: bad geometry");
    const int Lp = (L - K + 1) / pool;
    PK_REQUIRE(Lp > 0, "pk_conv1d_pool_fwd: empty output");
    const int span = LP_TILE * pool + K - 1;
    const size_t lds = sizeof(float) * ((size_t)CI_CHUNK * span + (size_t)CO_TILE * CI_CHUNK * K);
    PK_REQUIRE(lds <= 160 * 1024, "pk_conv1d_pool_fwd: filter length %d / pool %d exceed the LDS tile", K, pool);
    static bool attr_done = false;
    if (!attr_done) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)conv_pool_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
        attr_done = true;
    }
    dim3 grid((Lp + LP_TILE - 1) / LP_TILE, (Cout + CO_TILE - 1) / CO_TILE, B);
    hipLaunchKernelGGL(conv_pool_fwd_kernel, grid, dim3(LP_TILE), lds, pk_stream(stream), x, w, bias, B, Cin, L, Cout, K,
                       pool, Lp, y, argmax);
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t pk_conv_partial_floats(int B, int Cin, int L, int Cout, int K, int pool) {
    (void)Cin; (void)pool;
    return (int64_t)B * Cout * (L - K + 1);
}

extern "C" int pk_conv1d_pool_bwd(void* stream, const float* x, const float* w, const float* dy, const int32_t* argmax,
                                  int B, int Cin, int L, int Cout, int K, int pool, float* dw, float* dbias, float* dx,
                                  float* partial) {
    hipStream_t st = pk_stream(stream);
    const int Lc = L - K + 1, Lp = Lc / pool;
    {
        dim3 grid((Cin * K + 255) / 256, Cout);
        hipLaunchKernelGGL(conv_bwd_filter_kernel, grid, dim3(256), 0, st, x, dy, argmax, B, Cin, L, Cout, K, Lp, dw,
                           dbias);
        PK_LAUNCH_CHECK();
    }
    if (dx != nullptr) {
        PK_REQUIRE(partial != nullptr, "pk_conv1d_pool_bwd: dx needs the scratch buffer");
        const long n = (long)B * Cout * Lp;
        PK_CHECK_HIP(hipMemsetAsync(partial, 0, sizeof(float) * (size_t)B * Cout * Lc, st));
        long blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(unpool_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, argmax, n, Lp, Lc, partial);
        PK_LAUNCH_CHECK();
        const size_t lds = sizeof(float) * (size_t)Cout * K;
        PK_REQUIRE(lds <= 160 * 1024, "pk_conv1d_pool_bwd: Cout*K too large for the weight tile");
        static bool attr_bwd = false;
        if (!attr_bwd) {
            PK_CHECK_HIP(hipFuncSetAttribute((const void*)conv_bwd_data_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             160 * 1024));
            attr_bwd = true;
        }
        dim3 grid((L + 255) / 256, Cin, B);
        hipLaunchKernelGGL(conv_bwd_data_kernel, grid, dim3(256), lds, st, partial, w, B, Cin, L, Cout, K, Lc, dx);
        PK_LAUNCH_CHECK();
    }
    return 0;
}
