// pk_sinc.hip - the sinc band-pass bank of SincNet's first layer (neural_networks.py:1789-1800: SincConv.forward up to
// `self.filters`) as ONE launch each way.
//
// The bank is tiny (2 x 128 parameters -> 128 x 129 taps) and SURVEY.md K9 leaves its synthesis to torch autograd - but
// in a 3 ms launch-bound step its ~15 forward and ~25 backward element-wise launches (abs, outer product, sin, div,
// cat, flip, max, div, mul and their gradients) were 6 % of the step and a third of its stock torch launches
// (profiles/r05_timit_sincnet_kernel_stats_mid.csv).  Forward follows the reference operation by operation in fp32:
//   low = min_low + |low_hz_|, high = low + min_band + |band_hz_|
//   v(c, k) = ((2 pi) * (c * n_[k])) * sample_rate           (n_ = (k - (K-1)/2) / sample_rate, as the module stores it)
//   lp(c)[k] = (2 c) * s,  s = sin(v)/v left of the centre, 1 at the centre, the MIRRORED left value right of it
//   bp = lp(high) - lp(low);  filt = (bp / max_k bp) * window        (max: first index on ties, as torch.max)
// Backward is analytic: d lp(c)[k] / dc = 2 cos(v(c, k)) at every tap (the sin(v)/v terms cancel exactly: v is
// proportional to c), the max routes its gradient to the arg-max tap, |.| contributes sign().  One workgroup per filter.
#include <math.h>

#include "pk_common.h"

namespace {

constexpr int SINC_THREADS = 256;

__device__ __forceinline__ float sinc_tap(float c, const float* __restrict__ n_, int k, int K, float sr) {
    const int half = (K - 1) / 2;
    if (k == half) return 2.0f * c;
    const int kl = k < half ? k : K - 1 - k;  // the right half repeats the left values (torch.flip of the left half)
    const float v = ((float)(2.0 * M_PI) * (c * n_[kl])) * sr;
    return (2.0f * c) * (sinf(v) / v);
}
__device__ __forceinline__ float sinc_arg(float c, const float* __restrict__ n_, int k, int K, float sr) {
    const int half = (K - 1) / 2;
    const int kl = k < half ? k : (k == half ? half : K - 1 - k);
    return ((float)(2.0 * M_PI) * (c * n_[kl])) * sr;
}

// filt [N][K]; saves bp / max (the un-windowed normalised band pass) is not needed: backward recomputes from the
// parameters.  mx_o [N], kstar_o [N]: the maximum and its tap.
__global__ __launch_bounds__(SINC_THREADS) void sinc_bank_fwd_kernel(const float* __restrict__ low_hz, const float* __restrict__ band_hz,
                                                                     const float* __restrict__ n_, const float* __restrict__ window,
                                                                     int K, float sr, float min_low, float min_band,
                                                                     float* __restrict__ filt, float* __restrict__ mx_o,
                                                                     int* __restrict__ kstar_o) {
    __shared__ float s_v[SINC_THREADS];
    __shared__ int s_k[SINC_THREADS];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float low = min_low + fabsf(low_hz[i]);
    const float high = (low + min_band) + fabsf(band_hz[i]);
    float best = -INFINITY;
    int bk = 0x7FFFFFFF;
    for (int k = tid; k < K; k += SINC_THREADS) {
        const float bp = sinc_tap(high, n_, k, K, sr) - sinc_tap(low, n_, k, K, sr);
        if (bp > best || (bp == best && k < bk) || (bp != bp && best == best)) {  // (NaN wins, like torch.max)
            best = bp;
            bk = k;
        }
    }
    s_v[tid] = best;
    s_k[tid] = bk;
    __syncthreads();
    for (int off = SINC_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) {
            const float ov = s_v[tid + off];
            const int ok = s_k[tid + off];
            const float mv = s_v[tid];
            const int mk = s_k[tid];
            const bool take = (ov != ov && mv == mv) || (!(mv != mv) && (ov > mv || (ov == mv && ok < mk)));
            if (take) {
                s_v[tid] = ov;
                s_k[tid] = ok;
            }
        }
        __syncthreads();
    }
    const float mx = s_v[0];
    if (tid == 0) {
        mx_o[i] = mx;
        kstar_o[i] = s_k[0];
    }
    for (int k = tid; k < K; k += SINC_THREADS) {
        const float bp = sinc_tap(high, n_, k, K, sr) - sinc_tap(low, n_, k, K, sr);
        filt[(long)i * K + k] = (bp / mx) * window[k];
    }
}

// g [N][K] = dL/dfilt -> dlow [N], dband [N]
__global__ __launch_bounds__(SINC_THREADS) void sinc_bank_bwd_kernel(const float* __restrict__ g, const float* __restrict__ low_hz,
                                                                     const float* __restrict__ band_hz, const float* __restrict__ n_,
                                                                     const float* __restrict__ window, const float* __restrict__ mx_i,
                                                                     const int* __restrict__ kstar_i, int K, float sr,
                                                                     float min_low, float min_band, float* __restrict__ dlow,
                                                                     float* __restrict__ dband) {
    __shared__ float sh[3][SINC_THREADS];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float low = min_low + fabsf(low_hz[i]);
    const float high = (low + min_band) + fabsf(band_hz[i]);
    const float mx = mx_i[i];
    const int ks = kstar_i[i];
    // d bp[k] = w g / mx  -  [k == k*] sum_k' (w g bp)[k'] / mx^2 ;  d c_low = sum_k d bp[k] (-2 cos v_low), d c_high = sum_k d bp[k] 2 cos v_high
    float s_gb = 0.f, s_lo = 0.f, s_hi = 0.f;
    for (int k = tid; k < K; k += SINC_THREADS) {
        const float wg = window[k] * g[(long)i * K + k];
        const float bp = sinc_tap(high, n_, k, K, sr) - sinc_tap(low, n_, k, K, sr);
        s_gb += wg * bp;
        const float d = wg / mx;
        s_lo += d * (-2.0f * cosf(sinc_arg(low, n_, k, K, sr)));
        s_hi += d * (2.0f * cosf(sinc_arg(high, n_, k, K, sr)));
    }
    sh[0][tid] = s_gb;
    sh[1][tid] = s_lo;
    sh[2][tid] = s_hi;
    __syncthreads();
    for (int off = SINC_THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) {
#pragma unroll
            for (int q = 0; q < 3; ++q) sh[q][tid] += sh[q][tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float dmax = -sh[0][0] / (mx * mx);  // through bp / max: lands on the arg-max tap
        const float c_lo = sh[1][0] + dmax * (-2.0f * cosf(sinc_arg(low, n_, ks, K, sr)));
        const float c_hi = sh[2][0] + dmax * (2.0f * cosf(sinc_arg(high, n_, ks, K, sr)));
        const float lz = low_hz[i], bz = band_hz[i];
        // torch.abs backward: sign(x) (0 at 0)
        dlow[i] = (lz > 0.f ? 1.f : (lz < 0.f ? -1.f : 0.f)) * (c_lo + c_hi);
        dband[i] = (bz > 0.f ? 1.f : (bz < 0.f ? -1.f : 0.f)) * c_hi;
    }
}

}  // namespace

extern "C" int pk_sinc_bank_fwd(void* stream, const float* low_hz, const float* band_hz, const float* n_, const float* window,
                                int N, int K, float sample_rate, float min_low, float min_band, float* filt, float* mx,
                                int32_t* kstar) {
    if (N <= 0) return 0;
    PK_REQUIRE(K >= 3 && (K & 1) == 1, "pk_sinc_bank_fwd: the kernel size must be odd (SincConv makes it so), got %d", K);
    PK_REQUIRE(low_hz && band_hz && n_ && window && filt && mx && kstar, "pk_sinc_bank_fwd: null argument");
    hipLaunchKernelGGL(sinc_bank_fwd_kernel, dim3(N), dim3(SINC_THREADS), 0, pk_stream(stream), low_hz, band_hz, n_, window, K,
                       sample_rate, min_low, min_band, filt, mx, (int*)kstar);
    PK_LAUNCH_CHECK();
    return 0;
}
extern "C" int pk_sinc_bank_bwd(void* stream, const float* g, const float* low_hz, const float* band_hz, const float* n_,
                                const float* window, const float* mx, const int32_t* kstar, int N, int K, float sample_rate,
                                float min_low, float min_band, float* dlow, float* dband) {
    if (N <= 0) return 0;
    PK_REQUIRE(g && low_hz && band_hz && n_ && window && mx && kstar && dlow && dband, "pk_sinc_bank_bwd: null argument");
    hipLaunchKernelGGL(sinc_bank_bwd_kernel, dim3(N), dim3(SINC_THREADS), 0, pk_stream(stream), g, low_hz, band_hz, n_, window, mx,
                       (const int*)kstar, K, sample_rate, min_low, min_band, dlow, dband);
    PK_LAUNCH_CHECK();
    return 0;
}
