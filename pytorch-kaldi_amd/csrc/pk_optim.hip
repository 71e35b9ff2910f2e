// pk_optim.hip - fused optimizer steps on flat fp32 buckets:
// torch.optim.RMSprop / SGD / Adam exactly as utils.optimizer_init configures them
// (utils.py:2106-2164; momentum-free, non-centred RMSprop in every shipped cfg).
// One pass over {param, grad, state}: RMSprop 16 B/param read + 8 B/param written.
#include "pk_common.h"

namespace {

__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, long n, float lr,
                               float alpha, float eps, float wd) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gv = g[i];
        const float pv = p[i];
        if (wd != 0.f) gv += wd * pv;
        const float s = alpha * sq[i] + (1.f - alpha) * gv * gv;
        sq[i] = s;
        p[i] = pv - lr * gv / (sqrtf(s) + eps);
    }
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long n, float lr,
                           float momentum, float wd, int first) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gv = g[i];
        const float pv = p[i];
        if (wd != 0.f) gv += wd * pv;
        if (momentum != 0.f) {
            const float b = first ? gv : momentum * buf[i] + gv;
            buf[i] = b;
            gv = b;
        }
        p[i] = pv - lr * gv;
    }
}

// torch.optim.Adam (L2 weight decay folded into the gradient, optional amsgrad): bc1 = 1 - beta1^t, bc2s = sqrt(1 - beta2^t)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ vmax, long n, float lr, float beta1, float beta2, float eps, float wd,
                            float bc1, float bc2s) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gv = g[i];
        const float pv = p[i];
        if (wd != 0.f) gv += wd * pv;
        const float mv = beta1 * m[i] + (1.f - beta1) * gv;
        float vv = beta2 * v[i] + (1.f - beta2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        if (vmax != nullptr) {
            vv = fmaxf(vmax[i], vv);
            vmax[i] = vv;
        }
        const float denom = sqrtf(vv) / bc2s + eps;
        p[i] = pv - (lr / bc1) * (mv / denom);
    }
}

// ---- round 4: the same three steps with (a) the gradient zeroed on the way out - the zero_grad() in front of the next
// backward pass then has nothing left to do (one fill launch per optimizer and step less) - and (b) the bf16 copies the
// perf-mode GEMMs read of the 2-D weights written on the way: segs [nseg][5] = (first element in the flat buffer, rows,
// columns, pitch of the copy, first element of the copy in `shadow`), ascending; a thread owns four consecutive elements.
template <int KIND>
__global__ void fused_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ s0, float* __restrict__ s1,
                                  float* __restrict__ s2, long n, float lr, float h0, float h1, float h2, float wd, float bc1,
                                  float bc2s, int first, int zero_grad, const long* __restrict__ segs, int nseg,
                                  unsigned short* __restrict__ shadow) {
    for (long i = 4 * (blockIdx.x * (long)blockDim.x + threadIdx.x); i < n; i += 4 * (long)gridDim.x * blockDim.x) {
        f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
        f32x4 pv = *reinterpret_cast<const f32x4*>(p + i);
        if (wd != 0.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] += wd * pv[e];
        }
        if (KIND == 0) {  // RMSprop (momentum-free, non-centred): h0 = alpha, h1 = eps
            f32x4 sq = *reinterpret_cast<const f32x4*>(s0 + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sq[e] = h0 * sq[e] + (1.f - h0) * gv[e] * gv[e];
                pv[e] = pv[e] - lr * gv[e] / (sqrtf(sq[e]) + h1);
            }
            *reinterpret_cast<f32x4*>(s0 + i) = sq;
        } else if (KIND == 1) {  // SGD: h0 = momentum
            if (h0 != 0.f) {
                f32x4 b = *reinterpret_cast<const f32x4*>(s0 + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    b[e] = first ? gv[e] : h0 * b[e] + gv[e];
                    gv[e] = b[e];
                }
                *reinterpret_cast<f32x4*>(s0 + i) = b;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[e] = pv[e] - lr * gv[e];
        } else {  // Adam: h0 = beta1, h1 = beta2, h2 = eps; s2 = amsgrad maximum or null
            f32x4 m = *reinterpret_cast<const f32x4*>(s0 + i), v = *reinterpret_cast<const f32x4*>(s1 + i);
            f32x4 vm = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m[e] = h0 * m[e] + (1.f - h0) * gv[e];
                v[e] = h1 * v[e] + (1.f - h1) * gv[e] * gv[e];
                vm[e] = v[e];
            }
            *reinterpret_cast<f32x4*>(s0 + i) = m;
            *reinterpret_cast<f32x4*>(s1 + i) = v;
            if (s2 != nullptr) {
                const f32x4 old = *reinterpret_cast<const f32x4*>(s2 + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) vm[e] = fmaxf(old[e], v[e]);
                *reinterpret_cast<f32x4*>(s2 + i) = vm;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[e] = pv[e] - (lr / bc1) * (m[e] / (sqrtf(vm[e]) / bc2s + h2));
        }
        *reinterpret_cast<f32x4*>(p + i) = pv;
        if (zero_grad) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nseg > 0) {  // the segment that holds element i: the last one that starts at or before it
            int lo = 0, hi = nseg;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (segs[5 * mid] <= i) lo = mid;
                else hi = mid;
            }
            const long off = segs[5 * lo], rows = segs[5 * lo + 1], cols = segs[5 * lo + 2], pitch = segs[5 * lo + 3], soff = segs[5 * lo + 4];
            const long rel = i - off;
            if (rel >= 0 && rel < rows * cols) {  // (columns are a multiple of 4: the four elements share a row)
                const long row = rel / cols, col = rel - row * cols;
                uint2 pk;
                pk.x = pk_pack_bf2(pv[0], pv[1]);
                pk.y = pk_pack_bf2(pv[2], pv[3]);
                *reinterpret_cast<uint2*>(shadow + soff + row * pitch + col) = pk;
            }
        }
    }
}

inline int blocks_for(long n) {
    long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int pk_rmsprop_step(void* stream, float* p, const float* g, float* square_avg, int64_t n, float lr, float alpha,
                               float eps, float weight_decay) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(rmsprop_kernel, dim3(blocks_for(n)), dim3(256), 0, pk_stream(stream), p, g, square_avg, (long)n, lr,
                       alpha, eps, weight_decay);
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_sgd_step(void* stream, float* p, const float* g, float* momentum_buf, int64_t n, float lr,
                           float momentum, float weight_decay, int first_step) {
    if (n == 0) return 0;
    PK_REQUIRE(momentum == 0.f || momentum_buf != nullptr, "pk_sgd_step: momentum needs a buffer");
    hipLaunchKernelGGL(sgd_kernel, dim3(blocks_for(n)), dim3(256), 0, pk_stream(stream), p, g, momentum_buf, (long)n, lr,
                       momentum, weight_decay, first_step);
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int pk_adam_step(void* stream, float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                            int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step) {
    if (n == 0) return 0;
    PK_REQUIRE(step >= 1 && exp_avg != nullptr && exp_avg_sq != nullptr, "pk_adam_step: step counts from 1 and needs both moments");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(256), 0, pk_stream(stream), p, g, exp_avg, exp_avg_sq,
                       max_exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s);
    PK_LAUNCH_CHECK();
    return 0;
}

// kind: 0 = RMSprop (h0 = alpha, h1 = eps), 1 = SGD (h0 = momentum; s0 = momentum buffer or NULL), 2 = Adam (h0 / h1 = betas,
// h2 = eps, s0 / s1 = moments, s2 = amsgrad maximum or NULL); step counts from 1.  n and every pointer: multiples of 4
// elements / 16 bytes (optim.FlatParams aligns to 64).  zero_grad != 0: g is zero afterwards.  segs (device, [nseg][5]
// int64, may be NULL): see fused_step_kernel.
extern "C" int pk_fused_step(void* stream, int kind, float* p, float* g, float* s0, float* s1, float* s2, int64_t n, float lr,
                             float h0, float h1, float h2, float weight_decay, int step, int zero_grad, const int64_t* segs,
                             int nseg, uint16_t* shadow) {
    if (n == 0) return 0;
    PK_REQUIRE(kind >= 0 && kind <= 2 && step >= 1, "pk_fused_step: kind 0..2, step counts from 1");
    PK_REQUIRE((n & 3) == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)s0 | (uintptr_t)s1 | (uintptr_t)s2) & 15) == 0,
               "pk_fused_step: n and the buffers must be multiples of 4 elements / 16-byte aligned");
    PK_REQUIRE(kind != 0 || s0 != nullptr, "pk_fused_step: RMSprop needs its square average");
    PK_REQUIRE(kind != 1 || h0 == 0.f || s0 != nullptr, "pk_fused_step: momentum needs a buffer");
    PK_REQUIRE(kind != 2 || (s0 != nullptr && s1 != nullptr), "pk_fused_step: Adam needs both moments");
    PK_REQUIRE(nseg == 0 || (segs != nullptr && shadow != nullptr && ((uintptr_t)shadow & 7) == 0), "pk_fused_step: null segment table / copy");
    const float bc1 = kind == 2 ? 1.f - powf(h0, (float)step) : 1.f;
    const float bc2s = kind == 2 ? sqrtf(1.f - powf(h1, (float)step)) : 1.f;
    const dim3 grid(blocks_for(n / 4)), block(256);
    hipStream_t st = pk_stream(stream);
    const long* sg = reinterpret_cast<const long*>(segs);
    if (kind == 0) hipLaunchKernelGGL(fused_step_kernel<0>, grid, block, 0, st, p, g, s0, s1, s2, (long)n, lr, h0, h1, h2, weight_decay, bc1, bc2s, 0, zero_grad, sg, nseg, (unsigned short*)shadow);
    else if (kind == 1) hipLaunchKernelGGL(fused_step_kernel<1>, grid, block, 0, st, p, g, s0, s1, s2, (long)n, lr, h0, h1, h2, weight_decay, bc1, bc2s, step == 1 ? 1 : 0, zero_grad, sg, nseg, (unsigned short*)shadow);
    else hipLaunchKernelGGL(fused_step_kernel<2>, grid, block, 0, st, p, g, s0, s1, s2, (long)n, lr, h0, h1, h2, weight_decay, bc1, bc2s, 0, zero_grad, sg, nseg, (unsigned short*)shadow);
    PK_LAUNCH_CHECK();
    return 0;
}
