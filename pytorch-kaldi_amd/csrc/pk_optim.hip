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
