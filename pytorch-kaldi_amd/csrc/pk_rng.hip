// pk_rng.hip - the reference's drop-mask STREAM on the device.
//
// The reference draws a recurrent layer's drop mask with torch.bernoulli(torch.Tensor(rows, H).fill_(1 - p)) on the
// global CPU generator (neural_networks.py:1102-1107, :430-441, :604-615, ...): a scalar loop over rows * H elements,
// each taking ONE 32-bit output of an mt19937 engine (at::CPUGeneratorImpl -> at::mt19937), u = (y & 0xFFFFFF) * 2^-24,
// mask = u < 1 - p.  PK_MASK_RNG=reference reproduced that by making the very same call on the host (40 ms per training
// step at the BASELINE shape even with a helper thread: 5 x 140 800 scalar draws).  The stream does not depend on data,
// only on the generator's state - so the engine now carries a MIRROR of that state on the device and advances it there:
// one workgroup regenerates the 624-word state block by block (the twist has three dependent phases of <= 227 independent
// words each) and tempers / thresholds 624 outputs per block in parallel.  Bit-identical masks, bit-identical state
// afterwards (written back into torch's generator when the host needs it: functional._RefRng), ~0.2 us per block.
#include "pk_common.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;

__device__ __forceinline__ unsigned mt_twist(unsigned u, unsigned v) {
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// st: [626] = state words, left, next - the fields of at::mt19937's data (left counts down and the block is regenerated
// when it reaches 0; a freshly seeded engine has left = 1, next = 0).  out[i] = 1 if the i-th draw's uniform < keep else 0.
__global__ __launch_bounds__(256) void mt19937_bernoulli_kernel(unsigned* __restrict__ st, long n, float keep,
                                                                float* __restrict__ out) {
    __shared__ unsigned buf[2][MT_N];
    const int tid = threadIdx.x;
    int cur = 0;
    for (int j = tid; j < MT_N; j += 256) buf[0][j] = st[j];
    int left = (int)st[MT_N], next = (int)st[MT_N + 1];
    __syncthreads();
    long i = 0;
    while (i < n) {  // (uniform: every thread carries the same counters)
        left -= 1;
        if (left == 0) {
            const unsigned* p = buf[cur];
            unsigned* q = buf[cur ^ 1];
            for (int j = tid; j < MT_N - MT_M; j += 256) q[j] = p[j + MT_M] ^ mt_twist(p[j], p[j + 1]);
            __syncthreads();
            for (int j = MT_N - MT_M + tid; j < 2 * (MT_N - MT_M); j += 256) q[j] = q[j - (MT_N - MT_M)] ^ mt_twist(p[j], p[j + 1]);
            __syncthreads();
            for (int j = 2 * (MT_N - MT_M) + tid; j < MT_N - 1; j += 256) q[j] = q[j - (MT_N - MT_M)] ^ mt_twist(p[j], p[j + 1]);
            if (tid == 0) q[MT_N - 1] = q[MT_M - 1] ^ mt_twist(p[MT_N - 1], q[0]);
            __syncthreads();
            cur ^= 1;
            left = MT_N;
            next = 0;
        }
        const long rest = n - i;
        const int take = rest < (long)left ? (int)rest : left;
        const unsigned* p = buf[cur];
        for (int k = tid; k < take; k += 256) {
            const unsigned y = mt_temper(p[next + k]);
            const float u = (float)(y & 0xFFFFFFu) * 5.9604644775390625e-08f;  // 2^-24: at::uniform_real_distribution<float>
            out[i + k] = u < keep ? 1.0f : 0.0f;
        }
        i += take;
        next += take;
        left -= take - 1;
    }
    __syncthreads();
    for (int j = tid; j < MT_N; j += 256) st[j] = buf[cur][j];
    if (tid == 0) {
        st[MT_N] = (unsigned)left;
        st[MT_N + 1] = (unsigned)next;
    }
}

}  // namespace

extern "C" int pk_mt19937_bernoulli(void* stream, uint32_t* state, int64_t n, float keep, float* out) {
    PK_REQUIRE(state != nullptr && (n == 0 || out != nullptr) && n >= 0, "pk_mt19937_bernoulli: null argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(mt19937_bernoulli_kernel, dim3(1), dim3(256), 0, pk_stream(stream), (unsigned*)state, (long)n, keep, out);
    PK_LAUNCH_CHECK();
    return 0;
}
