// pk_gemm_bf16.hip - perf-mode GEMM for gfx950: bf16 operands resident in HBM,
// fp32 accumulate / output.  Replaces the same reference call sites as pk_gemm
// (nn.Linear forward/backward, neural_networks.py:111/139-148, :432-435,
// :609-611, :1114-1115, and the deferred dU = sum_t dgate_t^T . h_{t-1}).
//
//   C[M,N] = alpha * sum_k A(m,k) * B(k,n) + beta*C + bias
//
// Each operand is either "k-contiguous" (KC: A stored [M][lda], B stored
// [N][ldb] - activations x weights^T) or "k-major" (A stored [K][lda] with m
// contiguous, B stored [K][ldb] - the dW / dU shapes whose reduction runs over
// the T*B rows).  No operand is ever transposed in HBM:
//   * tiles go HBM -> LDS with the LDS-DMA (global_load_lds_dwordx4, 16 B per
//     lane, no VGPR round trip); the LDS image is lane-linear, so the bank
//     swizzle is applied to the per-lane SOURCE address and again on the read;
//   * KC fragments are one ds_read_b128; k-major fragments are two
//     ds_read_b64_tr_b16 (the gfx950 LDS transpose read) - the MFMA operand
//     wants 8 consecutive k per lane, the image has k as the row index.
// Block tile 128 x 128 x 64, 4 waves (2 x 2), wave tile 64 x 64 = 4 x 4
// v_mfma_f32_16x16x32_bf16, LDS double buffer (64 KB), one barrier per k-tile.
// blockIdx -> tile mapping is XCD-aware: the n-tiles of one 128-row A panel run
// back to back on ONE XCD so the panel is read from HBM once and re-read from
// that XCD's L2.
//
// Edges: rows beyond M / N are clamped (their results are never stored);
// 16-byte k-chunks beyond K come from a zero page, so K needs no padding
// beyond a multiple of 8 elements (producers zero-fill inside the last chunk).
#include <stdlib.h>

#include "pk_common.h"

namespace {

constexpr int TM = 128, TN = 128, TK = 64;
typedef short s4v __attribute__((ext_vector_type(4)));

struct BArgs {
    int M, N, K;
    float alpha, beta;
    const unsigned short* A;
    long lda;
    const unsigned short* B;
    long ldb;
    float* C;
    long ldc;
    const float* bias;
    float* ws;  // split-K slabs [splits][M][N] or null
    int k_per_split;
    int tiles_m, tiles_n;
    int items, per_xcd;  // work items = splits x tiles_m x tiles_n, and ceil(items / 8)
    const unsigned short* zeros;  // >= 16 bytes of zeros
    float* stats;  // 256-tile, no split-K: per (m-tile, column) (rows, mean, M2) of the output, [tiles_m][N][3], or null
};

__device__ unsigned short g_zero_page[64];

// swizzles (same involution on the staging source and on the fragment read)
__device__ __forceinline__ int swz_kc(int row) { return (row >> 1) & 7; }                  // 8 x 16-B slots per row
__device__ __forceinline__ int swz_km(int kr) { return 2 * ((kr & 3) | ((kr >> 1) & 4)); }  // 16 x 16-B slots per row

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// branch-free address select (a pointer ?: makes hipcc emit two exec-masked LDS-DMA loads)
__device__ __forceinline__ unsigned long long sel_addr(bool ok, const void* a, const void* z) {
    const unsigned long long m = 0ull - (unsigned long long)ok;
    return ((unsigned long long)a & m) | ((unsigned long long)z & ~m);
}

// Stage one 128 x 64 operand tile (16 KB) into `buf`.  KC image: [128 rows][8 slots]; k-major image: [64 k][16 slots].
template <bool KC, int NT = 256>
__device__ __forceinline__ void stage(const unsigned short* __restrict__ base, long ld, int r0, int rmax, int k0, int kmax,
                                      const unsigned short* zeros, unsigned char* buf, int tid, int wave) {
#pragma unroll
    for (int j = 0; j < 1024 / NT; ++j) {
        const int q = j * NT + tid;
        unsigned long long src;  // selected as an integer so that it stays ONE load (no exec-masked pair)
        if (KC) {
            const int row = q >> 3, ps = q & 7;
            const int ks = ps ^ swz_kc(row);
            int gr = r0 + row;
            gr = gr < rmax ? gr : rmax - 1;
            const int gk = k0 + ks * 8;
            src = sel_addr(gk < kmax, base + (long)gr * ld + gk, zeros);
        } else {
            const int kr = q >> 4, ps = q & 15;
            const int cs = ps ^ swz_km(kr);
            int gc = r0 + cs * 8;
            if (gc >= rmax) gc = (rmax - 1) & ~7;  // chunk entirely out of range: any in-range chunk (never stored)
            const int gk = k0 + kr;
            src = sel_addr(gk < kmax, base + (long)gk * ld + gc, zeros);
        }
        glds16(reinterpret_cast<const void*>(src), buf + (j * NT + wave * 64) * 16);
    }
}

// The same tile image, fetched through per-thread source addresses that are computed ONCE and advanced by a
// constant per k-tile: the address arithmetic of stage() (a 64-bit multiply per load for k-major operands) costs
// more VALU cycles per k-tile than the tile's MFMAs take.  Valid for k-tiles that lie entirely below kmax; the
// (single) ragged last tile of a split goes through stage().
template <bool KC, int NT = 256>
struct TileSrc {
    unsigned long long a[1024 / NT];
    unsigned long long step;
    __device__ __forceinline__ void init(const unsigned short* __restrict__ base, long ld, int r0, int rmax, int k0, int tid) {
#pragma unroll
        for (int j = 0; j < 1024 / NT; ++j) {
            const int q = j * NT + tid;
            if (KC) {
                const int row = q >> 3, ps = q & 7;
                const int ks = ps ^ swz_kc(row);
                int gr = r0 + row;
                gr = gr < rmax ? gr : rmax - 1;
                a[j] = (unsigned long long)(base + (long)gr * ld + k0 + ks * 8);
            } else {
                const int kr = q >> 4, ps = q & 15;
                const int cs = ps ^ swz_km(kr);
                int gc = r0 + cs * 8;
                if (gc >= rmax) gc = (rmax - 1) & ~7;
                a[j] = (unsigned long long)(base + (long)(k0 + kr) * ld + gc);
            }
        }
        step = KC ? (unsigned long long)(TK * 2) : (unsigned long long)ld * (TK * 2);
    }
    __device__ __forceinline__ void issue(unsigned char* buf, int wave) {
#pragma unroll
        for (int j = 0; j < 1024 / NT; ++j) {
            glds16(reinterpret_cast<const void*>(a[j]), buf + (j * NT + wave * 64) * 16);
            a[j] += step;
        }
    }
};

// MFMA 16x16x32 operand fragment for the 16 rows starting at `sub` (tile-local), k-step kk (0/1) of the 64-deep tile.
template <bool KC>
__device__ __forceinline__ bf16x8 frag(const unsigned char* buf, int sub, int kk, int lane) {
    if (KC) {
        const int row = sub + (lane & 15);
        const int ks = kk * 4 + (lane >> 4);
        const int ps = ks ^ swz_kc(row);
        return *reinterpret_cast<const bf16x8*>(buf + row * 128 + ps * 16);
    } else {
        const int g = lane >> 4, i = lane & 15;
        const int col = sub + (i & 3) * 4;  // tile-local m (or n) of this lane's 8-byte piece
        bf16x8 out;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kr = kk * 32 + g * 8 + h * 4 + (i >> 2);
            const int ps = (col >> 3) ^ swz_km(kr);
            const unsigned char* a = buf + kr * 256 + ps * 16 + (col & 7) * 2;
            const s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)a);
            out[4 * h + 0] = v[0];
            out[4 * h + 1] = v[1];
            out[4 * h + 2] = v[2];
            out[4 * h + 3] = v[3];
        }
        return out;
    }
}

// D' layout of a wave's 64 x 64 tile at (mw, nw): row (lane>>4)*4 + r is the COLUMN offset, lane&15 the ROW offset of C
__device__ __forceinline__ void epilogue(const BArgs& p, const f32x4 (&acc)[4][4], int mw, int nw, int split, int lane) {
    const bool vec_ok = (((uintptr_t)(p.ws ? p.ws : p.C) & 15) == 0) && (((p.ws ? (long)p.N : p.ldc) & 3) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = mw + i * 16 + (lane & 15);
            const int col = nw + j * 16 + (lane >> 4) * 4;
            if (row >= p.M || col >= p.N) continue;
            const f32x4 v = acc[i][j];
            float* dst = p.ws ? p.ws + ((long)split * p.M + row) * p.N + col : p.C + (long)row * p.ldc + col;
            if (vec_ok && col + 3 < p.N) {
                f32x4 o = v;
                if (p.ws == nullptr) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = p.alpha * v[r] + (p.bias ? p.bias[col + r] : 0.f);
                    if (p.beta != 0.f) {
                        const f32x4 c = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] += p.beta * c[r];
                    }
                }
                *reinterpret_cast<f32x4*>(dst) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (col + r >= p.N) continue;
                    float o = v[r];
                    if (p.ws == nullptr) {
                        o = p.alpha * o + (p.bias ? p.bias[col + r] : 0.f);
                        if (p.beta != 0.f) o += p.beta * dst[r];
                    }
                    dst[r] = o;
                }
            }
        }
}


// STAGES = 2: LDS double buffer, one barrier per k-tile, 2 workgroups per CU (64 KB each).
// STAGES = 1: single buffer, two barriers per k-tile, 3-4 workgroups per CU (32 KB each): the overlap of
//             loads and MFMA comes from the co-resident workgroups instead of from the software pipeline.
// (A ring of four buffers with three k-tiles in flight was tried for the small-batch shapes - 8 workgroups on 8 CUs - and
// changed nothing: 128 rows x 1024 x 1024 stayed at 16-25 us.  Those launches are bound by what ONE CU can pull through
// its memory pipe (~32 GB/s each), not by exposed latency: see gemm_bf16s_kernel, which spreads them over 32+ CUs.)
template <bool A_KC, bool B_KC, int STAGES>
__global__ __launch_bounds__(256, STAGES == 2 ? 2 : 4) void gemm_bf16x_kernel(BArgs p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // [STAGES][A 16 KB | B 16 KB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware work mapping (block b runs on XCD b % 8 - observed round-robin dispatch; speed only): the
    // (split, m-tile, n-tile) items, n fastest, are cut into 8 contiguous ranges, one per XCD.  Every XCD gets the
    // same number of items whatever tiles_m is, and the items that share an A panel / a k-range sit on one L2.
    const int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
    if (item >= p.items) return;
    const int tn = item % p.tiles_n;
    const int tm = (item / p.tiles_n) % p.tiles_m;
    const int split = item / (p.tiles_n * p.tiles_m);
    const int m0 = tm * TM, n0 = tn * TN;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + TK - 1) / TK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operands are swapped in the MFMA (D' = B.A^T): a lane then holds 4 consecutive COLUMNS n of one row m,
    // so the epilogue stores 16 bytes per lane instead of four scattered dwords
    TileSrc<A_KC> sa;
    TileSrc<B_KC> sb;
    sa.init(p.A, p.lda, m0, p.M, kbeg, tid);
    sb.init(p.B, p.ldb, n0, p.N, kbeg, tid);
    auto fetch = [&](int kt, unsigned char* buf) {
        const int k0 = kbeg + kt * TK;
        // incremental addresses pay for k-major operands only (their stage() address has a loop-variant 64-bit
        // multiply); every k-tile but a ragged last one takes that path
        const bool full = k0 + TK <= kend;
        if (!A_KC && full) sa.issue(buf, wave);
        else stage<A_KC>(p.A, p.lda, m0, p.M, k0, kend, p.zeros, buf, tid, wave);
        if (!B_KC && full) sb.issue(buf + 16384, wave);
        else stage<B_KC>(p.B, p.ldb, n0, p.N, k0, kend, p.zeros, buf + 16384, tid, wave);
    };
    if (STAGES == 2) {
        if (nk > 0) fetch(0, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int kt = 0; kt < nk; ++kt) {
        unsigned char* cur = smem + (STAGES == 2 ? (kt & 1) * 32768 : 0);
        if (STAGES == 2) {
            if (kt + 1 < nk) fetch(kt + 1, smem + ((kt + 1) & 1) * 32768);
        } else {
            fetch(kt, cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = frag<A_KC>(cur, wm * 64 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = frag<B_KC>(cur + 16384, wn * 64 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        if (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    epilogue(p, acc, m0 + wm * 64, n0 + wn * 64, split, lane);
}

// ============================================================================
// Small-batch products (round 3): M <= 128 rows, both operands k-contiguous - an MLP layer, an output layer or a SincNet
// fully-connected layer at batch 128.  On the 128 x 128 tile such a product is 8 workgroups on 8 CUs, and a CU pulls
// ~32 GB/s through its memory pipe whatever the software pipeline does (two stages, four stages and one stage all took
// 16-25 us for 128 x 1024 x 1024, 33 us with the layer epilogue below).  Here a workgroup owns ALL rows of 32 output
// columns: 32-61 workgroups, each re-reading the 128-row A panel from L2 (256 KB) next to its 64 KB of B.
//
// FUSED: one MLP layer in ONE launch - z = x W^T + b, BatchNorm1d in training mode over the batch, activation, drop
// mask, and the bf16 copy the next layer's GEMM reads (neural_networks.py:139-148, `drop(act(bn(wx(x))))`).  With every
// row of its columns in the accumulators of one workgroup the batch statistics are a reduction inside it (two passes
// over the registers: mean, then centred second moment - nothing cancels): replaces seven launches of the unfused
// pipeline (GEMM, 2 x statistics, finalize, affine + activation, mask, bf16 conversion of the next layer's input).
// ============================================================================
struct BnActArgs {
    const float *gamma, *beta, *mask;
    float eps, momentum, unbias;
    float *rmean, *rvar, *mean, *var;
    float *zout, *aout, *yout;  // fp32 [M][N]: pre-BN output, activation output (pre-mask), layer output (null: no mask)
    unsigned short* yb;         // bf16 [M][ldyb] layer output
    long ldyb;
    int act;
};

constexpr int SN = 32;                    // columns per workgroup
constexpr int SSTAGE = 16384 + SN * 128;  // bytes per stage: A 128 x 64 | B 32 x 64

// k-contiguous tile image of ROWS rows x 64 k (the [row][8 slots] image of stage<true>, ROWS * 8 chunks of 16 bytes)
template <int ROWS>
__device__ __forceinline__ void stage_kc_rows(const unsigned short* __restrict__ base, long ld, int r0, int rmax, int k0, int kmax,
                                              const unsigned short* zeros, unsigned char* buf, int tid, int wave) {
    static_assert(ROWS * 8 == 256, "one 16-byte chunk per thread");
    const int row = tid >> 3, ps = tid & 7;
    const int ks = ps ^ swz_kc(row);
    int gr = r0 + row;
    gr = gr < rmax ? gr : rmax - 1;
    const int gk = k0 + ks * 8;
    const unsigned long long src = sel_addr(gk < kmax, base + (long)gr * ld + gk, zeros);
    glds16(reinterpret_cast<const void*>(src), buf + (wave * 64) * 16);
}

// k-major tile image of 64 k x 32 columns: [64 k][4 slots of 8 columns] = 256 chunks, one per thread (B stored [K][ldb],
// n contiguous: the dX product of a layer, dz . W).  No swizzle: the four k-rows a 16-lane group of the transpose read
// touches are 256 contiguous bytes.
__device__ __forceinline__ void stage_km32(const unsigned short* __restrict__ base, long ld, int c0, int cmax, int k0, int kmax,
                                           const unsigned short* zeros, unsigned char* buf, int tid, int wave) {
    const int kr = tid >> 2, cs = tid & 3;
    int gc = c0 + cs * 8;
    if (gc >= cmax) gc = (cmax - 1) & ~7;  // chunk entirely out of range: any in-range chunk (never stored)
    const int gk = k0 + kr;
    const unsigned long long src = sel_addr(gk < kmax, base + (long)gk * ld + gc, zeros);
    glds16(reinterpret_cast<const void*>(src), buf + (wave * 64) * 16);
}
__device__ __forceinline__ bf16x8 frag_km32(const unsigned char* buf, int sub, int kk, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int col = sub + (i & 3) * 4;
    bf16x8 out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kr = kk * 32 + g * 8 + h * 4 + (i >> 2);
        const unsigned char* a = buf + kr * 64 + col * 2;
        const s4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)a);
        out[4 * h + 0] = v[0];
        out[4 * h + 1] = v[1];
        out[4 * h + 2] = v[2];
        out[4 * h + 3] = v[3];
    }
    return out;
}

// sum over the 16 lanes of a DPP row (quad_perm x 2, row_half_mirror, row_mirror): every lane ends with the total
__device__ __forceinline__ float dpp16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

template <bool FUSED, bool B_KC = true>
__global__ __launch_bounds__(256, 2) void gemm_bf16s_kernel(BArgs p, BnActArgs q) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // [2][A 16 KB | B 4 KB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * SN;
    const int nk = (p.K + TK - 1) / TK;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int kt, unsigned char* buf) {
        const int k0 = kt * TK;
        stage<true>(p.A, p.lda, 0, p.M, k0, p.K, p.zeros, buf, tid, wave);
        if (B_KC) stage_kc_rows<SN>(p.B, p.ldb, n0, p.N, k0, p.K, p.zeros, buf + 16384, tid, wave);
        else stage_km32(p.B, p.ldb, n0, p.N, k0, p.K, p.zeros, buf + 16384, tid, wave);
    };
    if (nk > 0) fetch(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        unsigned char* cur = smem + (kt & 1) * SSTAGE;
        if (kt + 1 < nk) fetch(kt + 1, smem + ((kt + 1) & 1) * SSTAGE);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = frag<true>(cur, wave * 32 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = B_KC ? frag<true>(cur + 16384, j * 16, kk, lane) : frag_km32(cur + 16384, j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // acc[i][j][r]: row = wave*32 + i*16 + (lane & 15), column = n0 + j*16 + (lane >> 4)*4 + r
    const int kq = lane >> 4, lr = lane & 15;
    const int cbase = n0 + kq * 4;
    if (!FUSED) {
        const bool vec_ok = (p.ldc & 3) == 0 && (((uintptr_t)p.C) & 15) == 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wave * 32 + i * 16 + lr;
            if (row >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = cbase + j * 16;
                float* dst = p.C + (long)row * p.ldc + col;
                if (vec_ok && col + 3 < p.N) {
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = p.alpha * acc[i][j][r] + (p.bias ? p.bias[col + r] : 0.f);
                    if (p.beta != 0.f) {
                        const f32x4 c = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] += p.beta * c[r];
                    }
                    *reinterpret_cast<f32x4*>(dst) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (col + r >= p.N) continue;
                        float o = p.alpha * acc[i][j][r] + (p.bias ? p.bias[col + r] : 0.f);
                        if (p.beta != 0.f) o += p.beta * dst[r];
                        dst[r] = o;
                    }
                }
            }
        }
        return;
    }
    // ---- the layer epilogue
    float* red = reinterpret_cast<float*>(smem);  // [pass 2][wave 4][32 columns] (all waves are past the main loop)
    bool rok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) rok[i] = wave * 32 + i * 16 + lr < p.M;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = cbase + j * 16 + r;
            const float bv = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j][r] += bv;
        }
    // sum over the 128 rows of every one of my 8 columns: over i in registers, over the 16 lanes of a DPP row, over the
    // four waves through LDS (every wave adds the four partial sums in the same order: bit-identical totals)
    auto column_total = [&](float (&s)[2][4], int pass) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[j][r] = dpp16_sum(s[j][r]);
        float* mine = red + (pass * 4 + wave) * SN;
        if (lr == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[j * 16 + kq * 4 + r] = s[j][r];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = j * 16 + kq * 4 + r;
                s[j][r] = ((red[(pass * 4 + 0) * SN + c] + red[(pass * 4 + 1) * SN + c]) + red[(pass * 4 + 2) * SN + c]) +
                          red[(pass * 4 + 3) * SN + c];
            }
    };
    float mean[2][4], var[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) t += rok[i] ? acc[i][j][r] : 0.f;
            mean[j][r] = t;
        }
    column_total(mean, 0);
    const float invM = 1.0f / (float)p.M;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mean[j][r] *= invM;
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float d = acc[i][j][r] - mean[j][r];
                t += rok[i] ? d * d : 0.f;
            }
            var[j][r] = t;
        }
    column_total(var, 1);
    float sc[2][4], sh[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = cbase + j * 16 + r;
            const bool cok = col < p.N;
            var[j][r] *= invM;
            const float inv = 1.0f / sqrtf(var[j][r] + q.eps);
            const float g = (q.gamma != nullptr && cok) ? q.gamma[col] : 1.f, b = (q.beta != nullptr && cok) ? q.beta[col] : 0.f;
            sc[j][r] = g * inv;
            sh[j][r] = b - mean[j][r] * sc[j][r];
            if (wave == 0 && lr == 0 && cok) {  // one lane per column: the statistics backward needs, the running statistics
                q.mean[col] = mean[j][r];
                q.var[col] = var[j][r];
                if (q.rmean != nullptr) {
                    q.rmean[col] = (1.f - q.momentum) * q.rmean[col] + q.momentum * mean[j][r];
                    q.rvar[col] = (1.f - q.momentum) * q.rvar[col] + q.momentum * (var[j][r] * q.unbias);
                }
            }
        }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + i * 16 + lr;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = cbase + j * 16;
            if (col >= p.N) continue;  // (N is a multiple of 4 on this path: a lane's four columns are in or out together)
            const f32x4 z = acc[i][j];
            f32x4 av, yv;
#pragma unroll
            for (int r = 0; r < 4; ++r) av[r] = pk_act(q.act, z[r] * sc[j][r] + sh[j][r]);
            const long o = (long)row * p.ldc + col;
            *reinterpret_cast<f32x4*>(q.zout + o) = z;
            *reinterpret_cast<f32x4*>(q.aout + o) = av;
            yv = av;
            if (q.mask != nullptr) {
                const f32x4 mk = *reinterpret_cast<const f32x4*>(q.mask + o);
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[r] = av[r] * mk[r];
                *reinterpret_cast<f32x4*>(q.yout + o) = yv;
            }
            uint2 pk;
            pk.x = pk_pack_bf2(yv[0], yv[1]);
            pk.y = pk_pack_bf2(yv[2], yv[3]);
            *reinterpret_cast<uint2*>(q.yb + (long)row * q.ldyb + col) = pk;
        }
    }
}


// ============================================================================
// Small-batch products, round 4: cut the per-workgroup footprint until the grid covers the chip.
//
// A launch of the kernels above is as long as its LARGEST per-workgroup footprint divided by what one CU pulls through
// its memory pipe (~26 GB/s, MI355X_MICROARCH.md: 10-11 B/clk/CU): all 128 rows x 32 columns over K = 1024 is 320 KB
// per workgroup on 32 CUs - 12-18 us for 0.27 GFLOP (profiles/r03_timit_mlp_kernel_stats.csv), whatever the pipeline
// depth (the k-tiles of one workgroup arrive one L2 round trip after the other).  Three kernels replace that shape:
//   gemm_bf16sk_kernel     the same 128 x 32 column tile, the reduction SPLIT over blockIdx.y: <= SK_TILES k-tiles per
//                          workgroup, every tile's LDS-DMA issued before the first wait (40-80 KB footprints, 200-500
//                          workgroups); fp32 slabs [split][M][N] summed by splitk_reduce_bf_kernel or by
//   linear_bn_act_epi_kernel  the MLP layer epilogue (bias, batch statistics, BatchNorm, activation, mask, bf16 copy) read
//                          straight from the slabs: 16 columns x all rows per workgroup;
//   gemm_bf16_t64_kernel   weight gradients of such a layer, C[n][k] (+)= sum_m dz[m][n] x[m][k] with the batch as the
//                          (short) reduction: 64 x 64 output tiles (64 KB footprints on 256 workgroups instead of 192 KB
//                          on 64).
// ============================================================================
constexpr int SK_TILES = 4;
template <bool B_KC>
__global__ __launch_bounds__(256) void gemm_bf16sk_kernel(BArgs p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // [SK_TILES][A 16 KB | B 4 KB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * SN, split = blockIdx.y;
    const int k_lo = split * p.k_per_split;
    int k_hi = k_lo + p.k_per_split;
    k_hi = k_hi < p.K ? k_hi : p.K;
    const int nk = (k_hi - k_lo + TK - 1) / TK;  // <= SK_TILES (host)
#pragma unroll
    for (int t = 0; t < SK_TILES; ++t) {
        if (t < nk) {
            unsigned char* buf = smem + t * SSTAGE;
            const int k0 = k_lo + t * TK;
            stage<true>(p.A, p.lda, 0, p.M, k0, k_hi, p.zeros, buf, tid, wave);
            if (B_KC) stage_kc_rows<SN>(p.B, p.ldb, n0, p.N, k0, k_hi, p.zeros, buf + 16384, tid, wave);
            else stage_km32(p.B, p.ldb, n0, p.N, k0, k_hi, p.zeros, buf + 16384, tid, wave);
        }
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const unsigned char* cur = smem + t * SSTAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = frag<true>(cur, wave * 32 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = B_KC ? frag<true>(cur + 16384, j * 16, kk, lane) : frag_km32(cur + 16384, j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    }
    // acc[i][j][r]: row = wave*32 + i*16 + (lane & 15), column = n0 + j*16 + (lane >> 4)*4 + r  -> my slab of p.ws
    float* slab = p.ws + (long)split * p.M * p.N;
    const bool vec_ok = (p.N & 3) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 32 + i * 16 + (lane & 15);
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + j * 16 + (lane >> 4) * 4;
            float* dst = slab + (long)row * p.N + col;
            if (vec_ok && col + 3 < p.N) {
                *reinterpret_cast<f32x4*>(dst) = acc[i][j];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col + r < p.N) dst[r] = acc[i][j][r];
            }
        }
    }
}

// Sum of one float4 per lane over the lanes that share (lane & 3): the 16 row groups of a wave in the 4-columns-per-thread
// layout of the epilogue kernels below
__device__ __forceinline__ f32x4 rows16_sum(f32x4 v) {
#pragma unroll
    for (int off = 4; off < 64; off <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], off, 64);
    return v;
}

// The MLP layer epilogue on split-K slabs (see gemm_bf16s_kernel<true> for the arithmetic - same two-pass statistics, same
// outputs): a workgroup owns EP_COLS columns and all M <= 128 rows; thread = (4 consecutive columns, row group rg of 64),
// rows rg and rg + 64.
constexpr int EP_COLS = 16;
__global__ __launch_bounds__(256) void linear_bn_act_epi_kernel(const float* __restrict__ ws, int splits, int M, int N,
                                                                 const float* __restrict__ bias, BnActArgs q) {
    __shared__ float red[2][4][EP_COLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = (tid & 3) * 4, rg = tid >> 2;
    const int col = blockIdx.x * EP_COLS + c4;
    const bool cok = col < N;  // (N is a multiple of 4: a thread's four columns are in or out together)
    const long slab = (long)M * N;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr && cok) bv = *reinterpret_cast<const f32x4*>(bias + col);
    f32x4 v[2];
    bool rok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = rg + 64 * k;
        rok[k] = cok && r < M;
        v[k] = bv;
    }
    // the slabs, eight loads in flight per row (a run-time trip count would make every slab a dependent round trip)
    for (int s0 = 0; s0 < splits; s0 += 8) {
        f32x4 t[2][8];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = rok[k] ? rg + 64 * k : 0;
            const float* src = ws + (long)r * N + (cok ? col : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int sj = s0 + j < splits ? s0 + j : splits - 1;
                t[k][j] = *reinterpret_cast<const f32x4*>(src + sj * slab);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] += (s0 + j < splits) ? t[k][j][e] : 0.f;
    }
    auto column_total = [&](f32x4 t, int pass) -> f32x4 {
        t = rows16_sum(t);
        if (lane < 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[pass][wave][c4 + e] = t[e];
        }
        __syncthreads();
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ((red[pass][0][c4 + e] + red[pass][1][c4 + e]) + red[pass][2][c4 + e]) + red[pass][3][c4 + e];
        return o;
    };
    const float invM = 1.0f / (float)M;
    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] += rok[k] ? v[k][e] : 0.f;
    f32x4 mean = column_total(t, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) mean[e] *= invM;
    t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[k][e] - mean[e];
            t[e] += rok[k] ? d * d : 0.f;
        }
    f32x4 var = column_total(t, 1);
    float sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        var[e] *= invM;
        const float inv = 1.0f / sqrtf(var[e] + q.eps);
        const float g = (q.gamma != nullptr && cok) ? q.gamma[col + e] : 1.f, b = (q.beta != nullptr && cok) ? q.beta[col + e] : 0.f;
        sc[e] = g * inv;
        sh[e] = b - mean[e] * sc[e];
    }
    if (rg == 0 && cok) {  // one thread per four columns: the statistics backward needs, the running statistics
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            q.mean[col + e] = mean[e];
            q.var[col + e] = var[e];
            if (q.rmean != nullptr) {
                q.rmean[col + e] = (1.f - q.momentum) * q.rmean[col + e] + q.momentum * mean[e];
                q.rvar[col + e] = (1.f - q.momentum) * q.rvar[col + e] + q.momentum * (var[e] * q.unbias);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (!rok[k]) continue;
        const int r = rg + 64 * k;
        const long o = (long)r * N + col;
        f32x4 av, yv;
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = pk_act(q.act, v[k][e] * sc[e] + sh[e]);
        *reinterpret_cast<f32x4*>(q.zout + o) = v[k];
        *reinterpret_cast<f32x4*>(q.aout + o) = av;
        yv = av;
        if (q.mask != nullptr) {
            const f32x4 mk = *reinterpret_cast<const f32x4*>(q.mask + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) yv[e] = av[e] * mk[e];
            *reinterpret_cast<f32x4*>(q.yout + o) = yv;
        }
        uint2 pk;
        pk.x = pk_pack_bf2(yv[0], yv[1]);
        pk.y = pk_pack_bf2(yv[2], yv[3]);
        *reinterpret_cast<uint2*>(q.yb + (long)r * q.ldyb + col) = pk;
    }
}

// C[M][N] = alpha * sum_k A[k][m] B[k][n] (+ beta C): both operands k-major, short reduction (the batch of an MLP step):
// one 64 x 64 output tile per workgroup (2 x 2 waves of 32 x 32), the operands as [64 k][32 columns] images (stage_km32 /
// frag_km32), up to T64_TILES k-tiles in flight before the first wait.
constexpr int T64_TILES = 2, T64_STAGE = 4 * 4096;  // per k-tile: A columns 0-31 | 32-63 | B columns 0-31 | 32-63
__global__ __launch_bounds__(256) void gemm_bf16_t64_kernel(BArgs p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // [T64_TILES][T64_STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int nk = (p.K + TK - 1) / TK;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int kt, unsigned char* buf) {
        const int k0 = kt * TK;
        stage_km32(p.A, p.lda, m0, p.M, k0, p.K, p.zeros, buf, tid, wave);
        stage_km32(p.A, p.lda, m0 + 32 < p.M ? m0 + 32 : m0, p.M, k0, p.K, p.zeros, buf + 4096, tid, wave);
        stage_km32(p.B, p.ldb, n0, p.N, k0, p.K, p.zeros, buf + 8192, tid, wave);
        stage_km32(p.B, p.ldb, n0 + 32 < p.N ? n0 + 32 : n0, p.N, k0, p.K, p.zeros, buf + 12288, tid, wave);
    };
    for (int base = 0; base < nk; base += T64_TILES) {
#pragma unroll
        for (int t = 0; t < T64_TILES; ++t)
            if (base + t < nk) fetch(base + t, smem + t * T64_STAGE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T64_TILES; ++t) {
            if (base + t >= nk) break;
            const unsigned char* cur = smem + t * T64_STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = frag_km32(cur + wm * 4096, i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = frag_km32(cur + 8192 + wn * 4096, j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();  // everyone is done reading before the next group of tiles lands
    }
    // acc[i][j][r]: row = m0 + wm*32 + i*16 + (lane & 15), column = n0 + wn*32 + j*16 + (lane >> 4)*4 + r
    const bool vec_ok = (p.ldc & 3) == 0 && (((uintptr_t)p.C) & 15) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + wm * 32 + i * 16 + (lane & 15);
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 32 + j * 16 + (lane >> 4) * 4;
            float* dst = p.C + (long)row * p.ldc + col;
            if (vec_ok && col + 3 < p.N) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = p.alpha * acc[i][j][r] + (p.bias ? p.bias[col + r] : 0.f);
                if (p.beta != 0.f) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] += p.beta * c[r];
                }
                *reinterpret_cast<f32x4*>(dst) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (col + r >= p.N) continue;
                    float o = p.alpha * acc[i][j][r] + (p.bias ? p.bias[col + r] : 0.f);
                    if (p.beta != 0.f) o += p.beta * dst[r];
                    dst[r] = o;
                }
            }
        }
    }
}

// ============================================================================
// 256 x 256 x 64 block tile, 8 waves, eight phases per pair of k-tiles.
//
// The 128 x 128 kernels above top out at 640-680 TFLOP/s: every k-tile is [stage, wait for ALL loads, barrier, 32
// MFMAs, barrier] and the waves of a workgroup idle together while the tile lands (PMC: matrix pipe 20-24 % busy,
// 55-65 % of wave cycles in s_waitcnt).  This kernel removes the drain:
//   * an operand k-tile is two PIECES of 128 rows x 64 k (16 KB - exactly the tile image of the kernels above, so the
//     staging / swizzle / fragment code is shared): A-h0, A-h1, B-h0, B-h1;
//   * a wave (2 x 4 grid: wm, wn) owns 64 rows of EACH A piece and 32 columns of EACH B piece, i.e. a 128 x 64 slice of
//     the output in four quadrants - so every piece is read by all waves in exactly one phase of a k-tile:
//       phase 1: read A-h0, B-h0  -> 16 MFMAs (A0 x B0)     phase 2: read B-h1 -> 16 MFMAs (A0 x B1)
//       phase 3: read A-h1        -> 16 MFMAs (A1 x B1)     phase 4: (registers) -> 16 MFMAs (A1 x B0)
//   * LDS holds two k-tiles = 8 piece slots (128 KB), filled by the LDS-DMA; a slot is refilled in the phase after the
//     one that read it, for the k-tile TWO ahead, so every wave keeps five pieces (10 loads per lane) in flight behind
//     the one it waits for, and the only wait in the loop is the counted `s_waitcnt vmcnt(10)` - never vmcnt(0);
//   * a phase is [L: issue one piece, read fragments] barrier [M: 16 MFMAs] barrier, and the two waves that share a SIMD
//     (w and w + 4: wm = 0 / 1) run ONE BARRIER = half a phase apart (the wm = 1 group takes one extra barrier before the
//     loop, the wm = 0 group one after it): while one wave of a SIMD is in its MFMA block its partner is in its
//     LDS-read / DMA-issue part instead of both wanting the matrix pipe at the same moment; s_setprio(1) around the
//     MFMA block.  A piece is waited for in the L part of the phase BEFORE the one that reads it, and fragment reads
//     are complete (lgkmcnt(0)) before the L part's barrier - that is what makes the half-phase offset safe in both
//     directions (the group ahead finds the other group's share of a piece landed; the group behind has finished
//     reading a slot before the group ahead refills it);
// Out-of-range quadrants (N = 1100 leaves the last column of tiles 76 of 256 columns) skip their MFMAs.
// ============================================================================
constexpr int PIECE = 16384;
template <int V>
struct BoolK {
    static constexpr int value = V;
};

// Fragment reads of the eight-phase kernel.  k-contiguous pieces: the compiler-visible ds_read_b128 of frag<true>.  k-major
// pieces: the LDS transpose read issued from INLINE ASM - hipcc puts an s_waitcnt vmcnt(0) in front of the
// __builtin_amdgcn_ds_read_tr16_b64 intrinsic whenever an LDS-DMA is in flight (it does not for plain LDS loads), which
// would drain the five-piece prefetch every phase.  The asm form is invisible to that logic; the caller waits with
// frag_wait() (lgkmcnt(0) + a scheduling barrier, so that no MFMA is hoisted above the wait) before the first use.
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
template <bool KC>
__device__ __forceinline__ bf16x8 frag8(const unsigned char* buf, int sub, int kk, int lane) {
    if (KC) {
        return frag<true>(buf, sub, kk, lane);
    } else {
        const int g = lane >> 4, i = lane & 15;
        const int col = sub + (i & 3) * 4;
        u32x2v h[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int kr = kk * 32 + g * 8 + hh * 4 + (i >> 2);
            const int ps = (col >> 3) ^ swz_km(kr);
            const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)(buf + kr * 256 + ps * 16 + (col & 7) * 2);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h[hh]) : "v"(addr) : "memory");
        }
        typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
        const u32x4v o = u32x4v{h[0][0], h[0][1], h[1][0], h[1][1]};
        return __builtin_bit_cast(bf16x8, o);
    }
}
template <bool ANY_ASM>
__device__ __forceinline__ void frag_wait() {
    if (ANY_ASM) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool KC>
struct PieceSrc {  // two 16-byte LDS-DMA loads per lane and piece; addresses advance by a constant per k-tile
    TileSrc<KC, 512> t;
    const unsigned short* base;
    long ld;
    int r0, rmax;
    __device__ __forceinline__ void init(const unsigned short* b, long l, int r0_, int rmax_, int k0, int tid) {
        base = b; ld = l; r0 = r0_; rmax = rmax_;
        t.init(b, l, r0_, rmax_, k0, tid);
    }
    // k-tile starting at k0 into `slot`.  CHECKED = false: the k-tile lies entirely below kend (the steady state: two
    // loads from the running addresses); CHECKED = true: ragged or past the end - element-checked addresses, the zero
    // page for k >= kend - so that the number of loads per phase is the same for every k-tile and the counted waits hold
    template <bool CHECKED>
    __device__ __forceinline__ void issue(int k0, int kend, const unsigned short* zeros, unsigned char* slot, int tid, int wave) {
        if (!CHECKED || k0 + TK <= kend) {
            t.issue(slot, wave);
        } else {
            stage<KC, 512>(base, ld, r0, rmax, k0, kend, zeros, slot, tid, wave);
#pragma unroll
            for (int j = 0; j < 2; ++j) t.a[j] += t.step;
        }
    }
};

__device__ __forceinline__ void store_frag(const BArgs& p, const f32x4 v, int row, int col, int split, bool vec_ok) {
    if (row >= p.M || col >= p.N) return;
    float* dst = p.ws ? p.ws + ((long)split * p.M + row) * p.N + col : p.C + (long)row * p.ldc + col;
    if (vec_ok && col + 3 < p.N) {
        f32x4 o = v;
        if (p.ws == nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = p.alpha * v[r] + (p.bias ? p.bias[col + r] : 0.f);
            if (p.beta != 0.f) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += p.beta * c[r];
            }
        }
        *reinterpret_cast<f32x4*>(dst) = o;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (col + r >= p.N) continue;
            float o = v[r];
            if (p.ws == nullptr) {
                o = p.alpha * o + (p.bias ? p.bias[col + r] : 0.f);
                if (p.beta != 0.f) o += p.beta * dst[r];
            }
            dst[r] = o;
        }
    }
}

// Column statistics of one 256 x 256 output tile, taken from the accumulators before they are stored (BatchNorm of a
// projection: neural_networks.py:1114-1124 normalises w*(x) over all T*B rows; the separate statistics pass re-read
// the 282 MB projection).  Two passes over the registers - sum, then squared deviations from the tile's own mean - so
// nothing cancels; lanes that share a column (the 16 rows of a fragment) fold with xor-shuffles, the two row halves
// of the workgroup through LDS.  Output: (rows, mean, M2) per column, the partial format of pk_bn_stats' merge.
__device__ __forceinline__ void tile_colstats(const BArgs& p, const f32x4 (&acc)[2][2][4][2], int m0, int n0, int tm, int wm,
                                              int wn, int lane, float* sh) {
    const int nrows = (p.M - m0) < 256 ? (p.M - m0) : 256;
    __syncthreads();  // every wave is past its last fragment read: the staging buffers are free
    float part[2][2][4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float mu[2][2][4];
        if (pass == 1) {
#pragma unroll
            for (int bh = 0; bh < 2; ++bh)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cl = bh * 128 + wn * 32 + j * 16 + (lane >> 4) * 4 + r;
                        mu[bh][j][r] = (sh[cl] + sh[256 + cl]) / (float)nrows;
                    }
        }
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = 0.f;
#pragma unroll
                    for (int ah = 0; ah < 2; ++ah)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool in = ah * 128 + wm * 64 + i * 16 + (lane & 15) < nrows;
                            const float v = acc[ah][bh][i][j][r];
                            const float d = pass == 0 ? v : v - mu[bh][j][r];
                            t += in ? (pass == 0 ? d : d * d) : 0.f;
                        }
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) t += __shfl_xor(t, off);
                    part[bh][j][r] = t;
                }
        if ((lane & 15) == 0) {
#pragma unroll
            for (int bh = 0; bh < 2; ++bh)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        sh[pass * 512 + wm * 256 + bh * 128 + wn * 32 + j * 16 + (lane >> 4) * 4 + r] = part[bh][j][r];
        }
        __syncthreads();
    }
    // 256 columns, 512 threads: thread c < 256 writes column c
    const int c = threadIdx.x;
    if (c < 256 && n0 + c < p.N) {
        const float mean_v = (sh[c] + sh[256 + c]) / (float)nrows;
        const float m2_v = sh[512 + c] + sh[768 + c];
        float* o = p.stats + ((long)tm * p.N + n0 + c) * 3;
        o[0] = (float)nrows;
        o[1] = p.alpha * mean_v + (p.bias ? p.bias[n0 + c] : 0.f);
        o[2] = p.alpha * p.alpha * m2_v;
    }
}

#define PK_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 1) void gemm_bf16_256_kernel(BArgs p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // [2 k-tiles][A-h0 | B-h0 | B-h1 | A-h1] x 16 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);  // XCD-aware mapping, as above
    if (item >= p.items) return;
    const int tn = item % p.tiles_n;
    const int tm = (item / p.tiles_n) % p.tiles_m;
    const int split = item / (p.tiles_n * p.tiles_m);
    const int m0 = tm * 256, n0 = tn * 256;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + TK - 1) / TK;
    const bool a1_live = m0 + 128 < p.M, b1_live = n0 + 128 < p.N;  // uniform: second halves entirely out of range?

    f32x4 acc[2][2][4][2];  // [A half][B half][i][j]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    PieceSrc<A_KC> sa0, sa1;
    PieceSrc<B_KC> sb0, sb1;
    sa0.init(p.A, p.lda, m0, p.M, kbeg, tid);
    sa1.init(p.A, p.lda, m0 + 128, p.M, kbeg, tid);
    sb0.init(p.B, p.ldb, n0, p.N, kbeg, tid);
    sb1.init(p.B, p.ldb, n0 + 128, p.N, kbeg, tid);
    auto slot = [&](int kt, int piece) { return smem + (((kt & 1) * 4 + piece) * PIECE); };
    // piece ids in slot order: 0 = A-h0, 1 = B-h0, 2 = B-h1, 3 = A-h1
    auto issueA0 = [&](int kt, auto C) { sa0.template issue<decltype(C)::value != 0>(kbeg + kt * TK, kend, p.zeros, slot(kt, 0), tid, wave); };
    auto issueB0 = [&](int kt, auto C) { sb0.template issue<decltype(C)::value != 0>(kbeg + kt * TK, kend, p.zeros, slot(kt, 1), tid, wave); };
    auto issueB1 = [&](int kt, auto C) { sb1.template issue<decltype(C)::value != 0>(kbeg + kt * TK, kend, p.zeros, slot(kt, 2), tid, wave); };
    auto issueA1 = [&](int kt, auto C) { sa1.template issue<decltype(C)::value != 0>(kbeg + kt * TK, kend, p.zeros, slot(kt, 3), tid, wave); };

    // prologue: k-tile 0 complete, k-tile 1 without its A-h1 (issued in phase 1 of k-tile 0)
    issueA0(0, BoolK<1>()); issueB0(0, BoolK<1>()); issueB1(0, BoolK<1>()); issueA1(0, BoolK<1>());
    issueA0(1, BoolK<1>()); issueB0(1, BoolK<1>()); issueB1(1, BoolK<1>());
    PK_VMCNT(10);  // A-h0, B-h0 of k-tile 0 (my share)
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // the second wave of every SIMD runs one barrier (half a phase) behind

    bf16x8 a[4][2], b[2][2][2];  // a: [i][kk] of the A half in use; b: [half][j][kk] (B0 lives until phase 4)
#define PK_MFMA_BLOCK(AH, BH)                                                                                         \
    do {                                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int i = 0; i < 4; ++i)                   \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
                acc[AH][BH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[BH][j][kk], a[i][kk], acc[AH][BH][i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                \
    } while (0)
    // end of an L part: fragment reads complete (also what lets the slot be refilled next phase), then the barrier
#define PK_END_L()                                         \
    do {                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                 \
        __builtin_amdgcn_s_barrier();                      \
    } while (0)
    // One k-tile = four phases.  FQ: every quadrant of the tile is in range (no conditional MFMA blocks: the common case
    // is straight-line code); CK: the loads issued here may be ragged / past the end (the last three k-tiles of a split).
    // Reads:   L1: A-h0, B-h0     L2: B-h1          L3: A-h1          L4: -
    // Refills: L1: A-h1(kt+1)     L2: A-h0(kt+2)    L3: B-h0(kt+2)    L4: B-h1(kt+2)     (the phase after the slot's read)
    // Waits:   L1: B-h1(kt)       L2: A-h1(kt)      L3: -             L4: A-h0, B-h0(kt+1)   (one phase before the read)
    auto ktile = [&](int kt, auto FQC, auto CKC) {
        constexpr bool FQ = decltype(FQC)::value != 0;
        // ---------------- phase 1: A0 x B0
        issueA1(kt + 1, CKC);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < 2; ++j) b[0][j][kk] = frag8<B_KC>(slot(kt, 1), wn * 32 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i][kk] = frag8<A_KC>(slot(kt, 0), wm * 64 + i * 16, kk, lane);
        }
        PK_VMCNT(10);
        PK_END_L();
        PK_MFMA_BLOCK(0, 0);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 2: A0 x B1
        issueA0(kt + 2, CKC);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) b[1][j][kk] = frag8<B_KC>(slot(kt, 2), wn * 32 + j * 16, kk, lane);
        PK_VMCNT(10);
        PK_END_L();
        if (FQ || b1_live) PK_MFMA_BLOCK(0, 1);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 3: A1 x B1
        issueB0(kt + 2, CKC);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i][kk] = frag8<A_KC>(slot(kt, 3), wm * 64 + i * 16, kk, lane);
        PK_END_L();
        if (FQ || (a1_live && b1_live)) PK_MFMA_BLOCK(1, 1);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 4: A1 x B0 (both from registers)
        issueB1(kt + 2, CKC);
        PK_VMCNT(10);
        __builtin_amdgcn_s_barrier();
        if (FQ || a1_live) PK_MFMA_BLOCK(1, 0);
        __builtin_amdgcn_s_barrier();
    };
    const int nk_full = (kend - kbeg) / TK;  // k-tiles that lie entirely below kend
    auto run = [&](auto FQC) {
        int kt = 0;
        for (; kt + 2 < nk_full; ++kt) ktile(kt, FQC, BoolK<0>());  // everything issued here (up to k-tile kt + 2) is full
        for (; kt < nk; ++kt) ktile(kt, FQC, BoolK<1>());
    };
    if (a1_live && b1_live) run(BoolK<1>());
    else run(BoolK<0>());
#undef PK_MFMA_BLOCK
#undef PK_END_L
    if (wm == 0) __builtin_amdgcn_s_barrier();  // pairs with the extra barrier the other group took before the loop
    PK_VMCNT(0);  // the zero-page DMAs of the two k-tiles past the end
    if constexpr (A_KC && B_KC) {  // (projections: both operands k-contiguous; the other instantiations stay as they were)
        if (p.stats != nullptr) tile_colstats(p, acc, m0, n0, tm, wm, wn, lane, reinterpret_cast<float*>(smem));
    }
    const bool vec_ok = (((uintptr_t)(p.ws ? p.ws : p.C) & 15) == 0) && (((p.ws ? (long)p.N : p.ldc) & 3) == 0);
    const bool interior = m0 + 256 <= p.M && n0 + 256 <= p.N;
    if (vec_ok && interior && (p.ws != nullptr || p.beta == 0.f)) {
        // the common case, straight-line: no bounds, no read-modify-write; 16 bytes per lane and store
        const bool direct = p.ws == nullptr;
        float* base = direct ? p.C : p.ws + (long)split * p.M * p.N;
        const long ld = direct ? p.ldc : (long)p.N;
        const float alpha = direct ? p.alpha : 1.f;
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + bh * 128 + wn * 32 + j * 16 + (lane >> 4) * 4;
                f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
                if (direct && p.bias != nullptr) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = p.bias[col + r];
                }
#pragma unroll
                for (int ah = 0; ah < 2; ++ah)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = m0 + ah * 128 + wm * 64 + i * 16 + (lane & 15);
                        const f32x4 v = acc[ah][bh][i][j];
                        f32x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = alpha * v[r] + bv[r];
                        *reinterpret_cast<f32x4*>(base + (long)row * ld + col) = o;
                    }
            }
        return;
    }
    // (fully unrolled: a run-time index into the accumulator array would move the whole array to scratch memory)
#pragma unroll
    for (int ah = 0; ah < 2; ++ah)
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    store_frag(p, acc[ah][bh][i][j], m0 + ah * 128 + wm * 64 + i * 16 + (lane & 15),
                               n0 + bh * 128 + wn * 32 + j * 16 + (lane >> 4) * 4, split, vec_ok);
}

__global__ void splitk_reduce_bf_kernel(const float* __restrict__ ws, int splitk, int M, int N, float alpha, float beta,
                                        const float* __restrict__ bias, float* __restrict__ C, long ldc) {
    const long total = (long)M * N;
    if ((N & 3) == 0 && (ldc & 3) == 0 && (((uintptr_t)C | (uintptr_t)ws) & 15) == 0) {
        // four columns per thread, eight slabs in flight (with a run-time trip count over single floats every slab was a
        // dependent round trip: 3.9 us for 8 slabs of 128 x 1024)
        const long quads = total >> 2;
        for (long qd = blockIdx.x * (long)blockDim.x + threadIdx.x; qd < quads; qd += (long)gridDim.x * blockDim.x) {
            const long i = qd << 2;
            const int row = (int)(i / N), col = (int)(i - (long)row * N);
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < splitk; k0 += 8) {
                f32x4 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const f32x4*>(ws + (long)(k0 + j < splitk ? k0 + j : splitk - 1) * total + i);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += (k0 + j < splitk) ? t[j][e] : 0.f;
            }
            float* c = C + (long)row * ldc + col;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = alpha * s[e] + (bias ? bias[col + e] : 0.f);
            if (beta != 0.f) {
                const f32x4 cv = *reinterpret_cast<const f32x4*>(c);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += beta * cv[e];
            }
            *reinterpret_cast<f32x4*>(c) = o;
        }
        return;
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / N), col = (int)(i % N);
        float s = 0.f;
        for (int k = 0; k < splitk; ++k) s += ws[(long)k * total + i];
        float o = alpha * s + (bias ? bias[col] : 0.f);
        float* c = C + (long)row * ldc + col;
        if (beta != 0.f) o += beta * (*c);
        *c = o;
    }
}

// fp32 [rows][lds] -> bf16 [rows][ldd]; source columns are `nseg` segments of `seglen`, each placed at
// a pitch of `segpad` in the destination; every other destination element of the row is zero.
__global__ void cvt_bf16_kernel(const float* __restrict__ src, long lds, long rows, int nseg, int seglen, int segpad,
                                unsigned short* __restrict__ dst, long ldd) {
    const long chunks_per_row = ldd >> 3;  // 8 bf16 = 16 B per thread
    const long total = rows * chunks_per_row;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / chunks_per_row;
        const int c0 = (int)(i - r * chunks_per_row) * 8;
        const float* s = src + r * lds;
        unsigned pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = c0 + 2 * e + h;
                const int sg = c / segpad, off = c - sg * segpad;
                v[h] = (sg < nseg && off < seglen) ? s[sg * seglen + off] : 0.f;
            }
            pk[e] = pk_pack_bf2(v[0], v[1]);
        }
        *reinterpret_cast<uint4*>(dst + r * ldd + c0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
}

}  // namespace

extern "C" int pk_cvt_bf16(void* stream, const float* src, int64_t ld_src, int64_t rows, int nseg, int seglen, int segpad,
                           uint16_t* dst, int64_t ld_dst) {
    if (rows <= 0) return 0;
    PK_REQUIRE(nseg >= 1 && seglen >= 1 && segpad >= seglen, "pk_cvt_bf16: bad segments");
    PK_REQUIRE((ld_dst % 8) == 0 && ld_dst >= (int64_t)nseg * segpad - (segpad - seglen) && ((uintptr_t)dst & 15) == 0,
               "pk_cvt_bf16: destination pitch must be a multiple of 8 elements and hold every segment");
    const long total = rows * (ld_dst >> 3);
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, pk_stream(stream), src, (long)ld_src,
                       (long)rows, nseg, seglen, segpad, (unsigned short*)dst, (long)ld_dst);
    PK_LAUNCH_CHECK();
    return 0;
}

// Which kernel a shape takes.  Measured on MI355X (tools/bench_gemm.py, profiles/r02_gemm_tiles.txt): both structures are
// held at 0.65-0.8 PFLOP/s by what the L2 delivers to the LDS-DMA (PMC: matrix pipe 26 % busy, 58 % of wave cycles
// parked in s_waitcnt / s_barrier, no LDS bank conflicts); the 256 x 256 tile halves the bytes per FLOP and wins where
// the reduction is long and the output small - the split-K weight-gradient shapes dW (1100 x 1104 x 64000: 690 -> 770)
// and the senone head's dW (1938 x 1100 x 64000: 670 -> 830) - while the row-streaming shapes (M = 64000, K ~ 1100:
// 18 k-tiles per output tile, 282 MB of fp32 output) gain nothing from it (prologue / epilogue are not overlapped with
// one workgroup per CU) and the 550 x 550 dU shape loses (3 x 3 tiles of which 28 % is padding).
// PK_EXPERIMENT gemm_tile=128|256 / pk_gemm_bf16_set_tile() force one of them.
static int g_gemm_tile_forced = -1;
extern "C" void pk_gemm_bf16_set_tile(int tile) { g_gemm_tile_forced = (tile == 128 || tile == 256) ? tile : 0; }
static int gemm_tile_for(int M, int N, int a_kc, int b_kc, int K = 1 << 30) {
    int& forced = g_gemm_tile_forced;
    if (forced < 0) {
        const char* e = pk_experiment("gemm_tile");
        forced = (e && atoi(e) == 128) ? 128 : (e && atoi(e) == 256) ? 256 : 0;
    }
    if (forced) return forced;
    // row-streaming shapes (projection, dX, output layers: M = T*B rows, k-contiguous A, N >= 1024): the 256-tile is worth
    // 3-10 % on the current kernel (652 vs 603, 775 vs 695 TFLOP/s: profiles/r02_gemm_tiles.txt) and 0.22 ms of the
    // 21.1 ms training step (A/B on one box, tools/gpu_ab3.sh); PK_EXPERIMENT gemm_tile_rows=0 keeps them on the 128-tile
    static int rows256 = -1;
    if (rows256 < 0) {
        const char* e = pk_experiment("gemm_tile_rows");
        rows256 = (e && atoi(e) == 0) ? 0 : 1;
    }
    if (rows256 && a_kc && M >= 16384 && N >= 1024) return 256;
    if (!a_kc && !b_kc && M >= 1024 && N >= 1024) {
        // a weight gradient over a short reduction (an MLP layer at batch 128: 1024 x 1024 over K = 128) has no split-K to
        // fill the chip with: 16 workgroups of 256-tiles each read-modify-write 256 KB of the fp32 gradient on their own
        // (17 us measured); the 128-tile puts the same traffic on 64 CUs
        const long tiles256 = (long)((M + 255) / 256) * ((N + 255) / 256);
        if (K < 2048 && tiles256 < 64) return 128;
        return 256;
    }
    return 128;
}

// kept for callers that sized split-K from the tile height (rows of the block tile of a k-contiguous shape)
extern "C" int pk_gemm_bf16_tile_m(int M) {
    (void)M;
    return TM;
}

// Split-K factor the library recommends for a k-major x k-major product C[M,N] = A^T.B over K (the dW / dU shapes: few
// output tiles, K = T*B rows): the reduction is cut so that the grid covers the chip about twice (128-tiles, several
// workgroups per CU) or once (256-tiles, one workgroup per CU), never below 512 k per slice.
// cus: the CUs the product will actually get (a weight-gradient GEMM on the side stream runs next to a persistent
// recurrence that holds 144 of the 256 CUs: sized for the whole chip its 250 items took three rounds on the 112 free ones)
extern "C" int pk_gemm_bf16_auto_splitk_cus(int M, int N, int K, int cus) {
    if (K < 2048 || M <= 0 || N <= 0) return 1;
    const int t = gemm_tile_for(M, N, 0, 0);
    const long tiles = (long)((M + t - 1) / t) * ((N + t - 1) / t);
    long ncu = pk_num_cu();
    if (cus > 0 && cus < ncu) ncu = cus;
    long s = (t == 256 ? ncu : 2 * ncu) / (tiles > 0 ? tiles : 1);
    if (s < 1) s = 1;
    if (s > 32) s = 32;
    const long kmax = K / 512 > 0 ? K / 512 : 1;
    if (s > kmax) s = kmax;
    return (int)s;
}
extern "C" int pk_gemm_bf16_auto_splitk(int M, int N, int K) { return pk_gemm_bf16_auto_splitk_cus(M, N, K, 0); }

static int gemm_bf16_impl(void* stream, int M, int N, int K, float alpha, const uint16_t* A, int64_t lda, int a_kc,
                          const uint16_t* B, int64_t ldb, int b_kc, float beta, float* C, int64_t ldc, const float* bias,
                          int splitk, float* workspace, float* stats) {
    if (M <= 0 || N <= 0) return 0;
    PK_REQUIRE(K >= 0, "pk_gemm_bf16: negative K");
    PK_REQUIRE((lda % 8) == 0 && (ldb % 8) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0,
               "pk_gemm_bf16: operands need 16-byte aligned bases and pitches that are multiples of 8 elements");
    PK_REQUIRE(!a_kc || (K % 8) == 0 || lda >= ((K + 7) & ~7), "pk_gemm_bf16: A pitch shorter than K rounded up to 8");
    PK_REQUIRE(!b_kc || (K % 8) == 0 || ldb >= ((K + 7) & ~7), "pk_gemm_bf16: B pitch shorter than K rounded up to 8");
    PK_REQUIRE(a_kc || lda >= ((M + 7) & ~7), "pk_gemm_bf16: k-major A needs a pitch of at least M rounded up to 8");
    PK_REQUIRE(b_kc || ldb >= ((N + 7) & ~7), "pk_gemm_bf16: k-major B needs a pitch of at least N rounded up to 8");
    hipStream_t st = pk_stream(stream);
    BArgs p;
    p.M = M; p.N = N; p.K = K;
    p.alpha = alpha; p.beta = beta;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb;
    p.C = C; p.ldc = ldc; p.bias = bias;
    p.stats = stats;
    const int tile = gemm_tile_for(M, N, a_kc, b_kc, K);
    PK_REQUIRE(stats == nullptr || (tile == 256 && a_kc && b_kc && splitk <= 1 && beta == 0.f), "pk_gemm_bf16_stats: internal: shape not covered");
    p.tiles_m = (M + tile - 1) / tile;
    p.tiles_n = (N + tile - 1) / tile;
    static void* zp = nullptr;  // looked up once (also keeps the call out of a HIP-graph capture)
    if (zp == nullptr) PK_CHECK_HIP(hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_page)));
    p.zeros = (const unsigned short*)zp;
    if (splitk < 1) splitk = 1;
    if (splitk > 1) {
        PK_REQUIRE(workspace != nullptr, "pk_gemm_bf16: split-K needs a workspace");
        int kps = (K + splitk - 1) / splitk;
        kps = ((kps + TK - 1) / TK) * TK;
        splitk = kps > 0 ? (K + kps - 1) / kps : 1;
        p.k_per_split = kps;
    }
    if (splitk <= 1) {
        splitk = 1;
        p.k_per_split = ((K + TK - 1) / TK) * TK;
        if (p.k_per_split == 0) p.k_per_split = TK;
    }
    p.ws = splitk > 1 ? workspace : nullptr;
    p.items = splitk * p.tiles_m * p.tiles_n;
    p.per_xcd = (p.items + 7) / 8;
    {   // small-batch products: all rows of 32 columns per workgroup (gemm_bf16s_kernel); PK_EXPERIMENT gemm_skinny=0 keeps the 128-tile
        static int skinny_on = -1;
        if (skinny_on < 0) {
            const char* e = pk_experiment("gemm_skinny");
            skinny_on = (e && e[0] == '0') ? 0 : 1;
        }
        if (skinny_on && !g_gemm_tile_forced && a_kc && M <= TM && splitk > 1 && p.k_per_split <= SK_TILES * TK && stats == nullptr) {
            // the reduction split over the grid (round 4): slabs, then the existing reduce
            static bool attr_done = false;
            if (!attr_done) {
                PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16sk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_TILES * SSTAGE));
                PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16sk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_TILES * SSTAGE));
                attr_done = true;
            }
            const dim3 sgrid((unsigned)((N + SN - 1) / SN), (unsigned)splitk);
            if (b_kc) hipLaunchKernelGGL((gemm_bf16sk_kernel<true>), sgrid, dim3(256), SK_TILES * SSTAGE, st, p);
            else hipLaunchKernelGGL((gemm_bf16sk_kernel<false>), sgrid, dim3(256), SK_TILES * SSTAGE, st, p);
            PK_LAUNCH_CHECK();
            const long total = (long)M * N;
            int blocks = (int)((total + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_bf_kernel, dim3(blocks), dim3(256), 0, st, workspace, splitk, M, N, alpha, beta,
                               bias, C, (long)ldc);
            PK_LAUNCH_CHECK();
            return 0;
        }
        if (skinny_on && !g_gemm_tile_forced && !a_kc && !b_kc && splitk == 1 && K <= 512 && stats == nullptr &&
            (long)((M + TM - 1) / TM) * ((N + TN - 1) / TN) <= 192 && M >= 64 && N >= 64) {
            // a weight gradient over a short reduction: 64 x 64 tiles put the same read-modify-write on four times the CUs
            const dim3 tgrid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
            hipLaunchKernelGGL(gemm_bf16_t64_kernel, tgrid, dim3(256), T64_TILES * T64_STAGE, st, p);
            PK_LAUNCH_CHECK();
            return 0;
        }
        if (skinny_on && !g_gemm_tile_forced && a_kc && M <= TM && N >= 4 * SN && splitk == 1 && stats == nullptr) {
            BnActArgs q = {};
            const dim3 sgrid((unsigned)((N + SN - 1) / SN));
            if (b_kc) hipLaunchKernelGGL((gemm_bf16s_kernel<false, true>), sgrid, dim3(256), 2 * SSTAGE, st, p, q);
            else hipLaunchKernelGGL((gemm_bf16s_kernel<false, false>), sgrid, dim3(256), 2 * SSTAGE, st, p, q);
            PK_LAUNCH_CHECK();
            return 0;
        }
    }
    if (tile == 256) {
        static bool attr_done = false;
        if (!attr_done) {
            PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PIECE));
            PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PIECE));
            PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PIECE));
            PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PIECE));
            attr_done = true;
        }
        dim3 grid256((unsigned)(p.per_xcd * 8)), block512(512);
        const size_t lds256 = 8 * PIECE;
        if (a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16_256_kernel<true, true>), grid256, block512, lds256, st, p);
        else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_bf16_256_kernel<true, false>), grid256, block512, lds256, st, p);
        else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16_256_kernel<false, true>), grid256, block512, lds256, st, p);
        else hipLaunchKernelGGL((gemm_bf16_256_kernel<false, false>), grid256, block512, lds256, st, p);
        PK_LAUNCH_CHECK();
        if (splitk > 1) {
            const long total = (long)M * N;
            int blocks = (int)((total + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_bf_kernel, dim3(blocks), dim3(256), 0, st, workspace, splitk, M, N, alpha, beta,
                               bias, C, (long)ldc);
            PK_LAUNCH_CHECK();
        }
        return 0;
    }
    dim3 grid((unsigned)(p.per_xcd * 8)), block(256);
    // measured on MI355X (tools/bench_gemm.py): the single-buffer variant with four workgroups per CU wins on the
    // row-streaming shapes (A k-contiguous: 640-680 vs 530-540 TFLOP/s at M = 64000), the double-buffered one on
    // the split-K k-major shapes (687 vs 661).  PK_EXPERIMENT gemm_stages=1|2 forces one of them.
    static int forced = -1;
    if (forced < 0) {
        const char* e = pk_experiment("gemm_stages");
        forced = (e && e[0] == '1') ? 1 : (e && e[0] == '2') ? 2 : 0;
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16x_kernel<true, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16x_kernel<true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16x_kernel<false, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16x_kernel<false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    }
    const int stages = forced ? forced : (a_kc ? 1 : 2);
    const size_t lds = stages == 2 ? 65536 : 32768;
#define PK_LAUNCH_BF(ST)                                                                                     \
    do {                                                                                                     \
        if (a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16x_kernel<true, true, ST>), grid, block, lds, st, p);    \
        else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_bf16x_kernel<true, false, ST>), grid, block, lds, st, p); \
        else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16x_kernel<false, true, ST>), grid, block, lds, st, p); \
        else hipLaunchKernelGGL((gemm_bf16x_kernel<false, false, ST>), grid, block, lds, st, p);              \
    } while (0)
    if (stages == 2) PK_LAUNCH_BF(2);
    else PK_LAUNCH_BF(1);
#undef PK_LAUNCH_BF
    PK_LAUNCH_CHECK();
    if (splitk > 1) {
        const long total = (long)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_bf_kernel, dim3(blocks), dim3(256), 0, st, workspace, splitk, M, N, alpha, beta,
                           bias, C, (long)ldc);
        PK_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int pk_gemm_bf16(void* stream, int M, int N, int K, float alpha, const uint16_t* A, int64_t lda, int a_kc,
                            const uint16_t* B, int64_t ldb, int b_kc, float beta, float* C, int64_t ldc,
                            const float* bias, int splitk, float* workspace) {
    return gemm_bf16_impl(stream, M, N, K, alpha, A, lda, a_kc, B, ldb, b_kc, beta, C, ldc, bias, splitk, workspace, nullptr);
}

// C = alpha * A.B + bias with the column statistics of C taken in the epilogue (shapes that run on the 256-tile; others:
// *row_blocks = 0 and the caller runs pk_bn_stats on C).  stats: pk_gemm_bf16_stats_floats(M, N) floats,
// [*row_blocks][N][3] = (rows, mean, M2) per 256-row tile and column - what pk_bn_stats_merge folds.
extern "C" int64_t pk_gemm_bf16_stats_floats(int M, int N) { return (int64_t)((M + 255) / 256) * N * 3; }

extern "C" int pk_gemm_bf16_stats(void* stream, int M, int N, int K, float alpha, const uint16_t* A, int64_t lda, int a_kc,
                                  const uint16_t* B, int64_t ldb, int b_kc, float* C, int64_t ldc, const float* bias,
                                  float* stats, int* row_blocks) {
    PK_REQUIRE(row_blocks != nullptr, "pk_gemm_bf16_stats: null row_blocks");
    const bool fused = stats != nullptr && M > 0 && N > 0 && K > 0 && a_kc && b_kc && gemm_tile_for(M, N, a_kc, b_kc) == 256;
    *row_blocks = fused ? (M + 255) / 256 : 0;
    return gemm_bf16_impl(stream, M, N, K, alpha, A, lda, a_kc, B, ldb, b_kc, 0.f, C, ldc, bias, 1, nullptr,
                          fused ? stats : nullptr);
}

// Split of the reduction for a small-batch product (M <= 128 rows, k-contiguous A): enough (column tile, split) workgroups
// to cover the chip, at most SK_TILES k-tiles per workgroup; 1 = the product is too small to be worth a second launch.
extern "C" int pk_gemm_bf16_small_splitk(int M, int N, int K) {
    if (M < 1 || M > TM || N < 1 || K < 2 * TK) return 1;
    const int col_tiles = (N + SN - 1) / SN;
    int s = 256 / col_tiles;               // workgroups ~ CUs
    const int need = (K + SK_TILES * TK - 1) / (SK_TILES * TK);  // the LDS holds SK_TILES k-tiles
    if (s < need) s = need;
    if (s > 16) s = 16;
    const int tiles = (K + TK - 1) / TK;
    if (s > tiles) s = tiles;
    if (s < 2) return 1;
    int kps = ((K + s - 1) / s + TK - 1) / TK * TK;  // what gemm_bf16_impl will use
    if (kps > SK_TILES * TK) return 1;
    return (K + kps - 1) / kps;
}

// One perf-mode MLP layer (see pk_linear_bn_act_bf16) in TWO launches that cover the chip: the product split along K into
// fp32 slabs ws[splitk][M][N] (splitk from pk_gemm_bf16_small_splitk, >= 2), then the layer epilogue straight from the slabs.
extern "C" int pk_linear_bn_act_bf16_sk(void* stream, int M, int N, int K, const uint16_t* xb, int64_t ldx, const uint16_t* wb,
                                        int64_t ldw, const float* bias, const float* gamma, const float* beta, float eps,
                                        float momentum, float* running_mean, float* running_var, int act, const float* mask,
                                        float* z, float* a, float* y, uint16_t* yb, int64_t ldyb, float* mean, float* var,
                                        int splitk, float* ws) {
    PK_REQUIRE(pk_linear_bn_act_bf16_covers(M, N, K), "pk_linear_bn_act_bf16_sk: needs 2 <= M <= 128 rows and N a multiple of 8 (got %d x %d)", M, N);
    PK_REQUIRE((ldx % 8) == 0 && (ldw % 8) == 0 && ((uintptr_t)xb & 15) == 0 && ((uintptr_t)wb & 15) == 0 && ldx >= ((K + 7) & ~7) &&
                   ldw >= ((K + 7) & ~7),
               "pk_linear_bn_act_bf16_sk: operands need 16-byte aligned bases and pitches that are multiples of 8 elements");
    PK_REQUIRE(z && a && yb && mean && var && (mask == nullptr || y != nullptr) && ldyb >= N && (ldyb % 4) == 0,
               "pk_linear_bn_act_bf16_sk: null output or bad pitch");
    PK_REQUIRE((((uintptr_t)z | (uintptr_t)a | (uintptr_t)y | (uintptr_t)mask | (uintptr_t)ws | (uintptr_t)bias) & 15) == 0 && ((uintptr_t)yb & 7) == 0,
               "pk_linear_bn_act_bf16_sk: fp32 matrices need 16-byte aligned bases");
    PK_REQUIRE(splitk >= 2 && ws != nullptr, "pk_linear_bn_act_bf16_sk: needs splitk >= 2 and a workspace of splitk x M x N floats");
    int kps = ((K + splitk - 1) / splitk + TK - 1) / TK * TK;
    PK_REQUIRE(kps <= SK_TILES * TK, "pk_linear_bn_act_bf16_sk: %d splits of K = %d exceed %d k-tiles per workgroup", splitk, K, SK_TILES);
    splitk = (K + kps - 1) / kps;
    hipStream_t st = pk_stream(stream);
    BArgs p;
    p.M = M; p.N = N; p.K = K;
    p.alpha = 1.f; p.beta = 0.f;
    p.A = xb; p.lda = ldx; p.B = wb; p.ldb = ldw;
    p.C = z; p.ldc = N; p.bias = nullptr;
    p.stats = nullptr; p.ws = ws;
    p.tiles_m = 1; p.tiles_n = (N + TN - 1) / TN;
    p.k_per_split = kps;
    p.items = p.tiles_n; p.per_xcd = (p.items + 7) / 8;
    static void* zp = nullptr;
    if (zp == nullptr) PK_CHECK_HIP(hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_page)));
    p.zeros = (const unsigned short*)zp;
    static bool attr_done = false;
    if (!attr_done) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16sk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SK_TILES * SSTAGE));
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_bf16sk_kernel<true>), dim3((unsigned)((N + SN - 1) / SN), (unsigned)splitk), dim3(256), SK_TILES * SSTAGE, st, p);
    PK_LAUNCH_CHECK();
    BnActArgs q;
    q.gamma = gamma; q.beta = beta; q.mask = mask;
    q.eps = eps; q.momentum = momentum; q.unbias = M > 1 ? (float)M / (float)(M - 1) : 1.f;
    q.rmean = running_mean; q.rvar = running_var; q.mean = mean; q.var = var;
    q.zout = z; q.aout = a; q.yout = y; q.yb = (unsigned short*)yb; q.ldyb = ldyb; q.act = act;
    hipLaunchKernelGGL(linear_bn_act_epi_kernel, dim3((unsigned)((N + EP_COLS - 1) / EP_COLS)), dim3(256), 0, st, ws, splitk, M, N, bias, q);
    PK_LAUNCH_CHECK();
    return 0;
}

// One perf-mode MLP layer of a batch of up to 128 rows in one launch: see gemm_bf16s_kernel<true>.
// xb [M][ldx] / wb [N][ldw] bf16 (k-contiguous), z / a / y fp32 [M][N] (y null = no mask: the layer output is a),
// yb bf16 [M][ldyb] (ldyb >= N, both multiples of 8), mean / var [N] (biased batch statistics, saved for backward).
// pk_linear_bn_act_bf16_covers says whether a shape takes this path.
extern "C" int pk_linear_bn_act_bf16_covers(int64_t M, int64_t N, int64_t K) {
    return M >= 2 && M <= TM && N >= 8 && (N % 8) == 0 && K >= 1;
}
extern "C" int pk_linear_bn_act_bf16(void* stream, int M, int N, int K, const uint16_t* xb, int64_t ldx, const uint16_t* wb,
                                     int64_t ldw, const float* bias, const float* gamma, const float* beta, float eps,
                                     float momentum, float* running_mean, float* running_var, int act, const float* mask,
                                     float* z, float* a, float* y, uint16_t* yb, int64_t ldyb, float* mean, float* var) {
    PK_REQUIRE(pk_linear_bn_act_bf16_covers(M, N, K), "pk_linear_bn_act_bf16: needs 2 <= M <= 128 rows and N a multiple of 8 (got %d x %d)", M, N);
    PK_REQUIRE((ldx % 8) == 0 && (ldw % 8) == 0 && ((uintptr_t)xb & 15) == 0 && ((uintptr_t)wb & 15) == 0 && ldx >= ((K + 7) & ~7) &&
                   ldw >= ((K + 7) & ~7),
               "pk_linear_bn_act_bf16: operands need 16-byte aligned bases and pitches that are multiples of 8 elements");
    PK_REQUIRE(z && a && yb && mean && var && (mask == nullptr || y != nullptr) && ldyb >= N && (ldyb % 4) == 0,
               "pk_linear_bn_act_bf16: null output or bad pitch");
    PK_REQUIRE((((uintptr_t)z | (uintptr_t)a | (uintptr_t)y | (uintptr_t)mask) & 15) == 0 && ((uintptr_t)yb & 7) == 0,
               "pk_linear_bn_act_bf16: fp32 matrices need 16-byte aligned bases");
    hipStream_t st = pk_stream(stream);
    BArgs p;
    p.M = M; p.N = N; p.K = K;
    p.alpha = 1.f; p.beta = 0.f;
    p.A = xb; p.lda = ldx; p.B = wb; p.ldb = ldw;
    p.C = z; p.ldc = N; p.bias = bias;
    p.stats = nullptr; p.ws = nullptr;
    p.tiles_m = 1; p.tiles_n = (N + TN - 1) / TN;
    p.k_per_split = ((K + TK - 1) / TK) * TK;
    p.items = p.tiles_n; p.per_xcd = (p.items + 7) / 8;
    static void* zp = nullptr;
    if (zp == nullptr) PK_CHECK_HIP(hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_page)));
    p.zeros = (const unsigned short*)zp;
    BnActArgs q;
    q.gamma = gamma; q.beta = beta; q.mask = mask;
    q.eps = eps; q.momentum = momentum; q.unbias = M > 1 ? (float)M / (float)(M - 1) : 1.f;
    q.rmean = running_mean; q.rvar = running_var; q.mean = mean; q.var = var;
    q.zout = z; q.aout = a; q.yout = y; q.yb = (unsigned short*)yb; q.ldyb = ldyb; q.act = act;
    hipLaunchKernelGGL((gemm_bf16s_kernel<true>), dim3((unsigned)((N + SN - 1) / SN)), dim3(256), 2 * SSTAGE, st, p, q);
    PK_LAUNCH_CHECK();
    return 0;
}
