// pk_gemm.hip - MFMA GEMMs for gfx950 (CDNA4), fp32 global operands.
//
// Replaces nn.Linear forward/backward of the reference hot path
// (neural_networks.py:111/139-148 MLP layers, :432-435 / :609-611 / :1114-1115
// input projections, per-step recurrent Linear) - see include/pk_amd.h.
//
//   C[M,N] = alpha * A(M,K) . B(K,N) + beta*C + bias
//
// Two operand precisions share the same staging skeleton:
//   PREC_F32  v_mfma_f32_32x32x2_f32   exact fp32 (bitwise an fmaf chain)  - parity mode
//   PREC_BF16 v_mfma_f32_32x32x16_bf16 operands rounded to bf16 while staging
//             fp32 tiles into LDS, fp32 accumulate                          - perf mode
//
// The exact-fp32 kernel has two forms with bit-identical results (every output element one fmaf chain over ascending
// k): the register-staged first form below (any strides / alignment) and the LDS-DMA form (gemm_f32_dma_kernel: aligned
// operands, 96-110 TFLOP/s of the pipe's 155 on the parity mode's shapes) - pk_gemm picks per call.
// Block tile 128x128, 4 waves (2x2), wave tile 64x64 = 2x2 MFMA 32x32 tiles.
// LDS layouts:  f32 : [BK=16][128+4] floats  (fragment reads = consecutive lanes, conflict free)
//               bf16: [128][BK=32+8] bf16    (fragment reads = ds_read_b128, odd 16-B row stride)
// Operands may be contiguous along k (KC) or along m/n; the loader vectorises
// along the contiguous direction and falls back to guarded scalar loads at
// ragged edges / unaligned leading dimensions.
#include "pk_common.h"

namespace {

struct GemmArgs {
    int M, N, K;
    float alpha, beta;
    const float* A;
    long a_rs, a_cs;
    const float* B;
    long b_rs, b_cs;
    float* C;
    long ldc;
    const float* bias;
    float* ws;      // split-K partials [splitk][M][N] or nullptr
    int k_per_split;  // multiple of BK
    int vecA, vecB;   // 16-B vector loads allowed for A / B
    int gx, gy, items, per_xcd;  // tiles along n / m, tiles x splits, items per XCD (1-D launch, see work_item)
};

constexpr int BM = 128, BN = 128;

// 1-D launch, XCD-aware: workgroup b runs on XCD b % 8, and each XCD has its own L2.  Items are numbered n-tile fastest,
// then m-tile, then split, and every XCD takes a CONTIGUOUS range of them - so the workgroups that share an operand tile
// (the n-tiles of one row block; all tiles of one split of a weight-gradient product, whose K range nobody else reads)
// share it through one L2 instead of fetching it once per XCD: the k-major x k-major split-K products were reading their
// operands up to nine times (2.3 ms for 155 GFLOP at 64 000 x 1100 x 1104).
__device__ __forceinline__ bool work_item(const GemmArgs& p, int& m0, int& n0, int& split) {
    const int item = p.per_xcd > 0 ? (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3) : (int)blockIdx.x;
    if (item >= p.items) return false;
    n0 = (item % p.gx) * BN;
    m0 = ((item / p.gx) % p.gy) * BM;
    split = item / (p.gx * p.gy);
    return true;
}

// ---------------------------------------------------------------------------
// global -> register staging of one operand tile (rows = m or n index, 128 of
// them; kdim = BK).  `KC` = contiguous along k.
//   element(r, k) = base[r * rs + k * cs]
// Each thread owns NV float4 "vectors" laid along the contiguous direction.
// ---------------------------------------------------------------------------
template <int BK, bool KC>
struct TileLoader {
    static constexpr int NV = (128 * BK) / (256 * 4);  // float4 per thread
    float4 v[NV];

    // vector index -> (row, k) of its first element
    __device__ __forceinline__ static void coord(int idx, int& r, int& k) {
        if (KC) {
            constexpr int QK = BK / 4;  // float4 per row
            r = idx / QK;
            k = (idx % QK) * 4;
        } else {
            r = (idx % 32) * 4;  // 32 float4 along the 128 rows
            k = idx / 32;
        }
    }

    __device__ __forceinline__ void load(const float* __restrict__ base, long rs, long cs, int r0, int k0, int rmax,
                                         int kmax, int vec_ok, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, k;
            coord(tid + 256 * i, r, k);
            const int gr = r0 + r, gk = k0 + k;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                if (gr < rmax) {
                    const float* p = base + (long)gr * rs + (long)gk * cs;
                    if (vec_ok && gk + 3 < kmax) {
                        t = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (gk + 0 < kmax) t.x = p[0];
                        if (gk + 1 < kmax) t.y = p[cs];
                        if (gk + 2 < kmax) t.z = p[2 * cs];
                        if (gk + 3 < kmax) t.w = p[3 * cs];
                    }
                }
            } else {
                if (gk < kmax) {
                    const float* p = base + (long)gr * rs + (long)gk * cs;
                    if (vec_ok && gr + 3 < rmax) {
                        t = *reinterpret_cast<const float4*>(p);
                    } else {
                        if (gr + 0 < rmax) t.x = p[0];
                        if (gr + 1 < rmax) t.y = p[rs];
                        if (gr + 2 < rmax) t.z = p[2 * rs];
                        if (gr + 3 < rmax) t.w = p[3 * rs];
                    }
                }
            }
            v[i] = t;
        }
    }

    // f32 LDS image: tile[k][row], leading dimension LD floats
    template <int LD>
    __device__ __forceinline__ void store_f32(float* __restrict__ tile, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, k;
            coord(tid + 256 * i, r, k);
            if (KC) {
                tile[(k + 0) * LD + r] = v[i].x;
                tile[(k + 1) * LD + r] = v[i].y;
                tile[(k + 2) * LD + r] = v[i].z;
                tile[(k + 3) * LD + r] = v[i].w;
            } else {
                *reinterpret_cast<float4*>(&tile[k * LD + r]) = v[i];
            }
        }
    }

    // bf16 LDS image: tile[row][k], leading dimension LD bf16 elements
    template <int LD>
    __device__ __forceinline__ void store_bf16(unsigned short* __restrict__ tile, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, k;
            coord(tid + 256 * i, r, k);
            if (KC) {
                uint2 pk;
                pk.x = pk_pack_bf2(v[i].x, v[i].y);
                pk.y = pk_pack_bf2(v[i].z, v[i].w);
                *reinterpret_cast<uint2*>(&tile[r * LD + k]) = pk;  // 8-B aligned: LD*2 and k*2 multiples of 8
            } else {
                tile[(r + 0) * LD + k] = pk_f2bf(v[i].x);
                tile[(r + 1) * LD + k] = pk_f2bf(v[i].y);
                tile[(r + 2) * LD + k] = pk_f2bf(v[i].z);
                tile[(r + 3) * LD + k] = pk_f2bf(v[i].w);
            }
        }
    }
};

// ---------------------------------------------------------------------------
// epilogue shared by both precisions
// ---------------------------------------------------------------------------
__device__ __forceinline__ void store_acc(const GemmArgs& p, const f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn,
                                          int lane, int split) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            if (col >= p.N) continue;
            const float bv = (p.bias != nullptr && p.ws == nullptr) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= p.M) continue;
                const float a = acc[i][j][r];
                if (p.ws != nullptr) {
                    p.ws[((long)split * p.M + row) * p.N + col] = a;
                } else {
                    float* c = p.C + (long)row * p.ldc + col;
                    float o = p.alpha * a + bv;
                    if (p.beta != 0.f) o += p.beta * (*c);
                    *c = o;
                }
            }
        }
}

// ---------------------------------------------------------------------------
// fp32 kernel
// ---------------------------------------------------------------------------
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    constexpr int BK = 16, LD = BM + 4;
    __shared__ __attribute__((aligned(16))) float As[2][BK * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int m0, n0, split;
    if (!work_item(p, m0, n0, split)) return;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TileLoader<BK, A_KC> la;
    TileLoader<BK, B_KC> lb;
    // B operand viewed as rows = n: element(n, k) = B[k*b_rs + n*b_cs]
    if (nk > 0) {
        la.load(p.A, p.a_rs, p.a_cs, m0, kbeg, p.M, kend, p.vecA, tid);
        lb.load(p.B, p.b_cs, p.b_rs, n0, kbeg, p.N, kend, p.vecB, tid);
        la.template store_f32<LD>(As[0], tid);
        lb.template store_f32<LD>(Bs[0], tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            la.load(p.A, p.a_rs, p.a_cs, m0, kbeg + (kt + 1) * BK, p.M, kend, p.vecA, tid);
            lb.load(p.B, p.b_cs, p.b_rs, n0, kbeg + (kt + 1) * BK, p.N, kend, p.vecB, tid);
        }
        const float* as = As[cur];
        const float* bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int kidx = 2 * kk + (lane >> 5);
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = as[kidx * LD + wm * 64 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = bs[kidx * LD + wn * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            la.template store_f32<LD>(As[cur ^ 1], tid);
            lb.template store_f32<LD>(Bs[cur ^ 1], tid);
        }
        __syncthreads();
    }
    store_acc(p, acc, m0, n0, wm, wn, lane, split);
}

// ---------------------------------------------------------------------------
// fp32 kernel, second form (round 6): the operand tiles go L2 -> LDS with the LDS-DMA (global_load_lds_dwordx4: no VGPR
// round trip, no ds_write pass, no per-k-tile address arithmetic - the per-thread source addresses are computed once and
// advanced by a constant).  With the register-staged form above, issuing the loads alone cost 20 % of a launch even when
// every load hit the cache (profiles/r06_fp32_gemm_dma.json).
//   LDS images (lane-linear, as the DMA writes them; 8 KB per operand and k-tile of 16):
//     k-contiguous operand  [128 rows][4 slots of 16 B]; slot = quad ^ ((row >> 2) & 3) - applied to the SOURCE address
//                           when staging and again when reading: lane half h reads quads h and 2 + h of its row with
//                           ds_read_b128 (conflict-free per 16-lane group); v_permlane32_swap_b32 then pairs the halves
//                           so that one register holds k (lanes 0-31) and k + 1 (lanes 32-63)
//     m/n-contiguous operand [16 k][128 rows]: fragments are ds_read_b32 over 32 consecutive floats
//   MFMA s of a k-tile multiplies k = 2 s (lanes 0-31) and 2 s + 1 (lanes 32-63): every output element is one fmaf chain
//   over ASCENDING k, bit for bit what the first form computes (and what the reference's CPU library does on the small
//   products of the trajectory fixture: with the quads taken as (0,4),(1,5).. instead, the 30-step CE-loss test drifted to
//   1e-3 at step 24 - exactly what one ulp on the inputs does to the reference's own run, tools/diag_fp32_trajectory_floor.py).
//   Pipeline: two stages; the DMA of tile kt + 1 is issued once tile kt's fragments sit in registers (all of a k-tile's
//   fragments are read up front), so it flies under the tile's 32 MFMAs; ONE barrier per k-tile.
// Covers operands with 16-byte aligned bases and leading dimensions that are multiples of 4 floats, and m/n-contiguous
// operands with 8-byte aligned rows when no piece hangs over a row end; everything else takes the first form.  Rows beyond M / N are clamped (never stored); 16-byte pieces beyond the end of the reduction come
// from a zero page.
// ---------------------------------------------------------------------------
__device__ float g_zero_page_f32[16];

__device__ __forceinline__ void f32_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int f32_swz(int row) { return (row >> 2) & 3; }

template <bool KC>
struct F32TileSrc {
    unsigned long long a[2];
    unsigned long long step;
    // ld: floats between consecutive rows (KC) or consecutive k (not KC)
    __device__ __forceinline__ void init(const float* __restrict__ base, long ld, int r0, int rmax, int k0, int tid) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = j * 256 + tid;
            if (KC) {
                const int row = q >> 2, ks = (q & 3) ^ f32_swz(row);
                int gr = r0 + row;
                gr = gr < rmax ? gr : rmax - 1;
                a[j] = (unsigned long long)(base + (long)gr * ld + k0 + ks * 4);
            } else {
                const int kr = q >> 5;
                int gc = r0 + (q & 31) * 4;
                if (gc >= rmax) gc = (rmax - 1) & ~3;  // (a piece entirely out of range: any in-range piece; never stored)
                a[j] = (unsigned long long)(base + (long)(k0 + kr) * ld + gc);
            }
        }
        step = KC ? 64ull : (unsigned long long)ld * 64ull;
    }
    __device__ __forceinline__ void issue(unsigned char* buf, int wave) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32_glds16(reinterpret_cast<const void*>(a[j]), buf + (j * 256 + wave * 64) * 16);
            a[j] += step;
        }
    }
    // the (single) ragged last k-tile of a split: pieces at or beyond kmax come from the zero page
    __device__ __forceinline__ void issue_tail(unsigned char* buf, int wave, int tid, int k0, int kmax, const float* zeros) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = j * 256 + tid;
            const int gk = KC ? k0 + (((q & 3) ^ f32_swz(q >> 2)) * 4) : k0 + (q >> 5);
            const unsigned long long m = 0ull - (unsigned long long)(gk < kmax);
            const unsigned long long src = (a[j] & m) | ((unsigned long long)zeros & ~m);
            f32_glds16(reinterpret_cast<const void*>(src), buf + (j * 256 + wave * 64) * 16);
        }
    }
};

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_dma_kernel(GemmArgs p, const float* zeros) {
    constexpr int BK = 16, OPB = 128 * BK * 4, STG = 2 * OPB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int m0, n0, split;
    if (!work_item(p, m0, n0, split)) return;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    F32TileSrc<A_KC> sa;
    F32TileSrc<B_KC> sb;
    sa.init(p.A, A_KC ? p.a_rs : p.a_cs, m0, p.M, kbeg, tid);
    sb.init(p.B, B_KC ? p.b_cs : p.b_rs, n0, p.N, kbeg, tid);
    auto fetch = [&](int kt, unsigned char* buf) {
        const int k0 = kbeg + kt * BK;
        if (k0 + BK <= kend) {
            sa.issue(buf, wave);
            sb.issue(buf + OPB, wave);
        } else {
            sa.issue_tail(buf, wave, tid, k0, kend, zeros);
            sb.issue_tail(buf + OPB, wave, tid, k0, kend, zeros);
        }
    };
    if (nk > 0) fetch(0, smem);
    const int h = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        unsigned char* cur = smem + (kt & 1) * STG;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((A_KC || B_KC) && kt == nk - 1 && (kend & 3) != 0) {
            // K is not a multiple of 4: the 16-byte piece of a k-contiguous operand that straddles the end of the reduction has
            // landed with what follows the row in memory - its elements at k >= K are zeroed here (the thread that fetched a
            // piece owns its LDS slot), so the tile adds exact zeros like the first form's
            const int k0 = kbeg + kt * BK;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = j * 256 + tid;
                const int gk = k0 + (((q & 3) ^ f32_swz(q >> 2)) * 4);
                if (gk < kend && gk + 4 > kend) {
                    for (int e = kend - gk; e < 4; ++e) {
                        if (A_KC) *reinterpret_cast<float*>(cur + q * 16 + e * 4) = 0.f;
                        if (B_KC) *reinterpret_cast<float*>(cur + OPB + q * 16 + e * 4) = 0.f;
                    }
                }
            }
        }
        __syncthreads();  // tile kt has landed for every wave; everybody is done reading the other buffer
        // all fragments of the k-tile up front.  MFMA s multiplies k = 2 s (lanes 0-31) and k = 2 s + 1 (lanes 32-63), in that
        // order: every output element is ONE fmaf chain over ascending k, as in the first form.  A k-contiguous operand is
        // read as quad 2 p + h per lane half (8 ds_read_b128 per k-tile for both fragments) and the halves are exchanged
        // in registers: v_permlane32_swap(e0, e1) leaves (k, k + 1) of the lower half's quad in the first register and of the
        // upper half's quad in the second
        float aop[8][2], bop[8][2];
        if (A_KC) {
            f32x4 aq[2][2];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = wm * 64 + i * 32 + l31;
                    aq[pp][i] = *reinterpret_cast<const f32x4*>(cur + row * 64 + (((2 * pp + h) ^ f32_swz(row)) * 16));
                }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(aq[pp][i][2 * e]),
                                                                        __float_as_uint(aq[pp][i][2 * e + 1]), false, false);
                        aop[4 * pp + e][i] = __uint_as_float(r[0]);
                        aop[4 * pp + 2 + e][i] = __uint_as_float(r[1]);
                    }
        } else {
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    aop[s_][i] = *reinterpret_cast<const float*>(cur + (2 * s_ + h) * 512 + (wm * 64 + i * 32 + l31) * 4);
        }
        if (B_KC) {
            f32x4 bq[2][2];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wn * 64 + j * 32 + l31;
                    bq[pp][j] = *reinterpret_cast<const f32x4*>(cur + OPB + row * 64 + (((2 * pp + h) ^ f32_swz(row)) * 16));
                }
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(bq[pp][j][2 * e]),
                                                                        __float_as_uint(bq[pp][j][2 * e + 1]), false, false);
                        bop[4 * pp + e][j] = __uint_as_float(r[0]);
                        bop[4 * pp + 2 + e][j] = __uint_as_float(r[1]);
                    }
        } else {
#pragma unroll
            for (int s_ = 0; s_ < 8; ++s_)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bop[s_][j] = *reinterpret_cast<const float*>(cur + OPB + (2 * s_ + h) * 512 + (wn * 64 + j * 32 + l31) * 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);  // the DMA of the next tile is issued behind this tile's reads, in front of its MFMAs
        if (kt + 1 < nk) fetch(kt + 1, smem + ((kt + 1) & 1) * STG);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[s_][i], bop[s_][j], acc[i][j], 0, 0, 0);
    }
    store_acc(p, acc, m0, n0, wm, wn, lane, split);
}

// ---------------------------------------------------------------------------
// bf16-operand kernel (fp32 in HBM, rounded to bf16 while staging)
// ---------------------------------------------------------------------------
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
    constexpr int BK = 32, LD = BK + 8;  // 80-byte rows: odd multiple of 16 B
    __shared__ __attribute__((aligned(16))) unsigned short As[2][BM * LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BN * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int m0, n0, split;
    if (!work_item(p, m0, n0, split)) return;
    const int kbeg = split * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TileLoader<BK, A_KC> la;
    TileLoader<BK, B_KC> lb;
    if (nk > 0) {
        la.load(p.A, p.a_rs, p.a_cs, m0, kbeg, p.M, kend, p.vecA, tid);
        lb.load(p.B, p.b_cs, p.b_rs, n0, kbeg, p.N, kend, p.vecB, tid);
        la.template store_bf16<LD>(As[0], tid);
        lb.template store_bf16<LD>(Bs[0], tid);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            la.load(p.A, p.a_rs, p.a_cs, m0, kbeg + (kt + 1) * BK, p.M, kend, p.vecA, tid);
            lb.load(p.B, p.b_cs, p.b_rs, n0, kbeg + (kt + 1) * BK, p.N, kend, p.vecB, tid);
        }
        const unsigned short* as = As[cur];
        const unsigned short* bs = Bs[cur];
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            // 32x32x16: lane supplies A[row = lane&31][k = (lane>>5)*8 .. +8], same for B columns
            const int koff = ks * 16 + (lane >> 5) * 8;
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(&as[(wm * 64 + i * 32 + (lane & 31)) * LD + koff]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(&bs[(wn * 64 + j * 32 + (lane & 31)) * LD + koff]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            la.template store_bf16<LD>(As[cur ^ 1], tid);
            lb.template store_bf16<LD>(Bs[cur ^ 1], tid);
        }
        __syncthreads();
    }
    store_acc(p, acc, m0, n0, wm, wn, lane, split);
}

// split-K reduction: C = alpha * sum_s ws[s] + beta*C + bias
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int splitk, int M, int N, float alpha, float beta,
                                     const float* __restrict__ bias, float* __restrict__ C, long ldc) {
    const long total = (long)M * N;
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

template <template <bool, bool> class K>
struct Dummy {};

}  // namespace

static int g_f32_first_form = 0;
extern "C" void pk_gemm_f32_set_form(int form) { g_f32_first_form = form == 1 ? 1 : 0; }

extern "C" int pk_gemm(void* stream, int prec, int M, int N, int K, float alpha, const float* A, int64_t a_rs,
                       int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float beta, float* C, int64_t ldc,
                       const float* bias, int splitk, float* workspace) {
    if (M <= 0 || N <= 0) return 0;
    PK_REQUIRE(K >= 0, "pk_gemm: negative K");
    PK_REQUIRE(a_rs == 1 || a_cs == 1, "pk_gemm: A must be contiguous along m or k (strides %ld,%ld)", (long)a_rs,
               (long)a_cs);
    PK_REQUIRE(b_rs == 1 || b_cs == 1, "pk_gemm: B must be contiguous along k or n (strides %ld,%ld)", (long)b_rs,
               (long)b_cs);
    PK_REQUIRE(prec == PK_PREC_F32 || prec == PK_PREC_BF16, "pk_gemm: bad prec %d", prec);
    hipStream_t st = pk_stream(stream);
    GemmArgs p;
    p.M = M; p.N = N; p.K = K;
    p.alpha = alpha; p.beta = beta;
    p.A = A; p.a_rs = a_rs; p.a_cs = a_cs;
    p.B = B; p.b_rs = b_rs; p.b_cs = b_cs;
    p.C = C; p.ldc = ldc; p.bias = bias;
    const bool a_kc = (a_cs == 1);   // contiguous along k
    const bool b_kc = (b_rs == 1);
    // 16-byte vector loads need an aligned base and a leading dimension that keeps rows aligned
    const long a_ld = a_kc ? a_rs : a_cs, b_ld = b_kc ? b_cs : b_rs;
    p.vecA = (((uintptr_t)A & 15) == 0 && (a_ld % 4) == 0) ? 1 : 0;
    p.vecB = (((uintptr_t)B & 15) == 0 && (b_ld % 4) == 0) ? 1 : 0;
    const int BKmax = 32;
    if (splitk < 1) splitk = 1;
    if (splitk > 1) {
        PK_REQUIRE(workspace != nullptr, "pk_gemm: split-K needs a workspace");
        int kps = (K + splitk - 1) / splitk;
        kps = ((kps + BKmax - 1) / BKmax) * BKmax;
        splitk = (K + kps - 1) / kps;
        p.k_per_split = kps;
    }
    if (splitk <= 1) {
        splitk = 1;
        p.k_per_split = ((K + BKmax - 1) / BKmax) * BKmax;
        if (p.k_per_split == 0) p.k_per_split = BKmax;
    }
    p.ws = (splitk > 1) ? workspace : nullptr;
    p.gx = (N + BN - 1) / BN;
    p.gy = (M + BM - 1) / BM;
    const long items = (long)p.gx * p.gy * splitk;
    PK_REQUIRE(items < (1L << 30), "pk_gemm: %ld tiles", items);
    p.items = (int)items;
    p.per_xcd = (p.items + 7) / 8;
    dim3 grid(p.per_xcd * 8), block(256);
    {
        static int flat = -1;  // PK_EXPERIMENT gemm_f32_flat=1: items in launch order (the mapping until round 5; A/B)
        if (flat < 0) {
            const char* e = pk_experiment("gemm_f32_flat");
            flat = e ? atoi(e) : 0;
        }
        if (flat) p.per_xcd = 0;
    }
#define PK_LAUNCH_GEMM(KERN)                                                              \
    do {                                                                                  \
        if (a_kc && b_kc) hipLaunchKernelGGL((KERN<true, true>), grid, block, 0, st, p);   \
        else if (a_kc && !b_kc) hipLaunchKernelGGL((KERN<true, false>), grid, block, 0, st, p); \
        else if (!a_kc && b_kc) hipLaunchKernelGGL((KERN<false, true>), grid, block, 0, st, p); \
        else hipLaunchKernelGGL((KERN<false, false>), grid, block, 0, st, p);              \
    } while (0)
    bool dma = false;
    if (prec == PK_PREC_F32) {
        static int dma_on = -1;  // PK_EXPERIMENT f32_dma=0: the register-staged first form for every shape (A/B)
        static void* zp = nullptr;
        if (dma_on < 0) {
            const char* e = pk_experiment("f32_dma");
            dma_on = (e && e[0] == '0') ? 0 : 1;
        }
        if (zp == nullptr) PK_CHECK_HIP(hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_page_f32)));
        // a 16-byte piece may hang over the end of the reduction (k-contiguous operand: zeroed in LDS) or over the last row /
        // column (m/n-contiguous operand: those lanes' results are never stored) - a 16-byte aligned 16-byte read that begins
        // inside the matrix never crosses a page, so it cannot fault
        // (an m/n-contiguous operand whose rows are only 8-byte aligned - the GRU's 1650-float gate-gradient rows - is taken
        // too when no piece hangs over the end of a row: the LDS-DMA fetches 8-byte aligned 16-byte pieces, and a piece that
        // lies inside the matrix cannot fault)
        const bool okA = p.vecA != 0 || (!a_kc && ((uintptr_t)A & 7) == 0 && (a_ld % 2) == 0 && (M % 4) == 0);
        const bool okB = p.vecB != 0 || (!b_kc && ((uintptr_t)B & 7) == 0 && (b_ld % 2) == 0 && (N % 4) == 0);
        dma = dma_on && !g_f32_first_form && K > 0 && okA && okB;
        if (dma) {
            const float* zeros = (const float*)zp;
            if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_dma_kernel<true, true>), grid, block, 0, st, p, zeros);
            else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_f32_dma_kernel<true, false>), grid, block, 0, st, p, zeros);
            else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_dma_kernel<false, true>), grid, block, 0, st, p, zeros);
            else hipLaunchKernelGGL((gemm_f32_dma_kernel<false, false>), grid, block, 0, st, p, zeros);
        }
    }
    if (dma) {
    } else if (prec == PK_PREC_F32) PK_LAUNCH_GEMM(gemm_f32_kernel);
    else PK_LAUNCH_GEMM(gemm_bf16_kernel);
#undef PK_LAUNCH_GEMM
    PK_LAUNCH_CHECK();
    if (splitk > 1) {
        const long total = (long)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, workspace, splitk, M, N, alpha, beta, bias,
                           C, (long)ldc);
        PK_LAUNCH_CHECK();
    }
    return 0;
}
