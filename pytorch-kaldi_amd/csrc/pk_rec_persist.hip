// pk_rec_persist.hip - PERSISTENT recurrent time loop for gfx950 (one launch per
// layer and direction-pair instead of T launches).
//
// Why: the reference's time loop (neural_networks.py:457-469 / 1130-1141 /
// 1438-1447) is T strictly dependent steps of a tiny GEMM [R x H].[H x G*H];
// at H = 550 the recurrent matrix (1.2 MB bf16 / 2.4 MB fp32) fits no single
// CU, and a kernel boundary per step costs more than the step itself.
//
// Design (CDNA4-first):
//  * the 2B rows (forward sequences, then time-reversed copies) are independent,
//    so they are split into 16-row MFMA tiles; a CLUSTER of `Pn` workgroups
//    owns a set of tiles for the whole sequence;
//  * inside a cluster the recurrent matrix is partitioned by hidden unit and
//    lives in REGISTERS for all T steps (MFMA B fragments; 512 KB of VGPRs per
//    CU > 160 KB LDS): bf16 -> 4 waves x 16 units (N-split), fp32 -> 4 waves
//    split K for the same 16 units and reduce through LDS in a fixed order;
//  * h_t is exchanged through L2 with NO barrier and NO flag: the layer output
//    buffer Y itself is the mailbox.  It is pre-filled with a NaN sentinel
//    (0xFFFFFFFF); producers store h with write-through agent-scope stores
//    (sc1), consumers poll the exact words they need with agent-scope loads
//    until no sentinel is left ("the data is the flag": one 4-byte granule per
//    value, placement independent - MI355X guide, Guideline 16 R2).  Backward
//    uses the gate-gradient buffer dP2 the same way;
//  * blocks of one cluster are congruent mod 8, i.e. land on one XCD under the
//    observed dispatch order, so the exchange stays in that XCD's L2 - a speed
//    assumption only, never a correctness one;
//  * every spin is bounded; a timeout raises a host-visible error word.
//
// Cells: liGRU, RNN, LSTM (single GEMM phase per step).  GRU / minimalGRU need
// two phases per step and use the step-wise algorithm.
#include <stdlib.h>

#include "pk_cell.h"

namespace {

constexpr unsigned SENT = 0xFFFFFFFFu;
constexpr int MTMAX = 4;  // 16-row tiles one cluster may own

struct PArgs {
    int T, B, R, H, YH, act;
    int C, Pn, tile0, ntiles;  // clusters, members per cluster, first tile / tile count of this launch
    const float *P, *pscale, *pshift, *U, *mask;
    float mask_scalar;
    float* Y;
    float* S;
    const float* dY;
    float* dP2;
    unsigned* err;  // host-mapped error counter
    int spin_limit;
};

__device__ __forceinline__ unsigned long long ld_pair_sc1(const float* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool has_sent(unsigned long long v) {
    return (unsigned)v == SENT || (unsigned)(v >> 32) == SENT;
}

// Load one 16-row A tile (rows of `len` floats at src_row(r)) into LDS, polling
// until every word is published.  Each wave stages rows 4*wave .. 4*wave+3.
//   BF16: LDS image [16][LDA] bf16, column index = col_of(c)
//   F32 : LDS image [16][LDA] f32
// `GATES`/`HP`: source column c = g*H + k maps to LDS column g*HP + k.
// shared spin-control: returns true when the wave must give up (timeout / peer failure)
__device__ __forceinline__ bool spin_check(int& spins, int spin_limit, unsigned* err, int lane) {
    if (++spins > spin_limit) {
        if (lane == 0) atomicAdd_system(err, 1u);
        return true;
    }
    if ((spins & 63) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return true;
    __builtin_amdgcn_s_sleep(1);
    return false;
}

// bf16 path: load one 16-row A tile (rows of GATES*H floats at src[r]) into the
// LDS image [16][LDA] bf16, polling until every word is published.  Each wave
// stages rows 4*wave .. 4*wave+3, RB rows per batch (bounds staging VGPRs).
// Source column c = g*H + k maps to LDS column g*HP + k.
template <int NIT, int LDA, int HP, int GATES>
__device__ __forceinline__ bool stage_tile_bf16(unsigned* lds, const float* const (&src)[4], const bool (&row_ok)[4],
                                                int H, int lane, int wave, unsigned* err, int spin_limit, bool dead) {
    constexpr int RB = (NIT > 9) ? 2 : 4;
    const int npairs = (GATES * H) >> 1;
#pragma unroll
    for (int r0 = 0; r0 < 4; r0 += RB) {
        unsigned long long v[RB][NIT];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cp = lane + 64 * it;
                v[rr][it] = 0ull;
                if (row_ok[r0 + rr] && cp < npairs) v[rr][it] = ld_pair_sc1(src[r0 + rr] + 2 * cp);
            }
        if (!dead) {
            int spins = 0;
            while (true) {
                bool bad = false;
#pragma unroll
                for (int rr = 0; rr < RB; ++rr)
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int cp = lane + 64 * it;
                        if (row_ok[r0 + rr] && cp < npairs && has_sent(v[rr][it])) {
                            v[rr][it] = ld_pair_sc1(src[r0 + rr] + 2 * cp);
                            bad = bad || has_sent(v[rr][it]);
                        }
                    }
                if (!__any(bad)) break;
                if (spin_check(spins, spin_limit, err, lane)) {
                    dead = true;
                    break;
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int cp = lane + 64 * it;
                if (cp < npairs) {
                    const int c = 2 * cp;
                    int g = 0;
                    if (GATES > 1) g = (c >= H) + (GATES > 2 ? (c >= 2 * H) : 0) + (GATES > 3 ? (c >= 3 * H) : 0);
                    const int col = g * HP + (c - g * H);
                    const float lo = __uint_as_float((unsigned)v[rr][it]);
                    const float hi = __uint_as_float((unsigned)(v[rr][it] >> 32));
                    const int row = wave * 4 + r0 + rr;
                    lds[(row * LDA + col) >> 1] = pk_pack_bf2(lo, hi);
                }
            }
    }
    return dead;
}

// fp32 path: every lane gathers ITS OWN MFMA A operands straight into registers
// (no LDS): lane (row = lane&15, kq = lane>>4) of K-slice wave `wk` owns the KF
// consecutive k values  kbase = (wk*4 + kq)*KF .. +KF  of each gate, i.e. MFMA
// step kk multiplies k = {kq*KF + kk}; the B fragments use the same k mapping
// (a dot product does not care about the order of its terms).
template <int KF, int GATES>
__device__ __forceinline__ bool gather_f32(float (&av)[GATES][KF], const float* rowp, bool row_ok, int kbase, int H,
                                           int lane, unsigned* err, int spin_limit, bool dead) {
    static_assert((KF % 2) == 0, "KF must be even");
    unsigned long long v[GATES][KF / 2];
#pragma unroll
    for (int g = 0; g < GATES; ++g)
#pragma unroll
        for (int q = 0; q < KF / 2; ++q) {
            const int k = kbase + 2 * q;
            v[g][q] = 0ull;
            if (row_ok && k < H) v[g][q] = ld_pair_sc1(rowp + g * H + k);
        }
    if (!dead) {
        int spins = 0;
        while (true) {
            bool bad = false;
#pragma unroll
            for (int g = 0; g < GATES; ++g)
#pragma unroll
                for (int q = 0; q < KF / 2; ++q) {
                    const int k = kbase + 2 * q;
                    if (row_ok && k < H && has_sent(v[g][q])) {
                        v[g][q] = ld_pair_sc1(rowp + g * H + k);
                        bad = bad || has_sent(v[g][q]);
                    }
                }
            if (!__any(bad)) break;
            if (spin_check(spins, spin_limit, err, lane)) {
                dead = true;
                break;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < GATES; ++g)
#pragma unroll
        for (int q = 0; q < KF / 2; ++q) {
            av[g][2 * q] = __uint_as_float((unsigned)v[g][q]);
            av[g][2 * q + 1] = __uint_as_float((unsigned)(v[g][q] >> 32));
        }
    return dead;
}

template <int CELL, int PREC, int KF>
struct PCfg {
    static constexpr bool BF = (PREC == PK_PREC_BF16);
    static constexpr int G = pk_cell_gates(CELL);
    static constexpr int NS = pk_cell_saved(CELL);
    static constexpr int WK = BF ? 1 : 4;     // K-split across the 4 waves (fp32)
    static constexpr int NR = 4 / WK;         // accumulator rows finalised per lane
    static constexpr int UPW = BF ? 64 : 16;  // hidden units per workgroup
    // padded K extent per gate: bf16 = KF MFMA steps of 32; fp32 = 4 waves x 4 lane-quarters x KF
    static constexpr int HP = BF ? KF * 32 : KF * 16;
};

// ============================================================================
// forward
// ============================================================================
template <int CELL, int PREC, int KF>
__global__ __launch_bounds__(256, 1) void rec_fwd_persist(PArgs a) {
    using Cf = PCfg<CELL, PREC, KF>;
    constexpr bool BF = Cf::BF;
    constexpr int G = Cf::G, NS = Cf::NS, WK = Cf::WK, NR = Cf::NR, HP = Cf::HP;
    constexpr int LDA = HP + 8;  // bf16 elements; 16-B row stride is odd -> conflict-free ds_read_b128
    constexpr int NIT = (HP / 2 + 63) / 64;
    __shared__ __attribute__((aligned(16))) unsigned smem[BF ? (2 * 16 * LDA / 2) : (2 * WK * G * 4 * 64)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int wn = BF ? wave : 0, wk = BF ? 0 : wave;
    const int H = a.H, B = a.B, T = a.T, GH = G * H;
    const int unit = p * Cf::UPW + wn * 16 + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;
    const int kbase = (wk * 4 + kq) * KF;  // fp32: first k of this lane's slice

    // ---- recurrent weights of my units -> registers (once).  B[k][n] = U_g[unit n][k]
    bf16x8 Bb[BF ? G : 1][BF ? KF : 1];
    float Bs[BF ? 1 : G][BF ? 1 : KF];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kk = 0; kk < KF; ++kk) {
            if (BF) {
                bf16x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = kk * 32 + kq * 8 + e;
                    const float w = (unit_ok && k < H) ? a.U[((long)(g * H + unit)) * H + k] : 0.f;
                    f[e] = (short)pk_f2bf(w);
                }
                Bb[BF ? g : 0][BF ? kk : 0] = f;
            } else {
                const int k = kbase + kk;
                Bs[BF ? 0 : g][BF ? 0 : kk] = (unit_ok && k < H) ? a.U[((long)(g * H + unit)) * H + k] : 0.f;
            }
        }
    float psc[G], psh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        psc[g] = unit_ok ? a.pscale[g * H + unit] : 0.f;
        psh[g] = unit_ok ? a.pshift[g * H + unit] : 0.f;
    }
    // zero LDS (bf16: pad columns and rows beyond R stay zero forever)
    for (int i = tid; i < (int)(sizeof(smem) / 4); i += 256) smem[i] = 0u;

    const int nmt = (a.ntiles > c) ? (a.ntiles - c + a.C - 1) / a.C : 0;  // tiles of my cluster (uniform per WG)
    float hprev[MTMAX][NR], cprev[MTMAX][NR], msk[MTMAX][NR];
#pragma unroll
    for (int i = 0; i < MTMAX; ++i)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            hprev[i][r] = 0.f;
            cprev[i][r] = 0.f;
            const int n = (a.tile0 + c + a.C * i) * 16 + kq * 4 + wk * NR + r;
            msk[i][r] = (a.mask != nullptr && i < nmt && n < a.R && unit_ok) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        }
    __syncthreads();

    bool dead = false;
    int iter = 0;
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int i = 0; i < MTMAX; ++i) {
            if (i < nmt) {
                const int n0 = (a.tile0 + c + a.C * i) * 16;
                // ---- projections of this step (independent of the recurrence: issue first)
                float pre[NR][G];
                long prow[NR];
                int dirs[NR];
                bool ok[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int n = n0 + kq * 4 + wk * NR + r;
                    const int dir = n >= B ? 1 : 0;
                    const int b = n - dir * B;
                    const int ts = dir ? (T - 1 - t) : t;
                    ok[r] = (n < a.R) && unit_ok;
                    dirs[r] = dir;
                    prow[r] = (long)ts * B + b;
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        pre[r][g] = ok[r] ? a.P[prow[r] * GH + g * H + unit] * psc[g] + psh[g] : 0.f;
                }
                f32x4 acc[G];
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (t > 0) {
                    if (BF) {
                        // ---- gather h_{t-1} of my 16 rows (all H units) from the mailbox into LDS
                        unsigned* Abuf = smem + (iter & 1) * (16 * LDA / 2);
                        const float* src[4];
                        bool row_ok[4];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int n = n0 + wave * 4 + rr;
                            const int dir = n >= B ? 1 : 0;
                            const int b = n - dir * B;
                            const int tsp = dir ? (T - t) : (t - 1);  // storage time of step t-1
                            row_ok[rr] = n < a.R;
                            src[rr] = a.Y + ((long)tsp * B + b) * a.YH + dir * H;
                        }
                        dead = stage_tile_bf16<NIT, LDA, HP, 1>(Abuf, src, row_ok, H, lane, wave, a.err, a.spin_limit, dead);
                        __syncthreads();
                        const unsigned short* Ab = reinterpret_cast<const unsigned short*>(Abuf);
#pragma unroll
                        for (int kk = 0; kk < KF; ++kk) {
                            const bf16x8 af = *reinterpret_cast<const bf16x8*>(&Ab[(lane & 15) * LDA + kk * 32 + kq * 8]);
#pragma unroll
                            for (int g = 0; g < G; ++g)
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bb[BF ? g : 0][BF ? kk : 0], acc[g], 0, 0, 0);
                        }
                    } else {
                        // ---- every lane gathers its own A operands (row = lane&15) straight to registers
                        const int n = n0 + (lane & 15);
                        const int dir = n >= B ? 1 : 0;
                        const int b = n - dir * B;
                        const int tsp = dir ? (T - t) : (t - 1);
                        const float* rowp = a.Y + ((long)tsp * B + b) * a.YH + dir * H;
                        float av[1][KF];
                        dead = gather_f32<KF, 1>(av, rowp, n < a.R, kbase, H, lane, a.err, a.spin_limit, dead);
#pragma unroll
                        for (int kk = 0; kk < KF; ++kk)
#pragma unroll
                            for (int g = 0; g < G; ++g)
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][kk], Bs[BF ? 0 : g][BF ? 0 : kk], acc[g], 0, 0, 0);
                        // fixed-order reduction of the 4 K-slices through LDS (double-buffered)
                        float* red = reinterpret_cast<float*>(smem) + (iter & 1) * (WK * G * 4 * 64);
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int r = 0; r < 4; ++r) red[((wk * G + g) * 4 + r) * 64 + lane] = acc[g][r];
                        __syncthreads();
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            float s = 0.f;
#pragma unroll
                            for (int w = 0; w < WK; ++w) s += red[((w * G + g) * 4 + wk) * 64 + lane];
                            acc[g][0] = s;  // the fp32 variant finalises one row per lane (accumulator row wk)
                        }
                    }
                }
                // ---- gate math for my (row, unit) pairs; publish h_t
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    float pr[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) pr[g] = pre[r][g] + acc[g][r];
                    float h, cc, s[NS];
                    pk_cell_fwd<CELL>(a.act, pr, hprev[i][r], cprev[i][r], msk[i][r], h, cc, s);
                    if (ok[r]) {
                        hprev[i][r] = h;
                        cprev[i][r] = cc;
                        st_sc1(a.Y + prow[r] * a.YH + dirs[r] * H + unit, h);
                        float* sp = a.S + ((long)dirs[r] * T * B + prow[r]) * (NS * H) + unit;
#pragma unroll
                        for (int k = 0; k < NS; ++k) sp[k * H] = s[k];
                    }
                }
                ++iter;
            }
        }
    }
}

// ============================================================================
// backward: dL/dh_{t-1} = direct + [dgates_t] . [U_0; U_1; ...]
// ============================================================================
template <int CELL, int PREC, int KF>
__global__ __launch_bounds__(256, 1) void rec_bwd_persist(PArgs a) {
    using Cf = PCfg<CELL, PREC, KF>;
    constexpr bool BF = Cf::BF;
    constexpr int G = Cf::G, NS = Cf::NS, WK = Cf::WK, NR = Cf::NR, HP = Cf::HP;
    constexpr int LDA = G * HP + 8;
    constexpr int NIT = (G * HP / 2 + 63) / 64;
    __shared__ __attribute__((aligned(16))) unsigned smem[BF ? (2 * 16 * LDA / 2) : (2 * WK * 4 * 64)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int wn = BF ? wave : 0, wk = BF ? 0 : wave;
    const int H = a.H, B = a.B, T = a.T, GH = G * H;
    const int unit = p * Cf::UPW + wn * 16 + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;
    const int kbase = (wk * 4 + kq) * KF;
    const long TB = (long)T * B;

    // B[kidx = (g,k)][n = unit] = U_g[k][unit]
    bf16x8 Bb[BF ? G : 1][BF ? KF : 1];
    float Bs[BF ? 1 : G][BF ? 1 : KF];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kk = 0; kk < KF; ++kk) {
            if (BF) {
                bf16x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = kk * 32 + kq * 8 + e;
                    const float w = (unit_ok && k < H) ? a.U[((long)(g * H + k)) * H + unit] : 0.f;
                    f[e] = (short)pk_f2bf(w);
                }
                Bb[BF ? g : 0][BF ? kk : 0] = f;
            } else {
                const int k = kbase + kk;
                Bs[BF ? 0 : g][BF ? 0 : kk] = (unit_ok && k < H) ? a.U[((long)(g * H + k)) * H + unit] : 0.f;
            }
        }
    for (int i = tid; i < (int)(sizeof(smem) / 4); i += 256) smem[i] = 0u;

    const int nmt = (a.ntiles > c) ? (a.ntiles - c + a.C - 1) / a.C : 0;
    float dh_dir[MTMAX][NR], dc_car[MTMAX][NR], msk[MTMAX][NR];
#pragma unroll
    for (int i = 0; i < MTMAX; ++i)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            dh_dir[i][r] = 0.f;
            dc_car[i][r] = 0.f;
            const int n = (a.tile0 + c + a.C * i) * 16 + kq * 4 + wk * NR + r;
            msk[i][r] = (a.mask != nullptr && i < nmt && n < a.R && unit_ok) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        }
    __syncthreads();

    bool dead = false;
    int iter = 0;
    for (int t = T - 1; t >= 0; --t) {
#pragma unroll
        for (int i = 0; i < MTMAX; ++i) {
            if (i < nmt) {
                const int n0 = (a.tile0 + c + a.C * i) * 16;
                // ---- everything that does not depend on the carry: issue first
                float sv[NR][NS], hp[NR], cp[NR], dy[NR];
                long prow[NR];
                int dirs[NR];
                bool ok[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int n = n0 + kq * 4 + wk * NR + r;
                    const int dir = n >= B ? 1 : 0;
                    const int b = n - dir * B;
                    const int ts = dir ? (T - 1 - t) : t;
                    ok[r] = (n < a.R) && unit_ok;
                    dirs[r] = dir;
                    prow[r] = (long)ts * B + b;
                    const long srow = (long)dir * TB + prow[r];
                    const long prev = (long)(dir ? ts + 1 : ts - 1) * B + b;  // storage row of step t-1
#pragma unroll
                    for (int k = 0; k < NS; ++k) sv[r][k] = ok[r] ? a.S[srow * (NS * H) + k * H + unit] : 0.f;
                    hp[r] = (ok[r] && t > 0) ? a.Y[prev * a.YH + dir * H + unit] : 0.f;
                    cp[r] = (CELL == PK_CELL_LSTM && ok[r] && t > 0)
                                ? a.S[((long)dir * TB + prev) * (NS * H) + 4 * H + unit]
                                : 0.f;
                    dy[r] = ok[r] ? a.dY[prow[r] * a.YH + dir * H + unit] : 0.f;
                }
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                if (t < T - 1) {
                    f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (BF) {
                        // ---- gather dgates_{t+1} of my 16 rows (all G*H columns) from the mailbox
                        unsigned* Abuf = smem + (iter & 1) * (16 * LDA / 2);
                        const float* src[4];
                        bool row_ok[4];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int n = n0 + wave * 4 + rr;
                            const int dir = n >= B ? 1 : 0;
                            const int b = n - dir * B;
                            const int tsn = dir ? (T - 2 - t) : (t + 1);  // storage time of step t+1
                            row_ok[rr] = n < a.R;
                            src[rr] = a.dP2 + ((long)dir * TB + (long)tsn * B + b) * GH;
                        }
                        dead = stage_tile_bf16<NIT, LDA, HP, G>(Abuf, src, row_ok, H, lane, wave, a.err, a.spin_limit, dead);
                        __syncthreads();
                        const unsigned short* Ab = reinterpret_cast<const unsigned short*>(Abuf);
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int kk = 0; kk < KF; ++kk) {
                                const bf16x8 af =
                                    *reinterpret_cast<const bf16x8*>(&Ab[(lane & 15) * LDA + g * HP + kk * 32 + kq * 8]);
                                if ((kk & 1) == 0)
                                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bb[BF ? g : 0][BF ? kk : 0], acc, 0, 0, 0);
                                else
                                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bb[BF ? g : 0][BF ? kk : 0], acc2, 0, 0, 0);
                            }
                    } else {
                        const int n = n0 + (lane & 15);
                        const int dir = n >= B ? 1 : 0;
                        const int b = n - dir * B;
                        const int tsn = dir ? (T - 2 - t) : (t + 1);
                        const float* rowp = a.dP2 + ((long)dir * TB + (long)tsn * B + b) * GH;
                        float av[G][KF];
                        dead = gather_f32<KF, G>(av, rowp, n < a.R, kbase, H, lane, a.err, a.spin_limit, dead);
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int kk = 0; kk < KF; ++kk) {
                                if ((kk & 1) == 0)
                                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][kk], Bs[BF ? 0 : g][BF ? 0 : kk], acc, 0, 0, 0);
                                else
                                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][kk], Bs[BF ? 0 : g][BF ? 0 : kk], acc2, 0, 0, 0);
                            }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
                    if (!BF) {
                        float* red = reinterpret_cast<float*>(smem) + (iter & 1) * (WK * 4 * 64);
#pragma unroll
                        for (int r = 0; r < 4; ++r) red[(wk * 4 + r) * 64 + lane] = acc[r];
                        __syncthreads();
                        float s = 0.f;
#pragma unroll
                        for (int w = 0; w < WK; ++w) s += red[(w * 4 + wk) * 64 + lane];
                        acc[0] = s;
                    }
                }
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const float dh = dy[r] + dh_dir[i][r] + acc[r];
                    float dg[G], dhd, dcp;
                    pk_cell_bwd<CELL>(a.act, sv[r], hp[r], cp[r], msk[i][r], dh, dc_car[i][r], dg, dhd, dcp);
                    if (ok[r]) {
                        dh_dir[i][r] = dhd;
                        dc_car[i][r] = dcp;
                        float* o = a.dP2 + ((long)dirs[r] * TB + prow[r]) * GH + unit;
#pragma unroll
                        for (int g = 0; g < G; ++g) st_sc1(o + g * H, dg[g]);
                    }
                }
                ++iter;
            }
        }
    }
}

unsigned* g_err_host = nullptr;
unsigned* g_err_dev = nullptr;

int ensure_err() {
    if (g_err_host) return 0;
    PK_CHECK_HIP(hipHostMalloc((void**)&g_err_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *g_err_host = 0;
    PK_CHECK_HIP(hipHostGetDevicePointer((void**)&g_err_dev, g_err_host, 0));
    return 0;
}

struct Plan {
    int Pn, C, launches, tiles_per_launch;
};

int make_plan(int prec, int R, int H, Plan& pl) {
    const int upw = (prec == PK_PREC_BF16) ? 64 : 16;
    pl.Pn = (H + upw - 1) / upw;
    const int ncu = pk_num_cu();
    const int tiles = (R + 15) / 16;
    int C = ncu / pl.Pn;
    PK_REQUIRE(C >= 1, "persistent recurrence: H=%d needs %d workgroups per cluster but the device has %d CUs", H, pl.Pn,
               ncu);
    if (C > tiles) C = tiles;
    if (C >= 8) C -= C % 8;  // members of one cluster congruent mod 8 (same XCD under round-robin dispatch)
    pl.C = C;
    pl.tiles_per_launch = C * MTMAX;
    pl.launches = (tiles + pl.tiles_per_launch - 1) / pl.tiles_per_launch;
    return 0;
}

template <int CELL, int PREC, int KF>
int launch_fwd(hipStream_t st, PArgs a, const Plan& pl, int tiles) {
    for (int l = 0; l < pl.launches; ++l) {
        a.tile0 = l * pl.tiles_per_launch;
        a.ntiles = tiles - a.tile0 < pl.tiles_per_launch ? tiles - a.tile0 : pl.tiles_per_launch;
        hipLaunchKernelGGL((rec_fwd_persist<CELL, PREC, KF>), dim3(pl.C * pl.Pn), dim3(256), 0, st, a);
        PK_LAUNCH_CHECK();
    }
    return 0;
}
template <int CELL, int PREC, int KF>
int launch_bwd(hipStream_t st, PArgs a, const Plan& pl, int tiles) {
    for (int l = 0; l < pl.launches; ++l) {
        a.tile0 = l * pl.tiles_per_launch;
        a.ntiles = tiles - a.tile0 < pl.tiles_per_launch ? tiles - a.tile0 : pl.tiles_per_launch;
        hipLaunchKernelGGL((rec_bwd_persist<CELL, PREC, KF>), dim3(pl.C * pl.Pn), dim3(256), 0, st, a);
        PK_LAUNCH_CHECK();
    }
    return 0;
}

int check_persist(const char* who, int cell, int H) {
    PK_REQUIRE(cell == PK_CELL_LIGRU || cell == PK_CELL_RNN || cell == PK_CELL_LSTM,
               "%s: the persistent algorithm covers liGRU/RNN/LSTM; use PK_REC_STEPWISE for cell %d", who, cell);
    PK_REQUIRE((H % 2) == 0, "%s: persistent algorithm needs an even H (got %d); use PK_REC_STEPWISE", who, H);
    PK_REQUIRE(H <= 576, "%s: persistent algorithm is built for H <= 576 (got %d); use PK_REC_STEPWISE", who, H);
    return 0;
}

// KF: bf16 -> k-steps of 32 (H <= 32*KF); f32 -> k-steps of 4 per wave (H <= 16*KF)
#define PK_DISPATCH_PERSIST(FN, ...)                                                                        \
    do {                                                                                                    \
        const bool small = H <= 128;                                                                        \
        if (prec == PK_PREC_BF16) {                                                                         \
            if (cell == PK_CELL_LIGRU) return small ? FN<PK_CELL_LIGRU, PK_PREC_BF16, 4>(__VA_ARGS__) : FN<PK_CELL_LIGRU, PK_PREC_BF16, 18>(__VA_ARGS__); \
            if (cell == PK_CELL_RNN) return small ? FN<PK_CELL_RNN, PK_PREC_BF16, 4>(__VA_ARGS__) : FN<PK_CELL_RNN, PK_PREC_BF16, 18>(__VA_ARGS__);       \
            return small ? FN<PK_CELL_LSTM, PK_PREC_BF16, 4>(__VA_ARGS__) : FN<PK_CELL_LSTM, PK_PREC_BF16, 18>(__VA_ARGS__);                            \
        } else {                                                                                            \
            if (cell == PK_CELL_LIGRU) return small ? FN<PK_CELL_LIGRU, PK_PREC_F32, 8>(__VA_ARGS__) : FN<PK_CELL_LIGRU, PK_PREC_F32, 36>(__VA_ARGS__);  \
            if (cell == PK_CELL_RNN) return small ? FN<PK_CELL_RNN, PK_PREC_F32, 8>(__VA_ARGS__) : FN<PK_CELL_RNN, PK_PREC_F32, 36>(__VA_ARGS__);        \
            return small ? FN<PK_CELL_LSTM, PK_PREC_F32, 8>(__VA_ARGS__) : FN<PK_CELL_LSTM, PK_PREC_F32, 36>(__VA_ARGS__);                             \
        }                                                                                                   \
    } while (0)

}  // namespace

extern "C" unsigned pk_persist_error_count(void) { return g_err_host ? *g_err_host : 0u; }
extern "C" void pk_persist_error_reset(void) {
    if (g_err_host) *g_err_host = 0u;
}

// second-generation exact-fp32 kernels (pk_rec_persist2_f32.hip): liGRU / RNN; their exchange buffer lives in `work`
int pk_rec2f_covers(int cell, int H);
int64_t pk_rec_work_base_floats(int cell, int B, int bidir, int H);
int pk_rec2f_fwd(hipStream_t st, int cell, int act, int T, int B, int bidir, int H, const float* P, const float* pscale,
                 const float* pshift, const float* U, const float* mask, float mask_scalar, float* Y, float* S, float* Yx,
                 const PkLnHost* ln);
int pk_rec2f_bwd(hipStream_t st, int cell, int act, int T, int B, int bidir, int H, const float* U, const float* mask,
                 float mask_scalar, const float* Y, const float* S, const float* dY, float* dP2, float* dGx, const PkLnHost* ln);
// fourth generation (pk_rec_persist4_f32.hip): LSTM / GRU / minimalGRU in exact fp32, with or without per-step LayerNorm
int pk_rec4f_covers(int cell, int H);
int pk_rec4f_fwd(hipStream_t st, int cell, int act, int T, int B, int bidir, int H, const float* P, const float* pscale,
                 const float* pshift, const float* U, const float* mask, float mask_scalar, float* Y, float* S, float* Yx,
                 const PkLnHost* ln);
int pk_rec4f_bwd(hipStream_t st, int cell, int act, int T, int B, int bidir, int H, const float* U, const float* mask,
                 float mask_scalar, const float* Y, const float* S, const float* dY, float* dP2, float* dGx, const PkLnHost* ln);
static bool use_gen4_f32(int prec, int cell, int H, const PkLnHost* ln) {
    (void)ln;
    return prec == PK_PREC_F32 && pk_rec4f_covers(cell, H);
}
static bool use_gen2_f32(int prec, int cell, int H) {
    static int off = -1;
    if (off < 0) {
        const char* e = pk_experiment("rec_f32_gen");  // 1 = keep the first-generation kernels (A/B measurements)
        off = (e && e[0] == '1') ? 1 : 0;
    }
    return !off && prec == PK_PREC_F32 && pk_rec2f_covers(cell, H);
}

int pk_rec_fwd_persistent(hipStream_t st, int prec, int cell, int act, int T, int B, int bidir, int H, const float* P,
                          const float* pscale, const float* pshift, const float* U, const float* mask,
                          float mask_scalar, float* Y, float* S, float* work, const PkLnHost* ln) {
    if (use_gen4_f32(prec, cell, H, ln)) {
        PK_REQUIRE(work != nullptr, "pk_rec_fwd: work buffer missing");
        return pk_rec4f_fwd(st, cell, act, T, B, bidir, H, P, pscale, pshift, U, mask, mask_scalar, Y, S,
                            work + pk_rec_work_base_floats(cell, B, bidir, H), ln);
    }
    if (use_gen2_f32(prec, cell, H)) {
        PK_REQUIRE(work != nullptr, "pk_rec_fwd: work buffer missing");
        return pk_rec2f_fwd(st, cell, act, T, B, bidir, H, P, pscale, pshift, U, mask, mask_scalar, Y, S,
                            work + pk_rec_work_base_floats(cell, B, bidir, H), ln);
    }
    PK_REQUIRE(ln == nullptr, "pk_rec_fwd: per-step LayerNorm in the persistent algorithm is covered for liGRU / RNN in fp32 "
               "(second-generation kernels) and for liGRU / RNN / LSTM in bf16 (pk_rec_fwd_bf16_ln); cell %d prec %d is not", cell, prec);
    int rc = check_persist("pk_rec_fwd", cell, H);
    if (rc) return rc;
    rc = ensure_err();
    if (rc) return rc;
    const int R = B * (1 + bidir), YH = (1 + bidir) * H;
    Plan pl;
    rc = make_plan(prec, R, H, pl);
    if (rc) return rc;
    PArgs a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.YH = YH; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.tile0 = 0; a.ntiles = 0;
    a.P = P; a.pscale = pscale; a.pshift = pshift; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = Y; a.S = S; a.dY = nullptr; a.dP2 = nullptr;
    a.err = g_err_dev; a.spin_limit = 400000;
    // the output buffer is the mailbox: poison it with the sentinel
    PK_CHECK_HIP(hipMemsetAsync(Y, 0xFF, sizeof(float) * (size_t)T * B * YH, st));
    const int tiles = (R + 15) / 16;
    PK_DISPATCH_PERSIST(launch_fwd, st, a, pl, tiles);
}

int pk_rec_bwd_persistent(hipStream_t st, int prec, int cell, int act, int T, int B, int bidir, int H, const float* U,
                          const float* mask, float mask_scalar, const float* Y, const float* S, const float* dY,
                          float* dP2, float* work, const PkLnHost* ln) {
    if (use_gen4_f32(prec, cell, H, ln)) {
        PK_REQUIRE(work != nullptr, "pk_rec_bwd: work buffer missing");
        return pk_rec4f_bwd(st, cell, act, T, B, bidir, H, U, mask, mask_scalar, Y, S, dY, dP2,
                            work + pk_rec_work_base_floats(cell, B, bidir, H), ln);
    }
    if (use_gen2_f32(prec, cell, H)) {
        PK_REQUIRE(work != nullptr, "pk_rec_bwd: work buffer missing");
        return pk_rec2f_bwd(st, cell, act, T, B, bidir, H, U, mask, mask_scalar, Y, S, dY, dP2,
                            work + pk_rec_work_base_floats(cell, B, bidir, H), ln);
    }
    PK_REQUIRE(ln == nullptr, "pk_rec_bwd: per-step LayerNorm in the persistent algorithm does not cover cell %d prec %d", cell, prec);
    int rc = check_persist("pk_rec_bwd", cell, H);
    if (rc) return rc;
    rc = ensure_err();
    if (rc) return rc;
    const int R = B * (1 + bidir), YH = (1 + bidir) * H, G = pk_cell_gates(cell);
    Plan pl;
    rc = make_plan(prec, R, H, pl);
    if (rc) return rc;
    PArgs a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.YH = YH; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.tile0 = 0; a.ntiles = 0;
    a.P = nullptr; a.pscale = nullptr; a.pshift = nullptr; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = const_cast<float*>(Y); a.S = const_cast<float*>(S); a.dY = dY; a.dP2 = dP2;
    a.err = g_err_dev; a.spin_limit = 400000;
    PK_CHECK_HIP(hipMemsetAsync(dP2, 0xFF, sizeof(float) * (size_t)(1 + bidir) * T * B * G * H, st));
    const int tiles = (R + 15) / 16;
    PK_DISPATCH_PERSIST(launch_bwd, st, a, pl, tiles);
}
