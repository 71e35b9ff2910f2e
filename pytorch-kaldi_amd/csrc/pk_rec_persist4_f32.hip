// pk_rec_persist4_f32.hip - EXACT-fp32 persistent forward and BPTT of the cells whose recurrent matrices do not fit one
// wave's registers in fp32: LSTM (neural_networks.py:457-469), GRU (:629-641), minimalGRU (:1291-1302) and their
// autograd - the parity-grade mode (PK_PRECISION=fp32, what the 1e-4 tests run in).  Until round 6 LSTM ran on the
// first-generation kernels (pk_rec_persist.hip: 4-byte granules, 32 ms per backward launch at the BASELINE geometry) and
// GRU / minimalGRU step by step (pk_rec.hip: 15 000 launches per training step).
//
// Same idea as every persistent generation here - clusters of workgroups, 16 rows (one MFMA M tile) per cluster, the
// recurrent matrix in REGISTERS for all T steps, h_t (dgates_t) exchanged through L2 in 16-byte chunks with the data as
// the flag - with two differences that make the bigger cells fit:
//
//   * K is split over a PAIR of waves.  A workgroup is two pairs = 32 hidden units; wave (tile, kh) holds the B fragments
//     of its tile's 16 units for HALF of the reduction (LSTM forward: 4 gates x 288 k = 288 registers; GRU: 216), both
//     waves of a pair sit on different SIMDs, exchange their partial sums through LDS (one workgroup barrier per MFMA
//     phase) and then do the SAME gate math on the same sums (a + b == b + a bitwise), so state never has to be handed
//     over; the pair splits the loads and stores of the step between its waves instead.
//   * The A operand never touches LDS.  Lane (row r = lane & 15, quarter kq = lane >> 4) of v_mfma_f32_16x16x4_f32
//     multiplies A[r][16 j + 4 kq + e] in step (j, e) - four consecutive k of one row: exactly one published 16-byte
//     chunk.  So every wave polls ITS chunks straight into the registers the MFMAs read (sentinel test on the loaded
//     registers, re-load of what has not arrived), and the poll -> LDS tile -> barrier -> ds_read path of the second
//     generation (and its 74-148 KB of LDS per workgroup) is gone.  The price: the two tiles of a workgroup read the same
//     rows from L2 twice (74 KB per CU and step) - L2 has that bandwidth many times over.
//
// A step is MFMA-bound by construction: 288 (LSTM) / 216 (GRU) x 32 clocks per wave and step on the exact-fp32 matrix
// pipe.  32 units per workgroup = 18 workgroups per cluster at H = 550, so a launch holds 8 clusters (144 CUs, every
// cluster on one XCD) and 256 rows take two launches; the other 112 CUs stay with the side-stream GEMMs.
//
// Two-phase cells: forward  1. poll h_{t-1} (Yx) -> MFMA [z(,r)] -> publish x_t = r*h / z*h (Xx)
//                           2. poll x_t (Xx)     -> MFMA candidate -> h_t -> publish (Yx)
//                  backward 1. poll [dz(,dr)]_{t+1} (dGx slots 0..G-2) -> MFMA carry -> dh_t -> da_t -> publish slot G-1
//                           2. poll da_t (dGx slot G-1) -> MFMA q = da.U_h -> dz(,dr) -> publish slots 0..G-2
// Per-step LayerNorm of h_t is not covered here (those layers keep the step-wise algorithm).
#include <stdlib.h>

#define PK_REC2_PRECISE 1
#include "pk_rec2_common.h"

namespace {

constexpr int KJ = KPAD / 16;    // 36 groups of 16 k per gate
constexpr int NJH = KJ / 2;      // 18: what one wave of a pair holds of one gate
constexpr int UW = 32;           // hidden units per workgroup (two pairs of waves)
constexpr int HS4 = 32;          // placement-handshake words per cluster (up to 18 members)

__device__ __forceinline__ u32x4 no_sentinel4(f32x4 v) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u = __float_as_uint(v[e]);
        o[e] = u == 0xFFFFFFFFu ? 0x7FC00000u : u;  // the one NaN encoding that reads as "not written yet"
    }
    return o;
}

// cluster_on_one_xcd (pk_rec2_common.h) for clusters of up to HS4 members
__device__ __forceinline__ bool cluster_on_one_xcd4(const R2Args& a, int c, int p, int tid, bool& dead) {
    const unsigned my = (a.hs_gen << 4) | (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu);  // HW_REG_XCC_ID
    unsigned* tab = a.xcd_tab + c * HS4;
    if (tid == 0) __hip_atomic_store(tab + p, my, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int same = 1;
    if (tid < a.Pn) {
        unsigned v = 0xFFFFFFFFu;
        bool current = false;
        for (int spins = 0; spins < a.spin_limit; ++spins) {
            v = __hip_atomic_load(tab + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            current = (v >> 4) == a.hs_gen;
            if (current) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!current) {
            atomicAdd_system(a.err, 1u);
            same = 0;
            dead = true;
        } else {
            same = (v == my) ? 1 : 0;
        }
    }
    return __syncthreads_and(same) != 0;
}

// Poll N 16-byte chunks per lane straight into registers until none holds the sentinel; off(i) = byte offset of chunk i.
// Chunks a lane does not need (k beyond Hp, rows beyond the cluster's) are redirected to a chunk it does need: their B
// fragments are zero (0 x finite = 0 exactly) or their rows are discarded, and a redirected read never waits for anything
// the lane would not wait for anyway.
template <int N, bool FAST, typename OffFn>
__device__ __forceinline__ bool poll_regs(__amdgpu_buffer_rsrc_t rs, OffFn off, u32x4 (&v)[N], unsigned* err, int spin_limit,
                                          int lane, bool dead) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = poll_load<FAST>(rs, off(i));
    bool bad = false;
#pragma unroll
    for (int i = 0; i < N; ++i) bad = bad | has_sent16(v[i]);
    if (__any(bad) && !dead) {
        int spins = 0;
        while (true) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                if (has_sent16(v[i])) v[i] = poll_load<FAST>(rs, off(i));
            bad = false;
#pragma unroll
            for (int i = 0; i < N; ++i) bad = bad | has_sent16(v[i]);
            if (!__any(bad)) break;
            if (spin_check2(spins, spin_limit, err, lane)) {
                dead = true;
                break;
            }
        }
    }
    return dead;
}

// A lane's chunks of one row: group j (16 k) of its quarter kq sits at lanebase + 64 j; the groups from jlim on lie beyond
// Hp and are read from group jlim - 1 instead (jl1 = jlim - 1; a lane whose quarter has no valid group at all reads
// quarter 0, group 0).
struct RowPoll {
    unsigned rbase, rstep, qoff, jl1;
    __device__ __forceinline__ void init(bool rok, unsigned base, unsigned step, int kq, int Hp) {
        const int jlim = Hp > 4 * kq ? (Hp - 4 * kq + 15) / 16 : 0;
        rbase = base;   // (an invalid row reads the cluster's row 0: the caller passes that row's base)
        rstep = step;
        qoff = jlim > 0 ? (unsigned)kq * 16u : 0u;
        jl1 = jlim > 0 ? (unsigned)(jlim - 1) : 0u;
        (void)rok;
    }
    __device__ __forceinline__ unsigned lanebase(int step) const { return rbase + (unsigned)step * rstep + qoff; }
    __device__ __forceinline__ unsigned at(unsigned lb, int j) const {
        const unsigned jj = (unsigned)j < jl1 ? (unsigned)j : jl1;
        return lb + (jj << 6);
    }
};

// the MFMAs of one polled batch: N groups x 4 k-steps, two accumulation chains
template <int N>
__device__ __forceinline__ void mfma_batch(const u32x4 (&v)[N], const float (*Bf)[4], f32x4& acc0, f32x4& acc1) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if ((e & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[i][e]), Bf[i][e], acc0, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(v[i][e]), Bf[i][e], acc1, 0, 0, 0);
        }
}

// which wave of a pair moves item k of n between registers and global memory
__host__ __device__ constexpr int owner_of(int k, int n) { return k < (n + 1) / 2 ? 0 : 1; }

// ============================================================================
// forward
// ============================================================================
template <int CELL, int ACT>
__global__ __launch_bounds__(256, 1) void rec4_fwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr bool TWO = pk_cell_two_phase(CELL);
    constexpr int G1 = TWO ? G - 1 : G;  // gates fed by h_{t-1}
    constexpr int NOUT = NS + 1;         // Y, saved slots
    // LDS (floats): per pair and parity: G input patches | 2 x G1 partial sums (| 2 candidate partial sums);
    // per wave: NOUT output patches + 1 publish patch for x_t
    constexpr int PAIR_PAR = (G + 2 * G1 + (TWO ? 2 : 0)) * 256;
    constexpr int WAVE_PRIV = (NOUT + 1) * 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = wave >> 1, kh = wave & 1;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int ubase = p * UW + tile * 16;
    const int unit = ubase + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;

    // ---- recurrent weights of my 16 units, my half of k (once): B1[g][jj][e] = U_g[unit][16 (kh*18 + jj) + 4 kq + e]
    float B1[G1][NJH][4];
    float B2[TWO ? NJH : 1][4];
    {
        const unsigned szU = (unsigned)((size_t)G * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int jj = 0; jj < NJH; ++jj) {
                const int k0 = (kh * NJH + jj) * 16 + kq * 4;
                const unsigned off = (unsigned)(((g * H + unit) * H + k0) * 4);
                // (rows of U are only 4-byte aligned when H is odd: four dword loads; the bounds check answers 0 beyond U)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsU, (unit_ok && k0 + e < H) ? off + 4u * e : szU, 0, 0));
                    if (g < G1) B1[g < G1 ? g : 0][jj][e] = w;
                    else if (TWO) B2[TWO ? jj : 0][e] = w;
                }
            }
    }
    float psc[G], psh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        psc[g] = unit_ok ? a.pscale[g * H + unit] : 0.f;
        psh[g] = unit_ok ? a.pshift[g * H + unit] : 0.f;
    }
    float* const lds = reinterpret_cast<float*>(smem);
    for (int i = tid; i < 2 * 2 * PAIR_PAR + 4 * WAVE_PRIV; i += 256) lds[i] = 0.f;

    // ---- poll geometry: my row of h_{t-1} / x_t (row lane & 15 of the cluster), my groups of k
    const unsigned TS = (unsigned)B * a.Ypitch * 4u;  // bytes per time slab of Yx / Xx
    const unsigned szYx = (unsigned)T * TS;
    RowPoll rp;
    {
        const int row = (lane & 15) < nrows ? (lane & 15) : 0;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        rp.init(true, ((unsigned)b * a.Ypitch + dir * Hp) * 4u + (unsigned)(dir ? (T - 1) : 0) * TS, dir ? 0u - TS : TS, kq, Hp);
    }
    float rvf[4], msk[4], hprev[4], cprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r, n = n_base + row;
        const bool ok = row < nrows && unit_ok;
        rvf[r] = ok ? 1.f : 0.f;
        msk[r] = (a.mask != nullptr && ok) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        hprev[r] = 0.f;
        cprev[r] = 0.f;
    }
    // ---- vector layout: row lane>>2, units ubase + (lane&3)*4 .. +3 (also the publish layout: one 16-byte chunk per lane)
    const int vrow = lane >> 2, vu0 = ubase + (lane & 3) * 4;
    const int vn = n_base + (vrow < nrows ? vrow : 0);
    const int vdir = vn >= B ? 1 : 0, vb = vn - vdir * B;
    int vnv = H - vu0;
    vnv = vnv > 4 ? 4 : (vnv < 0 ? 0 : vnv);
    const int edge = __any(vnv > 0 && vnv < 4) != 0 ? ((H & 1) ? 2 : 1) : 0;
    vnv = vrow < nrows ? vnv : 0;
    const unsigned vP0 = ((unsigned)vb * GH + vu0), vPs = (unsigned)B * GH;
    const unsigned vY0 = ((unsigned)vb * a.YH + vdir * H + vu0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)vdir * T * B + vb) * (NS * H) + vu0), vSs = (unsigned)B * NS * H;
    const bool pk_ok = vrow < nrows && vu0 < Hp;  // (padding units between H and Hp are published as zeros)
    const unsigned pbase = pk_ok ? ((unsigned)vb * a.Ypitch + vdir * Hp + vu0) * 4u : szYx;  // out of range: dropped

    float* const pairm = lds + tile * (2 * PAIR_PAR);                       // [parity][PAIR_PAR]
    float* const priv = lds + 2 * 2 * PAIR_PAR + wave * WAVE_PRIV;           // [NOUT + 1][256]
    float* const patchX = priv + NOUT * 256;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.Yx, szYx);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(TWO ? a.Xx : a.Yx, szYx);
    float* trash = a.trash + (tid & 63) * 4;

    // inputs: the G projections; wave owner_of(g, G) of the pair loads gate g (one step ahead) and puts it into the pair's patch
    f32x4 pv[G];
    auto load_proj = [&](int tt, auto E) {
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (owner_of(g, G) == kh) pv[g] = ld4<decltype(E)::value>(a.P, vP0 + ts * vPs + g * H, vnv);
    };
    // outputs: item 0 = h_t (Y), items 1.. = saved slots; wave owner_of(k, NOUT) stores item k (one step behind)
    auto flush_outputs = [&](int tt, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
        if (owner_of(0, NOUT) == kh) st4<EE>(a.Y, vY0 + ts * vYs, vnv, trash, patch_get_vec(priv, lane));
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (owner_of(k + 1, NOUT) == kh) st4<EE>(a.S, vS0 + ts * vSs + k * H, vnv, trash, patch_get_vec(priv + (k + 1) * 256, lane));
    };
#define PK4_LP0(E) load_proj(0, E)
    PK_EDGE_DISPATCH(PK4_LP0);
    __syncthreads();

    bool dead = false;
    const bool fast = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd4(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    // diagnostics (pk_persist2_set_empty_step; timing only, results are garbage): bit 0 = no MFMAs, bit 1 = no polls
    const bool no_mfma = (a.empty_step & 1) != 0;
    if ((a.empty_step & 2) != 0) dead = true;
    for (int t = 0; t < T; ++t) {
        float* const pm = pairm + (t & 1) * PAIR_PAR;
        float* const pin = pm;                                    // [G][256] projections of this step
        float* const xs1 = pm + G * 256;                          // [2 waves][G1][256] partial sums, phase 1
        float* const xs2 = pm + (G + 2 * G1) * 256;               // [2 waves][256] partial sums, phase 2
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (owner_of(g, G) == kh) patch_put_vec(pin + g * 256, lane, pv[g]);
        f32x4 acc[G1][2];
#pragma unroll
        for (int g = 0; g < G1; ++g) acc[g][0] = acc[g][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 av[NJH];
        if (t > 0) {
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
            const unsigned lb = rp.lanebase(t - 1);
            auto off = [&](int i) { return rp.at(lb, kh * NJH + i); };
            dead = fast ? poll_regs<NJH, true>(rs, off, av, a.err, a.spin_limit, lane, dead)
                        : poll_regs<NJH, false>(rs, off, av, a.err, a.spin_limit, lane, dead);
        }
        // off the dependency chain, behind the poll: fp32 outputs of the previous step, projections of the next one
        if (t > 0) {
#define PK4_FO(E) flush_outputs(t - 1, E)
            PK_EDGE_DISPATCH(PK4_FO);
        }
        if (t + 1 < T) {
#define PK4_LP1(E) load_proj(t + 1, E)
            PK_EDGE_DISPATCH(PK4_LP1);
        }
        if (t > 0 && !no_mfma) {
#pragma unroll
            for (int g = 0; g < G1; ++g) mfma_batch<NJH>(av, B1[g], acc[g][0], acc[g][1]);
        }
        // ---- the pair's partial sums meet in LDS: both waves end with the same totals (a + b == b + a)
        float sum1[G1][4];
#pragma unroll
        for (int g = 0; g < G1; ++g) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sum1[g][r] = acc[g][0][r] + acc[g][1][r];
            patch_put_cd(xs1 + (kh * G1 + g) * 256, kq, lane, sum1[g]);
        }
        PK_BARRIER_LDS();
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            float o[4];
            patch_get_cd(xs1 + ((kh ^ 1) * G1 + g) * 256, kq, lane, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) sum1[g][r] = kh == 0 ? sum1[g][r] + o[r] : o[r] + sum1[g][r];  // (same operand order in both waves)
        }
        float pre[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) patch_get_cd(pin + g * 256, kq, lane, pre[g]);
        float hv[4], sv[NS][4];
        if constexpr (!TWO) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr[G];
#pragma unroll
                for (int g = 0; g < G; ++g) pr[g] = pre[g][r] * psc[g] + psh[g] + sum1[g][r];
                float h, cc, s[NS];
                pk_cell_fwd<CELL>(act, pr, hprev[r], cprev[r], msk[r], h, cc, s);
                h = rvf[r] != 0.f ? h : 0.f;  // rows / units outside the layer carry exact zeros (published as padding)
                cc = rvf[r] != 0.f ? cc : 0.f;
                hprev[r] = h;
                cprev[r] = cc;
                hv[r] = h;
#pragma unroll
                for (int k = 0; k < NS; ++k) sv[k][r] = s[k];
            }
        } else {
            // ---- phase 1: the gates that depend on h_{t-1} only; x_t = r*h (GRU) / z*h (minimalGRU) goes to the cluster
            float xv[4], zt[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr[G1], s[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) s[k] = 0.f;
#pragma unroll
                for (int g = 0; g < G1; ++g) pr[g] = pre[g][r] * psc[g] + psh[g] + sum1[g][r];
                const float x = pk_cell_fwd_p1<CELL>(pr, hprev[r], s);
                xv[r] = rvf[r] != 0.f ? x : 0.f;
                zt[r] = s[0];
#pragma unroll
                for (int k = 0; k < NS; ++k) sv[k][r] = s[k];  // (the slots phase 1 fills: z (, r), r*h / z*h)
                sv[NS - 1][r] = xv[r];
            }
            if (kh == 0) {
                patch_put_cd(patchX, kq, lane, xv);
                PK_LDS_ORDER();
                const u32x4 o = no_sentinel4(patch_get_vec(patchX, lane));
                const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - t) : t) * TS : 0u);
                if (fast) pub_store<true>(rsX, off, o);
                else pub_store<false>(rsX, off, o);
            }
            // ---- phase 2: a_t = Wh_t + x_t . U_h^T
            f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
            {
                const unsigned lb = rp.lanebase(t);
                auto off = [&](int i) { return rp.at(lb, kh * NJH + i); };
                dead = fast ? poll_regs<NJH, true>(rsX, off, av, a.err, a.spin_limit, lane, dead)
                            : poll_regs<NJH, false>(rsX, off, av, a.err, a.spin_limit, lane, dead);
            }
            if (!no_mfma) mfma_batch<NJH>(av, B2, a0, a1);
            float sum2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) sum2[r] = a0[r] + a1[r];
            patch_put_cd(xs2 + kh * 256, kq, lane, sum2);
            PK_BARRIER_LDS();
            {
                float o[4];
                patch_get_cd(xs2 + (kh ^ 1) * 256, kq, lane, o);
#pragma unroll
                for (int r = 0; r < 4; ++r) sum2[r] = kh == 0 ? sum2[r] + o[r] : o[r] + sum2[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float at = pre[G - 1][r] * psc[G - 1] + psh[G - 1] + sum2[r];
                float h = pk_cell_fwd_p2<CELL>(act, at, zt[r], hprev[r], msk[r]);
                h = rvf[r] != 0.f ? h : 0.f;
                hprev[r] = h;
                hv[r] = h;
                sv[G - 1][r] = at;  // GRU: slot 2, minimalGRU: slot 1
            }
        }
        // ---- my share of the outputs into my patches; wave 0 of the pair publishes h_t: one 16-byte store per lane
        if (owner_of(0, NOUT) == kh) patch_put_cd(priv, kq, lane, hv);
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (owner_of(k + 1, NOUT) == kh) patch_put_cd(priv + (k + 1) * 256, kq, lane, sv[k]);
        PK_LDS_ORDER();
        if (kh == 0) {  // (owner_of(0, .) == 0: Y sits in wave 0's patch)
            const u32x4 o = no_sentinel4(patch_get_vec(priv, lane));
            const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - t) : t) * TS : 0u);
            if (fast) pub_store<true>(rs, off, o);
            else pub_store<false>(rs, off, o);
        }
    }
#define PK4_FOL(E) flush_outputs(T - 1, E)
    PK_EDGE_DISPATCH(PK4_FOL);
}

// ============================================================================
// backward: dL/dh_{t-1} = direct + [dgates_t] . [U_0; U_1; ...]
// ============================================================================
template <int CELL, int ACT>
__global__ __launch_bounds__(256, 1) void rec4_bwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr bool TWO = pk_cell_two_phase(CELL);
    constexpr bool LSTM = CELL == PK_CELL_LSTM;
    constexpr int Gh = TWO ? G - 1 : G;        // gates whose gradients go back through h_{t-1}
    constexpr int NJB = Gh * NJH;              // groups of the carry product one wave holds
    constexpr int NBQ = Gh >= 4 ? NJH / 2 : NJH;  // groups polled at a time (LSTM: 288 registers of U leave room for 9 chunks)
    constexpr int NBATCH = NJB / NBQ;
    // inputs of a step: saved slots, then (LSTM: c_{t-1}; the others: h_{t-1}), then dY
    constexpr int NIN = NS + 2;
    constexpr int PAIR_PAR = (NIN + 2 + (TWO ? 2 : 0)) * 256;  // input patches | carry partial sums (| q partial sums)
    constexpr int WAVE_PRIV = G * 256;                          // fp32 gate gradients (my share), also the publish patches
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = wave >> 1, kh = wave & 1;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const unsigned TB = (unsigned)T * B;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int ubase = p * UW + tile * 16;
    const int unit = ubase + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;

    // carry product: group jg = kh * NJB + jj -> gate g = jg / KJ, k = 16 (jg % KJ) + 4 kq + e: BB[jj][e] = U_g[k][unit]
    // two-phase cells: BA[jj][e] = U_{G-1}[16 (kh*18 + jj) + 4 kq + e][unit]  (q = da . U_h)
    float BB[NJB][4];
    float BA[TWO ? NJH : 1][4];
    {
        const unsigned szU = (unsigned)((size_t)G * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int jj = 0; jj < NJB; ++jj) {
            const int jg = kh * NJB + jj, g = jg / KJ, j = jg % KJ;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = j * 16 + kq * 4 + e;
                BB[jj][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    rsU, (unit_ok && k < H) ? (unsigned)(((g * H + k) * H + unit) * 4) : szU, 0, 0));  // out of range: 0
            }
        }
        if (TWO) {
#pragma unroll
            for (int jj = 0; jj < NJH; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = (kh * NJH + jj) * 16 + kq * 4 + e;
                    BA[TWO ? jj : 0][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rsU, (unit_ok && k < H) ? (unsigned)((((G - 1) * H + k) * H + unit) * 4) : szU, 0, 0));
                }
        }
    }
    float* const lds = reinterpret_cast<float*>(smem);
    for (int i = tid; i < 2 * 2 * PAIR_PAR + 4 * WAVE_PRIV; i += 256) lds[i] = 0.f;

    // ---- poll geometry: my row of the cluster's gate gradients
    const unsigned TS = (unsigned)B * a.Gpitch * 4u;  // bytes per time slab of dGx
    const unsigned ndir = (unsigned)(a.R / B);
    const unsigned szGx = ndir * (unsigned)T * TS;
    RowPoll rp;  // slab of step T-1 in loop order (it = 0), one slab per iteration; gate g of a row starts at g * Hp floats
    {
        const int row = (lane & 15) < nrows ? (lane & 15) : 0;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        rp.init(true, (unsigned)dir * (unsigned)T * TS + (unsigned)b * a.Gpitch * 4u + (unsigned)(dir ? 0 : (T - 1)) * TS,
                dir ? TS : 0u - TS, kq, Hp);
    }
    const unsigned gate_bytes = (unsigned)Hp * 4u;
    float rvf[4], msk[4], dh_dir[4], dc_car[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r, n = n_base + row;
        const bool ok = row < nrows && unit_ok;
        rvf[r] = ok ? 1.f : 0.f;
        msk[r] = (a.mask != nullptr && ok) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        dh_dir[r] = 0.f;
        dc_car[r] = 0.f;
    }
    const int vrow = lane >> 2, vu0 = ubase + (lane & 3) * 4;
    const int vn = n_base + (vrow < nrows ? vrow : 0);
    const int vdir = vn >= B ? 1 : 0, vb = vn - vdir * B;
    int vnv = H - vu0;
    vnv = vnv > 4 ? 4 : (vnv < 0 ? 0 : vnv);
    const int edge = __any(vnv > 0 && vnv < 4) != 0 ? ((H & 1) ? 2 : 1) : 0;
    vnv = vrow < nrows ? vnv : 0;
    const unsigned vY0 = ((unsigned)vb * a.YH + vdir * H + vu0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)vdir * TB + vb) * (NS * H) + vu0), vSs = (unsigned)B * NS * H;
    const unsigned vG0 = (((unsigned)vdir * TB + vb) * GH + vu0), vGs = (unsigned)B * GH;
    const bool pk_ok = vrow < nrows && vu0 < Hp;
    const unsigned pbase = pk_ok ? (unsigned)vdir * (unsigned)T * TS + ((unsigned)vb * a.Gpitch + vu0) * 4u : szGx;

    float* const pairm = lds + tile * (2 * PAIR_PAR);
    float* const priv = lds + 2 * 2 * PAIR_PAR + wave * WAVE_PRIV;  // [G][256]
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dGx, szGx);
    float* trash = a.trash + (tid & 63) * 4;

    // gate gradient g is published and stored by wave gate_owner(g) of the pair
    // (two-phase cells: da - slot G-1 - leaves in phase A from wave 0; GRU: dz wave 0, dr wave 1; minimalGRU: dz wave 1)
    auto gate_owner = [](int g) constexpr { return TWO ? (g == G - 1 ? 0 : (G == 3 ? g : 1)) : owner_of(g, G); };
    f32x4 iv[NIN];
    auto load_step_e = [&](int t, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(vdir ? (T - 1 - t) : t);
        const unsigned tp = t > 0 ? (vdir ? ts + 1 : ts - 1) : ts;
        const int nvp = t > 0 ? vnv : 0;
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (owner_of(k, NIN) == kh) iv[k] = ld4<EE>(a.S, vS0 + ts * vSs + k * H, vnv);
        if (owner_of(NS, NIN) == kh) {
            if (LSTM) iv[NS] = ld4<EE>(a.S, vS0 + tp * vSs + 4 * H, nvp);   // c_{t-1}: slot 4 of the previous step
            else iv[NS] = ld4<EE>(a.Y, vY0 + tp * vYs, nvp);                // h_{t-1}
            if (t == 0) iv[NS] = f32x4{0.f, 0.f, 0.f, 0.f};                 // c_{-1} = h_{-1} = 0
        }
        if (owner_of(NS + 1, NIN) == kh) iv[NS + 1] = ld4<EE>(a.dY, vY0 + ts * vYs, vnv);
    };
    auto flush_outputs_e = [&](int tt, auto E) {
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (gate_owner(g) == kh) st4<decltype(E)::value>(a.dP2, vG0 + ts * vGs + g * H, vnv, trash, patch_get_vec(priv + g * 256, lane));
    };
    auto publish_gate = [&](int g, int t, bool fast_) {
        const u32x4 o = no_sentinel4(patch_get_vec(priv + g * 256, lane));
        const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - t) : t) * TS + (unsigned)(g * Hp) * 4u : 0u);
        if (fast_) pub_store<true>(rs, off, o);
        else pub_store<false>(rs, off, o);
    };
#define PK4_LS(E) load_step_e(T - 1, E)
    PK_EDGE_DISPATCH(PK4_LS);
    __syncthreads();

    bool dead = false;
    const bool fast = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd4(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    const bool no_mfma = (a.empty_step & 1) != 0;  // diagnostics, as in the forward kernel
    if ((a.empty_step & 2) != 0) dead = true;
    int it = 0;
    for (int t = T - 1; t >= 0; --t, ++it) {
        float* const pm = pairm + (it & 1) * PAIR_PAR;
        float* const pin = pm;                          // [NIN][256]
        float* const xsB = pm + NIN * 256;              // [2][256] carry partial sums
        float* const xsA = pm + (NIN + 2) * 256;        // [2][256] q partial sums (two-phase cells)
#pragma unroll
        for (int k = 0; k < NIN; ++k)
            if (owner_of(k, NIN) == kh) patch_put_vec(pin + k * 256, lane, iv[k]);
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        if (t < T - 1) {
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
            const unsigned lb0 = rp.lanebase(it - 1);
#pragma unroll
            for (int bq = 0; bq < NBATCH; ++bq) {
                if (NBATCH > 2) __builtin_amdgcn_sched_barrier(0);  // (LSTM: keep one batch of chunks live at a time - 288 registers hold U)
                u32x4 av[NBQ];
                // batch bq of wave kh = groups jg = kh * NJB + bq * NBQ .. + NBQ - 1: gate jg / KJ, group jg % KJ
                const int jg0 = kh * NJB + bq * NBQ;
                const unsigned lb = lb0 + (unsigned)(jg0 / KJ) * gate_bytes;
                const int j0 = jg0 % KJ;
                auto off = [&](int i) { return rp.at(lb, j0 + i); };
                dead = fast ? poll_regs<NBQ, true>(rs, off, av, a.err, a.spin_limit, lane, dead)
                            : poll_regs<NBQ, false>(rs, off, av, a.err, a.spin_limit, lane, dead);
                if (bq == 0) {
                    // off the dependency chain, behind the first poll: fp32 gate gradients of the previous step, saved tensors of the next
#define PK4_FOB(E) flush_outputs_e(t + 1, E)
                    PK_EDGE_DISPATCH(PK4_FOB);
                    if (t > 0) {
#define PK4_LS1(E) load_step_e(t - 1, E)
                        PK_EDGE_DISPATCH(PK4_LS1);
                    }
                }
                if (!no_mfma) mfma_batch<NBQ>(av, &BB[bq * NBQ], acc0, acc1);
            }
        } else if (t > 0) {
#define PK4_LS2(E) load_step_e(t - 1, E)
            PK_EDGE_DISPATCH(PK4_LS2);
        }
        float car[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) car[r] = acc0[r] + acc1[r];
        patch_put_cd(xsB + kh * 256, kq, lane, car);
        PK_BARRIER_LDS();
        {
            float o[4];
            patch_get_cd(xsB + (kh ^ 1) * 256, kq, lane, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) car[r] = kh == 0 ? car[r] + o[r] : o[r] + car[r];
        }
        float sin[NIN][4];
#pragma unroll
        for (int k = 0; k < NIN; ++k) patch_get_cd(pin + k * 256, kq, lane, sin[k]);
        float dgv[G][4];
        if constexpr (!TWO) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) s[k] = sin[k][r];
                const float prev = sin[NS][r];  // LSTM: c_{t-1}
                const float dh = sin[NS + 1][r] + dh_dir[r] + car[r];
                float dg[G], dhd, dcp;
                pk_cell_bwd<CELL>(act, s, LSTM ? 0.f : prev, LSTM ? prev : 0.f, msk[r], dh, dc_car[r], dg, dhd, dcp);
                dh_dir[r] = rvf[r] != 0.f ? dhd : 0.f;
                dc_car[r] = rvf[r] != 0.f ? dcp : 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) dgv[g][r] = rvf[r] != 0.f ? dg[g] : 0.f;
            }
        } else {
            // ---- phase A: da_t (the operand of q = da . U_h) goes to the cluster
            float da4[4], dzp[4], dhd4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) s[k] = sin[k][r];
                const float dh = sin[NS + 1][r] + dh_dir[r] + car[r];
                const float da = pk_cell_bwd_pa<CELL>(act, s, sin[NS][r], msk[r], dh, dzp[r], dhd4[r]);
                da4[r] = rvf[r] != 0.f ? da : 0.f;
            }
            if (kh == 0) {
                patch_put_cd(priv + (G - 1) * 256, kq, lane, da4);
                PK_LDS_ORDER();
                publish_gate(G - 1, t, fast);
            }
            f32x4 q0 = f32x4{0.f, 0.f, 0.f, 0.f}, q1 = q0;
            {
                u32x4 av[NJH];
                const unsigned lb = rp.lanebase(it) + (unsigned)(G - 1) * gate_bytes;
                auto off = [&](int i) { return rp.at(lb, kh * NJH + i); };
                dead = fast ? poll_regs<NJH, true>(rs, off, av, a.err, a.spin_limit, lane, dead)
                            : poll_regs<NJH, false>(rs, off, av, a.err, a.spin_limit, lane, dead);
                if (!no_mfma) mfma_batch<NJH>(av, BA, q0, q1);
            }
            float q[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] = q0[r] + q1[r];
            patch_put_cd(xsA + kh * 256, kq, lane, q);
            PK_BARRIER_LDS();
            {
                float o[4];
                patch_get_cd(xsA + (kh ^ 1) * 256, kq, lane, o);
#pragma unroll
                for (int r = 0; r < 4; ++r) q[r] = kh == 0 ? q[r] + o[r] : o[r] + q[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) s[k] = sin[k][r];
                float dg[G];
                float dhd = dhd4[r];
                pk_cell_bwd_pb<CELL>(s, sin[NS][r], q[r], da4[r], dzp[r], dg, dhd);
                dh_dir[r] = rvf[r] != 0.f ? dhd : 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) dgv[g][r] = rvf[r] != 0.f ? dg[g] : 0.f;
            }
        }
        // ---- my share of the gate gradients through my patches into the vector layout, then publish what the next
        // step's carry needs: one 16-byte store per gate
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (gate_owner(g) == kh && !(TWO && g == G - 1)) patch_put_cd(priv + g * 256, kq, lane, dgv[g]);
        PK_LDS_ORDER();
#pragma unroll
        for (int g = 0; g < Gh; ++g)
            if (gate_owner(g) == kh) publish_gate(g, t, fast);
    }
#define PK4_FOBL(E) flush_outputs_e(0, E)
    PK_EDGE_DISPATCH(PK4_FOBL);
}

typedef void (*Rec4Kernel)(R2Args);
template <int CELL>
Rec4Kernel pick4_fwd(int act) {
    return act == PK_ACT_TANH ? rec4_fwd_kernel<CELL, PK_ACT_TANH> : act == PK_ACT_RELU ? rec4_fwd_kernel<CELL, PK_ACT_RELU>
                                                                                       : rec4_fwd_kernel<CELL, -1>;
}
template <int CELL>
Rec4Kernel pick4_bwd(int act) {
    return act == PK_ACT_TANH ? rec4_bwd_kernel<CELL, PK_ACT_TANH> : act == PK_ACT_RELU ? rec4_bwd_kernel<CELL, PK_ACT_RELU>
                                                                                       : rec4_bwd_kernel<CELL, -1>;
}
Rec4Kernel pick4(int cell, int act, bool backward) {
    switch (cell) {
        case PK_CELL_LSTM: return backward ? pick4_bwd<PK_CELL_LSTM>(act) : pick4_fwd<PK_CELL_LSTM>(act);
        case PK_CELL_GRU: return backward ? pick4_bwd<PK_CELL_GRU>(act) : pick4_fwd<PK_CELL_GRU>(act);
        default: return backward ? pick4_bwd<PK_CELL_MINGRU>(act) : pick4_fwd<PK_CELL_MINGRU>(act);
    }
}

size_t lds4(int cell, bool backward) {
    const int G = pk_cell_gates(cell), NS = pk_cell_saved(cell);
    const bool two = pk_cell_two_phase(cell);
    if (!backward) {
        const int G1 = two ? G - 1 : G;
        return ((size_t)2 * 2 * (G + 2 * G1 + (two ? 2 : 0)) * 256 + 4 * (size_t)(NS + 2) * 256) * 4;
    }
    return ((size_t)2 * 2 * (NS + 2 + 2 + (two ? 2 : 0)) * 256 + 4 * (size_t)G * 256) * 4;
}

int grant_lds4(Rec4Kernel k, size_t lds) {
    struct Entry { Rec4Kernel k; size_t lds; };
    static Entry granted[32];
    static int n = 0;
    for (int i = 0; i < n; ++i)
        if (granted[i].k == k && granted[i].lds >= lds) return 0;
    PK_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (n < 32) granted[n++] = Entry{k, lds};
    return 0;
}

// 32 units per workgroup; as many 16-row clusters per launch as fit the device, a multiple of 8 when there are 8 or more
// (members of one cluster congruent mod 8: one XCD under round-robin dispatch - speed only)
int make_plan4(int R, int H, Plan2& pl) {
    pl.Pn = (H + UW - 1) / UW;
    PK_REQUIRE(pl.Pn <= HS4, "persistent fp32 recurrence: H=%d needs %d workgroups per cluster (<= %d)", H, pl.Pn, HS4);
    const int ncu = pk_num_cu();
    int C = ncu / pl.Pn;
    PK_REQUIRE(C >= 1, "persistent fp32 recurrence: H=%d needs %d workgroups per cluster but the device has %d CUs", H, pl.Pn, ncu);
    if (C >= 8) C -= C % 8;
    const int need = (R + RMAX - 1) / RMAX;
    if (need < C) C = need;
    pl.C = C;
    pl.rpc = RMAX;
    pl.launches = (need + C - 1) / C;
    return 0;
}

}  // namespace

int pk_rec4f_covers(int cell, int H) {
    static int off = -1;
    if (off < 0) {
        const char* e = pk_experiment("rec_f32_gen4");  // 0 = keep the first-generation LSTM kernels / the step-wise GRU (A/B measurements)
        off = (e && e[0] == '0') ? 1 : 0;
    }
    return !off && (cell == PK_CELL_LSTM || cell == PK_CELL_GRU || cell == PK_CELL_MINGRU) && H >= 1 && H <= KPAD;
}
// floats of exchange buffer one call needs: forward T*B rows x ndir*Hp (twice for the two-phase cells: h and r*h / z*h),
// backward ndir*T*B rows x G*Hp (pitches rounded up to 16 floats); the larger of the two serves both passes
int64_t pk_rec4f_exchange_floats(int cell, int T, int B, int bidir, int H) {
    if (!pk_rec4f_covers(cell, H)) return 0;
    const int64_t ndir = 1 + bidir, Hp = (H + 7) & ~7, G = pk_cell_gates(cell);
    const int64_t yp = (ndir * Hp + 15) / 16 * 16, gp = (G * Hp + 15) / 16 * 16;
    const int64_t f = (int64_t)T * B * yp * (pk_cell_two_phase(cell) ? 2 : 1), b = ndir * T * B * gp;
    return (f > b ? f : b) + 64;
}

int pk_rec4f_fwd(hipStream_t st, int cell, int act, int T, int B, int bidir, int H, const float* P, const float* pscale,
                 const float* pshift, const float* U, const float* mask, float mask_scalar, float* Y, float* S, float* Yx) {
    int rc = pk_rec2_check("pk_rec_fwd (fp32, persistent, generation 4)", pk_rec4f_covers(cell, H), cell, T, B, bidir, H);
    if (rc) return rc;
    const int ndir = 1 + bidir, R = B * ndir, Hp = (H + 7) & ~7;
    const bool two = pk_cell_two_phase(cell);
    const int64_t y_pitch = ((int64_t)ndir * Hp + 15) / 16 * 16;
    const size_t slab = (size_t)T * B * y_pitch * 4;
    PK_REQUIRE(((uintptr_t)Yx & 15) == 0, "pk_rec_fwd: exchange buffer must be 16-byte aligned");
    PK_REQUIRE((double)slab < 4.0e9, "pk_rec_fwd (fp32, persistent): exchange buffer exceeds the 4 GB buffer-descriptor range");
    Plan2 pl;
    rc = make_plan4(R, H, pl);
    if (rc) return rc;
    R2Args a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.Hp = Hp; a.YH = ndir * H; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.rpc = pl.rpc; a.row0 = 0;
    a.P = P; a.pscale = pscale; a.pshift = pshift; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = Y; a.S = S; a.Yb = nullptr; a.Xb = nullptr; a.Ypitch = (int)y_pitch;
    a.dY = nullptr; a.dP2 = nullptr; a.dGb = nullptr; a.Gpitch = 0;
    a.Yx = Yx; a.dGx = nullptr; a.Xx = two ? Yx + slab / 4 : nullptr;
    a.ln_gamma = nullptr; a.ln_beta = nullptr;
    rc = pk_rec2_host_setup(a, false, cell);
    if (rc) return rc;
    PK_CHECK_HIP(hipMemsetAsync(Yx, 0xFF, slab * (two ? 2 : 1), st));  // the mailboxes: every dword "not written yet"
    const size_t lds = lds4(cell, false);
    const Rec4Kernel k = pick4(cell, act, false);
    rc = grant_lds4(k, lds);
    if (rc) return rc;
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        rc = pk_rec2_check_residency((const void*)k, 256, lds, pl.C * pl.Pn, "pk_rec_fwd (fp32, persistent, generation 4)");
        if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(pl.C * pl.Pn), dim3(256), lds, st, a);
        PK_LAUNCH_CHECK();
    }
    return 0;
}

int pk_rec4f_bwd(hipStream_t st, int cell, int act, int T, int B, int bidir, int H, const float* U, const float* mask,
                 float mask_scalar, const float* Y, const float* S, const float* dY, float* dP2, float* dGx) {
    int rc = pk_rec2_check("pk_rec_bwd (fp32, persistent, generation 4)", pk_rec4f_covers(cell, H), cell, T, B, bidir, H);
    if (rc) return rc;
    const int ndir = 1 + bidir, R = B * ndir, Hp = (H + 7) & ~7, G = pk_cell_gates(cell);
    const int64_t g_pitch = ((int64_t)G * Hp + 15) / 16 * 16;
    const size_t bytes = (size_t)ndir * T * B * g_pitch * 4;
    PK_REQUIRE(((uintptr_t)dGx & 15) == 0, "pk_rec_bwd: exchange buffer must be 16-byte aligned");
    PK_REQUIRE((double)bytes < 4.0e9, "pk_rec_bwd (fp32, persistent): exchange buffer exceeds the 4 GB buffer-descriptor range");
    Plan2 pl;
    rc = make_plan4(R, H, pl);
    if (rc) return rc;
    R2Args a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.Hp = Hp; a.YH = ndir * H; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.rpc = pl.rpc; a.row0 = 0;
    a.P = nullptr; a.pscale = nullptr; a.pshift = nullptr; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = const_cast<float*>(Y); a.S = const_cast<float*>(S); a.Yb = nullptr; a.Xb = nullptr; a.Ypitch = 0;
    a.dY = dY; a.dP2 = dP2; a.dGb = nullptr; a.Gpitch = (int)g_pitch;
    a.Yx = nullptr; a.dGx = dGx; a.Xx = nullptr;
    a.ln_gamma = nullptr; a.ln_beta = nullptr;
    rc = pk_rec2_host_setup(a, true, cell);
    if (rc) return rc;
    PK_CHECK_HIP(hipMemsetAsync(dGx, 0xFF, bytes, st));
    const size_t lds = lds4(cell, true);
    const Rec4Kernel k = pick4(cell, act, true);
    rc = grant_lds4(k, lds);
    if (rc) return rc;
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        rc = pk_rec2_check_residency((const void*)k, 256, lds, pl.C * pl.Pn, "pk_rec_bwd (fp32, persistent, generation 4)");
        if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(pl.C * pl.Pn), dim3(256), lds, st, a);
        PK_LAUNCH_CHECK();
    }
    return 0;
}
