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
//   * K is split over the FOUR waves of a workgroup.  A workgroup is 32 hidden units = two tiles of 16; wave w holds the B
//     fragments of BOTH tiles for quarter w of the reduction (LSTM: 2 tiles x 4 gates x 144 k = 288 registers; GRU: 216),
//     the four partial sums meet in LDS (one workgroup barrier per MFMA phase) and are added in a fixed order; the gate
//     math of a tile is then split by ROWS between two waves, so state never has to be handed over and nothing is
//     computed twice.
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
constexpr int UW = 32;           // hidden units per workgroup (two pairs of waves)
constexpr int HS4 = 32;          // placement-handshake words per cluster (up to 18 members)
constexpr int NPL = 5;           // per-step LayerNorm: 16-slot groups of the row-statistics exchange a lane polls (4 x 18 slots)

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

// The second half of poll_regs for loads that were issued earlier (two batches in flight): re-load what has not arrived.
template <int N, bool FAST, typename OffFn>
__device__ __forceinline__ bool settle_regs(__amdgpu_buffer_rsrc_t rs, OffFn off, u32x4 (&v)[N], unsigned* err, int spin_limit,
                                            int lane, bool dead) {
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

// ---- who does what in a workgroup (256 threads = 4 waves, 32 hidden units = 2 tiles of 16):
//   MFMA phase   wave w holds K-QUARTER w (groups 9 w .. 9 w + 8 of every gate's 36) of BOTH tiles' B fragments: it polls
//                only its quarter of the rows (no chunk is read twice by a workgroup) and feeds each polled register to
//                the MFMAs of both tiles;
//   reduction    every wave writes its partial sums of both tiles to LDS, one workgroup barrier, every wave adds the four
//                partials of ITS share in the fixed order 0, 1, 2, 3;
//   gate math    wave w owns tile w >> 1, accumulator row slots r = 2 (w & 1), 2 (w & 1) + 1 of every lane - 8 of the
//                cluster's 16 rows x 16 units: state, loads, stores and the publish of that share are its alone.
constexpr int NQ = KJ / 4;   // 9 groups of 16 k: a wave's quarter of one gate

// diagnostics (pk_persist2_set_trace, tools/trace_rec4.py): shader-clock stamps of (workgroup 0, thread 0), 8 per step
#define PK4_TRACE(step, slot)                                                                                      \
    do {                                                                                                           \
        if (a.trace != nullptr && blockIdx.x == 0 && tid == 0) a.trace[(long)(step) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// my two accumulator row slots of a [16 rows][16 units] patch, MFMA C/D layout: rows kq*4 + 2 rh + s, unit lane & 15
__device__ __forceinline__ void put_cd2(float* patch, int kq, int rh, int lane, const float (&v)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) patch[(kq * 4 + 2 * rh + s) * 16 + (lane & 15)] = v[s];
}
__device__ __forceinline__ void get_cd2(const float* patch, int kq, int rh, int lane, float (&v)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) v[s] = patch[(kq * 4 + 2 * rh + s) * 16 + (lane & 15)];
}
// vector layout of my share: lanes 0..31 -> (my row vr = lane >> 2 of 8, units (lane & 3) * 4 .. + 3); patch row of vr
__device__ __forceinline__ int share_row(int vr, int rh) { return (vr >> 1) * 4 + 2 * rh + (vr & 1); }

// ============================================================================
// forward
// ============================================================================
// LN: per-step LayerNorm of h_t (neural_networks.py:23-33 at :466-467, :638-639, :1300-1301): the row statistics take one more
// exchange inside the step (ln_row_allreduce, pk_rec2_common.h: every wave publishes the partial sums of the rows of its
// share - zeros for the rows of its partner wave - in its own slot, 4 x 18 slots per row quad); the normalised h_t is what is
// stored, published and fed back, the pre-LN value and (mean, 1 / (std + eps)) are saved for the backward pass.
template <int CELL, int ACT, bool LN = false>
__global__ __launch_bounds__(256, 1) void rec4_fwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr bool TWO = pk_cell_two_phase(CELL);
    constexpr int G1 = TWO ? G - 1 : G;  // gates fed by h_{t-1}
    constexpr int NOUT = NS + 1;         // Y, saved slots
    // LDS (floats): [parity][wave][tile][G1] partial sums (| [parity][wave][tile] candidate partial sums);
    // per wave: G input patches | NOUT output patches | 1 publish patch for x_t (| LN: the pre-LN h_t)
    constexpr int XS1 = 4 * 2 * G1 * 256, XS2 = TWO ? 4 * 2 * 256 : 0;
    constexpr int WAVE_PRIV = (G + NOUT + 1 + (LN ? 1 : 0)) * 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = wave >> 1, rh = wave & 1;  // my gate-math share: tile mt, row slots 2 rh, 2 rh + 1
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int kq = lane >> 4;

    // ---- recurrent weights, my quarter of k, both tiles (once): B1[tl][g][i][e] = U_g[unit(tl)][16 (9 w + i) + 4 kq + e]
    float B1[2][G1][NQ][4];
    float B2[2][TWO ? NQ : 1][4];
    {
        const unsigned szU = (unsigned)((size_t)G * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int un = p * UW + tl * 16 + (lane & 15);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const int k0 = (wave * NQ + i) * 16 + kq * 4;
                    const unsigned off = (unsigned)(((g * H + un) * H + k0) * 4);
                    // (rows of U are only 4-byte aligned when H is odd: four dword loads; the bounds check answers 0 beyond U)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsU, (un < H && k0 + e < H) ? off + 4u * e : szU, 0, 0));
                        if (g < G1) B1[tl][g < G1 ? g : 0][i][e] = w;
                        else if (TWO) B2[tl][TWO ? i : 0][e] = w;
                    }
                }
        }
    }
    const int ubase = p * UW + mt * 16;
    const int unit = ubase + (lane & 15);
    const bool unit_ok = unit < H;
    float psc[G], psh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        psc[g] = unit_ok ? a.pscale[g * H + unit] : 0.f;
        psh[g] = unit_ok ? a.pshift[g * H + unit] : 0.f;
    }
    float* const lds = reinterpret_cast<float*>(smem);
    for (int i = tid; i < 2 * (XS1 + XS2) + 4 * WAVE_PRIV; i += 256) lds[i] = 0.f;

    // ---- poll geometry: row lane & 15 of the cluster, my quarter of k
    const unsigned TS = (unsigned)B * a.Ypitch * 4u;  // bytes per time slab of Yx / Xx
    const unsigned szYx = (unsigned)T * TS;
    RowPoll rp;
    {
        const int row = (lane & 15) < nrows ? (lane & 15) : 0;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        rp.init(true, ((unsigned)b * a.Ypitch + dir * Hp) * 4u + (unsigned)(dir ? (T - 1) : 0) * TS, dir ? 0u - TS : TS, kq, Hp);
    }
    float rvf[2], msk[2], hprev[2], cprev[2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        const int row = kq * 4 + 2 * rh + s_, n = n_base + row;
        const bool ok = row < nrows && unit_ok;
        rvf[s_] = ok ? 1.f : 0.f;
        msk[s_] = (a.mask != nullptr && ok) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        hprev[s_] = 0.f;
        cprev[s_] = 0.f;
    }
    // ---- vector layout of my share (lanes 0..31): also the publish layout, one 16-byte chunk per lane
    const bool vlane = lane < 32;
    const int vr = (lane >> 2) & 7, vrow = share_row(vr, rh), vu0 = ubase + (lane & 3) * 4;
    const bool vrow_ok = vlane && vrow < nrows;
    const int vn = n_base + (vrow_ok ? vrow : 0);
    const int vdir = vn >= B ? 1 : 0, vb = vn - vdir * B;
    int vnv = H - vu0;
    vnv = vnv > 4 ? 4 : (vnv < 0 ? 0 : vnv);
    const int edge = __any(vnv > 0 && vnv < 4) != 0 ? ((H & 1) ? 2 : 1) : 0;
    vnv = vrow_ok ? vnv : 0;
    const unsigned vP0 = ((unsigned)vb * GH + vu0), vPs = (unsigned)B * GH;
    const unsigned vY0 = ((unsigned)vb * a.YH + vdir * H + vu0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)vdir * T * B + vb) * (NS * H) + vu0), vSs = (unsigned)B * NS * H;
    const bool pk_ok = vrow_ok && vu0 < Hp;  // (padding units between H and Hp are published as zeros)
    const unsigned pbase = pk_ok ? ((unsigned)vb * a.Ypitch + vdir * Hp + vu0) * 4u : szYx;  // out of range: dropped
    const int voff = vrow * 16 + (lane & 3) * 4;  // my chunk inside a patch

    float* const xs1 = lds;                                   // [parity][wave][tile][G1][256]
    float* const xs2 = lds + 2 * XS1;                         // [parity][wave][tile][256]
    float* const priv = lds + 2 * (XS1 + XS2) + wave * WAVE_PRIV;
    float* const pin = priv;                                  // [G] projections of this step (my share)
    float* const pout = priv + G * 256;                       // [NOUT] h_t, saved slots (my share)
    float* const patchX = priv + (G + NOUT) * 256;
    float* const patchL = priv + (G + NOUT + 1) * 256;        // (LN only)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.Yx, szYx);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(TWO ? a.Xx : a.Yx, szYx);
    float* trash = a.trash + (tid & 63) * 4;
    // ---- per-step LayerNorm state (rows kq*4 + r, r = 0..3: the totals of all four are known to every lane, my share is
    // rows 2 rh, 2 rh + 1)
    const LnSlotsN<NPL> ls = LN ? ln_slots_n<NPL>(a, c, p, wave, lane, T) : LnSlotsN<NPL>();
    const __amdgpu_buffer_rsrc_t rsl = make_rsrc(LN ? (const void*)a.lnx : (const void*)a.Yx, LN ? ls.size : 0u);
    const float gam = (LN && unit_ok) ? a.ln_gamma[unit] : 0.f, bet = (LN && unit_ok) ? a.ln_beta[unit] : 0.f;
    const float invH = 1.0f / (float)H, inv_nm1 = 1.0f / (float)(H > 1 ? H - 1 : 1);
    float piv[4] = {0.f, 0.f, 0.f, 0.f};  // pivot of the one-pass variance: the row's mean of the previous step
    // the statistics of row kq*4 + u are written by lane u (< 4) of each DPP row of (member 0, wave 0)
    pk_f32x2 st_val = {0.f, 0.f};
    const int st_row = kq * 4 + (lane & 3);
    const bool st_ok = LN && p == 0 && wave == 0 && (lane & 15) < 4 && st_row < nrows;
    float* const st_base = st_ok ? a.lnstat + (long)(n_base + st_row) * 2 : trash;
    const long st_step = st_ok ? (long)a.R * 2 : 0;

    f32x4 pv[G];
    auto load_proj = [&](int tt, auto E) {
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
#pragma unroll
        for (int g = 0; g < G; ++g) pv[g] = ld4<decltype(E)::value>(a.P, vP0 + ts * vPs + g * H, vnv);
    };
    auto flush_outputs = [&](int tt, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
        st4<EE>(a.Y, vY0 + ts * vYs, vnv, trash, *reinterpret_cast<const f32x4*>(pout + voff));
#pragma unroll
        for (int k = 0; k < NS; ++k) st4<EE>(a.S, vS0 + ts * vSs + k * H, vnv, trash, *reinterpret_cast<const f32x4*>(pout + (k + 1) * 256 + voff));
        if (LN) {
            st4<EE>(a.lnh, vY0 + ts * vYs, vnv, trash, *reinterpret_cast<const f32x4*>(patchL + voff));
            *reinterpret_cast<pk_f32x2*>(st_base + (long)tt * st_step) = st_val;
        }
    };
#define PK4_LP0(E) load_proj(0, E)
    PK_EDGE_DISPATCH(PK4_LP0);
    // self-filling exchange (pk_rec2_common.h): the "not written yet" pattern goes into my own chunk of step tt - the first
    // PK_R2_FILL_AHEAD slabs here, visible everywhere before the handshake lets anyone poll, the others that many steps ahead
    // of my publishes (same lane, same address, program order: the pattern can never overtake the data)
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto fill_slab = [&](int tt, auto FC) {
        const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - tt) : tt) * TS : 0u);
        pub_store<decltype(FC)::value != 0>(rs, off, sentinel);
        if (TWO) pub_store<decltype(FC)::value != 0>(rsX, off, sentinel);
    };
    if (a.self_fill) {
        for (int tt = 0; tt < PK_R2_FILL_AHEAD && tt < T; ++tt) fill_slab(tt, BoolC<0>());
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    const bool fast = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd4(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    // diagnostics (pk_persist2_set_empty_step; timing only, results are garbage): bit 0 = no MFMAs, bit 1 = no waiting in the polls
    const bool no_mfma = (a.empty_step & 1) != 0;
    if ((a.empty_step & 2) != 0) dead = true;
    // the time loop is instantiated per (XCD-local exchange?, does my tile straddle H?): no run-time choices inside it
    auto run = [&](auto FASTC, auto SEC) {
    constexpr bool FAST = decltype(FASTC)::value != 0;
    constexpr int SE = decltype(SEC)::value;
    // h_t = gamma * (x - mean) / (std + eps) + beta over the row's H units, unbiased std (neural_networks.py:23-33); in: the
    // cell's h_t of my two rows, out: the normalised values (also the state fed back)
    auto layer_norm = [&](float (&hv)[2], int t) {
        if (t == 0) {
            // first step: no previous mean to pivot the one-pass variance on - one more exchange, this step only, gives the
            // row's own mean first (the reference's two-pass form); it uses the extra slab behind the T step slabs
            const float m0 = rvf[0] != 0.f ? hv[0] : 0.f, m1 = rvf[1] != 0.f ? hv[1] : 0.f;
            float ma[4] = {rh ? 0.f : m0, rh ? 0.f : m1, rh ? m0 : 0.f, rh ? m1 : 0.f};
            float mb[4] = {0.f, 0.f, 0.f, 0.f};
            unsigned po0[NPL];
            ls.poll_at(T, po0);
            dead = ln_row_allreduce<FAST, NPL>(rsl, ls.pub_at(T), po0, ma, mb, a.err, a.spin_limit, lane, dead);
#pragma unroll
            for (int r = 0; r < 4; ++r) piv[r] = ma[r] * invH;
        }
        float d2[2], q2[2];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            const float d = hv[s_] - (rh ? piv[2 + s_] : piv[s_]);
            d2[s_] = rvf[s_] != 0.f ? d : 0.f;
            q2[s_] = rvf[s_] != 0.f ? d * d : 0.f;
        }
        float la[4] = {rh ? 0.f : d2[0], rh ? 0.f : d2[1], rh ? d2[0] : 0.f, rh ? d2[1] : 0.f};
        float lb[4] = {rh ? 0.f : q2[0], rh ? 0.f : q2[1], rh ? q2[0] : 0.f, rh ? q2[1] : 0.f};
        unsigned po[NPL];
        ls.poll_at(t, po);
        dead = ln_row_allreduce<FAST, NPL>(rsl, ls.pub_at(t), po, la, lb, a.err, a.spin_limit, lane, dead);
        put_cd2(patchL, kq, rh, lane, hv);  // the pre-LN value, saved for backward
        float mu4[4], ri4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float md = la[r] * invH;
            const float mu = piv[r] + md;
            const float var = fmaxf((lb[r] - la[r] * md) * inv_nm1, 0.f);
            mu4[r] = mu;
            ri4[r] = 1.0f / (sqrtf(var) + a.ln_eps);
            piv[r] = mu;
        }
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            const float mu = rh ? mu4[2 + s_] : mu4[s_], ri = rh ? ri4[2 + s_] : ri4[s_];
            hv[s_] = rvf[s_] != 0.f ? gam * ((hv[s_] - mu) * ri) + bet : 0.f;
        }
        const int u3 = lane & 3;
        st_val[0] = u3 == 0 ? mu4[0] : u3 == 1 ? mu4[1] : u3 == 2 ? mu4[2] : mu4[3];
        st_val[1] = u3 == 0 ? ri4[0] : u3 == 1 ? ri4[1] : u3 == 2 ? ri4[2] : ri4[3];
    };
    for (int t = 0; t < T; ++t) {
        float* const x1 = xs1 + (t & 1) * XS1;
        float* const x2 = xs2 + (t & 1) * XS2;
        PK4_TRACE(t, 0);
        if (vlane) {  // (lanes 32..63 hold no share: their copies of the loads are garbage)
#pragma unroll
            for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(pin + g * 256 + voff) = pv[g];
        }
        f32x4 acc[2][G1];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int g = 0; g < G1; ++g) acc[tl][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 av[NQ];
        if (t > 0) {
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
            const unsigned lb = rp.lanebase(t - 1);
            auto off = [&](int i) { return rp.at(lb, wave * NQ + i); };
            dead = poll_regs<NQ, FAST>(rs, off, av, a.err, a.spin_limit, lane, dead);
        }
        PK4_TRACE(t, 1);
        // off the dependency chain, behind the poll: fp32 outputs of the previous step, projections of the next one
        if (t > 0) {
#define PK4_FO(E) flush_outputs(t - 1, E)
            PK_EDGE_DISPATCH_S(PK4_FO);
        }
        if (t + 1 < T) {
#define PK4_LP1(E) load_proj(t + 1, E)
            PK_EDGE_DISPATCH_S(PK4_LP1);
        }
        if (a.self_fill && t + PK_R2_FILL_AHEAD < T) fill_slab(t + PK_R2_FILL_AHEAD, FASTC);
        if (t > 0 && !no_mfma) {
            // 2 x G1 independent accumulation chains: a chain comes round every 2 G1 x 32 clocks (dependent latency: 40)
#pragma unroll
            for (int i = 0; i < NQ; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < G1; ++g) {
                        pk4_mfma<false>(acc[0][g], av[i][e], B1[0][g][i][e]);
                        pk4_mfma<true>(acc[1][g], av[i][e], B1[1][g][i][e]);
                    }
            pk4_mfma_settle(acc[1]);
        }
        PK4_TRACE(t, 2);
        // ---- the four quarters' partial sums meet in LDS; every wave adds the four of its share in the same order
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int g = 0; g < G1; ++g) {
                float o[4] = {acc[tl][g][0], acc[tl][g][1], acc[tl][g][2], acc[tl][g][3]};
                patch_put_cd(x1 + ((wave * 2 + tl) * G1 + g) * 256, kq, lane, o);
            }
        PK_BARRIER_LDS();
        PK4_TRACE(t, 3);
        float sum1[G1][2];
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            float q0[2], q1[2], q2[2], q3[2];
            get_cd2(x1 + ((0 * 2 + mt) * G1 + g) * 256, kq, rh, lane, q0);
            get_cd2(x1 + ((1 * 2 + mt) * G1 + g) * 256, kq, rh, lane, q1);
            get_cd2(x1 + ((2 * 2 + mt) * G1 + g) * 256, kq, rh, lane, q2);
            get_cd2(x1 + ((3 * 2 + mt) * G1 + g) * 256, kq, rh, lane, q3);
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) sum1[g][s_] = ((q0[s_] + q1[s_]) + q2[s_]) + q3[s_];
        }
        float pre[G][2];
#pragma unroll
        for (int g = 0; g < G; ++g) get_cd2(pin + g * 256, kq, rh, lane, pre[g]);
        float hv[2], sv[NS][2];
        if constexpr (!TWO) {
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                float pr[G];
#pragma unroll
                for (int g = 0; g < G; ++g) pr[g] = pre[g][s_] * psc[g] + psh[g] + sum1[g][s_];
                float h, cc, sl[NS];
                pk_cell_fwd<CELL>(act, pr, hprev[s_], cprev[s_], msk[s_], h, cc, sl);
                h = rvf[s_] != 0.f ? h : 0.f;  // rows / units outside the layer carry exact zeros (published as padding)
                cc = rvf[s_] != 0.f ? cc : 0.f;
                hprev[s_] = h;
                cprev[s_] = cc;
                hv[s_] = h;
#pragma unroll
                for (int k = 0; k < NS; ++k) sv[k][s_] = sl[k];
            }
            if constexpr (LN) {
                layer_norm(hv, t);
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) hprev[s_] = hv[s_];
            }
        } else {
            // ---- phase 1: the gates that depend on h_{t-1} only; x_t = r*h (GRU) / z*h (minimalGRU) goes to the cluster
            float xv[2], zt[2];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                float pr[G1], sl[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) sl[k] = 0.f;
#pragma unroll
                for (int g = 0; g < G1; ++g) pr[g] = pre[g][s_] * psc[g] + psh[g] + sum1[g][s_];
                const float x = pk_cell_fwd_p1<CELL>(pr, hprev[s_], sl);
                xv[s_] = rvf[s_] != 0.f ? x : 0.f;
                zt[s_] = sl[0];
#pragma unroll
                for (int k = 0; k < NS; ++k) sv[k][s_] = sl[k];  // (the slots phase 1 fills: z (, r), r*h / z*h)
                sv[NS - 1][s_] = xv[s_];
            }
            put_cd2(patchX, kq, rh, lane, xv);
            PK_LDS_ORDER();
            {
                const u32x4 o = no_sentinel4(*reinterpret_cast<const f32x4*>(patchX + voff));
                const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - t) : t) * TS : 0u);
                pub_store<FAST>(rsX, off, o);
            }
            // ---- phase 2: a_t = Wh_t + x_t . U_h^T
            f32x4 a2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            f32x4 b2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            {
                const unsigned lb = rp.lanebase(t);
                auto off = [&](int i) { return rp.at(lb, wave * NQ + i); };
                dead = poll_regs<NQ, FAST>(rsX, off, av, a.err, a.spin_limit, lane, dead);
            }
            if (!no_mfma) {
#pragma unroll
                for (int i = 0; i < NQ; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                    {
                        if ((e & 1) == 0) {
                            pk4_mfma<false>(a2[0], av[i][e], B2[0][TWO ? i : 0][e]);
                            pk4_mfma<true>(a2[1], av[i][e], B2[1][TWO ? i : 0][e]);
                        } else {
                            pk4_mfma<false>(b2[0], av[i][e], B2[0][TWO ? i : 0][e]);
                            pk4_mfma<true>(b2[1], av[i][e], B2[1][TWO ? i : 0][e]);
                        }
                    }
                pk4_mfma_settle(a2[1], b2[1]);
            }
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = a2[tl][r] + b2[tl][r];
                patch_put_cd(x2 + (wave * 2 + tl) * 256, kq, lane, o);
            }
            PK_BARRIER_LDS();
            float sum2[2];
            {
                float q0[2], q1[2], q2[2], q3[2];
                get_cd2(x2 + (0 * 2 + mt) * 256, kq, rh, lane, q0);
                get_cd2(x2 + (1 * 2 + mt) * 256, kq, rh, lane, q1);
                get_cd2(x2 + (2 * 2 + mt) * 256, kq, rh, lane, q2);
                get_cd2(x2 + (3 * 2 + mt) * 256, kq, rh, lane, q3);
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) sum2[s_] = ((q0[s_] + q1[s_]) + q2[s_]) + q3[s_];
            }
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const float at = pre[G - 1][s_] * psc[G - 1] + psh[G - 1] + sum2[s_];
                float h = pk_cell_fwd_p2<CELL>(act, at, zt[s_], hprev[s_], msk[s_]);
                h = rvf[s_] != 0.f ? h : 0.f;
                hprev[s_] = h;
                hv[s_] = h;
                sv[G - 1][s_] = at;  // GRU: slot 2, minimalGRU: slot 1
            }
            if constexpr (LN) {
                layer_norm(hv, t);
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) hprev[s_] = hv[s_];
            }
        }
        PK4_TRACE(t, 4);
        // ---- my share of the outputs into my patches, then the publish of h_t: one 16-byte store per lane (lanes 0..31)
        put_cd2(pout, kq, rh, lane, hv);
#pragma unroll
        for (int k = 0; k < NS; ++k) put_cd2(pout + (k + 1) * 256, kq, rh, lane, sv[k]);
        PK_LDS_ORDER();
        {
            const u32x4 o = no_sentinel4(*reinterpret_cast<const f32x4*>(pout + voff));
            const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - t) : t) * TS : 0u);
            pub_store<FAST>(rs, off, o);
        }
        PK4_TRACE(t, 5);
    }
    };
    PK_RUN_SPECIALISED(run, fast);
#define PK4_FOL(E) flush_outputs(T - 1, E)
    PK_EDGE_DISPATCH(PK4_FOL);
}

// ============================================================================
// backward: dL/dh_{t-1} = direct + [dgates_t] . [U_0; U_1; ...]
// ============================================================================
// LN: the gradient arriving at h_t goes through the LayerNorm backward first (two more row sums -> ln_row_allreduce);
// d gamma / d beta are accumulated per lane over the steps and leave as per-cluster partial sums (a.lnpart).
template <int CELL, int ACT, bool LN = false>
__global__ __launch_bounds__(256, 1) void rec4_bwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr bool TWO = pk_cell_two_phase(CELL);
    constexpr bool LSTM = CELL == PK_CELL_LSTM;
    constexpr int Gh = TWO ? G - 1 : G;        // gates whose gradients go back through h_{t-1}
    constexpr int NJB = Gh * NQ;               // groups of the carry product one wave holds (its quarter of Gh x 36)
    constexpr int NBATCH = Gh;                 // polled NQ groups at a time, two batches in flight
    // inputs of a step: saved slots, then (LSTM: c_{t-1}; the others: h_{t-1}), then dY (, LN: the pre-LN h_t)
    constexpr int NIN = NS + 2 + (LN ? 1 : 0);
    constexpr int XSB = 4 * 2 * 256, XSA = TWO ? 4 * 2 * 256 : 0;  // [wave][tile] carry partial sums (| q partial sums)
    constexpr int WAVE_PRIV = (NIN + G) * 256;                      // inputs | fp32 gate gradients (also the publish patches)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt = wave >> 1, rh = wave & 1;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const unsigned TB = (unsigned)T * B;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int kq = lane >> 4;

    // carry product: my group jj = global group jg = wave * NJB + jj -> gate g = jg / KJ, k = 16 (jg % KJ) + 4 kq + e:
    // BB[tl][jj][e] = U_g[k][unit(tl)];  two-phase cells: BA[tl][i][e] = U_{G-1}[16 (9 w + i) + 4 kq + e][unit(tl)]  (q = da . U_h)
    float BB[2][NJB][4];
    float BA[2][TWO ? NQ : 1][4];
    {
        const unsigned szU = (unsigned)((size_t)G * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int un = p * UW + tl * 16 + (lane & 15);
#pragma unroll
            for (int jj = 0; jj < NJB; ++jj) {
                const int jg = wave * NJB + jj, g = jg / KJ, j = jg % KJ;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = j * 16 + kq * 4 + e;
                    BB[tl][jj][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rsU, (un < H && k < H) ? (unsigned)(((g * H + k) * H + un) * 4) : szU, 0, 0));  // out of range: 0
                }
            }
            if (TWO) {
#pragma unroll
                for (int i = 0; i < NQ; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = (wave * NQ + i) * 16 + kq * 4 + e;
                        BA[tl][TWO ? i : 0][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            rsU, (un < H && k < H) ? (unsigned)((((G - 1) * H + k) * H + un) * 4) : szU, 0, 0));
                    }
            }
        }
    }
    const int ubase = p * UW + mt * 16;
    const int unit = ubase + (lane & 15);
    const bool unit_ok = unit < H;
    float* const lds = reinterpret_cast<float*>(smem);
    for (int i = tid; i < 2 * (XSB + XSA) + 4 * WAVE_PRIV; i += 256) lds[i] = 0.f;

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
    float rvf[2], msk[2], dh_dir[2], dc_car[2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        const int row = kq * 4 + 2 * rh + s_, n = n_base + row;
        const bool ok = row < nrows && unit_ok;
        rvf[s_] = ok ? 1.f : 0.f;
        msk[s_] = (a.mask != nullptr && ok) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        dh_dir[s_] = 0.f;
        dc_car[s_] = 0.f;
    }
    const bool vlane = lane < 32;
    const int vr = (lane >> 2) & 7, vrow = share_row(vr, rh), vu0 = ubase + (lane & 3) * 4;
    const bool vrow_ok = vlane && vrow < nrows;
    const int vn = n_base + (vrow_ok ? vrow : 0);
    const int vdir = vn >= B ? 1 : 0, vb = vn - vdir * B;
    int vnv = H - vu0;
    vnv = vnv > 4 ? 4 : (vnv < 0 ? 0 : vnv);
    const int edge = __any(vnv > 0 && vnv < 4) != 0 ? ((H & 1) ? 2 : 1) : 0;
    vnv = vrow_ok ? vnv : 0;
    const unsigned vY0 = ((unsigned)vb * a.YH + vdir * H + vu0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)vdir * TB + vb) * (NS * H) + vu0), vSs = (unsigned)B * NS * H;
    const unsigned vG0 = (((unsigned)vdir * TB + vb) * GH + vu0), vGs = (unsigned)B * GH;
    const bool pk_ok = vrow_ok && vu0 < Hp;
    const unsigned pbase = pk_ok ? (unsigned)vdir * (unsigned)T * TS + ((unsigned)vb * a.Gpitch + vu0) * 4u : szGx;
    const int voff = vrow * 16 + (lane & 3) * 4;

    float* const xsB = lds;                          // [parity][wave][tile][256]
    float* const xsA = lds + 2 * XSB;                // [parity][wave][tile][256]
    float* const priv = lds + 2 * (XSB + XSA) + wave * WAVE_PRIV;
    float* const pin = priv;                         // [NIN]
    float* const pg = priv + NIN * 256;              // [G] gate gradients of my share
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dGx, szGx);
    float* trash = a.trash + (tid & 63) * 4;

    f32x4 iv[NIN];
    auto load_step_e = [&](int t, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(vdir ? (T - 1 - t) : t);
        const unsigned tp = t > 0 ? (vdir ? ts + 1 : ts - 1) : ts;
        const int nvp = t > 0 ? vnv : 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) iv[k] = ld4<EE>(a.S, vS0 + ts * vSs + k * H, vnv);
        if (LSTM) iv[NS] = ld4<EE>(a.S, vS0 + tp * vSs + 4 * H, nvp);   // c_{t-1}: slot 4 of the previous step
        else iv[NS] = ld4<EE>(a.Y, vY0 + tp * vYs, nvp);                // h_{t-1}
        if (t == 0) iv[NS] = f32x4{0.f, 0.f, 0.f, 0.f};                 // c_{-1} = h_{-1} = 0
        iv[NS + 1] = ld4<EE>(a.dY, vY0 + ts * vYs, vnv);
        if (LN) iv[NS + 2] = ld4<EE>(a.lnh, vY0 + ts * vYs, vnv);
    };
    // ---- per-step LayerNorm state: (mean, 1 / (std + eps)) of the four rows of my quad, loaded one step ahead like the
    // saved gates; my share is rows 2 rh, 2 rh + 1
    const LnSlotsN<NPL> ls = LN ? ln_slots_n<NPL>(a, c, p, wave, lane, T) : LnSlotsN<NPL>();
    const __amdgpu_buffer_rsrc_t rsl = make_rsrc(LN ? (const void*)a.lnx : (const void*)a.dGx, LN ? ls.size : 0u);
    const float gam = (LN && unit_ok) ? a.ln_gamma[unit] : 0.f;
    const float invH = 1.0f / (float)H, inv_nm1 = 1.0f / (float)(H > 1 ? H - 1 : 1);
    float accg = 0.f, accb = 0.f;
    pk_f32x2 stn[2];
    long st_off[2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        const int row = kq * 4 + 2 * rh + s_;
        st_off[s_] = (LN && row < nrows) ? (long)(n_base + row) * 2 : 0;
        stn[s_] = pk_f32x2{0.f, 1.f};
    }
    auto load_stats = [&](int t) {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) stn[s_] = *reinterpret_cast<const pk_f32x2*>(a.lnstat + (long)t * a.R * 2 + st_off[s_]);
    };
    if (LN) load_stats(T - 1);
    auto flush_outputs_e = [&](int tt, auto E) {
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
#pragma unroll
        for (int g = 0; g < G; ++g)
            st4<decltype(E)::value>(a.dP2, vG0 + ts * vGs + g * H, vnv, trash, *reinterpret_cast<const f32x4*>(pg + g * 256 + voff));
    };
    auto publish_gate = [&](int g, int t, auto FC) {
        const u32x4 o = no_sentinel4(*reinterpret_cast<const f32x4*>(pg + g * 256 + voff));
        const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - t) : t) * TS + (unsigned)(g * Hp) * 4u : 0u);
        pub_store<decltype(FC)::value != 0>(rs, off, o);
    };
#define PK4_LS(E) load_step_e(T - 1, E)
    PK_EDGE_DISPATCH(PK4_LS);
    // self-filling exchange, as in the forward kernel: every gate's chunk of step tt
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto fill_slab = [&](int tt, auto FC) {
        const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - tt) : tt) * TS : 0u);
#pragma unroll
        for (int g = 0; g < G; ++g) pub_store<decltype(FC)::value != 0>(rs, off + (pk_ok ? (unsigned)(g * Hp) * 4u : 0u), sentinel);
    };
    if (a.self_fill) {
        for (int k = 0; k < PK_R2_FILL_AHEAD && k < T; ++k) fill_slab(T - 1 - k, BoolC<0>());
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    const bool fast = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd4(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    const bool no_mfma = (a.empty_step & 1) != 0;  // diagnostics, as in the forward kernel
    if ((a.empty_step & 2) != 0) dead = true;
    auto run = [&](auto FASTC, auto SEC) {
    constexpr bool FAST = decltype(FASTC)::value != 0;
    constexpr int SE = decltype(SEC)::value;
    // dL/d(pre-LN h) = rinv * (g - mean(g)) - d * rinv^2 * sum(g d) / ((H - 1) std),  g = dh * gamma, d = x - mean
    // (in: dL/dh_t of my two rows, the pre-LN values and this step's statistics; out: dL/d(pre-LN h_t))
    auto layer_norm_bwd = [&](float (&dh2)[2], const float (&x2)[2], const float (&mu2)[2], const float (&ri2)[2], int it_) {
        float dd[2], gg[2], a2[2], b2[2];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            const bool ok = rvf[s_] != 0.f;
            dd[s_] = x2[s_] - mu2[s_];
            gg[s_] = dh2[s_] * gam;
            a2[s_] = ok ? gg[s_] : 0.f;
            b2[s_] = ok ? gg[s_] * dd[s_] : 0.f;
            accg += ok ? dh2[s_] * (dd[s_] * ri2[s_]) : 0.f;
            accb += ok ? dh2[s_] : 0.f;
        }
        float la[4] = {rh ? 0.f : a2[0], rh ? 0.f : a2[1], rh ? a2[0] : 0.f, rh ? a2[1] : 0.f};
        float lb[4] = {rh ? 0.f : b2[0], rh ? 0.f : b2[1], rh ? b2[0] : 0.f, rh ? b2[1] : 0.f};
        unsigned po[NPL];
        ls.poll_at(it_, po);
        dead = ln_row_allreduce<FAST, NPL>(rsl, ls.pub_at(it_), po, la, lb, a.err, a.spin_limit, lane, dead);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            const float sa = rh ? la[2 + s_] : la[s_], sb = rh ? lb[2 + s_] : lb[s_];
            const float sd = 1.0f / ri2[s_] - a.ln_eps;
            const float k2 = ri2[s_] * ri2[s_] * sb * inv_nm1 / sd;
            dh2[s_] = ri2[s_] * (gg[s_] - sa * invH) - k2 * dd[s_];
        }
    };
    int it = 0;
    for (int t = T - 1; t >= 0; --t, ++it) {
        float* const xB = xsB + (it & 1) * XSB;
        float* const xA = xsA + (it & 1) * XSA;
        PK4_TRACE(it, 0);
        if (vlane) {  // (lanes 32..63 hold no share: their copies of the loads are garbage)
#pragma unroll
            for (int k = 0; k < NIN; ++k) *reinterpret_cast<f32x4*>(pin + k * 256 + voff) = iv[k];
        }
        const float mu2[2] = {stn[0][0], stn[1][0]}, ri2[2] = {stn[0][1], stn[1][1]};  // (the loads are a whole step old)
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // one chain per tile: 64 clocks apart
        if (t < T - 1) {
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
            const unsigned lb0 = rp.lanebase(it - 1);
            // batch bq of wave w = groups jg = w * NJB + bq * NQ .. + NQ - 1: gate jg / KJ, group jg % KJ.  ONE set of 9 chunk
            // registers: the load of chunk i of the next batch is issued into a chunk's registers right behind the MFMAs that
            // consumed them, so the next batch streams in under this batch's MFMAs without a second buffer
            u32x4 av[NQ];
            auto lbase = [&](int bq) { return lb0 + (unsigned)((wave * NJB + bq * NQ) / KJ) * gate_bytes; };
            auto j0of = [&](int bq) { return (wave * NJB + bq * NQ) % KJ; };
            {
                const unsigned lb = lbase(0);
                const int j0 = j0of(0);
#pragma unroll
                for (int i = 0; i < NQ; ++i) av[i] = poll_load<FAST>(rs, rp.at(lb, j0 + i));
            }
#pragma unroll
            for (int bq = 0; bq < NBATCH; ++bq) {
                const unsigned lb = lbase(bq);
                const int j0 = j0of(bq);
                auto off = [&](int i) { return rp.at(lb, j0 + i); };
                dead = settle_regs<NQ, FAST>(rs, off, av, a.err, a.spin_limit, lane, dead);
                const unsigned lbn = bq + 1 < NBATCH ? lbase(bq + 1) : 0u;
                const int j0n = bq + 1 < NBATCH ? j0of(bq + 1) : 0;
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    if (!no_mfma) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            {
                                pk4_mfma<false>(acc[0], av[i][e], BB[0][bq * NQ + i][e]);
                                pk4_mfma<true>(acc[1], av[i][e], BB[1][bq * NQ + i][e]);
                            }
                    }
                    if (bq + 1 < NBATCH) av[i] = poll_load<FAST>(rs, rp.at(lbn, j0n + i));
                }
                if (bq == NBATCH - 1) {
                    // off the dependency chain, behind the LAST poll load of the step (nothing the MFMAs wait for may queue
                    // behind an HBM store): fp32 gate gradients of the previous step, saved tensors of the next
#define PK4_FOB(E) flush_outputs_e(t + 1, E)
                    PK_EDGE_DISPATCH_S(PK4_FOB);
                    if (t > 0) {
#define PK4_LS1(E) load_step_e(t - 1, E)
                        PK_EDGE_DISPATCH_S(PK4_LS1);
                        if (LN) load_stats(t - 1);
                    }
                    if (a.self_fill && t - PK_R2_FILL_AHEAD >= 0) fill_slab(t - PK_R2_FILL_AHEAD, FASTC);
                }
            }
        } else if (t > 0) {
#define PK4_LS2(E) load_step_e(t - 1, E)
            PK_EDGE_DISPATCH_S(PK4_LS2);
            if (LN) load_stats(t - 1);
            if (a.self_fill && t - PK_R2_FILL_AHEAD >= 0) fill_slab(t - PK_R2_FILL_AHEAD, FASTC);
        }
        pk4_mfma_settle(acc[1]);
        PK4_TRACE(it, 1);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            float o[4] = {acc[tl][0], acc[tl][1], acc[tl][2], acc[tl][3]};
            patch_put_cd(xB + (wave * 2 + tl) * 256, kq, lane, o);
        }
        PK_BARRIER_LDS();
        PK4_TRACE(it, 2);
        float car[2];
        {
            float q0[2], q1[2], q2[2], q3[2];
            get_cd2(xB + (0 * 2 + mt) * 256, kq, rh, lane, q0);
            get_cd2(xB + (1 * 2 + mt) * 256, kq, rh, lane, q1);
            get_cd2(xB + (2 * 2 + mt) * 256, kq, rh, lane, q2);
            get_cd2(xB + (3 * 2 + mt) * 256, kq, rh, lane, q3);
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) car[s_] = ((q0[s_] + q1[s_]) + q2[s_]) + q3[s_];
        }
        float sin[NIN][2];
#pragma unroll
        for (int k = 0; k < NIN; ++k) get_cd2(pin + k * 256, kq, rh, lane, sin[k]);
        float dgv[G][2];
        float dh2[2];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) dh2[s_] = sin[NS + 1][s_] + dh_dir[s_] + car[s_];
        if constexpr (LN) {
            const float x2[2] = {sin[NS + 2][0], sin[NS + 2][1]};
            layer_norm_bwd(dh2, x2, mu2, ri2, it);
        }
        if constexpr (!TWO) {
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                float sl[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) sl[k] = sin[k][s_];
                const float prev = sin[NS][s_];  // LSTM: c_{t-1}
                const float dh = dh2[s_];
                float dg[G], dhd, dcp;
                pk_cell_bwd<CELL>(act, sl, LSTM ? 0.f : prev, LSTM ? prev : 0.f, msk[s_], dh, dc_car[s_], dg, dhd, dcp);
                dh_dir[s_] = rvf[s_] != 0.f ? dhd : 0.f;
                dc_car[s_] = rvf[s_] != 0.f ? dcp : 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) dgv[g][s_] = rvf[s_] != 0.f ? dg[g] : 0.f;
            }
        } else {
            // ---- phase A: da_t (the operand of q = da . U_h) goes to the cluster
            float da2[2], dzp[2], dhd2[2];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                float sl[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) sl[k] = sin[k][s_];
                const float dh = dh2[s_];
                const float da = pk_cell_bwd_pa<CELL>(act, sl, sin[NS][s_], msk[s_], dh, dzp[s_], dhd2[s_]);
                da2[s_] = rvf[s_] != 0.f ? da : 0.f;
            }
            put_cd2(pg + (G - 1) * 256, kq, rh, lane, da2);
            PK_LDS_ORDER();
            publish_gate(G - 1, t, FASTC);
            PK4_TRACE(it, 3);
            f32x4 qa[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            f32x4 qb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            {
                u32x4 av[NQ];
                const unsigned lb = rp.lanebase(it) + (unsigned)(G - 1) * gate_bytes;
                auto off = [&](int i) { return rp.at(lb, wave * NQ + i); };
                dead = poll_regs<NQ, FAST>(rs, off, av, a.err, a.spin_limit, lane, dead);
                if (!no_mfma) {
#pragma unroll
                    for (int i = 0; i < NQ; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            {
                                if ((e & 1) == 0) {
                                    pk4_mfma<false>(qa[0], av[i][e], BA[0][TWO ? i : 0][e]);
                                    pk4_mfma<true>(qa[1], av[i][e], BA[1][TWO ? i : 0][e]);
                                } else {
                                    pk4_mfma<false>(qb[0], av[i][e], BA[0][TWO ? i : 0][e]);
                                    pk4_mfma<true>(qb[1], av[i][e], BA[1][TWO ? i : 0][e]);
                                }
                            }
                    pk4_mfma_settle(qa[1], qb[1]);
                }
            }
            PK4_TRACE(it, 4);
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = qa[tl][r] + qb[tl][r];
                patch_put_cd(xA + (wave * 2 + tl) * 256, kq, lane, o);
            }
            PK_BARRIER_LDS();
            PK4_TRACE(it, 5);
            float q[2];
            {
                float q0[2], q1[2], q2[2], q3[2];
                get_cd2(xA + (0 * 2 + mt) * 256, kq, rh, lane, q0);
                get_cd2(xA + (1 * 2 + mt) * 256, kq, rh, lane, q1);
                get_cd2(xA + (2 * 2 + mt) * 256, kq, rh, lane, q2);
                get_cd2(xA + (3 * 2 + mt) * 256, kq, rh, lane, q3);
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) q[s_] = ((q0[s_] + q1[s_]) + q2[s_]) + q3[s_];
            }
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                float sl[NS];
#pragma unroll
                for (int k = 0; k < NS; ++k) sl[k] = sin[k][s_];
                float dg[G];
                float dhd = dhd2[s_];
                pk_cell_bwd_pb<CELL>(sl, sin[NS][s_], q[s_], da2[s_], dzp[s_], dg, dhd);
                dh_dir[s_] = rvf[s_] != 0.f ? dhd : 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) dgv[g][s_] = rvf[s_] != 0.f ? dg[g] : 0.f;
            }
        }
        PK4_TRACE(it, 6);
        // ---- my share of the gate gradients through my patches into the vector layout, then publish what the next
        // step's carry needs: one 16-byte store per gate (lanes 0..31)
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (!(TWO && g == G - 1)) put_cd2(pg + g * 256, kq, rh, lane, dgv[g]);
        PK_LDS_ORDER();
#pragma unroll
        for (int g = 0; g < Gh; ++g) publish_gate(g, t, FASTC);
        PK4_TRACE(it, 7);
    }
    };
    PK_RUN_SPECIALISED(run, fast);
#define PK4_FOBL(E) flush_outputs_e(0, E)
    PK_EDGE_DISPATCH(PK4_FOBL);
    if constexpr (LN) {
        // my unit's share of d gamma / d beta over this cluster's rows and all steps: the four row quads of a wave fold with
        // shuffles, the two waves of a tile through LDS (the partial-sum area is free now); one owner per (cluster, unit)
        accg += __shfl_xor(accg, 16, 64);
        accg += __shfl_xor(accg, 32, 64);
        accb += __shfl_xor(accb, 16, 64);
        accb += __shfl_xor(accb, 32, 64);
        __syncthreads();
        if (rh == 1 && lane < 16) {
            lds[(mt * 2 + 0) * 16 + lane] = accg;
            lds[(mt * 2 + 1) * 16 + lane] = accb;
        }
        __syncthreads();
        if (rh == 0 && lane < 16 && unit < KPAD) {
            a.lnpart[(long)(a.ln_cg0 + c) * KPAD + unit] = accg + lds[(mt * 2 + 0) * 16 + lane];
            a.lnpart[(long)(a.ln_ncg + a.ln_cg0 + c) * KPAD + unit] = accb + lds[(mt * 2 + 1) * 16 + lane];
        }
    }
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
// (the LayerNorm variants exist with the run-time activation only: no shipped recipe normalises h_t)
template <int CELL>
Rec4Kernel pick4_ln(bool backward) {
    return backward ? (Rec4Kernel)rec4_bwd_kernel<CELL, -1, true> : (Rec4Kernel)rec4_fwd_kernel<CELL, -1, true>;
}
Rec4Kernel pick4(int cell, int act, bool backward, bool ln) {
    if (ln) {
        switch (cell) {
            case PK_CELL_LSTM: return pick4_ln<PK_CELL_LSTM>(backward);
            case PK_CELL_GRU: return pick4_ln<PK_CELL_GRU>(backward);
            default: return pick4_ln<PK_CELL_MINGRU>(backward);
        }
    }
    switch (cell) {
        case PK_CELL_LSTM: return backward ? pick4_bwd<PK_CELL_LSTM>(act) : pick4_fwd<PK_CELL_LSTM>(act);
        case PK_CELL_GRU: return backward ? pick4_bwd<PK_CELL_GRU>(act) : pick4_fwd<PK_CELL_GRU>(act);
        default: return backward ? pick4_bwd<PK_CELL_MINGRU>(act) : pick4_fwd<PK_CELL_MINGRU>(act);
    }
}

size_t lds4(int cell, bool backward, bool ln) {
    const int G = pk_cell_gates(cell), NS = pk_cell_saved(cell);
    const bool two = pk_cell_two_phase(cell);
    if (!backward) {
        const int G1 = two ? G - 1 : G;
        return ((size_t)2 * (4 * 2 * G1 * 256 + (two ? 4 * 2 * 256 : 0)) + 4 * (size_t)(G + NS + 2 + (ln ? 1 : 0)) * 256) * 4;
    }
    return ((size_t)2 * (4 * 2 * 256 + (two ? 4 * 2 * 256 : 0)) + 4 * (size_t)(NS + 2 + (ln ? 1 : 0) + G) * 256) * 4;
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

int self_fill4() {
    static int v = -1;
    if (v < 0) {
        const char* e = pk_experiment("rec4_self_fill");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
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

// The fourth-generation plan's share of pk_rec_ln_work_floats (pk_rec_persist2.hip): row-statistics exchange, per-cluster
// d gamma / d beta partial sums and the column-sum scratch, for twice as many workgroups per cluster as the second generation
int64_t pk_rec4f_ln_work_floats(int T, int B, int bidir, int H) {
    Plan2 pl;
    if (T <= 0 || B <= 0 || H <= 0 || H > KPAD || make_plan4(B * (1 + bidir), H, pl) != 0) return 0;
    const int64_t ncg = (int64_t)pl.launches * pl.C;
    return (int64_t)(T + 1) * ncg * 4 * (4 * pl.Pn) * 8 + 2 * ncg * KPAD + pk_bn_partial_floats(ncg, KPAD) + 256;
}

int pk_rec4f_fwd(hipStream_t st, int cell, int act, int T, int B, int bidir, int H, const float* P, const float* pscale,
                 const float* pshift, const float* U, const float* mask, float mask_scalar, float* Y, float* S, float* Yx,
                 const PkLnHost* ln) {
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
    rc = pk_rec2_host_setup(a, false, cell);
    if (rc) return rc;
    a.self_fill = self_fill4();
    // the mailboxes: every dword "not written yet" - written by the kernel itself a few steps ahead of its publishes, or
    // (PK_EXPERIMENT rec4_self_fill=0) by one fill of 282 / 565 MB in front of the launches
    if (!a.self_fill) PK_CHECK_HIP(hipMemsetAsync(Yx, 0xFF, slab * (two ? 2 : 1), st));
    rc = pk_rec2_ln_setup(st, a, pl, ln, false);
    if (rc) return rc;
    const size_t lds = lds4(cell, false, ln != nullptr);
    const Rec4Kernel k = pick4(cell, act, false, ln != nullptr);
    rc = grant_lds4(k, lds);
    if (rc) return rc;
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        a.ln_cg0 = l * pl.C;
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
                 float mask_scalar, const float* Y, const float* S, const float* dY, float* dP2, float* dGx, const PkLnHost* ln) {
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
    rc = pk_rec2_host_setup(a, true, cell);
    if (rc) return rc;
    a.self_fill = self_fill4();
    if (!a.self_fill) PK_CHECK_HIP(hipMemsetAsync(dGx, 0xFF, bytes, st));
    rc = pk_rec2_ln_setup(st, a, pl, ln, true);
    if (rc) return rc;
    const size_t lds = lds4(cell, true, ln != nullptr);
    const Rec4Kernel k = pick4(cell, act, true, ln != nullptr);
    rc = grant_lds4(k, lds);
    if (rc) return rc;
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        a.ln_cg0 = l * pl.C;
        rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        rc = pk_rec2_check_residency((const void*)k, 256, lds, pl.C * pl.Pn, "pk_rec_bwd (fp32, persistent, generation 4)");
        if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(pl.C * pl.Pn), dim3(256), lds, st, a);
        PK_LAUNCH_CHECK();
    }
    return pk_rec2_ln_finish(st, a, ln);
}
