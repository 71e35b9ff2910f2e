// pk_rec_persist2.hip - perf-mode (bf16 MFMA operands) persistent recurrent time loop
// for gfx950: forward and BPTT of liGRU / RNN / LSTM layers in ONE launch each.
//
// Replaces the reference's python time loops and their autograd
// (neural_networks.py:457-469 LSTM, :1130-1141 liGRU, :1438-1447 RNN) with the
// bidirectional cat/flip (:415-417, :475-478) folded into the indexing.
//
// Second-generation design (see pk_rec_persist.hip for the exact-fp32 kernels):
//  * rows (sequences of both directions) are split over CLUSTERS of Pn
//    workgroups; inside a cluster each wave owns 16 hidden units and keeps its
//    slice of the recurrent matrix in REGISTERS as MFMA B fragments for all T
//    steps (U is read from HBM once per layer);
//  * the per-step exchange of h_t inside a cluster goes through L2 in BF16,
//    16 bytes at a time: producers pack 8 units per lane through a wave-private
//    LDS patch and publish with ONE write-through (sc1) 16-byte store; consumers
//    poll with sc1 16-byte loads.  The data is the flag: the exchange buffer is
//    pre-filled with the bf16 NaN pattern 0xFFFF (never produced by the packer),
//    so there is no flag, no fence and no barrier between workgroups, and the
//    protocol is placement independent (MI355X guide, Guideline 16 R2);
//  * the exchange buffers are not scratch: Yb (bf16 copy of the layer output)
//    and dGb (bf16 gate gradients) are exactly the operands the dU / dW GEMMs
//    (pk_gemm_bf16) need, so nothing is converted afterwards;
//  * the A tile is staged once per workgroup in LDS (double buffered: one
//    barrier per step); projections / saved gates of step t+1 are prefetched
//    right after the poll of step t returns, so HBM latency is off the
//    dependency chain; every spin is bounded (host-visible error word).
#include "pk_cell.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KPAD = 576;    // K (hidden units) padded to 18 MFMA k-steps of 32
constexpr int KSTEPS = 18;
constexpr int RMAX = 16;     // rows per cluster (one MFMA M tile)

struct R2Args {
    int T, B, R, H, Hp, YH, act;
    int C, Pn, rpc, row0;  // clusters, workgroups per cluster, rows per cluster, first row of this launch
    const float *P, *pscale, *pshift, *U, *mask;
    float mask_scalar;
    float* Y;
    float* S;
    unsigned short* Yb;
    int Ypitch;  // elements per (t,b) row of Yb; direction d starts at d*Hp
    const float* dY;
    float* dP2;
    unsigned short* dGb;
    int Gpitch;  // elements per row of dGb; gate g starts at g*Hp
    unsigned* err;
    int spin_limit;
};

__device__ __forceinline__ bool has_sent16(const u32x4 v) {
    bool s = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) s = s || ((v[e] & 0xFFFFu) == 0xFFFFu) || ((v[e] >> 16) == 0xFFFFu);
    return s;
}
// bf16 with the sentinel pattern excluded (any NaN becomes the canonical quiet NaN 0x7FC0)
__device__ __forceinline__ unsigned short to_bf_pub(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)0x7FC0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ bool spin_check2(int& spins, int spin_limit, unsigned* err, int lane) {
    if (++spins > spin_limit) {
        if (lane == 0) atomicAdd_system(err, 1u);
        return true;
    }
    if ((spins & 63) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return true;
    __builtin_amdgcn_s_sleep(1);
    return false;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// Poll NCH 16-byte chunks per lane (write-through loads) until none holds the sentinel, then store them to LDS.
template <int NCH>
__device__ __forceinline__ bool poll_to_lds(__amdgpu_buffer_rsrc_t rs, const unsigned (&goff)[NCH], const bool (&cv)[NCH],
                                            const int (&loff)[NCH], unsigned char* tile, unsigned* err, int spin_limit,
                                            int lane, bool dead) {
    u32x4 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        v[i] = u32x4{0u, 0u, 0u, 0u};
        if (cv[i]) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i], 0, 16);
    }
    if (!dead) {
        int spins = 0;
        while (true) {
            bool bad = false;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (cv[i] && has_sent16(v[i])) {
                    v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i], 0, 16);
                    bad = bad || has_sent16(v[i]);
                }
            }
            if (!__any(bad)) break;
            if (spin_check2(spins, spin_limit, err, lane)) {
                dead = true;
                break;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (cv[i]) *reinterpret_cast<u32x4*>(tile + loff[i]) = v[i];
    return dead;
}

// ============================================================================
// forward
// ============================================================================
template <int CELL>
__global__ __launch_bounds__(256, 1) void rec2_fwd_kernel(R2Args a) {
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr int LDA = KPAD + 8;                 // bf16 elements per A-tile row (1168 B: odd multiple of 16 B)
    constexpr int ATILE = RMAX * LDA * 2;         // bytes
    constexpr int NCH = (RMAX * (KPAD / 8) + 255) / 256;  // 16-byte chunks polled per lane (5)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][ATILE] | pack [4][16][16] bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;  // whole workgroup (uniform): this cluster has no rows
    const int unit = p * 64 + wave * 16 + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;

    // ---- recurrent weights of my 16 units -> registers (once): B[k][n] = U_g[unit n][k]
    bf16x8 Bf[G][KSTEPS];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            bf16x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kk * 32 + kq * 8 + e;
                const float w = (unit_ok && k < H) ? a.U[((long)(g * H + unit)) * H + k] : 0.f;
                f[e] = (short)pk_f2bf(w);
            }
            Bf[g][kk] = f;
        }
    float psc[G], psh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        psc[g] = unit_ok ? a.pscale[g * H + unit] : 0.f;
        psh[g] = unit_ok ? a.pshift[g * H + unit] : 0.f;
    }
    for (int i = tid; i < (2 * ATILE + 4 * 512) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- poll descriptors: chunk ci = (row, col) of the cluster's [nrows][Hp/8] block of h_{t-1}
    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Ypitch * 2u;  // bytes per time slab of Yb
    unsigned cbase[NCH];
    int cdir[NCH], clds[NCH];
    bool cv[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = tid + 256 * i;
        cv[i] = ci < nrows * CPR;
        const int row = cv[i] ? ci / CPR : 0, col = cv[i] ? ci - row * CPR : 0;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        cdir[i] = dir;
        cbase[i] = ((unsigned)b * a.Ypitch + dir * Hp + col * 8) * 2u;
        clds[i] = row * (LDA * 2) + col * 16;
    }
    // ---- my (row, unit) pairs of the gate math: C/D layout row = kq*4 + r, col = lane&15
    bool rv[4];
    int rdir[4], rb[4];
    float msk[4], hprev[4], cprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r, n = n_base + row;
        rv[r] = row < nrows && unit_ok;
        rdir[r] = n >= B ? 1 : 0;
        rb[r] = n - rdir[r] * B;
        msk[r] = (a.mask != nullptr && rv[r]) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        hprev[r] = 0.f;
        cprev[r] = 0.f;
    }
    // ---- publish descriptors: lanes 0..31 store one 16-byte piece (row, 8 units) of the wave's patch
    const int prow = lane >> 1, phalf = lane & 1;
    const int pu0 = p * 64 + wave * 16 + phalf * 8;
    const bool pk_ok = lane < 32 && prow < nrows && pu0 < Hp;
    const int pn = n_base + (prow < nrows ? prow : 0);
    const int pdir = pn >= B ? 1 : 0, pb = pn - pdir * B;
    const unsigned pbase = ((unsigned)pb * a.Ypitch + pdir * Hp + pu0) * 2u;
    unsigned short* patch = reinterpret_cast<unsigned short*>(smem + 2 * ATILE + wave * 512);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.Yb, (unsigned)T * TS);

    float pre[4][G];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long pr0 = (long)(rdir[r] ? (T - 1) : 0) * B + rb[r];
#pragma unroll
        for (int g = 0; g < G; ++g) pre[r][g] = rv[r] ? a.P[pr0 * GH + g * H + unit] : 0.f;
    }
    __syncthreads();

    bool dead = false;
    for (int t = 0; t < T; ++t) {
        f32x4 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned char* At = smem + (t & 1) * ATILE;
        if (t > 0) {
            unsigned goff[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) goff[i] = cbase[i] + (unsigned)(cdir[i] ? (T - t) : (t - 1)) * TS;
            dead = poll_to_lds<NCH>(rs, goff, cv, clds, At, a.err, a.spin_limit, lane, dead);
        }
        // projections of step t+1: issued now, consumed after the next poll
        float pnx[4][G];
        if (t + 1 < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long pr1 = (long)(rdir[r] ? (T - 2 - t) : (t + 1)) * B + rb[r];
#pragma unroll
                for (int g = 0; g < G; ++g) pnx[r][g] = rv[r] ? a.P[pr1 * GH + g * H + unit] : 0.f;
            }
        }
        if (t > 0) {
            __syncthreads();
            const unsigned char* Ar = At + (lane & 15) * (LDA * 2) + kq * 16;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ar + kk * 64);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[g][kk], acc[g], 0, 0, 0);
            }
        }
        // ---- gate math for my (row, unit) pairs
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float pr[G];
#pragma unroll
            for (int g = 0; g < G; ++g) pr[g] = pre[r][g] * psc[g] + psh[g] + acc[g][r];
            float h, cc, s[NS];
            pk_cell_fwd<CELL>(a.act, pr, hprev[r], cprev[r], msk[r], h, cc, s);
            unsigned short hb = 0;
            if (rv[r]) {
                hprev[r] = h;
                cprev[r] = cc;
                hb = to_bf_pub(h);
                const long prw = (long)(rdir[r] ? (T - 1 - t) : t) * B + rb[r];
                a.Y[prw * a.YH + rdir[r] * H + unit] = h;
                float* sp = a.S + ((long)rdir[r] * T * B + prw) * (NS * H) + unit;
#pragma unroll
                for (int k = 0; k < NS; ++k) sp[k * H] = s[k];
            }
            patch[(kq * 4 + r) * 16 + (lane & 15)] = hb;
        }
        // ---- publish h_t: 8 units per lane, one write-through 16-byte store
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (pk_ok) {
            const u32x4 o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(patch) + prow * 32 + phalf * 16);
            const unsigned off = pbase + (unsigned)(pdir ? (T - 1 - t) : t) * TS;
            __builtin_amdgcn_raw_buffer_store_b128(o, rs, off, 0, 16);
        }
        if (t + 1 < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int g = 0; g < G; ++g) pre[r][g] = pnx[r][g];
        }
    }
}

// ============================================================================
// backward: dL/dh_{t-1} = direct + [dgates_t] . [U_0; U_1; ...]
// ============================================================================
template <int CELL>
__global__ __launch_bounds__(256, 1) void rec2_bwd_kernel(R2Args a) {
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr int LDA = G * KPAD + 8;
    constexpr int ATILE = RMAX * LDA * 2;
    constexpr int NCH = (RMAX * G * (KPAD / 8) + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][ATILE] | pack [4][G][16][16] bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const long TB = (long)T * B;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int unit = p * 64 + wave * 16 + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;

    // B[kidx = (g, j)][n = unit] = U_g[j][unit]
    bf16x8 Bf[G][KSTEPS];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            bf16x8 f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = kk * 32 + kq * 8 + e;
                const float w = (unit_ok && j < H) ? a.U[((long)(g * H + j)) * H + unit] : 0.f;
                f[e] = (short)pk_f2bf(w);
            }
            Bf[g][kk] = f;
        }
    for (int i = tid; i < (2 * ATILE + 4 * G * 512) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- poll descriptors: chunk ci = (row, gate, col) of the cluster's dgates_{t+1} block
    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Gpitch * 2u;  // bytes per time slab of dGb
    unsigned cbase[NCH];
    int cdir[NCH], clds[NCH];
    bool cv[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = tid + 256 * i;
        cv[i] = ci < nrows * G * CPR;
        const int row = cv[i] ? ci / (G * CPR) : 0;
        const int rem = cv[i] ? ci - row * (G * CPR) : 0;
        const int g = rem / CPR, col = rem - g * CPR;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        cdir[i] = dir;
        cbase[i] = (unsigned)dir * (unsigned)T * TS + ((unsigned)b * a.Gpitch + g * Hp + col * 8) * 2u;
        clds[i] = row * (LDA * 2) + (g * KPAD + col * 8) * 2;
    }
    bool rv[4];
    int rdir[4], rb[4];
    float msk[4], dh_dir[4], dc_car[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r, n = n_base + row;
        rv[r] = row < nrows && unit_ok;
        rdir[r] = n >= B ? 1 : 0;
        rb[r] = n - rdir[r] * B;
        msk[r] = (a.mask != nullptr && rv[r]) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        dh_dir[r] = 0.f;
        dc_car[r] = 0.f;
    }
    const int prow = lane >> 1, phalf = lane & 1;
    const int pu0 = p * 64 + wave * 16 + phalf * 8;
    const bool pk_ok = lane < 32 && prow < nrows && pu0 < Hp;
    const int pn = n_base + (prow < nrows ? prow : 0);
    const int pdir = pn >= B ? 1 : 0, pb = pn - pdir * B;
    const unsigned pbase = (unsigned)pdir * (unsigned)T * TS + ((unsigned)pb * a.Gpitch + pu0) * 2u;
    unsigned short* patch = reinterpret_cast<unsigned short*>(smem + 2 * ATILE + wave * (G * 512));
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dGb, (unsigned)(a.R / B) * (unsigned)T * TS);

    // saved tensors of one step for my (row, unit) pairs
    float sv[4][NS], hp[4], cp[4], dy[4];
    auto load_step = [&](int t, float (&sv_)[4][NS], float (&hp_)[4], float (&cp_)[4], float (&dy_)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ts = rdir[r] ? (T - 1 - t) : t;
            const long prw = (long)ts * B + rb[r];
            const long srow = (long)rdir[r] * TB + prw;
            const long prev = (long)(rdir[r] ? ts + 1 : ts - 1) * B + rb[r];
#pragma unroll
            for (int k = 0; k < NS; ++k) sv_[r][k] = rv[r] ? a.S[srow * (NS * H) + k * H + unit] : 0.f;
            hp_[r] = (rv[r] && t > 0) ? a.Y[prev * a.YH + rdir[r] * H + unit] : 0.f;
            cp_[r] = (CELL == PK_CELL_LSTM && rv[r] && t > 0) ? a.S[((long)rdir[r] * TB + prev) * (NS * H) + 4 * H + unit] : 0.f;
            dy_[r] = rv[r] ? a.dY[prw * a.YH + rdir[r] * H + unit] : 0.f;
        }
    };
    load_step(T - 1, sv, hp, cp, dy);
    __syncthreads();

    bool dead = false;
    int it = 0;
    for (int t = T - 1; t >= 0; --t, ++it) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned char* At = smem + (it & 1) * ATILE;
        if (t < T - 1) {
            unsigned goff[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) goff[i] = cbase[i] + (unsigned)(cdir[i] ? (T - 2 - t) : (t + 1)) * TS;
            dead = poll_to_lds<NCH>(rs, goff, cv, clds, At, a.err, a.spin_limit, lane, dead);
        }
        float svn[4][NS], hpn[4], cpn[4], dyn[4];
        if (t > 0) load_step(t - 1, svn, hpn, cpn, dyn);
        if (t < T - 1) {
            __syncthreads();
            const unsigned char* Ar = At + (lane & 15) * (LDA * 2) + kq * 16;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ar + (g * KPAD + kk * 32) * 2);
                    if ((kk & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[g][kk], acc0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[g][kk], acc1, 0, 0, 0);
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dh = dy[r] + dh_dir[r] + acc0[r] + acc1[r];
            float dg[G], dhd, dcp;
            pk_cell_bwd<CELL>(a.act, sv[r], hp[r], cp[r], msk[r], dh, dc_car[r], dg, dhd, dcp);
            if (rv[r]) {
                dh_dir[r] = dhd;
                dc_car[r] = dcp;
                const long prw = (long)(rdir[r] ? (T - 1 - t) : t) * B + rb[r];
                float* o = a.dP2 + ((long)rdir[r] * TB + prw) * GH + unit;
#pragma unroll
                for (int g = 0; g < G; ++g) o[g * H] = dg[g];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) patch[g * 256 + (kq * 4 + r) * 16 + (lane & 15)] = rv[r] ? to_bf_pub(dg[g]) : (unsigned short)0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (pk_ok) {
            const unsigned off = pbase + (unsigned)(pdir ? (T - 1 - t) : t) * TS;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u32x4 o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(patch) + g * 512 + prow * 32 + phalf * 16);
                __builtin_amdgcn_raw_buffer_store_b128(o, rs, off + (unsigned)(g * Hp) * 2u, 0, 16);
            }
        }
        if (t > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int k = 0; k < NS; ++k) sv[r][k] = svn[r][k];
                hp[r] = hpn[r];
                cp[r] = cpn[r];
                dy[r] = dyn[r];
            }
        }
    }
}

unsigned* g2_err_host = nullptr;
unsigned* g2_err_dev = nullptr;
int ensure_err2() {
    if (g2_err_host) return 0;
    PK_CHECK_HIP(hipHostMalloc((void**)&g2_err_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *g2_err_host = 0;
    PK_CHECK_HIP(hipHostGetDevicePointer((void**)&g2_err_dev, g2_err_host, 0));
    return 0;
}

struct Plan2 {
    int Pn, C, rpc, launches;
};
int make_plan2(int R, int H, Plan2& pl) {
    pl.Pn = (H + 63) / 64;
    const int ncu = pk_num_cu();
    int C = ncu / pl.Pn;
    PK_REQUIRE(C >= 1, "persistent recurrence: H=%d needs %d workgroups per cluster but the device has %d CUs", H, pl.Pn,
               ncu);
    if (C >= 8) C -= C % 8;  // members of one cluster congruent mod 8: one XCD under round-robin dispatch (speed only)
    int rpc = (R + C - 1) / C;
    if (rpc > RMAX) rpc = RMAX;
    if (rpc < 1) rpc = 1;
    int need = (R + rpc - 1) / rpc;          // clusters needed in total
    pl.launches = (need + C - 1) / C;
    if (pl.launches == 1) C = need;
    pl.C = C;
    pl.rpc = rpc;
    return 0;
}

int check2(const char* who, int cell, int T, int B, int bidir, int H) {
    PK_REQUIRE(cell == PK_CELL_LIGRU || cell == PK_CELL_RNN || cell == PK_CELL_LSTM,
               "%s: the bf16 persistent algorithm covers liGRU/RNN/LSTM (cell %d)", who, cell);
    PK_REQUIRE(T > 0 && B > 0 && H > 0 && H <= KPAD && (bidir == 0 || bidir == 1), "%s: bad geometry T=%d B=%d H=%d (H <= %d)",
               who, T, B, H, KPAD);
    return 0;
}

}  // namespace

extern "C" unsigned pk_persist2_error_count(void) { return g2_err_host ? *g2_err_host : 0u; }
extern "C" void pk_persist2_error_reset(void) {
    if (g2_err_host) *g2_err_host = 0u;
}

extern "C" int pk_rec_fwd_bf16(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                               const float* pscale, const float* pshift, const float* U, const float* mask,
                               float mask_scalar, float* Y, float* S, uint16_t* Yb, int64_t y_pitch) {
    int rc = check2("pk_rec_fwd_bf16", cell, T, B, bidir, H);
    if (rc) return rc;
    rc = ensure_err2();
    if (rc) return rc;
    hipStream_t st = pk_stream(stream);
    const int ndir = 1 + bidir, R = B * ndir, Hp = (H + 7) & ~7;
    PK_REQUIRE(y_pitch >= (int64_t)ndir * Hp && (y_pitch % 8) == 0 && ((uintptr_t)Yb & 15) == 0,
               "pk_rec_fwd_bf16: Yb pitch must be a multiple of 8 and hold %d x %d elements", ndir, Hp);
    PK_REQUIRE((double)T * B * y_pitch * 2.0 < 4.0e9, "pk_rec_fwd_bf16: exchange buffer exceeds the 4 GB buffer-descriptor range");
    Plan2 pl;
    rc = make_plan2(R, H, pl);
    if (rc) return rc;
    R2Args a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.Hp = Hp; a.YH = ndir * H; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.rpc = pl.rpc; a.row0 = 0;
    a.P = P; a.pscale = pscale; a.pshift = pshift; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = Y; a.S = S; a.Yb = (unsigned short*)Yb; a.Ypitch = (int)y_pitch;
    a.dY = nullptr; a.dP2 = nullptr; a.dGb = nullptr; a.Gpitch = 0;
    a.err = g2_err_dev; a.spin_limit = 400000;
    // the bf16 layer output is the mailbox: poison it with the sentinel
    PK_CHECK_HIP(hipMemsetAsync(Yb, 0xFF, (size_t)T * B * y_pitch * 2, st));
    const int G = pk_cell_gates(cell);
    (void)G;
    const size_t lds = 2 * (size_t)RMAX * (KPAD + 8) * 2 + 4 * 512;
    static bool attr_done = false;
    if (!attr_done) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)rec2_fwd_kernel<PK_CELL_LIGRU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)rec2_fwd_kernel<PK_CELL_RNN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)rec2_fwd_kernel<PK_CELL_LSTM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        dim3 grid(pl.C * pl.Pn), block(256);
        if (cell == PK_CELL_LIGRU) hipLaunchKernelGGL((rec2_fwd_kernel<PK_CELL_LIGRU>), grid, block, lds, st, a);
        else if (cell == PK_CELL_RNN) hipLaunchKernelGGL((rec2_fwd_kernel<PK_CELL_RNN>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((rec2_fwd_kernel<PK_CELL_LSTM>), grid, block, lds, st, a);
        PK_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int pk_rec_bwd_bf16(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U,
                               const float* mask, float mask_scalar, const float* Y, const float* S, const float* dY,
                               float* dP2, uint16_t* dGb, int64_t g_pitch) {
    int rc = check2("pk_rec_bwd_bf16", cell, T, B, bidir, H);
    if (rc) return rc;
    rc = ensure_err2();
    if (rc) return rc;
    hipStream_t st = pk_stream(stream);
    const int ndir = 1 + bidir, R = B * ndir, Hp = (H + 7) & ~7, G = pk_cell_gates(cell);
    PK_REQUIRE(g_pitch >= (int64_t)G * Hp && (g_pitch % 8) == 0 && ((uintptr_t)dGb & 15) == 0,
               "pk_rec_bwd_bf16: dGb pitch must be a multiple of 8 and hold %d x %d elements", G, Hp);
    PK_REQUIRE((double)ndir * T * B * g_pitch * 2.0 < 4.0e9, "pk_rec_bwd_bf16: exchange buffer exceeds the 4 GB buffer-descriptor range");
    Plan2 pl;
    rc = make_plan2(R, H, pl);
    if (rc) return rc;
    R2Args a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.Hp = Hp; a.YH = ndir * H; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.rpc = pl.rpc; a.row0 = 0;
    a.P = nullptr; a.pscale = nullptr; a.pshift = nullptr; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = const_cast<float*>(Y); a.S = const_cast<float*>(S); a.Yb = nullptr; a.Ypitch = 0;
    a.dY = dY; a.dP2 = dP2; a.dGb = (unsigned short*)dGb; a.Gpitch = (int)g_pitch;
    a.err = g2_err_dev; a.spin_limit = 400000;
    PK_CHECK_HIP(hipMemsetAsync(dGb, 0xFF, (size_t)ndir * T * B * g_pitch * 2, st));
    const size_t lds = 2 * (size_t)RMAX * (G * KPAD + 8) * 2 + 4 * (size_t)G * 512;
    static bool attr_done = false;
    if (!attr_done) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)rec2_bwd_kernel<PK_CELL_LIGRU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)rec2_bwd_kernel<PK_CELL_RNN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)rec2_bwd_kernel<PK_CELL_LSTM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        dim3 grid(pl.C * pl.Pn), block(256);
        if (cell == PK_CELL_LIGRU) hipLaunchKernelGGL((rec2_bwd_kernel<PK_CELL_LIGRU>), grid, block, lds, st, a);
        else if (cell == PK_CELL_RNN) hipLaunchKernelGGL((rec2_bwd_kernel<PK_CELL_RNN>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((rec2_bwd_kernel<PK_CELL_LSTM>), grid, block, lds, st, a);
        PK_LAUNCH_CHECK();
    }
    return 0;
}
