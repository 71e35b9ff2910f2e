// pk_rec_persist2_f32.hip - the EXACT-fp32 twin of pk_rec_persist2.hip: persistent forward and BPTT of liGRU / RNN
// layers (neural_networks.py:1130-1141, :1438-1447 and their autograd) for the parity-grade mode (PK_PRECISION=fp32,
// what the 1e-4 tests run in).
//
// Same second-generation structure as the bf16 kernels - clusters of workgroups, each wave owns 16 hidden units and
// keeps its slice of the recurrent matrix in REGISTERS for all T steps, h_t (dgates_t) exchanged through L2 in 16-byte
// chunks with the data as the flag (buffer pre-filled with 0xFFFFFFFF dwords, a pattern the publisher never emits),
// wave-private LDS patches between the MFMA C/D layout and the 16-byte "vector" layout, projections / saved gates
// prefetched one step ahead, fp32 outputs flushed one step behind - but
//   * the exchange carries fp32 (a chunk = 4 units), the A tile in LDS is fp32;
//   * the matrix products run on v_mfma_f32_16x16x4_f32 (exact fp32: bitwise an fmaf chain) - 1/16 of the bf16 MFMA
//     rate, so a step is MFMA-bound here: 2 x 144 MFMAs x 32 clocks = 9 200 clocks for liGRU at H = 550;
//   * the gate math uses the precise exp / division (pk_cell.h without PK_CELL_FAST_MATH).
// The first-generation kernels (pk_rec_persist.hip: 4-byte granules, K split over the waves of a workgroup) stay for
// LSTM, whose four gates do not fit one wave's registers in fp32.
//
// K ordering: a lane of the MFMA holds ONE k per step.  The A tile is read 16 bytes at a time - lane (row r, quarter q)
// reads k = 16 j + 4 q .. + 3 - and step (j, e) multiplies element e of that read with B[16 j + 4 q + e][n]: a
// permutation of k applied to both operands, so the sum runs over the same products (in another order than a
// left-to-right loop - like any blocked GEMM).
#include <stdlib.h>

#define PK_REC2_PRECISE 1
#include "pk_rec2_common.h"

namespace {

constexpr int KJ = KPAD / 16;  // 36 groups of 16 k
constexpr int JV = 22;   // k-groups below JV of the gates before the last keep their B fragments in VGPRs (pk4_mfma)

__device__ __forceinline__ u32x4 no_sentinel(f32x4 v) {
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u = __float_as_uint(v[e]);
        o[e] = u == 0xFFFFFFFFu ? 0x7FC00000u : u;  // the one NaN encoding that reads as "not written yet"
    }
    return o;
}

// ============================================================================
// forward
// ============================================================================
// LN: per-step LayerNorm of h_t (a.ln_gamma / ln_beta): the row statistics take one more exchange inside the step
// (ln_row_allreduce, pk_rec2_common.h); the normalised h_t is what is stored, published and fed back, the pre-LN value
// and (mean, 1 / (std + eps)) are saved for the backward pass.
template <int CELL, int ACT, bool LN>
__global__ __launch_bounds__(256, 1) void rec2f_fwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr int LDA = pk_r2_lda_f32(KPAD);       // floats per A-tile row (conflict-free b128 reads, pk_rec2_common.h)
    constexpr int ATILE = RMAX * LDA * 4;          // bytes
    constexpr int NCH = (RMAX * (KPAD / 4) + 255) / 256;  // 16-byte chunks polled per lane (9)
    constexpr int WAVE_LDS = (G + 1 + NS + (LN ? 1 : 0)) * 1024;  // P stage | Y | S slots (| pre-LN h)
    constexpr int LDS_TRASH = 2 * ATILE + 4 * WAVE_LDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][ATILE] | 4 x WAVE_LDS | trash

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int ubase = p * 64 + wave * 16;
    const int unit = ubase + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;

    // ---- recurrent weights of my 16 units -> registers (once): Bf[g][j][e] = U_g[unit][16 j + 4 kq + e]
    float Bf[G][KJ][4];
    {
        const unsigned szU = (unsigned)((size_t)G * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const int k0 = j * 16 + kq * 4;
                const unsigned off = (unsigned)(((g * H + unit) * H + k0) * 4);
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsU, (unit_ok && k0 < H) ? off : szU, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) Bf[g][j][e] = (k0 + e < H) ? __uint_as_float(raw[e]) : 0.f;  // beyond H: the next row
            }
    }
    float psc[G], psh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        psc[g] = unit_ok ? a.pscale[g * H + unit] : 0.f;
        psh[g] = unit_ok ? a.pshift[g * H + unit] : 0.f;
    }
    for (int i = tid; i < (LDS_TRASH + 16) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- poll descriptors: chunk ci = (row, 4-unit column) of the cluster's [nrows][Hp/4] block of h_{t-1}
    const int CPR = Hp >> 2;
    const unsigned TS = (unsigned)B * a.Ypitch * 4u;  // bytes per time slab of Yx
    const unsigned szYx = (unsigned)T * TS;
    unsigned cbase[NCH], cstep[NCH];
    int clds[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = tid + 256 * i;
        const bool ok = ci < nrows * CPR;
        const int row = ok ? ci / CPR : 0, col = ok ? ci - row * CPR : 0;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        cbase[i] = ok ? ((unsigned)b * a.Ypitch + dir * Hp + col * 4) * 4u + (unsigned)(dir ? (T - 1) : 0) * TS : szYx;
        cstep[i] = ok ? (dir ? 0u - TS : TS) : 0u;
        clds[i] = ok ? row * (LDA * 4) + col * 16 : LDS_TRASH;
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

    unsigned char* wl = smem + 2 * ATILE + wave * WAVE_LDS;
    float* patchP = reinterpret_cast<float*>(wl);
    float* patchY = reinterpret_cast<float*>(wl + G * 1024);
    float* patchS = reinterpret_cast<float*>(wl + (G + 1) * 1024);
    float* patchL = reinterpret_cast<float*>(wl + (G + 1 + NS) * 1024);  // (LN only)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.Yx, szYx);
    float* trash = a.trash + (tid & 63) * 4;
    // ---- per-step LayerNorm state
    const LnSlots ls = LN ? ln_slots(a, c, p, wave, lane, T) : LnSlots();
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(LN ? a.lnx : a.Yx, LN ? ls.size : 0u);
    const float gam = (LN && unit_ok) ? a.ln_gamma[unit] : 0.f, bet = (LN && unit_ok) ? a.ln_beta[unit] : 0.f;
    const float invH = 1.0f / (float)H, inv_nm1 = 1.0f / (float)(H - 1);
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
        st4<EE>(a.Y, vY0 + ts * vYs, vnv, trash, patch_get_vec(patchY, lane));
#pragma unroll
        for (int k = 0; k < NS; ++k) st4<EE>(a.S, vS0 + ts * vSs + k * H, vnv, trash, patch_get_vec(patchS + k * 256, lane));
        if (LN) {
            st4<EE>(a.lnh, vY0 + ts * vYs, vnv, trash, patch_get_vec(patchL, lane));
            *reinterpret_cast<pk_f32x2*>(st_base + (long)tt * st_step) = st_val;
        }
    };
#define PK_LP0(E) load_proj(0, E)
    PK_EDGE_DISPATCH(PK_LP0);
    // self-filling exchange (pk_rec2_common.h): the "not written yet" pattern goes into my own chunk - the first
    // PK_R2_FILL_AHEAD slabs here, in place before the handshake lets anyone poll, the others that many steps ahead of my
    // publishes (same lane, same address, program order: the pattern can never overtake the data)
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (a.self_fill) {
        for (int tt = 0; tt < PK_R2_FILL_AHEAD && tt < T; ++tt)
            pub_store<false>(rs, pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - tt) : tt) * TS : 0u), sentinel);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    const bool fast = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    for (int t = 0; t < T; ++t) {
        f32x4 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned char* At = smem + (t & 1) * ATILE;
        if (t > 0) {
            unsigned goff[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) goff[i] = cbase[i] + (unsigned)(t - 1) * cstep[i];
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
            int retries = 0;
            dead = fast ? poll_to_lds<NCH, true>(rs, goff, clds, At, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds<NCH, false>(rs, goff, clds, At, a.err, a.spin_limit, lane, dead, retries);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) patch_put_vec(patchP + g * 256, lane, pv[g]);
        if (t > 0) PK_BARRIER_LDS();
        else PK_LDS_ORDER();
        // off the dependency chain, behind the barrier: fp32 outputs of the previous step, projections of the next one
        if (t > 0) {
#define PK_FO(E) flush_outputs(t - 1, E)
            PK_EDGE_DISPATCH(PK_FO);
        }
        if (t + 1 < T) {
#define PK_LP1(E) load_proj(t + 1, E)
            PK_EDGE_DISPATCH(PK_LP1);
        }
        if (a.self_fill && t + PK_R2_FILL_AHEAD < T) {
            const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - (t + PK_R2_FILL_AHEAD)) : (t + PK_R2_FILL_AHEAD)) * TS : 0u);
            if (fast) pub_store<true>(rs, off, sentinel);
            else pub_store<false>(rs, off, sentinel);
        }
        if (t > 0) {
            const float* Ar = reinterpret_cast<const float*>(At) + (lane & 15) * LDA + kq * 4;
            // (the forward kernel keeps the builtin form: with its B fragments bound to AGPRs - pk4_mfma, as the backward kernel
            // below does - the allocator had no VGPR left to read the next LDS fragment ahead, every ds_read landed right in
            // front of its MFMAs and a launch went from 3.63 to 4.15 ms, round 6)
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(Ar + j * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], Bf[g][j][e], acc[g], 0, 0, 0);
            }
        }
        // ---- gate math for my (row, unit) pairs
        float pre[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) patch_get_cd(patchP + g * 256, kq, lane, pre[g]);
        float hv[4], sv[NS][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float pr[G];
#pragma unroll
            for (int g = 0; g < G; ++g) pr[g] = pre[g][r] * psc[g] + psh[g] + acc[g][r];
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
        if (LN) {
            // h_t = gamma * (x - mean) / (std + eps) + beta over the row's H units, unbiased std (neural_networks.py:23-33)
            if (t == 0) {
                // first step: there is no previous mean to pivot the one-pass variance on (a pivot of 0 costs eps * (mean /
                // std)^2 of relative accuracy) - one more exchange, this step only, gives the row's own mean first: the
                // reference's two-pass form (neural_networks.py:23-33).  It uses the extra slab behind the T step slabs.
                float ma[4], mb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ma[r] = rvf[r] != 0.f ? hv[r] : 0.f;
                    mb[r] = 0.f;
                }
                unsigned po0[3];
                ls.poll_at(T, po0);
                dead = fast ? ln_row_allreduce<true>(rsx, ls.pub_at(T), po0, ma, mb, a.err, a.spin_limit, lane, dead)
                            : ln_row_allreduce<false>(rsx, ls.pub_at(T), po0, ma, mb, a.err, a.spin_limit, lane, dead);
#pragma unroll
                for (int r = 0; r < 4; ++r) piv[r] = ma[r] * invH;
            }
            float la[4], lb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = hv[r] - piv[r];
                la[r] = rvf[r] != 0.f ? d : 0.f;
                lb[r] = rvf[r] != 0.f ? d * d : 0.f;
            }
            unsigned po[3];
            ls.poll_at(t, po);
            dead = fast ? ln_row_allreduce<true>(rsx, ls.pub_at(t), po, la, lb, a.err, a.spin_limit, lane, dead)
                        : ln_row_allreduce<false>(rsx, ls.pub_at(t), po, la, lb, a.err, a.spin_limit, lane, dead);
            patch_put_cd(patchL, kq, lane, hv);  // the pre-LN value, saved for backward
            float mu4[4], ri4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float md = la[r] * invH;
                const float mu = piv[r] + md;
                const float var = fmaxf((lb[r] - la[r] * md) * inv_nm1, 0.f);
                const float ri = 1.0f / (sqrtf(var) + a.ln_eps);
                const float hn = rvf[r] != 0.f ? gam * ((hv[r] - mu) * ri) + bet : 0.f;
                hv[r] = hn;
                hprev[r] = hn;
                piv[r] = mu;
                mu4[r] = mu;
                ri4[r] = ri;
            }
            const int u3 = lane & 3;
            st_val[0] = u3 == 0 ? mu4[0] : u3 == 1 ? mu4[1] : u3 == 2 ? mu4[2] : mu4[3];
            st_val[1] = u3 == 0 ? ri4[0] : u3 == 1 ? ri4[1] : u3 == 2 ? ri4[2] : ri4[3];
        }
        // ---- h_t through the wave's Y patch into the vector layout, then publish: one 16-byte store per lane
        patch_put_cd(patchY, kq, lane, hv);
        PK_LDS_ORDER();
        {
            const u32x4 o = no_sentinel(patch_get_vec(patchY, lane));
            const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - t) : t) * TS : 0u);
            if (fast) pub_store<true>(rs, off, o);
            else pub_store<false>(rs, off, o);
        }
        // ---- the saved gates go to the wave patches; they (and Y) are written to HBM at the top of the next step
#pragma unroll
        for (int k = 0; k < NS; ++k) patch_put_cd(patchS + k * 256, kq, lane, sv[k]);
        PK_LDS_ORDER();
    }
#define PK_FOL(E) flush_outputs(T - 1, E)
    PK_EDGE_DISPATCH(PK_FOL);
}

// ============================================================================
// backward: dL/dh_{t-1} = direct + [dgates_t] . [U_0; U_1; ...]
// ============================================================================
// LN: the gradient arriving at h_t goes through the LayerNorm backward first (two more row sums -> ln_row_allreduce);
// d gamma / d beta are accumulated per lane over the steps and leave as per-cluster partial sums (a.lnpart).
template <int CELL, int ACT, bool LN>
__global__ __launch_bounds__(256, 1) void rec2f_bwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr int LDA = pk_r2_lda_f32(G * KPAD);   // floats per A-tile row
    constexpr int ATILE = RMAX * LDA * 4;
    constexpr int NBUF = (2 * ATILE > 100 * 1024) ? 1 : 2;  // two gates: one 74 KB tile + an extra barrier per step
    constexpr int NCH = (RMAX * G * (KPAD / 4) + 255) / 256;  // 9 per gate
    constexpr int NIN = NS + 2 + (LN ? 1 : 0);     // saved gates, h_{t-1}, dY (, pre-LN h_t)
    constexpr int WAVE_LDS = (NIN + G) * 1024;     // input patches | fp32 gate-gradient patches
    constexpr int LDS_TRASH = NBUF * ATILE + 4 * WAVE_LDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const unsigned TB = (unsigned)T * B;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int ubase = p * 64 + wave * 16;
    const int unit = ubase + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;

    // Bf[g][j][e] = U_g[16 j + 4 kq + e][unit]
    float Bf[G][KJ][4];
    {
        const unsigned szU = (unsigned)((size_t)G * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < KJ; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = j * 16 + kq * 4 + e;
                    Bf[g][j][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rsU, (unit_ok && k < H) ? (unsigned)(((g * H + k) * H + unit) * 4) : szU, 0, 0));  // out of range: 0
                }
    }
    for (int i = tid; i < (LDS_TRASH + 16) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- poll descriptors: chunk ci = (row, gate, 4-unit column) of the cluster's dgates_{t+1} block
    const int CPR = Hp >> 2;
    const unsigned TS = (unsigned)B * a.Gpitch * 4u;  // bytes per time slab of dGx
    const unsigned ndir = (unsigned)(a.R / B);
    const unsigned szGx = ndir * (unsigned)T * TS;
    unsigned cbase[NCH], cstep[NCH];
    int clds[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = tid + 256 * i;
        const bool ok = ci < nrows * G * CPR;
        const int row = ok ? ci / (G * CPR) : 0;
        const int rem = ok ? ci - row * (G * CPR) : 0;
        const int g = rem / CPR, col = rem - g * CPR;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        cbase[i] = ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)b * a.Gpitch + g * Hp + col * 4) * 4u +
                            (unsigned)(dir ? 0 : (T - 1)) * TS
                      : szGx;
        cstep[i] = ok ? (dir ? TS : 0u - TS) : 0u;
        clds[i] = ok ? row * (LDA * 4) + (g * KPAD + col * 4) * 4 : LDS_TRASH;
    }
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

    unsigned char* wl = smem + NBUF * ATILE + wave * WAVE_LDS;
    float* patchI = reinterpret_cast<float*>(wl);               // [NIN][256]: S slots, hp, dY
    float* patchG = reinterpret_cast<float*>(wl + NIN * 1024);  // [G][256] fp32 gate gradients
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
        iv[NS] = ld4<EE>(a.Y, vY0 + tp * vYs, nvp);
        iv[NS + 1] = ld4<EE>(a.dY, vY0 + ts * vYs, vnv);
        if (LN) iv[NS + 2] = ld4<EE>(a.lnh, vY0 + ts * vYs, vnv);
        if (t == 0) iv[NS] = f32x4{0.f, 0.f, 0.f, 0.f};  // h_{-1} = 0
    };
    // ---- per-step LayerNorm state: (mean, 1/(std+eps)) of my four rows, loaded one step ahead like the saved gates
    const LnSlots ls = LN ? ln_slots(a, c, p, wave, lane, T) : LnSlots();
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(LN ? a.lnx : a.dGx, LN ? ls.size : 0u);
    const float gam = (LN && unit_ok) ? a.ln_gamma[unit] : 0.f;
    const float invH = 1.0f / (float)H, inv_nm1 = 1.0f / (float)(H - 1);
    float accg = 0.f, accb = 0.f;
    pk_f32x2 stn[4];
    long st_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r;
        st_off[r] = (LN && row < nrows) ? (long)(n_base + row) * 2 : 0;
        stn[r] = pk_f32x2{0.f, 1.f};
    }
    auto load_stats = [&](int t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) stn[r] = *reinterpret_cast<const pk_f32x2*>(a.lnstat + (long)t * a.R * 2 + st_off[r]);
    };
    if (LN) load_stats(T - 1);
    auto flush_outputs_e = [&](int tt, auto E) {
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
#pragma unroll
        for (int g = 0; g < G; ++g)
            st4<decltype(E)::value>(a.dP2, vG0 + ts * vGs + g * H, vnv, trash, patch_get_vec(patchG + g * 256, lane));
    };
#define PK_LS(E) load_step_e(T - 1, E)
    PK_EDGE_DISPATCH(PK_LS);
    // self-filling exchange, as in the forward kernel: every gate's chunk of step tt
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto fill_slab = [&](int tt, bool fast_) {
        const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - tt) : tt) * TS : 0u);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned og = off + (pk_ok ? (unsigned)(g * Hp) * 4u : 0u);
            if (fast_) pub_store<true>(rs, og, sentinel);
            else pub_store<false>(rs, og, sentinel);
        }
    };
    if (a.self_fill) {
        for (int k = 0; k < PK_R2_FILL_AHEAD && k < T; ++k) fill_slab(T - 1 - k, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    const bool fast = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    int it = 0;
    for (int t = T - 1; t >= 0; --t, ++it) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned char* At = smem + (NBUF == 2 ? (it & 1) * ATILE : 0);
        if (t < T - 1) {
            unsigned goff[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) goff[i] = cbase[i] + (unsigned)(it - 1) * cstep[i];
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
            int retries = 0;
            dead = fast ? poll_to_lds<NCH, true>(rs, goff, clds, At, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds<NCH, false>(rs, goff, clds, At, a.err, a.spin_limit, lane, dead, retries);
        }
        float mu4[4], ri4[4];  // this step's row statistics (the loads are a whole step old)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mu4[r] = stn[r][0];
            ri4[r] = stn[r][1];
        }
#pragma unroll
        for (int k = 0; k < NIN; ++k) patch_put_vec(patchI + k * 256, lane, iv[k]);
        if (t < T - 1) PK_BARRIER_LDS();
        else PK_LDS_ORDER();
        // off the dependency chain, behind the barrier: fp32 gate gradients of the previous step, saved tensors of the next
        if (t < T - 1) {
#define PK_FOB(E) flush_outputs_e(t + 1, E)
            PK_EDGE_DISPATCH(PK_FOB);
        }
        if (t > 0) {
#define PK_LS1(E) load_step_e(t - 1, E)
            PK_EDGE_DISPATCH(PK_LS1);
            if (LN) load_stats(t - 1);
        }
        if (a.self_fill && t - PK_R2_FILL_AHEAD >= 0) fill_slab(t - PK_R2_FILL_AHEAD, fast);
        if (t < T - 1) {
            const float* Ar = reinterpret_cast<const float*>(At) + (lane & 15) * LDA + kq * 4;
#pragma unroll
            for (int g = 0; g < G - 1; ++g)
                pk_static_for<0, KJ>([&](auto JC) {
                    constexpr int j = decltype(JC)::value;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(Ar + g * KPAD + j * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if ((e & 1) == 0) pk4_mfma<(G > 1 && j >= JV)>(acc0, __float_as_uint(av[e]), Bf[g][j][e]);
                        else pk4_mfma<(G > 1 && j >= JV)>(acc1, __float_as_uint(av[e]), Bf[g][j][e]);
                    }
                });
            // (the last gate's B fragments and the k-groups from JV on of the others live in AGPRs and are read from there:
            // pk4_mfma, pk_rec2_common.h)
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(Ar + (G - 1) * KPAD + j * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if ((e & 1) == 0) pk4_mfma<(G > 1)>(acc0, __float_as_uint(av[e]), Bf[G - 1][j][e]);
                    else pk4_mfma<(G > 1)>(acc1, __float_as_uint(av[e]), Bf[G - 1][j][e]);
                }
            }
            if (G > 1) pk4_mfma_settle(acc0, acc1);
        }
        if (NBUF == 1 && t < T - 1) PK_BARRIER_LDS();  // single A tile: everyone is done reading before the next poll refills it
        float sin[NIN][4];
#pragma unroll
        for (int k = 0; k < NIN; ++k) patch_get_cd(patchI + k * 256, kq, lane, sin[k]);
        float dgv[G][4];
        float dh4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dh4[r] = sin[NS + 1][r] + dh_dir[r] + acc0[r] + acc1[r];
        if (LN) {
            // dL/d(pre-LN h) = rinv * (g - mean(g)) - d * rinv^2 * sum(g d) / ((H - 1) std),  g = dh * gamma, d = x - mean
            float la[4], lb[4], dd[4], gg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = rvf[r] != 0.f;
                dd[r] = sin[NS + 2][r] - mu4[r];
                gg[r] = dh4[r] * gam;
                la[r] = ok ? gg[r] : 0.f;
                lb[r] = ok ? gg[r] * dd[r] : 0.f;
                accg += ok ? dh4[r] * (dd[r] * ri4[r]) : 0.f;
                accb += ok ? dh4[r] : 0.f;
            }
            unsigned po[3];
            ls.poll_at(it, po);
            dead = fast ? ln_row_allreduce<true>(rsx, ls.pub_at(it), po, la, lb, a.err, a.spin_limit, lane, dead)
                        : ln_row_allreduce<false>(rsx, ls.pub_at(it), po, la, lb, a.err, a.spin_limit, lane, dead);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sd = 1.0f / ri4[r] - a.ln_eps;
                const float k2 = ri4[r] * ri4[r] * lb[r] * inv_nm1 / sd;
                dh4[r] = ri4[r] * (gg[r] - la[r] * invH) - k2 * dd[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) s[k] = sin[k][r];
            const float hp = sin[NS][r];
            const float dh = dh4[r];
            float dg[G], dhd, dcp;
            pk_cell_bwd<CELL>(act, s, hp, 0.f, msk[r], dh, dc_car[r], dg, dhd, dcp);
            dh_dir[r] = rvf[r] != 0.f ? dhd : 0.f;
            dc_car[r] = rvf[r] != 0.f ? dcp : 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) dgv[g][r] = rvf[r] != 0.f ? dg[g] : 0.f;
        }
        // ---- gate gradients through the wave's patches into the vector layout, then publish: one 16-byte store per gate
#pragma unroll
        for (int g = 0; g < G; ++g) patch_put_cd(patchG + g * 256, kq, lane, dgv[g]);
        PK_LDS_ORDER();
        {
            const unsigned off = pbase + (pk_ok ? (unsigned)(vdir ? (T - 1 - t) : t) * TS : 0u);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u32x4 o = no_sentinel(patch_get_vec(patchG + g * 256, lane));
                const unsigned og = off + (pk_ok ? (unsigned)(g * Hp) * 4u : 0u);
                if (fast) pub_store<true>(rs, og, o);
                else pub_store<false>(rs, og, o);
            }
        }
    }
#define PK_FOBL(E) flush_outputs_e(0, E)
    PK_EDGE_DISPATCH(PK_FOBL);
    if (LN) {  // my unit's share of d gamma / d beta over this cluster's rows and all steps: one owner per (cluster, unit)
        accg += __shfl_xor(accg, 16, 64);
        accg += __shfl_xor(accg, 32, 64);
        accb += __shfl_xor(accb, 16, 64);
        accb += __shfl_xor(accb, 32, 64);
        if (lane < 16) {
            a.lnpart[(long)(a.ln_cg0 + c) * KPAD + unit] = accg;
            a.lnpart[(long)(a.ln_ncg + a.ln_cg0 + c) * KPAD + unit] = accb;
        }
    }
}

typedef void (*Rec2fKernel)(R2Args);
// (the LayerNorm variants exist with the run-time activation only: no shipped recipe normalises h_t)
template <int CELL>
Rec2fKernel pickf_fwd(int act, bool ln) {
    if (ln) return rec2f_fwd_kernel<CELL, -1, true>;
    return act == PK_ACT_RELU ? rec2f_fwd_kernel<CELL, PK_ACT_RELU, false>
         : act == PK_ACT_TANH ? rec2f_fwd_kernel<CELL, PK_ACT_TANH, false> : rec2f_fwd_kernel<CELL, -1, false>;
}
template <int CELL>
Rec2fKernel pickf_bwd(int act, bool ln) {
    if (ln) return rec2f_bwd_kernel<CELL, -1, true>;
    return act == PK_ACT_RELU ? rec2f_bwd_kernel<CELL, PK_ACT_RELU, false>
         : act == PK_ACT_TANH ? rec2f_bwd_kernel<CELL, PK_ACT_TANH, false> : rec2f_bwd_kernel<CELL, -1, false>;
}

int self_fill2f() {
    static int v = -1;
    if (v < 0) {
        const char* e = pk_experiment("rec4_self_fill");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v;
}

int grant_lds(Rec2fKernel k, size_t lds) {
    struct Entry { Rec2fKernel k; size_t lds; };
    static Entry granted[16];
    static int n = 0;
    for (int i = 0; i < n; ++i)
        if (granted[i].k == k && granted[i].lds >= lds) return 0;
    PK_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (n < 16) granted[n++] = Entry{k, lds};
    return 0;
}

}  // namespace

// Does the exact-fp32 second-generation path cover this layer?  (liGRU / RNN up to 576 units; LSTM keeps the
// first-generation kernels.)
int pk_rec2f_covers(int cell, int H) { return (cell == PK_CELL_LIGRU || cell == PK_CELL_RNN) && H >= 1 && H <= KPAD; }
// floats of exchange buffer one call needs: forward T*B rows x ndir*Hp, backward ndir*T*B rows x G*Hp (pitches rounded
// up to 16 floats); the larger of the two so that one work buffer serves both passes
int64_t pk_rec2f_exchange_floats(int cell, int T, int B, int bidir, int H) {
    if (!pk_rec2f_covers(cell, H)) return 0;
    const int64_t ndir = 1 + bidir, Hp = (H + 7) & ~7, G = pk_cell_gates(cell);
    const int64_t yp = (ndir * Hp + 15) / 16 * 16, gp = (G * Hp + 15) / 16 * 16;
    const int64_t f = (int64_t)T * B * yp, b = ndir * T * B * gp;
    return (f > b ? f : b) + 64;
}

int pk_rec2f_fwd(hipStream_t st, int cell, int act, int T, int B, int bidir, int H, const float* P, const float* pscale,
                 const float* pshift, const float* U, const float* mask, float mask_scalar, float* Y, float* S, float* Yx,
                 const PkLnHost* ln) {
    int rc = pk_rec2_check("pk_rec_fwd (fp32, persistent)", pk_rec2f_covers(cell, H), cell, T, B, bidir, H);
    if (rc) return rc;
    const int ndir = 1 + bidir, R = B * ndir, Hp = (H + 7) & ~7;
    const int64_t y_pitch = ((int64_t)ndir * Hp + 15) / 16 * 16;
    PK_REQUIRE(((uintptr_t)Yx & 15) == 0, "pk_rec_fwd: exchange buffer must be 16-byte aligned");
    PK_REQUIRE((double)T * B * y_pitch * 4.0 < 4.0e9, "pk_rec_fwd (fp32, persistent): exchange buffer exceeds the 4 GB buffer-descriptor range");
    Plan2 pl;
    rc = pk_rec2_make_plan(R, H, pl);
    if (rc) return rc;
    R2Args a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.Hp = Hp; a.YH = ndir * H; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.rpc = pl.rpc; a.row0 = 0;
    a.P = P; a.pscale = pscale; a.pshift = pshift; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = Y; a.S = S; a.Yb = nullptr; a.Xb = nullptr; a.Ypitch = (int)y_pitch;
    a.dY = nullptr; a.dP2 = nullptr; a.dGb = nullptr; a.Gpitch = 0;
    a.Yx = Yx; a.dGx = nullptr;
    rc = pk_rec2_host_setup(a, false, cell);
    if (rc) return rc;
    a.self_fill = self_fill2f();
    // the mailbox: every dword "not written yet" - written by the kernel itself a few steps ahead of its publishes, or
    // (PK_EXPERIMENT rec4_self_fill=0, one switch for both fp32 generations) by one fill in front of the launches
    if (!a.self_fill) PK_CHECK_HIP(hipMemsetAsync(Yx, 0xFF, (size_t)T * B * y_pitch * 4, st));
    rc = pk_rec2_ln_setup(st, a, pl, ln, false);
    if (rc) return rc;
    const int G = pk_cell_gates(cell);
    const size_t lds = 2 * (size_t)RMAX * pk_r2_lda_f32(KPAD) * 4 + 4 * ((size_t)(G + 1 + pk_cell_saved(cell) + (ln ? 1 : 0)) * 1024) + 16;
    const Rec2fKernel k = cell == PK_CELL_LIGRU ? pickf_fwd<PK_CELL_LIGRU>(act, ln != nullptr) : pickf_fwd<PK_CELL_RNN>(act, ln != nullptr);
    rc = grant_lds(k, lds);
    if (rc) return rc;
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        a.ln_cg0 = l * pl.C;
        rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        rc = pk_rec2_check_residency((const void*)k, 256, lds, pl.C * pl.Pn, "pk_rec_fwd (fp32, persistent)");
        if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(pl.C * pl.Pn), dim3(256), lds, st, a);
        PK_LAUNCH_CHECK();
    }
    return 0;
}

int pk_rec2f_bwd(hipStream_t st, int cell, int act, int T, int B, int bidir, int H, const float* U, const float* mask,
                 float mask_scalar, const float* Y, const float* S, const float* dY, float* dP2, float* dGx, const PkLnHost* ln) {
    int rc = pk_rec2_check("pk_rec_bwd (fp32, persistent)", pk_rec2f_covers(cell, H), cell, T, B, bidir, H);
    if (rc) return rc;
    const int ndir = 1 + bidir, R = B * ndir, Hp = (H + 7) & ~7, G = pk_cell_gates(cell);
    const int64_t g_pitch = ((int64_t)G * Hp + 15) / 16 * 16;
    PK_REQUIRE(((uintptr_t)dGx & 15) == 0, "pk_rec_bwd: exchange buffer must be 16-byte aligned");
    PK_REQUIRE((double)ndir * T * B * g_pitch * 4.0 < 4.0e9, "pk_rec_bwd (fp32, persistent): exchange buffer exceeds the 4 GB buffer-descriptor range");
    Plan2 pl;
    rc = pk_rec2_make_plan(R, H, pl);
    if (rc) return rc;
    R2Args a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.Hp = Hp; a.YH = ndir * H; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.rpc = pl.rpc; a.row0 = 0;
    a.P = nullptr; a.pscale = nullptr; a.pshift = nullptr; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = const_cast<float*>(Y); a.S = const_cast<float*>(S); a.Yb = nullptr; a.Xb = nullptr; a.Ypitch = 0;
    a.dY = dY; a.dP2 = dP2; a.dGb = nullptr; a.Gpitch = (int)g_pitch;
    a.Yx = nullptr; a.dGx = dGx;
    rc = pk_rec2_host_setup(a, true, cell);
    if (rc) return rc;
    a.self_fill = self_fill2f();
    if (!a.self_fill) PK_CHECK_HIP(hipMemsetAsync(dGx, 0xFF, (size_t)ndir * T * B * g_pitch * 4, st));
    rc = pk_rec2_ln_setup(st, a, pl, ln, true);
    if (rc) return rc;
    const size_t atile = (size_t)RMAX * pk_r2_lda_f32(G * KPAD) * 4;
    const size_t lds = (2 * atile > 100 * 1024 ? 1 : 2) * atile + 4 * ((size_t)(pk_cell_saved(cell) + 2 + (ln ? 1 : 0) + G) * 1024) + 16;
    const Rec2fKernel k = cell == PK_CELL_LIGRU ? pickf_bwd<PK_CELL_LIGRU>(act, ln != nullptr) : pickf_bwd<PK_CELL_RNN>(act, ln != nullptr);
    rc = grant_lds(k, lds);
    if (rc) return rc;
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        a.ln_cg0 = l * pl.C;
        rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        rc = pk_rec2_check_residency((const void*)k, 256, lds, pl.C * pl.Pn, "pk_rec_bwd (fp32, persistent)");
        if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(pl.C * pl.Pn), dim3(256), lds, st, a);
        PK_LAUNCH_CHECK();
    }
    return pk_rec2_ln_finish(st, a, ln);
}
