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
#include <atomic>
#include <stdlib.h>

#include "pk_rec2_common.h"

namespace {

// ============================================================================
// forward
// ============================================================================
// ACT >= 0: the activation is a compile-time constant (relu, tanh: what the shipped recipes use) and the switch in
// pk_act folds away; ACT < 0: run-time a.act.  With the run-time switch the gate math of one step executes ~40 scalar
// branches (one switch per row and use), which shows up as several hundred clocks on the dependency chain.
// LN: per-step LayerNorm of h_t (a.ln_gamma / ln_beta; neural_networks.py:23-33 at :466-467, :1138-1139, :1444-1445): the
// row statistics take one more exchange inside the step (ln_row_allreduce, pk_rec2_common.h: fp32 partial sums, no
// barrier); the normalised h_t is what is stored, published and fed back, the pre-LN value and (mean, 1 / (std + eps))
// are saved for the backward pass.
// NOSAVE: the forward pass of a validation / forward chunk (torch.no_grad, core.py:644-671): nothing is saved for a backward
// pass (a.S is null), and the fp32 output is written only when a.Y is not null - the inner layers of a stack hand their
// output on as the bf16 copy Yb alone.  Three to four of the five 16-byte stores a lane issues per step go away.
template <int CELL, int ACT, bool TR, bool LN = false, bool NOSAVE = false>
__global__ __launch_bounds__(256, 1) void rec2_fwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr int LDA = pk_r2_lda_bf16(KPAD);      // bf16 elements per A-tile row (1312 B: conflict-free b128 reads)
    constexpr int ATILE = RMAX * LDA * 2;         // bytes
    constexpr int NCH = (RMAX * (KPAD / 8) + 255) / 256;  // 16-byte chunks polled per lane (5)
    constexpr int NF = G + 1 + NS + (LN ? 1 : 0);         // fp32 patches of a wave
    constexpr int WAVE_LDS = NF * 1024 + 512;             // P stage | Y | S slots (| pre-LN h) | bf16 publish patch
    constexpr int LDS_TRASH = 2 * ATILE + 4 * WAVE_LDS;   // 16-byte dump slot for chunks a lane does not own
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][ATILE] | 4 x WAVE_LDS | trash

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;  // whole workgroup (uniform): this cluster has no rows
    const int ubase = p * 64 + wave * 16;
    const int unit = ubase + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;

    // ---- recurrent weights of my 16 units -> registers (once): B[k][n] = U_g[unit n][k]
    bf16x8 Bf[G][KSTEPS];
    {
        const unsigned szU = (unsigned)((size_t)G * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            u32x4 raw[KSTEPS][2];
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int k0 = kk * 32 + kq * 8;
                const unsigned off = (unsigned)(((g * H + unit) * H + k0) * 4);
                raw[kk][0] = __builtin_amdgcn_raw_buffer_load_b128(rsU, (unit_ok && k0 < H) ? off : szU, 0, 0);
                raw[kk][1] = __builtin_amdgcn_raw_buffer_load_b128(rsU, (unit_ok && k0 + 4 < H) ? off + 16 : szU, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                bf16x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = kk * 32 + kq * 8 + e;
                    const float w = (k < H) ? __uint_as_float(raw[kk][e >> 2][e & 3]) : 0.f;  // beyond H: the next row's data
                    f[e] = (short)pk_f2bf(w);
                }
                Bf[g][kk] = f;
            }
        }
    }
    float psc[G], psh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        psc[g] = unit_ok ? a.pscale[g * H + unit] : 0.f;
        psh[g] = unit_ok ? a.pshift[g * H + unit] : 0.f;
    }
    for (int i = tid; i < (LDS_TRASH + 16) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- poll descriptors: chunk ci = (row, col) of the cluster's [nrows][Hp/8] block of h_{t-1}
    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Ypitch * 2u;  // bytes per time slab of Yb
    const unsigned szYb = (unsigned)T * TS;
    unsigned cbase[NCH], cstep[NCH];  // byte offset at step 1 and its increment per step (out of range: not my chunk)
    int clds[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = tid + 256 * i;
        const bool ok = ci < nrows * CPR;
        const int row = ok ? ci / CPR : 0, col = ok ? ci - row * CPR : 0;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        // step t reads storage time (dir ? T-t : t-1); a slot I do not own stays out of range
        cbase[i] = ok ? ((unsigned)b * a.Ypitch + dir * Hp + col * 8) * 2u + (unsigned)(dir ? (T - 1) : 0) * TS : szYb;
        cstep[i] = ok ? (dir ? 0u - TS : TS) : 0u;
        clds[i] = ok ? row * (LDA * 2) + col * 16 : LDS_TRASH;
    }
    // ---- gate-math (C/D) layout: rows kq*4 + r, unit lane&15
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
    // ---- vector layout: row lane>>2, units ubase + (lane&3)*4 .. +3
    const int vrow = lane >> 2, vu0 = ubase + (lane & 3) * 4;
    const int vn = n_base + (vrow < nrows ? vrow : 0);
    const int vdir = vn >= B ? 1 : 0, vb = vn - vdir * B;
    int vnv = H - vu0;
    vnv = vnv > 4 ? 4 : (vnv < 0 ? 0 : vnv);
    // wave-uniform: 0 = my 16 units do not straddle H, 1 = they do and H is even, 2 = H is odd
    const int edge = __any(vnv > 0 && vnv < 4) != 0 ? ((H & 1) ? 2 : 1) : 0;
    vnv = vrow < nrows ? vnv : 0;
    // element offsets of my 4 units at storage time 0 / per unit of storage time, for P, Y and S
    const unsigned vP0 = ((unsigned)vb * GH + vu0), vPs = (unsigned)B * GH;
    const unsigned vY0 = ((unsigned)vb * a.YH + vdir * H + vu0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)vdir * T * B + vb) * (NS * H) + vu0), vSs = (unsigned)B * NS * H;
    // ---- publish descriptors: lanes 0..31 store one 16-byte piece (row, 8 units) of the wave's bf16 patch
    const int prow = lane >> 1, phalf = lane & 1;
    const int pu0 = ubase + phalf * 8;
    const bool pk_ok = lane < 32 && prow < nrows && pu0 < Hp;
    const int pn = n_base + (prow < nrows ? prow : 0);
    const int pdir = pn >= B ? 1 : 0, pb = pn - pdir * B;
    const unsigned pbase = pk_ok ? ((unsigned)pb * a.Ypitch + pdir * Hp + pu0) * 2u : szYb;  // out of range: dropped

    unsigned char* wl = smem + 2 * ATILE + wave * WAVE_LDS;
    float* patchP = reinterpret_cast<float*>(wl);                     // [G][256]
    float* patchY = reinterpret_cast<float*>(wl + G * 1024);          // [256]
    float* patchS = reinterpret_cast<float*>(wl + (G + 1) * 1024);    // [NS][256]
    float* patchL = reinterpret_cast<float*>(wl + (G + 1 + NS) * 1024);  // (LN only) [256]
    unsigned short* patchB = reinterpret_cast<unsigned short*>(wl + NF * 1024);  // [16][16] bf16
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.Yb, szYb);
    float* trash = a.trash + (tid & 63) * 4;
    // ---- per-step LayerNorm state
    const LnSlots ls = LN ? ln_slots(a, c, p, wave, lane, T) : LnSlots();
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(LN ? (const void*)a.lnx : (const void*)a.Yb, LN ? ls.size : 0u);
    const float gam = (LN && unit_ok) ? a.ln_gamma[unit] : 0.f, bet = (LN && unit_ok) ? a.ln_beta[unit] : 0.f;
    const float invH = 1.0f / (float)H, inv_nm1 = 1.0f / (float)(H - 1);
    float piv[4] = {0.f, 0.f, 0.f, 0.f};  // pivot of the one-pass variance: the row's mean of the previous step
    // the statistics of row kq*4 + u are written by lane u (< 4) of each DPP row of (member 0, wave 0)
    pk_f32x2 st_val = {0.f, 0.f};
    const int st_row = kq * 4 + (lane & 3);
    const bool st_ok = LN && p == 0 && wave == 0 && (lane & 15) < 4 && st_row < nrows;
    float* const st_base = st_ok ? a.lnstat + (long)(n_base + st_row) * 2 : trash;
    const long st_step = st_ok ? (long)a.R * 2 : 0;

    // projections of step tt (vector layout; staged to the gate-math layout at the top of that step)
    f32x4 pv[G];
    auto load_proj = [&](int tt, auto E) {
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
#pragma unroll
        for (int g = 0; g < G; ++g) pv[g] = ld4<decltype(E)::value>(a.P, vP0 + ts * vPs + g * H, vnv);
    };
    // layer output and saved gates of step tt: wave patches -> HBM, 16 bytes per lane
    const bool want_y = a.Y != nullptr;
    auto flush_outputs = [&](int tt, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
        if (!NOSAVE || want_y) st4<EE>(a.Y, vY0 + ts * vYs, vnv, trash, patch_get_vec(patchY, lane));
        if constexpr (!NOSAVE) {
#pragma unroll
            for (int k = 0; k < NS; ++k) st4<EE>(a.S, vS0 + ts * vSs + k * H, vnv, trash, patch_get_vec(patchS + k * 256, lane));
        }
        if (LN) {
            st4<EE>(a.lnh, vY0 + ts * vYs, vnv, trash, patch_get_vec(patchL, lane));
            *reinterpret_cast<pk_f32x2*>(st_base + (long)tt * st_step) = st_val;
        }
    };
#define PK_LP0(E) load_proj(0, E)
    PK_EDGE_DISPATCH(PK_LP0);
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (a.self_fill) {  // my chunks of the first slabs, visible everywhere before the handshake lets anyone poll
        for (int tt = 0; tt < PK_R2_FILL_AHEAD && tt < T; ++tt)
            pub_store<false>(rs, pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - tt) : tt) * TS : 0u), sentinel);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    // (made explicitly wave-uniform: the specialised loops below must be entered through scalar branches)
    const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    // the time loop, instantiated per (XCD-local fast path?, static edge case?) so that neither choice is a branch in it
    auto run = [&](auto FASTC, auto SEC) {  // FASTC < 0: run-time choice (the LayerNorm variants: one copy of the loop)
    const bool fast = decltype(FASTC)::value < 0 ? fast_rt : decltype(FASTC)::value != 0;
    constexpr int SE = decltype(SEC)::value;
    for (int t = 0; t < T; ++t) {
        const int step_idx = t;
        PK_TRACE(0);
        f32x4 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned char* At = smem + (t & 1) * ATILE;
        if (t > 0) {
            unsigned goff[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) goff[i] = cbase[i] + (unsigned)(t - 1) * cstep[i];
            // a poll that arrives before the other members' stores costs a whole extra round trip
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
            int retries = 0;
            dead = fast ? poll_to_lds<NCH, true>(rs, goff, clds, At, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds<NCH, false>(rs, goff, clds, At, a.err, a.spin_limit, lane, dead, retries);
            if (a.trace != nullptr && blockIdx.x == 0 && tid == 0) a.trace[(long)step_idx * 8 + 6] = (unsigned long long)retries;
        }
        PK_TRACE(1);
        // stage this step's projections (loaded one step ago) into the gate-math layout
#pragma unroll
        for (int g = 0; g < G; ++g) patch_put_vec(patchP + g * 256, lane, pv[g]);
        if (t > 0) PK_BARRIER_LDS();
        else PK_LDS_ORDER();
        PK_TRACE(2);
        // off the dependency chain, behind the barrier (they overlap the MFMA phase and have the rest of the step
        // to drain before the next poll): fp32 outputs of the previous step, projections of the next one
        if (t > 0) {
#define PK_FO(E) flush_outputs(t - 1, E)
            PK_EDGE_DISPATCH_S(PK_FO);
        }
        if (t + 1 < T) {
#define PK_LP1(E) load_proj(t + 1, E)
            PK_EDGE_DISPATCH_S(PK_LP1);
        }
        if (a.self_fill && t + PK_R2_FILL_AHEAD < T) {
            const unsigned off = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - (t + PK_R2_FILL_AHEAD)) : (t + PK_R2_FILL_AHEAD)) * TS : 0u);
            if (fast) pub_store<true>(rs, off, sentinel);
            else pub_store<false>(rs, off, sentinel);
        }
        const bool empty = TR && a.empty_step != 0;  // diagnostics: the step without its arithmetic
        if (t > 0 && !empty) {
            const unsigned char* Ar = At + (lane & 15) * (LDA * 2) + kq * 16;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ar + kk * 64);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[g][kk], acc[g], 0, 0, 0);
            }
        }
        PK_TRACE(3);
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
            if (empty) {
                h = 0.25f;
                cc = 0.f;
#pragma unroll
                for (int k = 0; k < NS; ++k) s[k] = pr[0];
            } else {
                pk_cell_fwd<CELL>(act, pr, hprev[r], cprev[r], msk[r], h, cc, s);
            }
            h = rvf[r] != 0.f ? h : 0.f;  // rows / units outside the layer carry exact zeros (published as padding)
            cc = rvf[r] != 0.f ? cc : 0.f;
            hprev[r] = h;
            cprev[r] = cc;
            hv[r] = h;
#pragma unroll
            for (int k = 0; k < NS; ++k) sv[k][r] = s[k];
            if (!LN) patchB[(kq * 4 + r) * 16 + (lane & 15)] = to_bf_pub(h);
        }
        if (LN) {
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
                patchB[(kq * 4 + r) * 16 + (lane & 15)] = to_bf_pub(hn);
            }
            const int u3 = lane & 3;
            st_val[0] = u3 == 0 ? mu4[0] : u3 == 1 ? mu4[1] : u3 == 2 ? mu4[2] : mu4[3];
            st_val[1] = u3 == 0 ? ri4[0] : u3 == 1 ? ri4[1] : u3 == 2 ? ri4[2] : ri4[3];
        }
        PK_TRACE(4);
        // ---- publish h_t first: it is what the other workgroups of the cluster wait for
        PK_LDS_ORDER();
        {
            const u32x4 o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(patchB) + (prow & 15) * 32 + phalf * 16);
            const unsigned off = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - t) : t) * TS : 0u);
            if (fast) pub_store<true>(rs, off, o);
            else pub_store<false>(rs, off, o);
        }
        // ---- the fp32 outputs (layer output, gates saved for backward) go to the wave patches; they are
        // written to HBM at the top of the next step, behind its poll loads
        if (!NOSAVE || want_y) patch_put_cd(patchY, kq, lane, hv);
        if constexpr (!NOSAVE) {
#pragma unroll
            for (int k = 0; k < NS; ++k) patch_put_cd(patchS + k * 256, kq, lane, sv[k]);
        }
        PK_LDS_ORDER();
        PK_TRACE(5);
    }
    };
    if constexpr (LN) {
        run(BoolC<-1>(), BoolC<-1>());
    } else {
        PK_RUN_SPECIALISED(run, fast_rt);
    }
#define PK_FOL(E) flush_outputs(T - 1, E)
    PK_EDGE_DISPATCH(PK_FOL);
}

// ============================================================================
// backward: dL/dh_{t-1} = direct + [dgates_t] . [U_0; U_1; ...]
// ============================================================================
// LN: the gradient arriving at h_t goes through the LayerNorm backward first (two more row sums -> ln_row_allreduce);
// d gamma / d beta are accumulated per lane over the steps and leave as per-cluster partial sums (a.lnpart).
template <int CELL, int ACT, bool TR, bool LN = false>
__global__ __launch_bounds__(256, 1) void rec2_bwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr bool LSTM = (CELL == PK_CELL_LSTM);
    constexpr int LDA = pk_r2_lda_bf16(G * KPAD);
    constexpr int ATILE = RMAX * LDA * 2;
    constexpr int NBUF = (2 * ATILE > 96 * 1024) ? 1 : 2;  // LSTM: one A tile (74 KB) + an extra barrier per step
    constexpr int NCH = (RMAX * G * (KPAD / 8) + 255) / 256;
    constexpr int NIN = NS + 2 + (LSTM ? 1 : 0) + (LN ? 1 : 0);  // saved gates, h_{t-1}, dY (, c_{t-1}) (, pre-LN h_t)
    constexpr int ILN = NS + 2 + (LSTM ? 1 : 0);            // slot of the pre-LN h_t
    constexpr int WAVE_LDS = (NIN + G) * 1024 + G * 512;    // input patches | dgate fp32 patches | dgate bf16 patches
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

    // B[kidx = (g, j)][n = unit] = U_g[j][unit]
    bf16x8 Bf[G][KSTEPS];
    {
        const unsigned szU = (unsigned)((size_t)G * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                unsigned raw[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = kk * 32 + kq * 8 + e;
                    raw[e] = __builtin_amdgcn_raw_buffer_load_b32(rsU, (unit_ok && j < H) ? (unsigned)(((g * H + j) * H + unit) * 4) : szU, 0, 0);
                }
                bf16x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (short)pk_f2bf(__uint_as_float(raw[e]));
                Bf[g][kk] = f;
            }
    }
    for (int i = tid; i < (LDS_TRASH + 16) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- poll descriptors: chunk ci = (row, gate, col) of the cluster's dgates_{t+1} block
    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Gpitch * 2u;  // bytes per time slab of dGb
    const unsigned ndir = (unsigned)(a.R / B);
    const unsigned szGb = ndir * (unsigned)T * TS;
    unsigned cbase[NCH], cstep[NCH];  // byte offset of the first polled step and its increment per iteration
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
        // iteration it (t = T-1-it, it >= 1) reads storage time (dir ? T-2-t : t+1) = (dir ? it-1 : T-it)
        cbase[i] = ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)b * a.Gpitch + g * Hp + col * 8) * 2u +
                            (unsigned)(dir ? 0 : (T - 1)) * TS
                      : szGb;
        cstep[i] = ok ? (dir ? TS : 0u - TS) : 0u;
        clds[i] = ok ? row * (LDA * 2) + (g * KPAD + col * 8) * 2 : LDS_TRASH;
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
    // wave-uniform: 0 = my 16 units do not straddle H, 1 = they do and H is even, 2 = H is odd
    const int edge = __any(vnv > 0 && vnv < 4) != 0 ? ((H & 1) ? 2 : 1) : 0;
    vnv = vrow < nrows ? vnv : 0;
    const unsigned vY0 = ((unsigned)vb * a.YH + vdir * H + vu0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)vdir * TB + vb) * (NS * H) + vu0), vSs = (unsigned)B * NS * H;
    const unsigned vG0 = (((unsigned)vdir * TB + vb) * GH + vu0), vGs = (unsigned)B * GH;
    const int prow = lane >> 1, phalf = lane & 1;
    const int pu0 = ubase + phalf * 8;
    const bool pk_ok = lane < 32 && prow < nrows && pu0 < Hp;
    const int pn = n_base + (prow < nrows ? prow : 0);
    const int pdir = pn >= B ? 1 : 0, pb = pn - pdir * B;
    const unsigned pbase = pk_ok ? (unsigned)pdir * (unsigned)T * TS + ((unsigned)pb * a.Gpitch + pu0) * 2u : szGb;

    unsigned char* wl = smem + NBUF * ATILE + wave * WAVE_LDS;
    float* patchI = reinterpret_cast<float*>(wl);                       // [NIN][256]: S slots, hp, dY (, cp)
    float* patchG = reinterpret_cast<float*>(wl + NIN * 1024);          // [G][256] fp32 gate gradients
    unsigned short* patchB = reinterpret_cast<unsigned short*>(wl + (NIN + G) * 1024);  // [G][16][16] bf16
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dGb, szGb);
    float* trash = a.trash + (tid & 63) * 4;
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(LN ? (const void*)a.lnx : (const void*)a.dGb,
                                                 LN ? (unsigned)T * ((unsigned)a.ln_ncg * 4u * 4u * (unsigned)a.Pn * 32u) : 0u);

    // saved tensors of one step in the vector layout: [0..NS) gates, NS = h_{t-1}, NS+1 = dY, NS+2 = c_{t-1}
    f32x4 iv[NIN];
    auto load_step_e = [&](int t, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(vdir ? (T - 1 - t) : t);
        const unsigned tp = t > 0 ? (vdir ? ts + 1 : ts - 1) : ts;  // storage time of step t-1 (any valid row when t == 0)
        const int nvp = t > 0 ? vnv : 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) iv[k] = ld4<EE>(a.S, vS0 + ts * vSs + k * H, vnv);
        iv[NS] = ld4<EE>(a.Y, vY0 + tp * vYs, nvp);
        iv[NS + 1] = ld4<EE>(a.dY, vY0 + ts * vYs, vnv);
        if (LSTM) iv[NS + 2] = ld4<EE>(a.S, vS0 + tp * vSs + 4 * H, nvp);
        if (LN) iv[ILN] = ld4<EE>(a.lnh, vY0 + ts * vYs, vnv);
        if (t == 0) {  // h_{-1} = c_{-1} = 0
            iv[NS] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (LSTM) iv[NS + 2] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // ---- per-step LayerNorm state: (mean, 1/(std+eps)) of my four rows, loaded one step ahead like the saved gates
    const LnSlots ls = LN ? ln_slots(a, c, p, wave, lane, T) : LnSlots();
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
    auto load_step = [&](int t, auto SEC) {  // SEC: BoolC<0> = static no-edge case, BoolC<-1> = run-time dispatch
        constexpr int SE = decltype(SEC)::value;
#define PK_LS(E) load_step_e(t, E)
        PK_EDGE_DISPATCH_S(PK_LS);
    };
    auto flush_outputs_e = [&](int tt, auto E) {
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
#pragma unroll
        for (int g = 0; g < G; ++g)
            st4<decltype(E)::value>(a.dP2, vG0 + ts * vGs + g * H, vnv, trash, patch_get_vec(patchG + g * 256, lane));
    };
    auto flush_outputs = [&](int tt, auto SEC) {
        constexpr int SE = decltype(SEC)::value;
        if (a.dP2 == nullptr) return;  // perf mode: BatchNorm backward works from the bf16 copy
#define PK_FOB(E) flush_outputs_e(tt, E)
        PK_EDGE_DISPATCH_S(PK_FOB);
    };
    load_step(T - 1, BoolC<-1>());
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto fill_slab = [&](int tt, auto FASTC) {  // my G chunks of the slab that step tt will publish
        const unsigned off = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - tt) : tt) * TS : 0u);
#pragma unroll
        for (int g = 0; g < G; ++g) pub_store<decltype(FASTC)::value != 0>(rs, off + (pk_ok ? (unsigned)(g * Hp) * 2u : 0u), sentinel);
    };
    if (a.self_fill) {  // the first slabs (the steps run T-1, T-2, ...), in place before the handshake lets anyone poll
        for (int k = 0; k < PK_R2_FILL_AHEAD && k < T; ++k) fill_slab(T - 1 - k, BoolC<0>());
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    // (made explicitly wave-uniform: the specialised loops below must be entered through scalar branches)
    const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    auto run = [&](auto FASTC, auto SEC) {  // see the forward kernel; FASTC < 0: run-time choice
    const bool fast = decltype(FASTC)::value < 0 ? fast_rt : decltype(FASTC)::value != 0;
    constexpr int SE = decltype(SEC)::value;
    int it = 0;
    for (int t = T - 1; t >= 0; --t, ++it) {
        const int step_idx = it;
        PK_TRACE(0);
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
            if (a.trace != nullptr && blockIdx.x == 0 && tid == 0) a.trace[(long)step_idx * 8 + 6] = (unsigned long long)retries;
        }
        PK_TRACE(1);
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
        PK_TRACE(2);
        // off the dependency chain, behind the barrier: fp32 gate gradients of the previous step (if wanted) and
        // the saved tensors of the next one
        if (t < T - 1) flush_outputs(t + 1, SEC);
        if (t > 0) load_step(t - 1, SEC);
        if (LN && t > 0) load_stats(t - 1);
        if (a.self_fill && t - PK_R2_FILL_AHEAD >= 0) {
            if (fast) fill_slab(t - PK_R2_FILL_AHEAD, BoolC<1>());
            else fill_slab(t - PK_R2_FILL_AHEAD, BoolC<0>());
        }
        const bool empty = TR && a.empty_step != 0;
        if (t < T - 1 && !empty) {
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
        if (NBUF == 1 && t < T - 1) PK_BARRIER_LDS();  // single A tile: everyone is done reading before the next poll refills it
        PK_TRACE(3);
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
                dd[r] = sin[ILN][r] - mu4[r];
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
            const float cp = LSTM ? sin[NS + 2][r] : 0.f;
            const float dh = dh4[r];
            float dg[G], dhd, dcp;
            if (empty) {
                dhd = 0.f;
                dcp = 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) dg[g] = 0.125f;
            } else {
                pk_cell_bwd<CELL>(act, s, hp, cp, msk[r], dh, dc_car[r], dg, dhd, dcp);
            }
            // rows / units outside the layer: exact zeros (select, not multiply: their inputs are arbitrary)
            dh_dir[r] = rvf[r] != 0.f ? dhd : 0.f;
            dc_car[r] = rvf[r] != 0.f ? dcp : 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float d = rvf[r] != 0.f ? dg[g] : 0.f;
                dgv[g][r] = d;
                patchB[g * 256 + (kq * 4 + r) * 16 + (lane & 15)] = to_bf_pub(d);
            }
        }
        PK_TRACE(4);
        PK_LDS_ORDER();
        {
            const unsigned off = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - t) : t) * TS : 0u);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u32x4 o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(patchB) + g * 512 + (prow & 15) * 32 + phalf * 16);
                const unsigned og = off + (pk_ok ? (unsigned)(g * Hp) * 2u : 0u);
                if (fast) pub_store<true>(rs, og, o);
                else pub_store<false>(rs, og, o);
            }
        }
        if (a.dP2 != nullptr) {
#pragma unroll
            for (int g = 0; g < G; ++g) patch_put_cd(patchG + g * 256, kq, lane, dgv[g]);
            PK_LDS_ORDER();
        }
        PK_TRACE(5);
    }
    };
    // (one copy, run-time choices: with this larger loop instantiated per fast / edge case the backward kernel measured
    // 2-4 % SLOWER on the same box, while the forward kernel gains 4 % from it)
    run(BoolC<-1>(), BoolC<-1>());
    flush_outputs(0, BoolC<-1>());
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

unsigned long long* g2_trace = nullptr;  // set by pk_persist2_set_trace (diagnostics only)
unsigned* g2_err_host = nullptr;
unsigned* g2_err_dev = nullptr;
int g2_force_safe = 0;
int g2_poll_delay = -1;  // < 0: per-pass defaults (pk_rec2_host_setup)
int g2_empty_step = 0;
unsigned* g2_xcd_tab = nullptr;  // [256][16] handshake words (library-owned scratch, one launch at a time)
float* g2_trash = nullptr;        // write-only dump page for masked-off vector stores
constexpr size_t XCD_TAB_BYTES = 256 * 16 * sizeof(unsigned);
int ensure_err2() {
    if (g2_err_host) return 0;
    PK_CHECK_HIP(hipMalloc((void**)&g2_xcd_tab, XCD_TAB_BYTES));
    PK_CHECK_HIP(hipMemset(g2_xcd_tab, 0xFF, XCD_TAB_BYTES));  // no generation is current yet (pk_rec2_reset_handshake)
    PK_CHECK_HIP(hipDeviceSynchronize());
    PK_CHECK_HIP(hipMalloc((void**)&g2_trash, 4096));
    PK_CHECK_HIP(hipHostMalloc((void**)&g2_err_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *g2_err_host = 0;
    PK_CHECK_HIP(hipHostGetDevicePointer((void**)&g2_err_dev, g2_err_host, 0));
    return 0;
}

}  // namespace
int pk_rec2_make_plan(int R, int H, Plan2& pl) {
    pl.Pn = (H + 63) / 64;
    const int ncu = pk_num_cu();
    int C = ncu / pl.Pn;
    PK_REQUIRE(C >= 1, "persistent recurrence: H=%d needs %d workgroups per cluster but the device has %d CUs", H, pl.Pn,
               ncu);
    if (C > 256) C = 256;
    if (C >= 8) C -= C % 8;  // members of one cluster congruent mod 8: one XCD under round-robin dispatch (speed only)
    {   // no more clusters than full 16-row MFMA tiles need (rounded up to the XCD count): a step costs the same for 11
        // rows as for 16, and the CUs left over go to the weight-gradient GEMMs of the side stream (256 rows: 16 clusters
        // instead of 24, 19.09 -> 18.99 ms per step; a count that is not a multiple of 8 straddles XCDs: 23.6 ms).
        // PK_EXPERIMENT rec_clusters overrides the cap.
        static int cap = -1;
        if (cap < 0) {
            const char* e = pk_experiment("rec_clusters");
            cap = e ? atoi(e) : 0;
        }
        const int full = (((R + RMAX - 1) / RMAX) + 7) & ~7;
        const int want = cap > 0 ? cap : full;
        if (want < C) C = want;
    }
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

// ---- per-step LayerNorm: layout of the saved tensor and of the scratch (one definition for kernels, host and callers)
static inline int64_t ln_stat_offset(int64_t T, int64_t R, int64_t H) { return (T * R * H + 3) / 4 * 4; }
extern "C" int64_t pk_rec_ln_saved_floats(int T, int B, int bidir, int H) {
    const int64_t R = (int64_t)B * (1 + bidir);
    const int64_t pers = ln_stat_offset(T, R, H) + (int64_t)T * R * 2 + 64;  // pre-LN h in Y's layout | (mean, rinv) per (step, row)
    const int64_t step = (int64_t)T * R * (H + 2);                           // step-wise algorithm: [mean, rinv, pre-LN h] rows
    return pers > step ? pers : step;
}
static int64_t ln_exchange_floats(int T, const Plan2& pl) { return (int64_t)(T + 1) * pl.launches * pl.C * 4 * (4 * pl.Pn) * 8; }
int64_t pk_rec4f_ln_work_floats(int T, int B, int bidir, int H);  // pk_rec_persist4_f32.hip: 18 workgroups per cluster
extern "C" int64_t pk_rec_ln_work_floats(int T, int B, int bidir, int H) {
    Plan2 pl;
    if (T <= 0 || B <= 0 || H <= 0 || H > KPAD || pk_rec2_make_plan(B * (1 + bidir), H, pl) != 0) return 0;
    const int64_t ncg = (int64_t)pl.launches * pl.C;
    const int64_t gen2 = ln_exchange_floats(T, pl) + 2 * ncg * KPAD + pk_bn_partial_floats(ncg, KPAD) + 256;
    const int64_t gen4 = pk_rec4f_ln_work_floats(T, B, bidir, H);  // (the caller does not say which kernels will run)
    return gen2 > gen4 ? gen2 : gen4;
}
int pk_rec2_ln_setup(hipStream_t st, R2Args& a, const Plan2& pl, const PkLnHost* ln, bool backward) {
    a.ln_gamma = a.ln_beta = nullptr;
    a.lnh = a.lnstat = a.lnx = a.lnpart = nullptr;
    a.ln_eps = 0.f; a.ln_cg0 = 0; a.ln_ncg = 0;
    if (ln == nullptr) return 0;
    PK_REQUIRE(ln->gamma && ln->LNS && ln->lnwork && (backward ? (ln->dgamma && ln->dbeta) : ln->beta != nullptr),
               "persistent recurrence with per-step LayerNorm: gamma / beta / LNS / scratch pointers missing");
    PK_REQUIRE(a.H > 1, "per-step LayerNorm needs H > 1");
    const int64_t xf = ln_exchange_floats(a.T, pl);
    PK_REQUIRE((double)xf * 4.0 < 4.0e9, "per-step LayerNorm: the row-statistics exchange exceeds the 4 GB buffer-descriptor range");
    PK_REQUIRE((((uintptr_t)ln->LNS | (uintptr_t)ln->lnwork) & 15) == 0, "per-step LayerNorm: LNS / scratch must be 16-byte aligned");
    a.ln_gamma = ln->gamma; a.ln_beta = ln->beta; a.ln_eps = ln->eps;
    a.lnh = ln->LNS;
    a.lnstat = ln->LNS + ln_stat_offset(a.T, a.R, a.H);
    a.lnx = ln->lnwork;
    a.lnpart = ln->lnwork + xf;
    a.ln_ncg = pl.launches * pl.C;
    PK_CHECK_HIP(hipMemsetAsync(a.lnx, 0xFF, (size_t)xf * 4, st));  // every dword "not written yet"
    if (backward) PK_CHECK_HIP(hipMemsetAsync(a.lnpart, 0, (size_t)2 * a.ln_ncg * KPAD * 4, st));  // clusters without rows add nothing
    return 0;
}
int pk_rec2_ln_finish(hipStream_t st, const R2Args& a, const PkLnHost* ln) {
    if (ln == nullptr) return 0;
    float* part = a.lnpart + (size_t)2 * a.ln_ncg * KPAD;
    int rc = pk_colsum((void*)st, a.lnpart, nullptr, KPAD, a.ln_ncg, a.H, part, ln->dgamma);
    if (rc) return rc;
    return pk_colsum((void*)st, a.lnpart + (size_t)a.ln_ncg * KPAD, nullptr, KPAD, a.ln_ncg, a.H, part, ln->dbeta);
}

int pk_rec2_check(const char* who, int cell_ok, int cell, int T, int B, int bidir, int H) {
    PK_REQUIRE(cell_ok, "%s: cell %d is not covered by this entry point", who, cell);
    PK_REQUIRE(T > 0 && B > 0 && H > 0 && H <= KPAD && (bidir == 0 || bidir == 1), "%s: bad geometry T=%d B=%d H=%d (H <= %d)",
               who, T, B, H, KPAD);
    return 0;
}

// Idle time between a workgroup's publish and its first poll of the next step, in s_sleep units of 64 clocks
// (PK_EXPERIMENT poll_delay_fwd / PK_EXPERIMENT poll_delay_bwd override).  A poll that arrives before the other members' stores is repeated;
// while the sentinel test was expensive that cost ~2300 clocks and 6 units were best (re-polls 0.4 -> 0.04 per step);
// with the dword-level test a re-poll is cheap and the DELAY sweep of tools/trace_rec2.py (Li-GRU, BASELINE geometry) is
// flat between 0 and 2 units (forward 5160-5200, backward 5750-5870 clocks per step) and rises beyond.
static int default_poll_delay(bool backward, int cell) {
    static int env[2] = {-2, -2};
    int& e = env[backward ? 1 : 0];
    if (e == -2) {
        const char* v = pk_experiment(backward ? "poll_delay_bwd" : "poll_delay_fwd");
        e = v ? atoi(v) : -1;
    }
    if (e >= 0) return e;
    // Per cell and pass.  Li-GRU (end of round 2, 16 clusters + self-filling exchange, whole training step on one box):
    // backward 1 -> 18.47 ms, 0 -> 18.36, 2 -> 18.24; forward 1 vs 2: 18.45 vs 18.47.  The eight-wave LSTM backward polls
    // from TWO waves per SIMD whose helper wave is delayed separately (PK_EXPERIMENT lstm_helper_delay): any idle time in front of
    // its first poll only delays the hand-over - round 3, timit_lstm step on one box, two rounds each: 0 -> 26.5 / 26.5 ms,
    // 1 -> 40.8 / 36.6, 2 -> 33.9 / 36.0, 3 -> 35.2 / 39.7 (the round-2 default of 2 for every cell is what made the
    // driver's LSTM line 33.2 ms against the 26.7 ms measured before that commit).
    if (backward && cell == PK_CELL_LSTM) return 0;
    return 2;
}

int pk_rec2_host_setup(R2Args& a, bool backward, int cell) {
    int rc = ensure_err2();
    if (rc) return rc;
    a.err = g2_err_dev; a.spin_limit = 400000; a.trace = g2_trace; a.xcd_tab = g2_xcd_tab; a.force_safe = g2_force_safe;
    a.trash = g2_trash; a.poll_delay = g2_poll_delay >= 0 ? g2_poll_delay : default_poll_delay(backward, cell);
    a.helper_delay = 0;
    a.empty_step = g2_empty_step;
    a.self_fill = 0;
    {   // PK_EXPERIMENT rec_flush_late=1: the third-generation kernels issue a step's output stores / next-step loads behind its MFMA
        // block instead of right behind the barrier (A/B switch)
        static int fl = -1;
        if (fl < 0) {
            const char* e = pk_experiment("rec_flush_late");
            fl = e ? atoi(e) : 0;  // (third generation: 1 = late; role-split kernels: bit 0 no priorities, bit 1 nt stores, bit 2 nt loads)
        }
        a.flush_late = fl;
    }
    return 0;
}
int pk_rec2_check_residency(const void* kernel, int threads, size_t lds, int grid, const char* who) {
    struct Entry { const void* k; size_t lds; int threads; int cap; };
    static Entry cache[64];
    static int n_cache = 0;
    int cap = -1;
    for (int i = 0; i < n_cache; ++i)
        if (cache[i].k == kernel && cache[i].lds == lds && cache[i].threads == threads) cap = cache[i].cap;
    if (cap < 0) {
        int per_cu = 0;
        PK_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds));
        cap = per_cu * pk_num_cu();
        if (n_cache < 64) cache[n_cache++] = Entry{kernel, lds, threads, cap};
    }
    PK_REQUIRE(grid <= cap, "%s: a grid of %d workgroups cannot be co-resident on this device (%d fit): the persistent "
               "recurrence would dead-lock into spin time-outs", who, grid, cap);
    return 0;
}

int pk_rec2_reset_handshake(hipStream_t st, R2Args& a) {
    // generations 1 .. 2^28 - 17 (the table starts out as 0xFF bytes: generation 2^28 - 1, never handed out).  Inside a
    // stream capture the number is baked into the graph, so every replay would find its own words from the replay
    // before: there the table is reset by a memset node in front of the kernel, as it used to be everywhere.
    static std::atomic<unsigned> gen{0};  // (atomic: two host threads must never hand out the same number)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
    unsigned g = gen.fetch_add(1u) + 1u;
    if (g >= 0x0FFFFFEFu) {  // wrap: nothing of the old numbering may survive
        // (one handshake table per process, used by ONE persistent launch at a time: every caller launches on the stream
        // torch made current; the wrap needs a device-wide sync, which a stream capture does not allow)
        PK_REQUIRE(cs == hipStreamCaptureStatusNone, "persistent recurrence: the handshake generation wrapped inside a stream "
                   "capture (2^28 launches): end the capture, launch once eagerly, capture again");
        PK_CHECK_HIP(hipDeviceSynchronize());
        PK_CHECK_HIP(hipMemset(g2_xcd_tab, 0xFF, XCD_TAB_BYTES));
        PK_CHECK_HIP(hipDeviceSynchronize());
        gen.store(1u);
        g = 1u;
    }
    if (cs != hipStreamCaptureStatusNone) PK_CHECK_HIP(hipMemsetAsync(g2_xcd_tab, 0xFF, XCD_TAB_BYTES, st));
    a.hs_gen = g;
    return 0;
}

// Does the bf16 persistent kernel of this cell keep its exchange buffer filled by itself (prefilled = 2)?
extern "C" int pk_rec_self_fill(int cell) {
    (void)cell;
    return 1;  // every bf16 persistent kernel (liGRU / RNN / LSTM: pk_rec_*_bf16; GRU / minimalGRU: pk_rec2p_*_bf16)
}

// CUs a bf16 persistent recurrence over R rows of H units occupies (one workgroup per CU): what is left is what a
// kernel on another stream can get while it runs
extern "C" int pk_rec_plan_cus(int R, int H) {
    Plan2 pl;
    if (R <= 0 || H <= 0 || H > KPAD || pk_rec2_make_plan(R, H, pl) != 0) return 0;
    return pl.C * pl.Pn;
}

extern "C" void pk_persist2_set_mode(int force_safe) { g2_force_safe = force_safe ? 1 : 0; }
extern "C" void pk_persist2_set_poll_delay(int units) { g2_poll_delay = units; }  // < 0: back to the per-pass defaults
extern "C" void pk_persist2_set_trace(void* dev_buf) { g2_trace = (unsigned long long*)dev_buf; }
extern "C" void pk_persist2_set_empty_step(int on) { g2_empty_step = on; }  // 1: no arithmetic; 2 (role-split kernels): no HBM traffic either
extern "C" unsigned pk_persist2_error_count(void) { return g2_err_host ? *g2_err_host : 0u; }
extern "C" void pk_persist2_error_reset(void) {
    if (g2_err_host) *g2_err_host = 0u;
}

typedef void (*Rec2Kernel)(R2Args);
inline int act_slot(int act) { return act == PK_ACT_RELU ? 0 : act == PK_ACT_TANH ? 1 : 2; }
// the phase trace (pk_persist2_set_trace, tools/trace_rec2.py) is compiled into the Li-GRU / relu kernels only
template <int CELL>
Rec2Kernel pick_fwd(int act) {
    return act == PK_ACT_RELU ? rec2_fwd_kernel<CELL, PK_ACT_RELU, false>
         : act == PK_ACT_TANH ? rec2_fwd_kernel<CELL, PK_ACT_TANH, false> : rec2_fwd_kernel<CELL, -1, false>;
}
template <int CELL>
Rec2Kernel pick_fwd_nosave(int act) {  // (S == null: validation / forward chunks)
    return act == PK_ACT_RELU ? rec2_fwd_kernel<CELL, PK_ACT_RELU, false, false, true>
         : act == PK_ACT_TANH ? rec2_fwd_kernel<CELL, PK_ACT_TANH, false, false, true> : rec2_fwd_kernel<CELL, -1, false, false, true>;
}
template <int CELL>
Rec2Kernel pick_bwd(int act) {
    return act == PK_ACT_RELU ? rec2_bwd_kernel<CELL, PK_ACT_RELU, false>
         : act == PK_ACT_TANH ? rec2_bwd_kernel<CELL, PK_ACT_TANH, false> : rec2_bwd_kernel<CELL, -1, false>;
}
inline bool traced(int cell, int act) { return g2_trace != nullptr && cell == PK_CELL_LIGRU && act == PK_ACT_RELU; }
// (the LayerNorm variants exist with the run-time activation only: no shipped recipe normalises h_t)
inline Rec2Kernel pick_fwd_ln(int cell) {
    return cell == PK_CELL_LIGRU ? rec2_fwd_kernel<PK_CELL_LIGRU, -1, false, true>
         : cell == PK_CELL_RNN ? rec2_fwd_kernel<PK_CELL_RNN, -1, false, true> : rec2_fwd_kernel<PK_CELL_LSTM, -1, false, true>;
}
inline Rec2Kernel pick_bwd_ln(int cell) {
    return cell == PK_CELL_LIGRU ? rec2_bwd_kernel<PK_CELL_LIGRU, -1, false, true>
         : cell == PK_CELL_RNN ? rec2_bwd_kernel<PK_CELL_RNN, -1, false, true> : rec2_bwd_kernel<PK_CELL_LSTM, -1, false, true>;
}
inline Rec2Kernel pick_fwd(int cell, int act) {
    if (traced(cell, act)) return rec2_fwd_kernel<PK_CELL_LIGRU, PK_ACT_RELU, true>;
    return cell == PK_CELL_LIGRU ? pick_fwd<PK_CELL_LIGRU>(act) : cell == PK_CELL_RNN ? pick_fwd<PK_CELL_RNN>(act) : pick_fwd<PK_CELL_LSTM>(act);
}
inline Rec2Kernel pick_bwd(int cell, int act) {
    if (traced(cell, act)) return rec2_bwd_kernel<PK_CELL_LIGRU, PK_ACT_RELU, true>;
    return cell == PK_CELL_LIGRU ? pick_bwd<PK_CELL_LIGRU>(act) : cell == PK_CELL_RNN ? pick_bwd<PK_CELL_RNN>(act) : pick_bwd<PK_CELL_LSTM>(act);
}

static int rec_fwd_bf16_impl(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                             const float* pscale, const float* pshift, const float* U, const float* mask,
                             float mask_scalar, float* Y, float* S, uint16_t* Yb, int64_t y_pitch, int prefilled,
                             const PkLnHost* ln) {
    int rc = pk_rec2_check("pk_rec_fwd_bf16", 1, cell, T, B, bidir, H);
    if (rc) return rc;
    // S == null: a validation / forward chunk, nothing is saved for a backward pass; Y == null with it: an inner layer of a
    // stack, whose output is the bf16 copy Yb alone
    PK_REQUIRE(S != nullptr || (ln == nullptr && !traced(cell, act)), "pk_rec_fwd_bf16: S may be null only without per-step LayerNorm");
    PK_REQUIRE(Y != nullptr || S == nullptr, "pk_rec_fwd_bf16: Y may be null only together with S");
    hipStream_t st = pk_stream(stream);
    const int ndir = 1 + bidir, R = B * ndir, Hp = (H + 7) & ~7;
    PK_REQUIRE(y_pitch >= (int64_t)ndir * Hp && (y_pitch % 8) == 0 && ((uintptr_t)Yb & 15) == 0,
               "pk_rec_fwd_bf16: Yb pitch must be a multiple of 8 and hold %d x %d elements", ndir, Hp);
    PK_REQUIRE((double)T * B * y_pitch * 2.0 < 4.0e9, "pk_rec_fwd_bf16: exchange buffer exceeds the 4 GB buffer-descriptor range");
    Plan2 pl;
    rc = pk_rec2_make_plan(R, H, pl);
    if (rc) return rc;
    R2Args a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.Hp = Hp; a.YH = ndir * H; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.rpc = pl.rpc; a.row0 = 0;
    a.P = P; a.pscale = pscale; a.pshift = pshift; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = Y; a.S = S; a.Yb = (unsigned short*)Yb; a.Xb = nullptr; a.Ypitch = (int)y_pitch;
    a.dY = nullptr; a.dP2 = nullptr; a.dGb = nullptr; a.Gpitch = 0;
    rc = pk_rec2_host_setup(a, false, cell);
    if (rc) return rc;
    // the bf16 layer output is the mailbox: it must hold the sentinel wherever a poll can arrive before its data.
    // prefilled: 1 = the caller filled it, 2 = fill it on the way if this kernel can (pk_rec_self_fill), 0 = fill here
    const bool lstm8 = cell == PK_CELL_LSTM && pk_rec2l_enabled();
    a.self_fill = prefilled == 2 ? 1 : 0;
    if (prefilled != 1 && !a.self_fill) PK_CHECK_HIP(hipMemsetAsync(Yb, 0xFF, (size_t)T * B * y_pitch * 2, st));
    rc = pk_rec2_ln_setup(st, a, pl, ln, false);
    if (rc) return rc;
    // (per-step LayerNorm lives in the four-wave second-generation kernels of this file, for every cell)
    if (lstm8 && !ln) return pk_rec2l_launch(st, a, pl, act, false);
    if (!ln && pk_rec3_covers(cell, 0)) return pk_rec3_launch(st, a, pl, cell, act, false, traced(cell, act));
    const int G = pk_cell_gates(cell);
    const size_t lds = 2 * (size_t)RMAX * pk_r2_lda_bf16(KPAD) * 2 + 4 * ((size_t)(G + 1 + pk_cell_saved(cell) + (ln ? 1 : 0)) * 1024 + 512) + 16;
    const bool nosave = S == nullptr;
    const Rec2Kernel k = ln ? pick_fwd_ln(cell)
                       : nosave ? (cell == PK_CELL_LIGRU ? pick_fwd_nosave<PK_CELL_LIGRU>(act) : cell == PK_CELL_RNN ? pick_fwd_nosave<PK_CELL_RNN>(act)
                                                                                                                      : pick_fwd_nosave<PK_CELL_LSTM>(act))
                                : pick_fwd(cell, act);
    {   // dynamic LDS above the 64 KB default needs the opt-in (exact size: the kernels also hold a little static LDS)
        static size_t granted[3][8] = {{0}, {0}, {0}};  // hipFuncSetAttribute is slow (milliseconds): once per kernel and size
        const int slot = cell == PK_CELL_LIGRU ? 0 : cell == PK_CELL_RNN ? 1 : 2;
        const int as = ln ? 4 : nosave ? 5 + act_slot(act) : traced(cell, act) ? 3 : act_slot(act);
        if (granted[slot][as] < lds) {
            PK_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            granted[slot][as] = lds;
        }
    }
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        a.ln_cg0 = l * pl.C;
        rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        dim3 grid(pl.C * pl.Pn), block(256);
        rc = pk_rec2_check_residency((const void*)k, 256, lds, pl.C * pl.Pn, "pk_rec_fwd_bf16");
        if (rc) return rc;
        const int help = ln ? 0 : pk_rec_helper_wanted(false, pl.launches, cell);
        if (help && (rc = pk_rec_helper_fork(st)) != 0) return rc;
        hipLaunchKernelGGL(k, grid, block, lds, st, a);
        PK_LAUNCH_CHECK();
        if (help && (rc = pk_rec_helper_launch(st, a, pl, G, pk_cell_saved(cell), false, true, help)) != 0) return rc;
    }
    return 0;
}
extern "C" int pk_rec_fwd_bf16(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                               const float* pscale, const float* pshift, const float* U, const float* mask,
                               float mask_scalar, float* Y, float* S, uint16_t* Yb, int64_t y_pitch, int prefilled) {
    return rec_fwd_bf16_impl(stream, cell, act, T, B, bidir, H, P, pscale, pshift, U, mask, mask_scalar, Y, S, Yb, y_pitch,
                             prefilled, nullptr);
}
// ... with per-step LayerNorm of h_t (neural_networks.py:466-467, :1138-1139, :1444-1445; liGRU / RNN / LSTM):
// LNS >= pk_rec_ln_saved_floats floats (saved for backward), lnwork >= pk_rec_ln_work_floats floats of scratch.
extern "C" int pk_rec_fwd_bf16_ln(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                                  const float* pscale, const float* pshift, const float* U, const float* mask,
                                  float mask_scalar, const float* ln_gamma, const float* ln_beta, float ln_eps, float* Y,
                                  float* S, float* LNS, uint16_t* Yb, int64_t y_pitch, int prefilled, float* lnwork) {
    PK_REQUIRE(ln_gamma && ln_beta && LNS && lnwork, "pk_rec_fwd_bf16_ln: null LayerNorm argument");
    const PkLnHost ln = {ln_gamma, ln_beta, ln_eps, LNS, lnwork, nullptr, nullptr};
    return rec_fwd_bf16_impl(stream, cell, act, T, B, bidir, H, P, pscale, pshift, U, mask, mask_scalar, Y, S, Yb, y_pitch,
                             prefilled, &ln);
}

static int rec_bwd_bf16_impl(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U,
                             const float* mask, float mask_scalar, const float* Y, const float* S, const float* dY,
                             float* dP2, uint16_t* dGb, int64_t g_pitch, int prefilled, const PkLnHost* ln) {
    int rc = pk_rec2_check("pk_rec_bwd_bf16", 1, cell, T, B, bidir, H);
    if (rc) return rc;
    hipStream_t st = pk_stream(stream);
    const int ndir = 1 + bidir, R = B * ndir, Hp = (H + 7) & ~7, G = pk_cell_gates(cell);
    PK_REQUIRE(g_pitch >= (int64_t)G * Hp && (g_pitch % 8) == 0 && ((uintptr_t)dGb & 15) == 0,
               "pk_rec_bwd_bf16: dGb pitch must be a multiple of 8 and hold %d x %d elements", G, Hp);
    PK_REQUIRE((double)ndir * T * B * g_pitch * 2.0 < 4.0e9, "pk_rec_bwd_bf16: exchange buffer exceeds the 4 GB buffer-descriptor range");
    Plan2 pl;
    rc = pk_rec2_make_plan(R, H, pl);
    if (rc) return rc;
    R2Args a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.Hp = Hp; a.YH = ndir * H; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.rpc = pl.rpc; a.row0 = 0;
    a.P = nullptr; a.pscale = nullptr; a.pshift = nullptr; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = const_cast<float*>(Y); a.S = const_cast<float*>(S); a.Yb = nullptr; a.Xb = nullptr; a.Ypitch = 0;
    a.dY = dY; a.dP2 = dP2; a.dGb = (unsigned short*)dGb; a.Gpitch = (int)g_pitch;
    rc = pk_rec2_host_setup(a, true, cell);
    if (rc) return rc;
    const bool lstm8 = cell == PK_CELL_LSTM && pk_rec2l_enabled();
    a.self_fill = prefilled == 2 ? 1 : 0;
    if (prefilled != 1 && !a.self_fill) PK_CHECK_HIP(hipMemsetAsync(dGb, 0xFF, (size_t)ndir * T * B * g_pitch * 2, st));
    rc = pk_rec2_ln_setup(st, a, pl, ln, true);
    if (rc) return rc;
    if (lstm8 && !ln) return pk_rec2l_launch(st, a, pl, act, true);
    if (!ln && pk_rec3_covers(cell, 1)) return pk_rec3_launch(st, a, pl, cell, act, true, traced(cell, act));
    const size_t atile = (size_t)RMAX * pk_r2_lda_bf16(G * KPAD) * 2;
    const int nin = pk_cell_saved(cell) + 2 + (cell == PK_CELL_LSTM ? 1 : 0) + (ln ? 1 : 0);
    const size_t lds = (2 * atile > 96 * 1024 ? 1 : 2) * atile + 4 * ((size_t)(nin + G) * 1024 + (size_t)G * 512) + 16;
    const Rec2Kernel k = ln ? pick_bwd_ln(cell) : pick_bwd(cell, act);
    {
        static size_t granted[3][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};  // hipFuncSetAttribute is slow (milliseconds): once per kernel and size
        const int slot = cell == PK_CELL_LIGRU ? 0 : cell == PK_CELL_RNN ? 1 : 2;
        const int as = ln ? 4 : traced(cell, act) ? 3 : act_slot(act);
        if (granted[slot][as] < lds) {
            PK_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            granted[slot][as] = lds;
        }
    }
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        a.ln_cg0 = l * pl.C;
        rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        dim3 grid(pl.C * pl.Pn), block(256);
        rc = pk_rec2_check_residency((const void*)k, 256, lds, pl.C * pl.Pn, "pk_rec_bwd_bf16");
        if (rc) return rc;
        const int help = ln ? 0 : pk_rec_helper_wanted(true, pl.launches, cell);
        if (help && (rc = pk_rec_helper_fork(st)) != 0) return rc;
        hipLaunchKernelGGL(k, grid, block, lds, st, a);
        PK_LAUNCH_CHECK();
        if (help && (rc = pk_rec_helper_launch(st, a, pl, G, pk_cell_saved(cell), true, true, help)) != 0) return rc;
    }
    return pk_rec2_ln_finish(st, a, ln);
}
extern "C" int pk_rec_bwd_bf16(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U,
                               const float* mask, float mask_scalar, const float* Y, const float* S, const float* dY,
                               float* dP2, uint16_t* dGb, int64_t g_pitch, int prefilled) {
    return rec_bwd_bf16_impl(stream, cell, act, T, B, bidir, H, U, mask, mask_scalar, Y, S, dY, dP2, dGb, g_pitch, prefilled,
                             nullptr);
}
extern "C" int pk_rec_bwd_bf16_ln(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U,
                                  const float* mask, float mask_scalar, const float* ln_gamma, float ln_eps, const float* Y,
                                  const float* S, const float* LNS, const float* dY, float* dP2, uint16_t* dGb,
                                  int64_t g_pitch, int prefilled, float* lnwork, float* dln_gamma, float* dln_beta) {
    PK_REQUIRE(ln_gamma && LNS && lnwork && dln_gamma && dln_beta, "pk_rec_bwd_bf16_ln: null LayerNorm argument");
    const PkLnHost ln = {ln_gamma, nullptr, ln_eps, const_cast<float*>(LNS), lnwork, dln_gamma, dln_beta};
    return rec_bwd_bf16_impl(stream, cell, act, T, B, bidir, H, U, mask, mask_scalar, Y, S, dY, dP2, dGb, g_pitch, prefilled,
                             &ln);
}
