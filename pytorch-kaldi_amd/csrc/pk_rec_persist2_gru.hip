// pk_rec_persist2_gru.hip - perf-mode persistent time loop of the TWO-PHASE cells:
// GRU (neural_networks.py:629-641) and minimalGRU (:1291-1302), forward and BPTT, one launch each.
//
// Their candidate GEMM consumes a gate of the same step (a_t = Wh_t + (r_t*h_{t-1}).U_h^T for GRU,
// (z_t*h_{t-1}).U_h^T for minimalGRU), so a step is two dependent MFMA phases and two cluster-wide
// exchanges instead of one.  Everything else is the single-phase design of pk_rec_persist2.hip
// (U as MFMA B fragments in registers for all T steps, 16-byte bf16 exchange through L2 with the data
// as the flag, XCD-local fast path chosen by a placement handshake, 16-byte vector I/O through
// wave-private LDS patches, non-critical traffic issued behind the workgroup barrier):
//
//   forward, step t    1. poll h_{t-1} (Yb)          -> MFMA gates [z(,r)]  -> publish x_t = r*h (GRU) / z*h
//                      2. poll x_t     (Xb)          -> MFMA candidate      -> h_t -> publish (Yb)
//   backward, step t   1. poll [dz(,dr)]_{t+1} (dGb) -> MFMA carry          -> dh_t -> da_t -> publish gate slot G-1
//                      2. poll da_t    (dGb)         -> MFMA q = da.U_h     -> dz(,dr) -> publish gate slots 0..G-2
//
// Xb (bf16 r*h / z*h) and dGb are outputs as well: they are the k-major operands of the dU GEMMs.
#include "pk_rec2_common.h"

namespace {

// ============================================================================
// forward
// ============================================================================
// ACT: compile-time activation (relu / tanh) or -1 = run-time a.act, as in pk_rec_persist2.hip
// LN: per-step LayerNorm of h_t (neural_networks.py:638-639, :1299-1300) - a third exchange in the step, of the rows'
// partial sums (ln_row_allreduce, pk_rec2_common.h), between the blend and the publish of h_t.
template <int CELL, int ACT, bool LN = false>
__global__ __launch_bounds__(256, 1) void rec2g_fwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr bool TR = false;  // no phase trace in the two-phase kernels
    constexpr int G = pk_cell_gates(CELL), G1 = G - 1, NS = pk_cell_saved(CELL);
    constexpr bool GRU = (CELL == PK_CELL_GRU);
    constexpr int LDA = pk_r2_lda_bf16(KPAD);
    constexpr int ATILE = RMAX * LDA * 2;
    constexpr int NCH = (RMAX * (KPAD / 8) + 255) / 256;
    constexpr int NF = G + 1 + G + (LN ? 1 : 0);
    constexpr int WAVE_LDS = NF * 1024 + 512;  // P stage | Y | saved z(,r),a (| pre-LN h) | bf16 publish patch
    constexpr int LDS_TRASH = 2 * ATILE + 4 * WAVE_LDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // tile(phase 1) | tile(phase 2) | 4 x WAVE_LDS | trash

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

    // recurrent weights of my 16 units: B[k][n] = U_g[unit n][k]; gate G-1 is the candidate (U_h)
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
                    const float w = (k < H) ? __uint_as_float(raw[kk][e >> 2][e & 3]) : 0.f;
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

    // poll descriptors (shared by both phases: phase 1 of step t reads Yb at offset cbase + (t-1)*cstep,
    // phase 2 reads Xb at cbase + t*cstep)
    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Ypitch * 2u;
    const unsigned szYb = (unsigned)T * TS;
    unsigned cbase[NCH], cstep[NCH];
    int clds[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = tid + 256 * i;
        const bool ok = ci < nrows * CPR;
        const int row = ok ? ci / CPR : 0, col = ok ? ci - row * CPR : 0;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        cbase[i] = ok ? ((unsigned)b * a.Ypitch + dir * Hp + col * 8) * 2u + (unsigned)(dir ? (T - 1) : 0) * TS : szYb;
        cstep[i] = ok ? (dir ? 0u - TS : TS) : 0u;
        clds[i] = ok ? row * (LDA * 2) + col * 16 : LDS_TRASH;
    }
    float rvf[4], msk[4], hprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r, n = n_base + row;
        const bool ok = row < nrows && unit_ok;
        rvf[r] = ok ? 1.f : 0.f;
        msk[r] = (a.mask != nullptr && ok) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        hprev[r] = 0.f;
    }
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
    const int prow = lane >> 1, phalf = lane & 1;
    const int pu0 = ubase + phalf * 8;
    const bool pk_ok = lane < 32 && prow < nrows && pu0 < Hp;
    const int pn = n_base + (prow < nrows ? prow : 0);
    const int pdir = pn >= B ? 1 : 0, pb = pn - pdir * B;
    const unsigned pbase = pk_ok ? ((unsigned)pb * a.Ypitch + pdir * Hp + pu0) * 2u : szYb;

    unsigned char* wl = smem + 2 * ATILE + wave * WAVE_LDS;
    float* patchP = reinterpret_cast<float*>(wl);                  // [G][256]
    float* patchY = reinterpret_cast<float*>(wl + G * 1024);       // [256]
    float* patchS = reinterpret_cast<float*>(wl + (G + 1) * 1024); // [G][256]: z(,r), a
    float* patchL = reinterpret_cast<float*>(wl + (2 * G + 1) * 1024);  // (LN only)
    unsigned short* patchB = reinterpret_cast<unsigned short*>(wl + NF * 1024);
    const __amdgpu_buffer_rsrc_t rsY = make_rsrc(a.Yb, szYb);
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(a.Xb, szYb);
    float* trash = a.trash + (tid & 63) * 4;
    // ---- per-step LayerNorm state (as in pk_rec_persist2.hip)
    const LnSlots ls = LN ? ln_slots(a, c, p, wave, lane, T) : LnSlots();
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(LN ? (const void*)a.lnx : (const void*)a.Yb, LN ? ls.size : 0u);
    const float gam = (LN && unit_ok) ? a.ln_gamma[unit] : 0.f, bet = (LN && unit_ok) ? a.ln_beta[unit] : 0.f;
    const float invH = 1.0f / (float)H, inv_nm1 = 1.0f / (float)(H - 1);
    float piv[4] = {0.f, 0.f, 0.f, 0.f};
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
    // layer output and the saved gates z(,r),a (slots 0..G-1 of S; the r*h / z*h slot is only kept as bf16 in Xb)
    auto flush_outputs = [&](int tt, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
        // (a.S == null: a validation / forward chunk - nothing is saved; a.Y == null with it: an inner layer, Yb is its output)
        if (a.Y != nullptr) st4<EE>(a.Y, vY0 + ts * vYs, vnv, trash, patch_get_vec(patchY, lane));
        if (a.S != nullptr) {
#pragma unroll
            for (int k = 0; k < G; ++k) st4<EE>(a.S, vS0 + ts * vSs + k * H, vnv, trash, patch_get_vec(patchS + k * 256, lane));
        }
        if (LN) {
            st4<EE>(a.lnh, vY0 + ts * vYs, vnv, trash, patch_get_vec(patchL, lane));
            *reinterpret_cast<pk_f32x2*>(st_base + (long)tt * st_step) = st_val;
        }
    };
#define PKG_LP0(E) load_proj(0, E)
    PK_EDGE_DISPATCH(PKG_LP0);
    // self-filling exchange (pk_rec2_common.h), both mailboxes: h in Yb, r*h (z*h) in Xb
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (a.self_fill) {
        for (int tt = 0; tt < PK_R2_FILL_AHEAD && tt < T; ++tt) {
            const unsigned off = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - tt) : tt) * TS : 0u);
            pub_store<false>(rsY, off, sentinel);
            pub_store<false>(rsX, off, sentinel);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    unsigned char* A0 = smem;
    unsigned char* A1 = smem + ATILE;
    // the time loop, instantiated per (XCD-local fast path?, static edge case?) as in pk_rec_persist2.hip (forward)
    auto run = [&](auto FASTC, auto SEC) {
    constexpr bool fast = decltype(FASTC)::value != 0;
    constexpr int SE = decltype(SEC)::value;
    for (int t = 0; t < T; ++t) {
        const int step_idx = t;
        PK_TRACE(0);
        unsigned goff[NCH];
        int retries = 0;
        // ---------------- phase 1: gates that see h_{t-1} only
        f32x4 acc[G1];
#pragma unroll
        for (int g = 0; g < G1; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) goff[i] = cbase[i] + (unsigned)(t - 1) * cstep[i];
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);  // see pk_rec2_host_setup
            dead = fast ? poll_to_lds<NCH, true>(rsY, goff, clds, A0, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds<NCH, false>(rsY, goff, clds, A0, a.err, a.spin_limit, lane, dead, retries);
        }
        PK_TRACE(1);
#pragma unroll
        for (int g = 0; g < G; ++g) patch_put_vec(patchP + g * 256, lane, pv[g]);
        if (t > 0) PK_BARRIER_LDS();
        else PK_LDS_ORDER();
        if (t > 0) {
#define PKG_FO(E) flush_outputs(t - 1, E)
            PK_EDGE_DISPATCH_S(PKG_FO);
        }
        if (t + 1 < T) {
#define PKG_LP1(E) load_proj(t + 1, E)
            PK_EDGE_DISPATCH_S(PKG_LP1);
        }
        if (a.self_fill && t + PK_R2_FILL_AHEAD < T) {
            const unsigned off = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - (t + PK_R2_FILL_AHEAD)) : (t + PK_R2_FILL_AHEAD)) * TS : 0u);
            pub_store<fast>(rsY, off, sentinel);
            pub_store<fast>(rsX, off, sentinel);
        }
        if (t > 0) {
            const unsigned char* Ar = A0 + (lane & 15) * (LDA * 2) + kq * 16;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ar + kk * 64);
#pragma unroll
                for (int g = 0; g < G1; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[g][kk], acc[g], 0, 0, 0);
            }
        }
        PK_TRACE(2);
        float pre[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) patch_get_cd(patchP + g * 256, kq, lane, pre[g]);
        float zt[4], rt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            zt[r] = pk_sig(pre[0][r] * psc[0] + psh[0] + acc[0][r]);
            rt[r] = GRU ? pk_sig(pre[1][r] * psc[1] + psh[1] + acc[G1 - 1][r]) : zt[r];
            const float x = rvf[r] != 0.f ? rt[r] * hprev[r] : 0.f;  // r*h (GRU) or z*h (minimalGRU): what U_h sees
            patchB[(kq * 4 + r) * 16 + (lane & 15)] = to_bf_pub(x);
        }
        PK_LDS_ORDER();
        {
            const u32x4 o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(patchB) + (prow & 15) * 32 + phalf * 16);
            const unsigned off = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - t) : t) * TS : 0u);
            if (fast) pub_store<true>(rsX, off, o);
            else pub_store<false>(rsX, off, o);
        }
        PK_TRACE(3);
        // ---------------- phase 2: candidate GEMM on x_t, blend
        f32x4 acca = f32x4{0.f, 0.f, 0.f, 0.f}, accb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {  // (x_0 = 0: nothing to multiply at the first step)
#pragma unroll
            for (int i = 0; i < NCH; ++i) goff[i] = cbase[i] + (unsigned)t * cstep[i];
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);  // see pk_rec2_host_setup
            dead = fast ? poll_to_lds<NCH, true>(rsX, goff, clds, A1, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds<NCH, false>(rsX, goff, clds, A1, a.err, a.spin_limit, lane, dead, retries);
            PK_BARRIER_LDS();
            const unsigned char* Ar = A1 + (lane & 15) * (LDA * 2) + kq * 16;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ar + kk * 64);
                if ((kk & 1) == 0) acca = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[G1][kk], acca, 0, 0, 0);
                else accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[G1][kk], accb, 0, 0, 0);
            }
        }
        PK_TRACE(4);
        float hv[4], av[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float at = pre[G1][r] * psc[G1] + psh[G1] + acca[r] + accb[r];
            float h = pk_cell_fwd_p2<CELL>(act, at, zt[r], hprev[r], msk[r]);
            h = rvf[r] != 0.f ? h : 0.f;
            hprev[r] = h;
            hv[r] = h;
            av[r] = at;
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
                dead = ln_row_allreduce<fast>(rsx, ls.pub_at(T), po0, ma, mb, a.err, a.spin_limit, lane, dead);
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
            dead = ln_row_allreduce<fast>(rsx, ls.pub_at(t), po, la, lb, a.err, a.spin_limit, lane, dead);
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
        PK_LDS_ORDER();
        {
            const u32x4 o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(patchB) + (prow & 15) * 32 + phalf * 16);
            const unsigned off = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - t) : t) * TS : 0u);
            if (fast) pub_store<true>(rsY, off, o);
            else pub_store<false>(rsY, off, o);
        }
        patch_put_cd(patchY, kq, lane, hv);
        patch_put_cd(patchS, kq, lane, zt);
        if (GRU) patch_put_cd(patchS + 256, kq, lane, rt);
        patch_put_cd(patchS + G1 * 256, kq, lane, av);
        PK_LDS_ORDER();
        PK_TRACE(5);
    }
    };
    PK_RUN_SPECIALISED(run, fast_rt);
#define PKG_FOL(E) flush_outputs(T - 1, E)
    PK_EDGE_DISPATCH(PKG_FOL);
}

// ============================================================================
// backward
// ============================================================================
template <int CELL, int ACT, bool LN = false>
__global__ __launch_bounds__(256, 1) void rec2g_bwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr bool TR = false;  // no phase trace in the two-phase kernels
    constexpr int G = pk_cell_gates(CELL), G1 = G - 1, NS = pk_cell_saved(CELL);
    constexpr bool GRU = (CELL == PK_CELL_GRU);
    constexpr int LDB = pk_r2_lda_bf16(G1 * KPAD);  // tile of [dz(,dr)]_{t+1}
    constexpr int LDA = pk_r2_lda_bf16(KPAD);       // tile of da_t
    constexpr int BTILE = RMAX * LDB * 2, ATILE = RMAX * LDA * 2;
    constexpr int NCHB = (RMAX * G1 * (KPAD / 8) + 255) / 256;
    constexpr int NCHA = (RMAX * (KPAD / 8) + 255) / 256;
    constexpr int NIN = G + 2 + (LN ? 1 : 0);    // saved z(,r),a | h_{t-1} | dY (| pre-LN h_t)
    constexpr int WAVE_LDS = NIN * 1024 + G * 512;
    constexpr int LDS_TRASH = BTILE + ATILE + 4 * WAVE_LDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T;
    const unsigned TB = (unsigned)T * B;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int ubase = p * 64 + wave * 16;
    const int unit = ubase + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;

    // B[kidx = (g, j)][n = unit] = U_g[j][unit]; gate G-1 = U_h (the q = da.U_h product of phase 2)
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

    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Gpitch * 2u;
    const unsigned ndir = (unsigned)(a.R / B);
    const unsigned szGb = ndir * (unsigned)T * TS;
    // phase 1 (iteration it >= 1, t = T-1-it) reads gates 0..G-2 of step t+1: storage time (dir ? it-1 : T-it)
    unsigned cbB[NCHB], csB[NCHB];
    int clB[NCHB];
#pragma unroll
    for (int i = 0; i < NCHB; ++i) {
        const int ci = tid + 256 * i;
        const bool ok = ci < nrows * G1 * CPR;
        const int row = ok ? ci / (G1 * CPR) : 0;
        const int rem = ok ? ci - row * (G1 * CPR) : 0;
        const int g = rem / CPR, col = rem - g * CPR;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        cbB[i] = ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)b * a.Gpitch + g * Hp + col * 8) * 2u +
                          (unsigned)(dir ? 0 : (T - 1)) * TS
                    : szGb;
        csB[i] = ok ? (dir ? TS : 0u - TS) : 0u;
        clB[i] = ok ? row * (LDB * 2) + (g * KPAD + col * 8) * 2 : LDS_TRASH;
    }
    // phase 2 (iteration it, t = T-1-it) reads gate G-1 of step t: storage time (dir ? it : T-1-it)
    unsigned cbA[NCHA], csA[NCHA];
    int clA[NCHA];
#pragma unroll
    for (int i = 0; i < NCHA; ++i) {
        const int ci = tid + 256 * i;
        const bool ok = ci < nrows * CPR;
        const int row = ok ? ci / CPR : 0, col = ok ? ci - row * CPR : 0;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        cbA[i] = ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)b * a.Gpitch + G1 * Hp + col * 8) * 2u +
                          (unsigned)(dir ? 0 : (T - 1)) * TS
                    : szGb;
        csA[i] = ok ? (dir ? TS : 0u - TS) : 0u;
        clA[i] = ok ? BTILE + row * (LDA * 2) + col * 16 : LDS_TRASH;
    }
    float rvf[4], msk[4], dh_dir[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r, n = n_base + row;
        const bool ok = row < nrows && unit_ok;
        rvf[r] = ok ? 1.f : 0.f;
        msk[r] = (a.mask != nullptr && ok) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        dh_dir[r] = 0.f;
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
    const int prow = lane >> 1, phalf = lane & 1;
    const int pu0 = ubase + phalf * 8;
    const bool pk_ok = lane < 32 && prow < nrows && pu0 < Hp;
    const int pn = n_base + (prow < nrows ? prow : 0);
    const int pdir = pn >= B ? 1 : 0, pb = pn - pdir * B;
    const unsigned pbase = pk_ok ? (unsigned)pdir * (unsigned)T * TS + ((unsigned)pb * a.Gpitch + pu0) * 2u : szGb;

    unsigned char* wl = smem + BTILE + ATILE + wave * WAVE_LDS;
    float* patchI = reinterpret_cast<float*>(wl);  // [NIN][256]: z(,r), a, h_{t-1}, dY
    unsigned short* patchB = reinterpret_cast<unsigned short*>(wl + NIN * 1024);  // [G][16][16] bf16
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dGb, szGb);

    f32x4 iv[NIN];
    auto load_step_e = [&](int t, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(vdir ? (T - 1 - t) : t);
        const unsigned tp = t > 0 ? (vdir ? ts + 1 : ts - 1) : ts;
#pragma unroll
        for (int k = 0; k < G; ++k) iv[k] = ld4<EE>(a.S, vS0 + ts * vSs + k * H, vnv);
        iv[G] = ld4<EE>(a.Y, vY0 + tp * vYs, t > 0 ? vnv : 0);
        iv[G + 1] = ld4<EE>(a.dY, vY0 + ts * vYs, vnv);
        if (LN) iv[G + 2] = ld4<EE>(a.lnh, vY0 + ts * vYs, vnv);
        if (t == 0) iv[G] = f32x4{0.f, 0.f, 0.f, 0.f};  // h_{-1} = 0
    };
    auto load_step = [&](int t) {
#define PKG_LS(E) load_step_e(t, E)
        PK_EDGE_DISPATCH(PKG_LS);
    };
    load_step(T - 1);
    // ---- per-step LayerNorm state (as in pk_rec_persist2.hip)
    const LnSlots ls = LN ? ln_slots(a, c, p, wave, lane, T) : LnSlots();
    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(LN ? (const void*)a.lnx : (const void*)a.dGb, LN ? ls.size : 0u);
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
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto fill_slab = [&](int tt, auto FASTC) {  // my G chunks of the slab that step tt will publish
        const unsigned off = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - tt) : tt) * TS : 0u);
#pragma unroll
        for (int g = 0; g < G; ++g) pub_store<decltype(FASTC)::value != 0>(rs, off + (pk_ok ? (unsigned)(g * Hp) * 2u : 0u), sentinel);
    };
    if (a.self_fill) {
        for (int k = 0; k < PK_R2_FILL_AHEAD && k < T; ++k) fill_slab(T - 1 - k, BoolC<0>());
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    const bool fast = cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0;
    int it = 0;
    for (int t = T - 1; t >= 0; --t, ++it) {
        const int step_idx = it;
        PK_TRACE(0);
        int retries = 0;
        // ---------------- phase 1: carry GEMM over [dz(,dr)]_{t+1}, then da_t
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < T - 1) {
            unsigned goff[NCHB];
#pragma unroll
            for (int i = 0; i < NCHB; ++i) goff[i] = cbB[i] + (unsigned)(it - 1) * csB[i];
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);  // see pk_rec2_host_setup
            dead = fast ? poll_to_lds<NCHB, true>(rs, goff, clB, smem, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds<NCHB, false>(rs, goff, clB, smem, a.err, a.spin_limit, lane, dead, retries);
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
        if (t > 0) load_step(t - 1);
        if (LN && t > 0) load_stats(t - 1);
        if (a.self_fill && t - PK_R2_FILL_AHEAD >= 0) {
            if (fast) fill_slab(t - PK_R2_FILL_AHEAD, BoolC<1>());
            else fill_slab(t - PK_R2_FILL_AHEAD, BoolC<0>());
        }
        if (t < T - 1) {
            const unsigned char* Ar = smem + (lane & 15) * (LDB * 2) + kq * 16;
#pragma unroll
            for (int g = 0; g < G1; ++g)
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ar + (g * KPAD + kk * 32) * 2);
                    if ((kk & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[g][kk], acc0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[g][kk], acc1, 0, 0, 0);
                }
        }
        PK_TRACE(2);
        float sin[NIN][4];
#pragma unroll
        for (int k = 0; k < NIN; ++k) patch_get_cd(patchI + k * 256, kq, lane, sin[k]);
        float da[4], dzp[4], dhd[4];
        float dh4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dh4[r] = sin[G + 1][r] + dh_dir[r] + acc0[r] + acc1[r];
        if (LN) {  // through the LayerNorm first (see pk_rec_persist2.hip)
            float la[4], lb[4], dd[4], gg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = rvf[r] != 0.f;
                dd[r] = sin[G + 2][r] - mu4[r];
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
            const float z = sin[0][r], at = sin[G1][r], hp = sin[G][r];
            const float dh = dh4[r];
            const float cand = pk_act(act, at) * msk[r];
            const bool ok = rvf[r] != 0.f;
            dzp[r] = ok ? dh * (hp - cand) * z * (1.f - z) : 0.f;
            dhd[r] = ok ? dh * z : 0.f;
            da[r] = ok ? dh * (1.f - z) * msk[r] * pk_act_grad_from_in(act, at) : 0.f;
            patchB[G1 * 256 + (kq * 4 + r) * 16 + (lane & 15)] = to_bf_pub(da[r]);
        }
        PK_LDS_ORDER();
        const unsigned poff = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - t) : t) * TS : 0u);
        {
            const u32x4 o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(patchB) + G1 * 512 + (prow & 15) * 32 + phalf * 16);
            const unsigned og = poff + (pk_ok ? (unsigned)(G1 * Hp) * 2u : 0u);
            if (fast) pub_store<true>(rs, og, o);
            else pub_store<false>(rs, og, o);
        }
        PK_TRACE(3);
        // ---------------- phase 2: q = da_t . U_h, then dz(,dr) and the direct carry
        f32x4 qa = f32x4{0.f, 0.f, 0.f, 0.f}, qb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {  // (at t = 0 q only multiplies h_{-1} = 0 and feeds a carry nobody reads)
            unsigned goff[NCHA];
#pragma unroll
            for (int i = 0; i < NCHA; ++i) goff[i] = cbA[i] + (unsigned)it * csA[i];
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);  // see pk_rec2_host_setup
            dead = fast ? poll_to_lds<NCHA, true>(rs, goff, clA, smem, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds<NCHA, false>(rs, goff, clA, smem, a.err, a.spin_limit, lane, dead, retries);
            PK_BARRIER_LDS();
            const unsigned char* Ar = smem + BTILE + (lane & 15) * (LDA * 2) + kq * 16;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ar + kk * 64);
                if ((kk & 1) == 0) qa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[G1][kk], qa, 0, 0, 0);
                else qb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[G1][kk], qb, 0, 0, 0);
            }
        }
        PK_TRACE(4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = rvf[r] != 0.f;
            const float q = ok ? qa[r] + qb[r] : 0.f;
            const float z = sin[0][r], hp = sin[G][r];
            float dz, dr = 0.f;
            if (GRU) {
                const float rg = sin[1][r];
                dz = dzp[r];
                dr = q * hp * rg * (1.f - rg);
                dh_dir[r] = dhd[r] + q * rg;
                patchB[256 + (kq * 4 + r) * 16 + (lane & 15)] = to_bf_pub(dr);
            } else {
                dz = dzp[r] + q * hp * z * (1.f - z);
                dh_dir[r] = dhd[r] + q * z;
            }
            patchB[(kq * 4 + r) * 16 + (lane & 15)] = to_bf_pub(dz);
        }
        PK_LDS_ORDER();
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            const u32x4 o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(patchB) + g * 512 + (prow & 15) * 32 + phalf * 16);
            const unsigned og = poff + (pk_ok ? (unsigned)(g * Hp) * 2u : 0u);
            if (fast) pub_store<true>(rs, og, o);
            else pub_store<false>(rs, og, o);
        }
        PK_TRACE(5);
    }
    if (LN) {  // my unit's share of d gamma / d beta over this cluster's rows and all steps
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

// ============================================================================
// backward, third generation (round 3): the same two-phase step with the MFMA operands swapped (pk_rec_persist3.hip has
// the reasoning).  A lane holds four consecutive units of one row: the saved tensors (z(, r), a, h_{t-1}, dY) are loaded
// with one 16-byte access per lane straight into the layout the gate math works in, the bf16 chunks of da and dz(, dr)
// are assembled with v_permlane16_swap_b32 - no wave-private LDS patches, no LDS drain in front of either publish.
// ============================================================================
__device__ __forceinline__ u32x4 pack_chunk_g(unsigned lo, unsigned hi) {
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return u32x4{a[0], b[0], a[1], b[1]};
}
__device__ __forceinline__ unsigned pack2_g(float x, float y) { return (unsigned)to_bf_pub(x) | ((unsigned)to_bf_pub(y) << 16); }

template <int CELL, int ACT>
__global__ __launch_bounds__(256, 1) void rec3g_bwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), G1 = G - 1, NS = pk_cell_saved(CELL);
    constexpr bool GRU = (CELL == PK_CELL_GRU);
    constexpr int LDB = pk_r2_lda_bf16(G1 * KPAD);  // tile of [dz(,dr)]_{t+1}
    constexpr int LDA = pk_r2_lda_bf16(KPAD);       // tile of da_t
    constexpr int BTILE = RMAX * LDB * 2, ATILE = RMAX * LDA * 2;
    constexpr int NCHB = (RMAX * G1 * (KPAD / 8) + 255) / 256;
    constexpr int NCHA = (RMAX * (KPAD / 8) + 255) / 256;
    constexpr int NIN = G + 2;                   // saved z(,r),a | h_{t-1} | dY
    constexpr int LDS_TRASH = BTILE + ATILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T;
    const unsigned TB = (unsigned)T * B;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int ubase = p * 64 + wave * 16;
    const int kq = lane >> 4;
    const int frag_unit = ubase + (lane & 15);
    const bool frag_ok = frag_unit < H;

    // A[m = unit][kidx = (g, j)] = U_g[j][unit]; gate G-1 = U_h (the q = da.U_h product of phase 2)
    bf16x8 Uf[G][KSTEPS];
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
                    raw[e] = __builtin_amdgcn_raw_buffer_load_b32(rsU, (frag_ok && j < H) ? (unsigned)(((g * H + j) * H + frag_unit) * 4) : szU, 0, 0);
                }
                bf16x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (short)pk_f2bf(__uint_as_float(raw[e]));
                Uf[g][kk] = f;
            }
    }
    for (int i = tid; i < (LDS_TRASH + 16) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Gpitch * 2u;
    const unsigned ndir = (unsigned)(a.R / B);
    const unsigned szGb = ndir * (unsigned)T * TS;
    // phase 1 (iteration it >= 1, t = T-1-it) reads gates 0..G-2 of step t+1: storage time (dir ? it-1 : T-it)
    unsigned cbB[NCHB], csB[NCHB];
    int clB[NCHB];
#pragma unroll
    for (int i = 0; i < NCHB; ++i) {
        const int ci = tid + 256 * i;
        const bool ok = ci < nrows * G1 * CPR;
        const int row = ok ? ci / (G1 * CPR) : 0;
        const int rem = ok ? ci - row * (G1 * CPR) : 0;
        const int g = rem / CPR, col = rem - g * CPR;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        cbB[i] = ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)b * a.Gpitch + g * Hp + col * 8) * 2u +
                          (unsigned)(dir ? 0 : (T - 1)) * TS
                    : szGb;
        csB[i] = ok ? (dir ? TS : 0u - TS) : 0u;
        clB[i] = ok ? row * (LDB * 2) + (g * KPAD + col * 8) * 2 : LDS_TRASH;
    }
    // phase 2 (iteration it, t = T-1-it) reads gate G-1 of step t: storage time (dir ? it : T-1-it)
    unsigned cbA[NCHA], csA[NCHA];
    int clA[NCHA];
#pragma unroll
    for (int i = 0; i < NCHA; ++i) {
        const int ci = tid + 256 * i;
        const bool ok = ci < nrows * CPR;
        const int row = ok ? ci / CPR : 0, col = ok ? ci - row * CPR : 0;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        cbA[i] = ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)b * a.Gpitch + G1 * Hp + col * 8) * 2u +
                          (unsigned)(dir ? 0 : (T - 1)) * TS
                    : szGb;
        csA[i] = ok ? (dir ? TS : 0u - TS) : 0u;
        clA[i] = ok ? BTILE + row * (LDA * 2) + col * 16 : LDS_TRASH;
    }
    // ---- my (row, 4 units): row = lane & 15, units u0 .. u0 + 3
    const int row = lane & 15, u0 = ubase + kq * 4;
    const bool row_ok = row < nrows;
    const int n = n_base + (row_ok ? row : 0);
    const int dir = n >= B ? 1 : 0, bb = n - dir * B;
    int nv = H - u0;
    nv = nv > 4 ? 4 : (nv < 0 ? 0 : nv);
    const int edge = __any(nv > 0 && nv < 4) != 0 ? ((H & 1) ? 2 : 1) : 0;
    nv = row_ok ? nv : 0;
    float msk[4], dh_dir[4];
    bool ok4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ok4[r] = r < nv;
        msk[r] = (a.mask != nullptr && ok4[r]) ? a.mask[(long)n * H + u0 + r] : a.mask_scalar;
        dh_dir[r] = 0.f;
    }
    const unsigned vY0 = ((unsigned)bb * a.YH + dir * H + u0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)dir * TB + bb) * (NS * H) + u0), vSs = (unsigned)B * NS * H;
    const int pu0 = ubase + (kq >> 1) * 8;
    const bool pk_ok = (kq & 1) == 0 && row_ok && pu0 < Hp;
    const unsigned pbase = pk_ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)bb * a.Gpitch + pu0) * 2u : szGb;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dGb, szGb);

    f32x4 iv[NIN], inext[NIN];  // this step / the next one (loaded a step ahead)
    auto load_step_e = [&](f32x4 (&dst)[NIN], int t, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(dir ? (T - 1 - t) : t);
        const unsigned tp = t > 0 ? (dir ? ts + 1 : ts - 1) : ts;
#pragma unroll
        for (int k = 0; k < G; ++k) dst[k] = ld4<EE>(a.S, vS0 + ts * vSs + k * H, nv);
        dst[G] = ld4<EE>(a.Y, vY0 + tp * vYs, t > 0 ? nv : 0);
        dst[G + 1] = ld4<EE>(a.dY, vY0 + ts * vYs, nv);
        if (t == 0) dst[G] = f32x4{0.f, 0.f, 0.f, 0.f};  // h_{-1} = 0
    };
#define PKG3_LS0(E) load_step_e(iv, T - 1, E)
    PK_EDGE_DISPATCH(PKG3_LS0);
#pragma unroll
    for (int k = 0; k < NIN; ++k) inext[k] = iv[k];
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto fill_slab = [&](int tt, auto FASTC) {  // my G chunks of the slab that step tt will publish
        const unsigned off = pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - tt) : tt) * TS : 0u);
#pragma unroll
        for (int g = 0; g < G; ++g) pub_store<decltype(FASTC)::value != 0>(rs, off + (pk_ok ? (unsigned)(g * Hp) * 2u : 0u), sentinel);
    };
    if (a.self_fill) {
        for (int k = 0; k < PK_R2_FILL_AHEAD && k < T; ++k) fill_slab(T - 1 - k, BoolC<0>());
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    const bool fast = cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing in flight when the time loop is entered (pk_rec_persist3.hip)
    int it = 0;
    for (int t = T - 1; t >= 0; --t, ++it) {
        int retries = 0;
        // ---------------- phase 1: carry GEMM over [dz(,dr)]_{t+1}, then da_t
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < T - 1) {
            unsigned goff[NCHB];
#pragma unroll
            for (int i = 0; i < NCHB; ++i) goff[i] = cbB[i] + (unsigned)(it - 1) * csB[i];
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);  // see pk_rec2_host_setup
            dead = fast ? poll_to_lds<NCHB, true>(rs, goff, clB, smem, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds<NCHB, false>(rs, goff, clB, smem, a.err, a.spin_limit, lane, dead, retries);
            PK_BARRIER_LDS();
        }
        // off the dependency chain: the saved tensors of the next step (loads first), the fill pattern ahead
        if (t > 0) {
#define PKG3_LS1(E) load_step_e(inext, t - 1, E)
            PK_EDGE_DISPATCH(PKG3_LS1);
        }
        if (a.self_fill && t - PK_R2_FILL_AHEAD >= 0) {
            if (fast) fill_slab(t - PK_R2_FILL_AHEAD, BoolC<1>());
            else fill_slab(t - PK_R2_FILL_AHEAD, BoolC<0>());
        }
        if (t < T - 1) {
            const unsigned char* Ar = smem + (lane & 15) * (LDB * 2) + kq * 16;
#pragma unroll
            for (int g = 0; g < G1; ++g)
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const bf16x8 df = *reinterpret_cast<const bf16x8*>(Ar + (g * KPAD + kk * 32) * 2);
                    if ((kk & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Uf[g][kk], df, acc0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Uf[g][kk], df, acc1, 0, 0, 0);
                }
        }
        float da[4], dzp[4], dhd[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = iv[0][r], at = iv[G1][r], hp = iv[G][r];
            const float dh = iv[G + 1][r] + dh_dir[r] + acc0[r] + acc1[r];
            const float cand = pk_act(act, at) * msk[r];
            dzp[r] = ok4[r] ? dh * (hp - cand) * z * (1.f - z) : 0.f;
            dhd[r] = ok4[r] ? dh * z : 0.f;
            da[r] = ok4[r] ? dh * (1.f - z) * msk[r] * pk_act_grad_from_in(act, at) : 0.f;
        }
        const unsigned poff = pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - t) : t) * TS : 0u);
        {
            const u32x4 o = pack_chunk_g(pack2_g(da[0], da[1]), pack2_g(da[2], da[3]));
            const unsigned og = poff + (pk_ok ? (unsigned)(G1 * Hp) * 2u : 0u);
            if (fast) pub_store<true>(rs, og, o);
            else pub_store<false>(rs, og, o);
        }
        // ---------------- phase 2: q = da_t . U_h, then dz(,dr) and the direct carry
        f32x4 qa = f32x4{0.f, 0.f, 0.f, 0.f}, qb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t > 0) {  // (at t = 0 q only multiplies h_{-1} = 0 and feeds a carry nobody reads)
            unsigned goff[NCHA];
#pragma unroll
            for (int i = 0; i < NCHA; ++i) goff[i] = cbA[i] + (unsigned)it * csA[i];
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);  // see pk_rec2_host_setup
            dead = fast ? poll_to_lds<NCHA, true>(rs, goff, clA, smem, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds<NCHA, false>(rs, goff, clA, smem, a.err, a.spin_limit, lane, dead, retries);
            PK_BARRIER_LDS();
            const unsigned char* Ar = smem + BTILE + (lane & 15) * (LDA * 2) + kq * 16;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const bf16x8 df = *reinterpret_cast<const bf16x8*>(Ar + kk * 64);
                if ((kk & 1) == 0) qa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Uf[G1][kk], df, qa, 0, 0, 0);
                else qb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Uf[G1][kk], df, qb, 0, 0, 0);
            }
        }
        float dzv[4], drv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float q = ok4[r] ? qa[r] + qb[r] : 0.f;
            const float z = iv[0][r], hp = iv[G][r];
            drv[r] = 0.f;
            if (GRU) {
                const float rg = iv[1][r];
                dzv[r] = dzp[r];
                drv[r] = q * hp * rg * (1.f - rg);
                dh_dir[r] = dhd[r] + q * rg;
            } else {
                dzv[r] = dzp[r] + q * hp * z * (1.f - z);
                dh_dir[r] = dhd[r] + q * z;
            }
        }
        {
            const u32x4 o = pack_chunk_g(pack2_g(dzv[0], dzv[1]), pack2_g(dzv[2], dzv[3]));
            if (fast) pub_store<true>(rs, poff, o);
            else pub_store<false>(rs, poff, o);
        }
        if (GRU) {
            const u32x4 o = pack_chunk_g(pack2_g(drv[0], drv[1]), pack2_g(drv[2], drv[3]));
            const unsigned og = poff + (pk_ok ? (unsigned)Hp * 2u : 0u);
            if (fast) pub_store<true>(rs, og, o);
            else pub_store<false>(rs, og, o);
        }
#pragma unroll
        for (int k = 0; k < NIN; ++k) iv[k] = inext[k];
    }
}

size_t granted_lds[2][2][3] = {{{0, 0, 0}, {0, 0, 0}}, {{0, 0, 0}, {0, 0, 0}}};
typedef void (*Rec2gKernel)(R2Args);
inline int act_slot(int act) { return act == PK_ACT_RELU ? 0 : act == PK_ACT_TANH ? 1 : 2; }
template <int CELL>
Rec2gKernel pick_fwd(int act) {
    return act == PK_ACT_RELU ? rec2g_fwd_kernel<CELL, PK_ACT_RELU> : act == PK_ACT_TANH ? rec2g_fwd_kernel<CELL, PK_ACT_TANH>
                                                                                        : rec2g_fwd_kernel<CELL, -1>;
}
template <int CELL>
Rec2gKernel pick_bwd(int act) {
    return act == PK_ACT_RELU ? rec2g_bwd_kernel<CELL, PK_ACT_RELU> : act == PK_ACT_TANH ? rec2g_bwd_kernel<CELL, PK_ACT_TANH>
                                                                                        : rec2g_bwd_kernel<CELL, -1>;
}

template <int CELL>
Rec2gKernel pick_bwd3(int act) {
    return act == PK_ACT_RELU ? rec3g_bwd_kernel<CELL, PK_ACT_RELU> : act == PK_ACT_TANH ? rec3g_bwd_kernel<CELL, PK_ACT_TANH>
                                                                                        : rec3g_bwd_kernel<CELL, -1>;
}
// PK_EXPERIMENT gru_bwd_gen=3 selects the third-generation backward kernel.  NOT the default: on the two-phase step it measured
// slower (libri_gru 35.7 vs 34.7 ms per step, two rounds on one box) - the loads and fill stores issued behind barrier 1
// sit in front of the vmcnt(0) of the phase-2 poll half a step later, and without the patch traffic in between the fill
// stores' HBM acknowledge is still outstanding there.  (The one-hop kernels gain: liGRU -12 %, LSTM -5 % of the step.)
inline bool bwd_gen3() {
    static int g = -1;
    if (g < 0) {
        const char* e = pk_experiment("gru_bwd_gen");
        g = (e && e[0] == '3') ? 3 : 2;
    }
    return g == 3;
}
size_t granted_lds3[2][3] = {{0, 0, 0}, {0, 0, 0}};
size_t granted_ln[2][2] = {{0, 0}, {0, 0}};  // [fwd / bwd][GRU / minimalGRU]: the LayerNorm variants (run-time activation only)

int rec2p_fwd_impl(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                   const float* pscale, const float* pshift, const float* U, const float* mask,
                   float mask_scalar, float* Y, float* S, uint16_t* Yb, uint16_t* Xb, int64_t y_pitch, int prefilled,
                   const PkLnHost* ln) {
    int rc = pk_rec2_check("pk_rec2p_fwd_bf16", cell == PK_CELL_GRU || cell == PK_CELL_MINGRU, cell, T, B, bidir, H);
    if (rc) return rc;
    hipStream_t st = pk_stream(stream);
    const int ndir = 1 + bidir, R = B * ndir, Hp = (H + 7) & ~7, G = pk_cell_gates(cell);
    PK_REQUIRE(y_pitch >= (int64_t)ndir * Hp && (y_pitch % 8) == 0 && ((uintptr_t)Yb & 15) == 0 && ((uintptr_t)Xb & 15) == 0,
               "pk_rec2p_fwd_bf16: Yb / Xb pitch must be a multiple of 8 and hold %d x %d elements", ndir, Hp);
    PK_REQUIRE((double)T * B * y_pitch * 2.0 < 4.0e9, "pk_rec2p_fwd_bf16: exchange buffer exceeds the 4 GB buffer-descriptor range");
    Plan2 pl;
    rc = pk_rec2_make_plan(R, H, pl);
    if (rc) return rc;
    R2Args a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.Hp = Hp; a.YH = ndir * H; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.rpc = pl.rpc; a.row0 = 0;
    a.P = P; a.pscale = pscale; a.pshift = pshift; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = Y; a.S = S; a.Yb = (unsigned short*)Yb; a.Xb = (unsigned short*)Xb; a.Ypitch = (int)y_pitch;
    a.dY = nullptr; a.dP2 = nullptr; a.dGb = nullptr; a.Gpitch = 0;
    rc = pk_rec2_host_setup(a, false, cell);
    if (rc) return rc;
    a.self_fill = prefilled == 2 ? 1 : 0;  // (prefilled: see pk_rec_fwd_bf16)
    if (!prefilled) PK_CHECK_HIP(hipMemsetAsync(Yb, 0xFF, (size_t)T * B * y_pitch * 2, st));
    if (!prefilled) PK_CHECK_HIP(hipMemsetAsync(Xb, 0xFF, (size_t)T * B * y_pitch * 2, st));
    rc = pk_rec2_ln_setup(st, a, pl, ln, false);
    if (rc) return rc;
    const size_t lds = 2 * (size_t)RMAX * pk_r2_lda_bf16(KPAD) * 2 + 4 * ((size_t)(2 * G + 1 + (ln ? 1 : 0)) * 1024 + 512) + 16;
    const int slot = cell == PK_CELL_GRU ? 0 : 1;
    const Rec2gKernel fn = ln ? (cell == PK_CELL_GRU ? rec2g_fwd_kernel<PK_CELL_GRU, -1, true> : rec2g_fwd_kernel<PK_CELL_MINGRU, -1, true>)
                              : (cell == PK_CELL_GRU ? pick_fwd<PK_CELL_GRU>(act) : pick_fwd<PK_CELL_MINGRU>(act));
    size_t& granted = ln ? granted_ln[0][slot] : granted_lds[0][slot][act_slot(act)];
    if (granted < lds) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted = lds;
    }
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        a.ln_cg0 = l * pl.C;
        rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        dim3 grid(pl.C * pl.Pn), block(256);
        rc = pk_rec2_check_residency((const void*)fn, 256, lds, pl.C * pl.Pn, "pk_rec2p_*_bf16");
        if (rc) return rc;
        // (forward: the projections only - this cell's S has its own layout)
        const int help = ln ? 0 : (pk_rec_helper_wanted(false, pl.launches, cell) & 1);
        if (help && (rc = pk_rec_helper_fork(st)) != 0) return rc;
        hipLaunchKernelGGL(fn, grid, block, lds, st, a);
        PK_LAUNCH_CHECK();
        if (help && (rc = pk_rec_helper_launch(st, a, pl, G, 0, false, false, help)) != 0) return rc;
    }
    return 0;
}

int rec2p_bwd_impl(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U,
                   const float* mask, float mask_scalar, const float* Y, const float* S, const float* dY,
                   uint16_t* dGb, int64_t g_pitch, int prefilled, const PkLnHost* ln) {
    int rc = pk_rec2_check("pk_rec2p_bwd_bf16", cell == PK_CELL_GRU || cell == PK_CELL_MINGRU, cell, T, B, bidir, H);
    if (rc) return rc;
    hipStream_t st = pk_stream(stream);
    const int ndir = 1 + bidir, R = B * ndir, Hp = (H + 7) & ~7, G = pk_cell_gates(cell), G1 = G - 1;
    PK_REQUIRE(g_pitch >= (int64_t)G * Hp && (g_pitch % 8) == 0 && ((uintptr_t)dGb & 15) == 0,
               "pk_rec2p_bwd_bf16: dGb pitch must be a multiple of 8 and hold %d x %d elements", G, Hp);
    PK_REQUIRE((double)ndir * T * B * g_pitch * 2.0 < 4.0e9, "pk_rec2p_bwd_bf16: exchange buffer exceeds the 4 GB buffer-descriptor range");
    Plan2 pl;
    rc = pk_rec2_make_plan(R, H, pl);
    if (rc) return rc;
    R2Args a;
    a.T = T; a.B = B; a.R = R; a.H = H; a.Hp = Hp; a.YH = ndir * H; a.act = act;
    a.C = pl.C; a.Pn = pl.Pn; a.rpc = pl.rpc; a.row0 = 0;
    a.P = nullptr; a.pscale = nullptr; a.pshift = nullptr; a.U = U; a.mask = mask; a.mask_scalar = mask_scalar;
    a.Y = const_cast<float*>(Y); a.S = const_cast<float*>(S); a.Yb = nullptr; a.Xb = nullptr; a.Ypitch = 0;
    a.dY = dY; a.dP2 = nullptr; a.dGb = (unsigned short*)dGb; a.Gpitch = (int)g_pitch;
    rc = pk_rec2_host_setup(a, true, cell);
    if (rc) return rc;
    a.self_fill = prefilled == 2 ? 1 : 0;
    if (!prefilled) PK_CHECK_HIP(hipMemsetAsync(dGb, 0xFF, (size_t)ndir * T * B * g_pitch * 2, st));
    rc = pk_rec2_ln_setup(st, a, pl, ln, true);
    if (rc) return rc;
    const bool g3 = bwd_gen3() && !ln;  // (the LayerNorm variant exists in the second-generation kernel)
    const size_t tiles = (size_t)RMAX * pk_r2_lda_bf16(G1 * KPAD) * 2 + (size_t)RMAX * pk_r2_lda_bf16(KPAD) * 2;
    const size_t lds = g3 ? tiles + 32 : tiles + 4 * ((size_t)(G + 2 + (ln ? 1 : 0)) * 1024 + (size_t)G * 512) + 16;
    const int slot = cell == PK_CELL_GRU ? 0 : 1;
    const Rec2gKernel fn = ln ? (cell == PK_CELL_GRU ? rec2g_bwd_kernel<PK_CELL_GRU, -1, true> : rec2g_bwd_kernel<PK_CELL_MINGRU, -1, true>)
                         : g3 ? (cell == PK_CELL_GRU ? pick_bwd3<PK_CELL_GRU>(act) : pick_bwd3<PK_CELL_MINGRU>(act))
                              : (cell == PK_CELL_GRU ? pick_bwd<PK_CELL_GRU>(act) : pick_bwd<PK_CELL_MINGRU>(act));
    size_t& granted = ln ? granted_ln[1][slot] : g3 ? granted_lds3[slot][act_slot(act)] : granted_lds[1][slot][act_slot(act)];
    if (granted < lds) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        granted = lds;
    }
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        a.ln_cg0 = l * pl.C;
        rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        dim3 grid(pl.C * pl.Pn), block(256);
        rc = pk_rec2_check_residency((const void*)fn, 256, lds, pl.C * pl.Pn, "pk_rec2p_*_bf16");
        if (rc) return rc;
        hipLaunchKernelGGL(fn, grid, block, lds, st, a);
        PK_LAUNCH_CHECK();
    }
    return pk_rec2_ln_finish(st, a, ln);
}

}  // namespace

extern "C" int pk_rec2p_fwd_bf16(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                                 const float* pscale, const float* pshift, const float* U, const float* mask,
                                 float mask_scalar, float* Y, float* S, uint16_t* Yb, uint16_t* Xb, int64_t y_pitch, int prefilled) {
    return rec2p_fwd_impl(stream, cell, act, T, B, bidir, H, P, pscale, pshift, U, mask, mask_scalar, Y, S, Yb, Xb, y_pitch,
                          prefilled, nullptr);
}
extern "C" int pk_rec2p_bwd_bf16(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U,
                                 const float* mask, float mask_scalar, const float* Y, const float* S, const float* dY,
                                 uint16_t* dGb, int64_t g_pitch, int prefilled) {
    return rec2p_bwd_impl(stream, cell, act, T, B, bidir, H, U, mask, mask_scalar, Y, S, dY, dGb, g_pitch, prefilled, nullptr);
}
// ... with per-step LayerNorm of h_t (GRU: neural_networks.py:638-639, minimalGRU: :1299-1300); arguments as
// pk_rec_fwd_bf16_ln / pk_rec_bwd_bf16_ln
extern "C" int pk_rec2p_fwd_bf16_ln(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                                    const float* pscale, const float* pshift, const float* U, const float* mask,
                                    float mask_scalar, const float* ln_gamma, const float* ln_beta, float ln_eps, float* Y,
                                    float* S, float* LNS, uint16_t* Yb, uint16_t* Xb, int64_t y_pitch, int prefilled,
                                    float* lnwork) {
    PK_REQUIRE(ln_gamma && ln_beta && LNS && lnwork, "pk_rec2p_fwd_bf16_ln: null LayerNorm argument");
    const PkLnHost ln = {ln_gamma, ln_beta, ln_eps, LNS, lnwork, nullptr, nullptr};
    return rec2p_fwd_impl(stream, cell, act, T, B, bidir, H, P, pscale, pshift, U, mask, mask_scalar, Y, S, Yb, Xb, y_pitch,
                          prefilled, &ln);
}
extern "C" int pk_rec2p_bwd_bf16_ln(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U,
                                    const float* mask, float mask_scalar, const float* ln_gamma, float ln_eps, const float* Y,
                                    const float* S, const float* LNS, const float* dY, uint16_t* dGb, int64_t g_pitch,
                                    int prefilled, float* lnwork, float* dln_gamma, float* dln_beta) {
    PK_REQUIRE(ln_gamma && LNS && lnwork && dln_gamma && dln_beta, "pk_rec2p_bwd_bf16_ln: null LayerNorm argument");
    const PkLnHost ln = {ln_gamma, nullptr, ln_eps, const_cast<float*>(LNS), lnwork, dln_gamma, dln_beta};
    return rec2p_bwd_impl(stream, cell, act, T, B, bidir, H, U, mask, mask_scalar, Y, S, dY, dGb, g_pitch, prefilled, &ln);
}
