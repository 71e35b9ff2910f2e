// pk_rec_persist2_lstm.hip - the LSTM time loops of pk_rec_persist2.hip with EIGHT waves per workgroup.
//
// Why: a wave that owns 16 hidden units of an LSTM holds 4 gates x 18 k-steps of MFMA B fragments = 288 registers per
// lane.  With the four-wave kernels (one wave per SIMD) the compiler leaves a part of that array in scratch memory
// (forward: 490 spilled VGPRs; backward: the whole array, 108 scratch_load_dwordx4 per step).  Here two
// waves share each group of 16 units and each holds HALF of the fragments (144 registers), so that both fit the 256
// registers a wave gets at two waves per SIMD:
//
//   forward   gate split: wave h of a pair multiplies h_{t-1} with U of gates (2h, 2h+1) = (f, i) / (o, g), applies
//             the gate non-linearities to its own two, and the (o, g) wave hands its ACTIVATED values to the (f, i)
//             wave through LDS; the (f, i) wave finishes c_t, h_t and publishes.  Per gate the MFMA accumulation order
//             is the one of the four-wave kernel, so the two kernels produce the same h_t.
//   backward  K split: dh_{t-1} = sum_g dgate_g . U_g; wave h accumulates gates (2h, 2h+1), the second wave hands its
//             partial sum (4 floats per lane) over, the first one does the gate math and publishes.  The hand-over
//             barrier is the one the single A tile (74 KB) needs anyway.
//
// The two waves of a pair sit on the same SIMD (wave w and w+4), so the split does not change the MFMA or VALU work per
// SIMD: the gain is the scratch traffic that disappears and MFMA issue from two waves instead of one.
// Everything else (cluster exchange through L2, sentinel protocol, patches, prefetch order) is pk_rec_persist2.hip's;
// see the notes there.  Reference time loop: neural_networks.py:457-469.
#include <stdlib.h>

#include "pk_rec2_common.h"

namespace {

constexpr int LG = 4;    // gates [f, i, o, g]
constexpr int LNS = 5;   // saved slots f, i, o, g, c
constexpr int GW = 2;    // gates per wave
constexpr int NWV = 8;   // waves per workgroup

// poll_to_lds (pk_rec2_common.h) with the LDS destinations read from a per-thread LDS table (stride 512 entries per
// slot) once the chunks have landed, instead of being held in registers across the whole step
template <int NCH, bool FAST>
__device__ __forceinline__ bool poll_to_lds_tab(__amdgpu_buffer_rsrc_t rs, const unsigned (&goff)[NCH], const int* ltab,
                                                unsigned char* tile, unsigned* err, int spin_limit, int lane, bool dead,
                                                int& retries) {
    u32x4 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = poll_load<FAST>(rs, goff[i]);
    bool bad = false;
#pragma unroll
    for (int i = 0; i < NCH; ++i) bad = bad | has_sent16(v[i]);
    if (__any(bad) && !dead) {
        int spins = 0;
        while (true) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (has_sent16(v[i])) v[i] = poll_load<FAST>(rs, goff[i]);
            bad = false;
#pragma unroll
            for (int i = 0; i < NCH; ++i) bad = bad | has_sent16(v[i]);
            ++retries;
            if (!__any(bad)) break;
            if (spin_check2(spins, spin_limit, err, lane)) {
                dead = true;
                break;
            }
        }
    }
    int lo[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) lo[i] = ltab[i * 512];
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(tile + lo[i]) = v[i];
    return dead;
}

// ============================================================================
// forward
// ============================================================================
template <int ACT>
__global__ __launch_bounds__(512, 1) void rec2l_fwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int LDA = pk_r2_lda_bf16(KPAD);
    constexpr int ATILE = RMAX * LDA * 2;
    constexpr int NCH = (RMAX * (KPAD / 8) + 511) / 512;   // 16-byte chunks polled per lane (3)
    constexpr int WAVE_LDS = (GW + 1 + 3) * 1024 + 512;    // P stage (2 gates) | Y | up to 3 S slots | bf16 publish patch
    constexpr int XCH = 2 * ATILE + NWV * WAVE_LDS;        // pair hand-over: [4 pairs][2 gates][64 lanes] x 16 bytes
    constexpr int LDS_TRASH = XCH + 4 * GW * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, uw = wave & 3;
    const int half = __builtin_amdgcn_readfirstlane(wave >> 2);  // 0: gates f, i (+ cell state, publish); 1: gates o, g
    const int g0 = half * GW;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = LG * H;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int ubase = p * 64 + uw * 16;
    const int unit = ubase + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;

    // ---- recurrent weights of my 16 units, my two gates -> registers: B[k][n] = U_g[unit n][k]
    bf16x8 Bf[GW][KSTEPS];
    {
        const unsigned szU = (unsigned)((size_t)LG * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
        // six k-steps at a time, fenced: left alone the scheduler issues all 72 loads of the pair of gates first and
        // the 288 registers they land in push every long-lived value of the kernel into scratch memory
        constexpr int KB = 6;
#pragma unroll
        for (int gg = 0; gg < GW; ++gg) {
            const int g = g0 + gg;
#pragma unroll
            for (int kb = 0; kb < KSTEPS; kb += KB) {
                u32x4 raw[KB][2];
#pragma unroll
                for (int kj = 0; kj < KB; ++kj) {
                    const int k0 = (kb + kj) * 32 + kq * 8;
                    const unsigned off = (unsigned)(((g * H + unit) * H + k0) * 4);
                    raw[kj][0] = __builtin_amdgcn_raw_buffer_load_b128(rsU, (unit_ok && k0 < H) ? off : szU, 0, 0);
                    raw[kj][1] = __builtin_amdgcn_raw_buffer_load_b128(rsU, (unit_ok && k0 + 4 < H) ? off + 16 : szU, 0, 0);
                }
#pragma unroll
                for (int kj = 0; kj < KB; ++kj) {
                    bf16x8 f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = (kb + kj) * 32 + kq * 8 + e;
                        const float w = (k < H) ? __uint_as_float(raw[kj][e >> 2][e & 3]) : 0.f;  // beyond H: the next row's data
                        f[e] = (short)pk_f2bf(w);
                    }
                    Bf[gg][kb + kj] = f;
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float psc[GW], psh[GW];
#pragma unroll
    for (int gg = 0; gg < GW; ++gg) {
        psc[gg] = unit_ok ? a.pscale[(g0 + gg) * H + unit] : 0.f;
        psh[gg] = unit_ok ? a.pshift[(g0 + gg) * H + unit] : 0.f;
    }
    for (int i = tid; i < (LDS_TRASH + 16) / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- poll descriptors: chunk ci = (row, col) of the cluster's [nrows][Hp/8] block of h_{t-1}
    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Ypitch * 2u;
    const unsigned szYb = (unsigned)T * TS;
    unsigned cbase[NCH], cstep[NCH];
    int clds[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = tid + 512 * i;
        const bool ok = ci < nrows * CPR;
        const int row = ok ? ci / CPR : 0, col = ok ? ci - row * CPR : 0;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        cbase[i] = ok ? ((unsigned)b * a.Ypitch + dir * Hp + col * 8) * 2u + (unsigned)(dir ? (T - 1) : 0) * TS : szYb;
        cstep[i] = ok ? (dir ? 0u - TS : TS) : 0u;
        clds[i] = ok ? row * (LDA * 2) + col * 16 : LDS_TRASH;
    }
    // ---- gate-math (C/D) layout: rows kq*4 + r, unit lane&15
    float rvf[4], msk[4], cprev[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r, n = n_base + row;
        const bool ok = row < nrows && unit_ok;
        rvf[r] = ok ? 1.f : 0.f;
        msk[r] = (a.mask != nullptr && ok) ? a.mask[(long)n * H + unit] : a.mask_scalar;
        cprev[r] = 0.f;
    }
    // ---- vector layout: row lane>>2, units ubase + (lane&3)*4 .. +3
    const int vrow = lane >> 2, vu0 = ubase + (lane & 3) * 4;
    const int vn = n_base + (vrow < nrows ? vrow : 0);
    const int vdir = vn >= B ? 1 : 0, vb = vn - vdir * B;
    int vnv = H - vu0;
    vnv = vnv > 4 ? 4 : (vnv < 0 ? 0 : vnv);
    const int edge = __any(vnv > 0 && vnv < 4) != 0 ? ((H & 1) ? 2 : 1) : 0;
    vnv = vrow < nrows ? vnv : 0;
    const unsigned vP0 = ((unsigned)vb * GH + vu0), vPs = (unsigned)B * GH;
    const unsigned vY0 = ((unsigned)vb * a.YH + vdir * H + vu0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)vdir * T * B + vb) * (LNS * H) + vu0), vSs = (unsigned)B * LNS * H;
    // ---- publish descriptors (first wave of a pair): lanes 0..31 store one 16-byte piece (row, 8 units)
    const int prow = lane >> 1, phalf = lane & 1;
    const int pu0 = ubase + phalf * 8;
    const bool pk_ok = lane < 32 && prow < nrows && pu0 < Hp;
    const int pn = n_base + (prow < nrows ? prow : 0);
    const int pdir = pn >= B ? 1 : 0, pb = pn - pdir * B;
    const unsigned pbase = pk_ok ? ((unsigned)pb * a.Ypitch + pdir * Hp + pu0) * 2u : szYb;

    unsigned char* wl = smem + 2 * ATILE + wave * WAVE_LDS;
    float* patchP = reinterpret_cast<float*>(wl);                     // [GW][256]
    float* patchY = reinterpret_cast<float*>(wl + GW * 1024);         // [256]           (first wave)
    float* patchS = reinterpret_cast<float*>(wl + (GW + 1) * 1024);   // first wave: f, i, c; second wave: o, g
    unsigned short* patchB = reinterpret_cast<unsigned short*>(wl + (GW + 1 + 3) * 1024);  // [16][16] bf16
    unsigned char* xch = smem + XCH + uw * (GW * 1024) + lane * 16;   // + gg * 1024
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.Yb, szYb);
    float* trash = a.trash + (tid & 63) * 4;

    f32x4 pv[GW];
    auto load_proj = [&](int tt, auto E) {
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
#pragma unroll
        for (int gg = 0; gg < GW; ++gg) pv[gg] = ld4<decltype(E)::value>(a.P, vP0 + ts * vPs + (g0 + gg) * H, vnv);
    };
    // layer output and saved gates of step tt: wave patches -> HBM, 16 bytes per lane.  First wave: Y and the slots
    // f, i, c (0, 1, 4); second wave: o, g (2, 3)
    auto flush_outputs = [&](int tt, auto E, auto HALFC) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
        // (a.S == null: a validation / forward chunk - nothing is saved; a.Y == null with it: an inner layer, Yb is its output)
        if constexpr (decltype(HALFC)::value == 0) {
            if (a.Y != nullptr) st4<EE>(a.Y, vY0 + ts * vYs, vnv, trash, patch_get_vec(patchY, lane));
            if (a.S != nullptr) {
                st4<EE>(a.S, vS0 + ts * vSs + 0 * H, vnv, trash, patch_get_vec(patchS, lane));
                st4<EE>(a.S, vS0 + ts * vSs + 1 * H, vnv, trash, patch_get_vec(patchS + 256, lane));
                st4<EE>(a.S, vS0 + ts * vSs + 4 * H, vnv, trash, patch_get_vec(patchS + 512, lane));
            }
        } else if (a.S != nullptr) {
            st4<EE>(a.S, vS0 + ts * vSs + 2 * H, vnv, trash, patch_get_vec(patchS, lane));
            st4<EE>(a.S, vS0 + ts * vSs + 3 * H, vnv, trash, patch_get_vec(patchS + 256, lane));
        }
    };
#define PK_LLP0(E) load_proj(0, E)
    PK_EDGE_DISPATCH(PK_LLP0);
    // self-filling exchange (pk_rec2_common.h): the publishing lanes (first wave of a pair) pattern their own chunks
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (a.self_fill && half == 0) {
        for (int tt = 0; tt < PK_R2_FILL_AHEAD && tt < T; ++tt)
            pub_store<false>(rs, pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - tt) : tt) * TS : 0u), sentinel);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    // the time loop, instantiated per (XCD-local fast path?, static edge case?, which wave of the pair?)
    auto run = [&](auto FASTC, auto SEC, auto HALFC) {
    constexpr bool fast = decltype(FASTC)::value != 0;
    constexpr int SE = decltype(SEC)::value;
    constexpr int HALF = decltype(HALFC)::value;
    for (int t = 0; t < T; ++t) {
        f32x4 acc[GW];
#pragma unroll
        for (int gg = 0; gg < GW; ++gg) acc[gg] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned char* At = smem + (t & 1) * ATILE;
        // this step's projections (loaded a step ago) -> the wave's patch, before the poll takes its registers
#pragma unroll
        for (int gg = 0; gg < GW; ++gg) patch_put_vec(patchP + gg * 256, lane, pv[gg]);
        if (t > 0) {
            unsigned goff[NCH];
#pragma unroll
            for (int i = 0; i < NCH; ++i) goff[i] = cbase[i] + (unsigned)(t - 1) * cstep[i];
            // the second wave of a pair is done earlier than the first: it waits a little longer before it polls
            const int nap = a.poll_delay + (HALF ? a.helper_delay : 0);
            for (int d = 0; d < nap; ++d) __builtin_amdgcn_s_sleep(1);
            int retries = 0;
            dead = poll_to_lds<NCH, fast>(rs, goff, clds, At, a.err, a.spin_limit, lane, dead, retries);
        }
        PK_BARRIER_LDS();  // A: the polled h_{t-1} tile is complete
        // off the dependency chain: fp32 outputs of the previous step, projections of the next one
        if (t > 0) {
#define PK_LFO(E) flush_outputs(t - 1, E, HALFC)
            PK_EDGE_DISPATCH_S(PK_LFO);
        }
        if (t + 1 < T) {
#define PK_LLP1(E) load_proj(t + 1, E)
            PK_EDGE_DISPATCH_S(PK_LLP1);
        }
        if constexpr (HALF == 0) {
            if (a.self_fill && t + PK_R2_FILL_AHEAD < T)
                pub_store<fast>(rs, pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - (t + PK_R2_FILL_AHEAD)) : (t + PK_R2_FILL_AHEAD)) * TS : 0u), sentinel);
        }
        if (t > 0) {
            const unsigned char* Ar = At + (lane & 15) * (LDA * 2) + kq * 16;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ar + kk * 64);
#pragma unroll
                for (int gg = 0; gg < GW; ++gg) acc[gg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[gg][kk], acc[gg], 0, 0, 0);
            }
        }
        // ---- my two gates for my (row, unit) pairs
        float pre[GW][4];
#pragma unroll
        for (int gg = 0; gg < GW; ++gg) patch_get_cd(patchP + gg * 256, kq, lane, pre[gg]);
        float g_a[4], g_b[4];  // first wave: f, i;  second wave: o, g
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pa = pre[0][r] * psc[0] + psh[0] + acc[0][r];
            const float pb2 = pre[1][r] * psc[1] + psh[1] + acc[1][r];
            g_a[r] = pk_sig(pa);
            g_b[r] = HALF == 0 ? pk_sig(pb2) : pk_act(act, pb2);
        }
        if constexpr (HALF == 1) {
            *reinterpret_cast<f32x4*>(xch) = f32x4{g_a[0], g_a[1], g_a[2], g_a[3]};
            *reinterpret_cast<f32x4*>(xch + 1024) = f32x4{g_b[0], g_b[1], g_b[2], g_b[3]};
        }
        PK_BARRIER_LDS();  // B: (o, g) of this step are in the pair's hand-over slots
        if constexpr (HALF == 1) {
            // saved o, g for backward; written to HBM at the top of the next step
            patch_put_cd(patchS, kq, lane, g_a);
            patch_put_cd(patchS + 256, kq, lane, g_b);
            PK_LDS_ORDER();
        } else {
            const f32x4 o4 = *reinterpret_cast<const f32x4*>(xch);
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(xch + 1024);
            float hv[4], cv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // neural_networks.py:460-464 with pk_cell_fwd<LSTM>'s expression order
                float cc = g_b[r] * g4[r] * msk[r] + g_a[r] * cprev[r];
                float h = o4[r] * pk_act(act, cc);
                h = rvf[r] != 0.f ? h : 0.f;  // rows / units outside the layer carry exact zeros (published as padding)
                cc = rvf[r] != 0.f ? cc : 0.f;
                cprev[r] = cc;
                hv[r] = h;
                cv[r] = cc;
                patchB[(kq * 4 + r) * 16 + (lane & 15)] = to_bf_pub(h);
            }
            // ---- publish h_t first: it is what the other workgroups of the cluster wait for
            PK_LDS_ORDER();
            {
                const u32x4 o = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(patchB) + (prow & 15) * 32 + phalf * 16);
                const unsigned off = pbase + (pk_ok ? (unsigned)(pdir ? (T - 1 - t) : t) * TS : 0u);
                pub_store<fast>(rs, off, o);
            }
            patch_put_cd(patchY, kq, lane, hv);
            patch_put_cd(patchS, kq, lane, g_a);
            patch_put_cd(patchS + 256, kq, lane, g_b);
            patch_put_cd(patchS + 512, kq, lane, cv);
            PK_LDS_ORDER();
        }
    }
    };
    if (half == 0) {
        auto run0 = [&](auto F, auto E) { run(F, E, BoolC<0>()); };
        PK_RUN_SPECIALISED(run0, fast_rt);
#define PK_LFOL0(E) flush_outputs(T - 1, E, BoolC<0>())
        PK_EDGE_DISPATCH(PK_LFOL0);
    } else {
        auto run1 = [&](auto F, auto E) { run(F, E, BoolC<1>()); };
        PK_RUN_SPECIALISED(run1, fast_rt);
#define PK_LFOL1(E) flush_outputs(T - 1, E, BoolC<1>())
        PK_EDGE_DISPATCH(PK_LFOL1);
    }
}

// ============================================================================
// backward: dL/dh_{t-1} = [dgates_t] . [U_f; U_i; U_o; U_g]
// ============================================================================
template <int ACT>
__global__ __launch_bounds__(512, 1) void rec2l_bwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int LDA = pk_r2_lda_bf16(LG * KPAD);
    constexpr int ATILE = RMAX * LDA * 2;                        // one tile (74 KB): barrier B of a step frees it
    constexpr int NCH = (RMAX * LG * (KPAD / 8) + 511) / 512;    // 9
    constexpr int NIN = LNS + 3;                                 // f, i, o, g, c | h_{t-1}, dY, c_{t-1}
    constexpr int NLD = NIN / 2;                                 // slots each wave of a pair loads (4)
    constexpr int PAIR_LDS = (NIN + LG) * 1024 + LG * 512;       // input patches | dgate fp32 patches | dgate bf16 patches
    constexpr int XCH = ATILE + 4 * PAIR_LDS;                    // [4 pairs][64 lanes] x 16 bytes: partial dh
    constexpr int LTAB = XCH + 4 * 1024;                         // [NCH][512] LDS byte offsets of the polled chunks
    constexpr int LDS_TRASH = LTAB + NCH * 512 * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, uw = wave & 3;
    const int half = __builtin_amdgcn_readfirstlane(wave >> 2);  // 0: gates f, i of K + the gate math; 1: gates o, g of K
    const int g0 = half * GW;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = LG * H;
    const unsigned TB = (unsigned)T * B;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int ubase = p * 64 + uw * 16;
    const int unit = ubase + (lane & 15);
    const bool unit_ok = unit < H;
    const int kq = lane >> 4;

    // B[kidx = (g, j)][n = unit] = U_g[j][unit], my two gates
    bf16x8 Bf[GW][KSTEPS];
    {
        const unsigned szU = (unsigned)((size_t)LG * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int gg = 0; gg < GW; ++gg)
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                unsigned raw[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = kk * 32 + kq * 8 + e;
                    raw[e] = __builtin_amdgcn_raw_buffer_load_b32(
                        rsU, (unit_ok && j < H) ? (unsigned)((((g0 + gg) * H + j) * H + unit) * 4) : szU, 0, 0);
                }
                bf16x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (short)pk_f2bf(__uint_as_float(raw[e]));
                Bf[gg][kk] = f;
                if (kk % 3 == 2) {  // fenced in groups of 24 loads (see the forward kernel)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    }
    for (int i = tid; i < (LDS_TRASH + 16) / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- poll descriptors: chunk ci = (row, gate, col) of the cluster's dgates_{t+1} block
    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Gpitch * 2u;
    const unsigned ndir = (unsigned)(a.R / B);
    const unsigned szGb = ndir * (unsigned)T * TS;
    // goff[i] = byte offset of chunk i at the step about to be polled; it moves by one time slab per step, up for the
    // backward direction's rows and down for the forward one's (bit i of dirbits).  Chunk slots a lane does not own
    // alias the cluster's first chunk (a harmless duplicate read of a real chunk) and land in the LDS trash slot: that
    // keeps the per-step update to a select and an add per chunk and the descriptors to two registers per chunk.
    // The LDS destinations of the chunks (nine more loop-invariant registers, which the allocator kept in scratch
    // memory and reloaded one by one behind every poll) live in an LDS table instead: [slot][thread], conflict free.
    unsigned goff[NCH], dirbits = 0u;
    int clds[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = tid + 512 * i;
        const bool ok = ci < nrows * LG * CPR;
        const int row = ok ? ci / (LG * CPR) : 0;
        const int rem = ok ? ci - row * (LG * CPR) : 0;
        const int g = rem / CPR, col = rem - g * CPR;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        // iteration it (t = T-1-it, it >= 1) reads storage time (dir ? it-1 : T-it)
        goff[i] = (unsigned)dir * (unsigned)T * TS + ((unsigned)b * a.Gpitch + g * Hp + col * 8) * 2u +
                  (unsigned)(dir ? 0 : (T - 1)) * TS;
        dirbits |= (unsigned)dir << i;
        clds[i] = ok ? row * (LDA * 2) + (g * KPAD + col * 8) * 2 : LDS_TRASH;
    }
    float rvf[4], msk[4], dc_car[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = kq * 4 + r, n = n_base + row;
        const bool ok = row < nrows && unit_ok;
        rvf[r] = ok ? 1.f : 0.f;
        msk[r] = (a.mask != nullptr && ok) ? a.mask[(long)n * H + unit] : a.mask_scalar;
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
    const unsigned vS0 = (((unsigned)vdir * TB + vb) * (LNS * H) + vu0), vSs = (unsigned)B * LNS * H;
    const unsigned vG0 = (((unsigned)vdir * TB + vb) * GH + vu0), vGs = (unsigned)B * GH;
    // ---- publish descriptors (first wave): 4 gates x 16 rows x 2 halves = 128 16-byte pieces, two per lane
    unsigned pbase[2];
    int plds[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = j * 64 + lane;
        const int g = idx >> 5, prow = (idx >> 1) & 15, phalf = idx & 1;
        const int pu0 = ubase + phalf * 8;
        const bool ok = prow < nrows && pu0 < Hp;
        const int pn = n_base + (prow < nrows ? prow : 0);
        const int pdir = pn >= B ? 1 : 0, pb = pn - pdir * B;
        // direction slab + row + gate + 8-unit piece; the time slab is added per step.  plds packs what the step needs:
        // bits 0..15 = byte offset of the piece in the bf16 patches, bit 29 = this lane stores, bit 30 = direction
        pbase[j] = ok ? (unsigned)pdir * (unsigned)T * TS + ((unsigned)pb * a.Gpitch + g * Hp + pu0) * 2u : szGb;
        plds[j] = (pdir << 30) | (ok ? 1 << 29 : 0) | (g * 512 + prow * 32 + phalf * 16);
    }

    unsigned char* pl = smem + ATILE + uw * PAIR_LDS;
    float* patchI = reinterpret_cast<float*>(pl);                       // [NIN][256]: f, i, o, g, c, hp, dY, cp
    float* patchG = reinterpret_cast<float*>(pl + NIN * 1024);          // [LG][256] fp32 gate gradients
    unsigned char* patchB = pl + (NIN + LG) * 1024;                     // [LG][16][16] bf16
    unsigned char* xch = smem + XCH + uw * 1024 + lane * 16;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dGb, szGb);
    float* trash = a.trash + (tid & 63) * 4;

    // saved tensors of one step in the vector layout.  First wave: slots 0..3 = f, i, o, g; second wave: c, h_{t-1},
    // dY, c_{t-1}.  Steps are loaded in order T-1, T-2, ...: the element offsets of the step to load next are running
    // values (oS into S, oY into Y / dY) that move by one time slab per load, up for the backward direction's rows and
    // down for the forward one's - an add instead of the 64-bit multiply-add per tensor the closed form compiles to.
    const unsigned dS = vdir ? vSs : 0u - vSs, dYs = vdir ? vYs : 0u - vYs;
    unsigned oS = vS0 + (unsigned)(vdir ? 0 : T - 1) * vSs, oY = vY0 + (unsigned)(vdir ? 0 : T - 1) * vYs;
    f32x4 iv[NLD];
    auto load_step_e = [&](int t, auto E, auto HALFC) {
        constexpr int EE = decltype(E)::value;
        if constexpr (decltype(HALFC)::value == 0) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) iv[k] = ld4<EE>(a.S, oS + k * H, vnv);
        } else {
            const int nvp = t > 0 ? vnv : 0;  // step t-1 sits one slab further along (nothing there when t == 0)
            iv[0] = ld4<EE>(a.S, oS + 4 * H, vnv);
            iv[1] = ld4<EE>(a.Y, oY + dYs, nvp);
            iv[2] = ld4<EE>(a.dY, oY, vnv);
            iv[3] = ld4<EE>(a.S, oS + dS + 4 * H, nvp);
            if (t == 0) {  // h_{-1} = c_{-1} = 0
                iv[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                iv[3] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        oS += dS;
        oY += dYs;
    };
    // fp32 gate gradients of step tt (only when the caller wants them): wave h writes gates 2h, 2h+1
    auto flush_outputs_e = [&](int tt, auto E) {
        const unsigned ts = (unsigned)(vdir ? (T - 1 - tt) : tt);
#pragma unroll
        for (int gg = 0; gg < GW; ++gg)
            st4<decltype(E)::value>(a.dP2, vG0 + ts * vGs + (g0 + gg) * H, vnv, trash, patch_get_vec(patchG + (g0 + gg) * 256, lane));
    };
    auto flush_outputs = [&](int tt) {
        if (a.dP2 == nullptr) return;  // perf mode: BatchNorm backward works from the bf16 copy
#define PK_LFOB(E) flush_outputs_e(tt, E)
        PK_EDGE_DISPATCH(PK_LFOB);
    };
    if (half == 0) {
#define PK_LLS0(E) load_step_e(T - 1, E, BoolC<0>())
        PK_EDGE_DISPATCH(PK_LLS0);
    } else {
#define PK_LLS1(E) load_step_e(T - 1, E, BoolC<1>())
        PK_EDGE_DISPATCH(PK_LLS1);
    }
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto fill_slab = [&](int tt, auto FASTC) {  // my pieces of the slab that step tt will publish
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = (plds[j] >> 29) & 1;
            const int pdir = (plds[j] >> 30) & 1;
            pub_store<decltype(FASTC)::value != 0>(rs, pbase[j] + (ok ? (unsigned)(pdir ? (T - 1 - tt) : tt) * TS : 0u), sentinel);
        }
    };
    if (a.self_fill && half == 0) {
        for (int k = 0; k < PK_R2_FILL_AHEAD && k < T; ++k) fill_slab(T - 1 - k, BoolC<0>());
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int* ltab = reinterpret_cast<int*>(smem + LTAB) + tid;
#pragma unroll
    for (int i = 0; i < NCH; ++i) ltab[i * 512] = clds[i];

    bool dead = false;
    const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    auto run = [&](auto HALFC) {
    constexpr int HALF = decltype(HALFC)::value;
    const bool fast = fast_rt;
    int it = 0;
    for (int t = T - 1; t >= 0; --t, ++it) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned char* At = smem;
        // this step's saved tensors (loaded a step ago) go to the pair's patches BEFORE the poll: their registers are
        // free while the poll holds its chunks (the patches were last read before barrier B of the previous step)
#pragma unroll
        for (int k = 0; k < NLD; ++k) patch_put_vec(patchI + (HALF * NLD + k) * 256, lane, iv[k]);
        if (t < T - 1) {
            const int nap = a.poll_delay + (HALF ? a.helper_delay : 0);
            for (int d = 0; d < nap; ++d) __builtin_amdgcn_s_sleep(1);
            int retries = 0;
            dead = fast ? poll_to_lds_tab<NCH, true>(rs, goff, ltab, At, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds_tab<NCH, false>(rs, goff, ltab, At, a.err, a.spin_limit, lane, dead, retries);
        }
        PK_BARRIER_LDS();  // A: the polled dgates_{t+1} tile and the pair's input patches are complete
        // off the dependency chain: fp32 gate gradients of the previous step (if wanted)
        if (t < T - 1) flush_outputs(t + 1);
        if constexpr (HALF == 0) {
            if (a.self_fill && t - PK_R2_FILL_AHEAD >= 0) {
                if (fast) fill_slab(t - PK_R2_FILL_AHEAD, BoolC<1>());
                else fill_slab(t - PK_R2_FILL_AHEAD, BoolC<0>());
            }
        }
        if (t < T - 1) {
            const unsigned char* Ar = At + (lane & 15) * (LDA * 2) + kq * 16;
#pragma unroll
            for (int gg = 0; gg < GW; ++gg)
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ar + ((HALF * GW + gg) * KPAD + kk * 32) * 2);
                    if ((kk & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[gg][kk], acc0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, Bf[gg][kk], acc1, 0, 0, 0);
                }
        }
        // saved tensors of the next step: issued BEHIND the MFMA block, so that their registers are free during it
        // (with them in flight two B fragments lived in scratch memory, and the s_waitcnt vmcnt(0) of their reloads
        // put the HBM latency of these loads on the dependency chain); they have the gate math, the publish and the
        // cluster hand-over to land
        if (t > 0) {
#define PK_LLSN(E) load_step_e(t - 1, E, HALFC)
            PK_EDGE_DISPATCH(PK_LLSN);
        }
        if constexpr (HALF == 1) *reinterpret_cast<f32x4*>(xch) = acc0 + acc1;
        // the first wave reads its inputs before the barrier: they were complete at barrier A
        float sin[NIN][4];
        if constexpr (HALF == 0) {
#pragma unroll
            for (int k = 0; k < NIN; ++k) patch_get_cd(patchI + k * 256, kq, lane, sin[k]);
        }
        PK_BARRIER_LDS();  // B: the partial sums are handed over, and everybody is done reading the A tile
        if constexpr (HALF == 0) {
            const f32x4 part = *reinterpret_cast<const f32x4*>(xch);
            float dgv[LG][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s[LNS];
#pragma unroll
                for (int k = 0; k < LNS; ++k) s[k] = sin[k][r];
                const float hp = sin[LNS][r], dy = sin[LNS + 1][r], cp = sin[LNS + 2][r];
                const float dh = dy + acc0[r] + acc1[r] + part[r];
                float dg[LG], dhd, dcp;
                pk_cell_bwd<PK_CELL_LSTM>(act, s, hp, cp, msk[r], dh, dc_car[r], dg, dhd, dcp);
                dc_car[r] = rvf[r] != 0.f ? dcp : 0.f;
#pragma unroll
                for (int g = 0; g < LG; ++g) {
                    const float d = rvf[r] != 0.f ? dg[g] : 0.f;
                    dgv[g][r] = d;
                    reinterpret_cast<unsigned short*>(patchB)[g * 256 + (kq * 4 + r) * 16 + (lane & 15)] = to_bf_pub(d);
                }
            }
            PK_LDS_ORDER();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 o = *reinterpret_cast<const u32x4*>(patchB + (plds[j] & 0xFFFF));
                const bool ok = (plds[j] >> 29) & 1;
                const int pdir = (plds[j] >> 30) & 1;
                const unsigned off = pbase[j] + (ok ? (unsigned)(pdir ? (T - 1 - t) : t) * TS : 0u);
                if (fast) pub_store<true>(rs, off, o);
                else pub_store<false>(rs, off, o);
            }
            if (a.dP2 != nullptr) {
#pragma unroll
                for (int g = 0; g < LG; ++g) patch_put_cd(patchG + g * 256, kq, lane, dgv[g]);
                PK_LDS_ORDER();
            }
        }
        // next step's poll offsets, behind the publish.  (The empty asm keeps the nine selects in the loop: hoisted,
        // they are nine more loop-invariant registers, which the allocator then keeps in scratch memory.)
        if (t < T - 1) {
            asm volatile("" : "+v"(dirbits));
#pragma unroll
            for (int i = 0; i < NCH; ++i) goff[i] += ((dirbits >> i) & 1u) ? TS : 0u - TS;
        }
    }
    };
    if (half == 0) run(BoolC<0>());
    else run(BoolC<1>());
    __syncthreads();  // the second wave of a pair flushes gate gradients the first one wrote
    flush_outputs(0);
}

// ============================================================================
// backward, third generation (round 3): MFMA operands swapped - see pk_rec_persist3.hip.  A lane holds FOUR CONSECUTIVE
// UNITS of ONE row (row = lane & 15, units 4 * (lane >> 4) .. + 3), which is the layout of every tensor the step reads
// (S, Y, dY rows) and of half a publish chunk: the eight saved tensors come in with one 16-byte load per lane and are
// handed inside the pair through lane-indexed LDS slots (same lane mapping in both waves: no transposition), the bf16
// gate gradients leave through two v_permlane16_swap_b32 per chunk.  What is gone per step: 8 transposing patch
// writes + 32 four-byte patch reads (inputs), 16 two-byte patch writes + an LDS drain + 2 patch reads (publish).
// Same pair structure as rec2l_bwd_kernel: wave h of a pair multiplies gates (2h, 2h+1) of K, the second wave hands its
// partial sum over at barrier B, the first does the gate math and publishes.
// ============================================================================
__device__ __forceinline__ u32x4 pack_chunk_l(unsigned lo, unsigned hi) {
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return u32x4{a[0], b[0], a[1], b[1]};
}
__device__ __forceinline__ unsigned pack2_l(float x, float y) { return (unsigned)to_bf_pub(x) | ((unsigned)to_bf_pub(y) << 16); }

template <int ACT>
__global__ __launch_bounds__(512, 1) void rec3l_bwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int LDA = pk_r2_lda_bf16(LG * KPAD);
    constexpr int ATILE = RMAX * LDA * 2;                        // one tile (74 KB): barrier B of a step frees it
    constexpr int NCH = (RMAX * LG * (KPAD / 8) + 511) / 512;    // 9
    constexpr int NIN = LNS + 3;                                 // f, i, o, g, c | h_{t-1}, dY, c_{t-1}
    constexpr int NLD = NIN / 2;                                 // slots each wave of a pair loads (4)
    constexpr int XIN = ATILE;                                   // [4 pairs][NIN slots][64 lanes] x 16 bytes: the pair's inputs
    constexpr int XCH = XIN + 4 * NIN * 1024;                    // [4 pairs][64 lanes] x 16 bytes: partial dh
    constexpr int LTAB = XCH + 4 * 1024;                         // [NCH][512] LDS byte offsets of the polled chunks
    constexpr int LDS_TRASH = LTAB + NCH * 512 * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, uw = wave & 3;
    const int half = __builtin_amdgcn_readfirstlane(wave >> 2);  // 0: gates f, i of K + the gate math; 1: gates o, g of K
    const int g0 = half * GW;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = LG * H;
    const unsigned TB = (unsigned)T * B;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int ubase = p * 64 + uw * 16;
    const int kq = lane >> 4;
    const int frag_unit = ubase + (lane & 15);
    const bool frag_ok = frag_unit < H;

    // A[m = unit][kidx = (g, j)] = U_g[j][unit], my two gates
    bf16x8 Uf[GW][KSTEPS];
    {
        const unsigned szU = (unsigned)((size_t)LG * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int gg = 0; gg < GW; ++gg)
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                unsigned raw[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = kk * 32 + kq * 8 + e;
                    raw[e] = __builtin_amdgcn_raw_buffer_load_b32(
                        rsU, (frag_ok && j < H) ? (unsigned)((((g0 + gg) * H + j) * H + frag_unit) * 4) : szU, 0, 0);
                }
                bf16x8 f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (short)pk_f2bf(__uint_as_float(raw[e]));
                Uf[gg][kk] = f;
                if (kk % 3 == 2) {  // fenced in groups of 24 loads (see the forward kernel)
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    }
    for (int i = tid; i < (LDS_TRASH + 16) / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- poll descriptors (as rec2l_bwd_kernel): running offsets, LDS destinations in an LDS table
    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Gpitch * 2u;
    const unsigned ndir = (unsigned)(a.R / B);
    const unsigned szGb = ndir * (unsigned)T * TS;
    unsigned goff[NCH], dirbits = 0u;
    int clds[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ci = tid + 512 * i;
        const bool ok = ci < nrows * LG * CPR;
        const int row = ok ? ci / (LG * CPR) : 0;
        const int rem = ok ? ci - row * (LG * CPR) : 0;
        const int g = rem / CPR, col = rem - g * CPR;
        const int n = n_base + row;
        const int dir = n >= B ? 1 : 0, b = n - dir * B;
        goff[i] = (unsigned)dir * (unsigned)T * TS + ((unsigned)b * a.Gpitch + g * Hp + col * 8) * 2u +
                  (unsigned)(dir ? 0 : (T - 1)) * TS;
        dirbits |= (unsigned)dir << i;
        clds[i] = ok ? row * (LDA * 2) + (g * KPAD + col * 8) * 2 : LDS_TRASH;
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
    float msk[4], dc_car[4];
    bool ok4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ok4[r] = r < nv;
        msk[r] = (a.mask != nullptr && ok4[r]) ? a.mask[(long)n * H + u0 + r] : a.mask_scalar;
        dc_car[r] = 0.f;
    }
    const unsigned vY0 = ((unsigned)bb * a.YH + dir * H + u0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)dir * TB + bb) * (LNS * H) + u0), vSs = (unsigned)B * LNS * H;
    const unsigned vG0 = (((unsigned)dir * TB + bb) * GH + u0), vGs = (unsigned)B * GH;
    // ---- publish (first wave): the lanes of the even 16-lane rows store one chunk per gate (my row, 8 units from pu0)
    const int pu0 = ubase + (kq >> 1) * 8;
    const bool pk_ok = (kq & 1) == 0 && row_ok && pu0 < Hp;
    const unsigned pbase = pk_ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)bb * a.Gpitch + pu0) * 2u : szGb;
    unsigned char* xin = smem + XIN + uw * (NIN * 1024) + lane * 16;  // + slot * 1024
    unsigned char* xch = smem + XCH + uw * 1024 + lane * 16;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dGb, szGb);
    float* trash = a.trash + (tid & 63) * 4;

    // saved tensors of one step, one 16-byte load each.  First wave: f, i, o, g; second wave: c, h_{t-1}, dY, c_{t-1}.
    // Running element offsets (oS into S, oY into Y / dY), one time slab per load, up for the backward direction's rows
    // and down for the forward one's.
    const unsigned dS = dir ? vSs : 0u - vSs, dYs = dir ? vYs : 0u - vYs;
    unsigned oS = vS0 + (unsigned)(dir ? 0 : T - 1) * vSs, oY = vY0 + (unsigned)(dir ? 0 : T - 1) * vYs;
    f32x4 iv[NLD];
    auto load_step_e = [&](int t, auto E, auto HALFC) {
        constexpr int EE = decltype(E)::value;
        if constexpr (decltype(HALFC)::value == 0) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) iv[k] = ld4<EE>(a.S, oS + k * H, nv);
        } else {
            const int nvp = t > 0 ? nv : 0;  // step t-1 sits one slab further along (nothing there when t == 0)
            iv[0] = ld4<EE>(a.S, oS + 4 * H, nv);
            iv[1] = ld4<EE>(a.Y, oY + dYs, nvp);
            iv[2] = ld4<EE>(a.dY, oY, nv);
            iv[3] = ld4<EE>(a.S, oS + dS + 4 * H, nvp);
            if (t == 0) {  // h_{-1} = c_{-1} = 0
                iv[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                iv[3] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        oS += dS;
        oY += dYs;
    };
    if (half == 0) {
#define PK_L3S0(E) load_step_e(T - 1, E, BoolC<0>())
        PK_EDGE_DISPATCH(PK_L3S0);
    } else {
#define PK_L3S1(E) load_step_e(T - 1, E, BoolC<1>())
        PK_EDGE_DISPATCH(PK_L3S1);
    }
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto fill_slab = [&](int tt, auto FASTC) {  // my LG chunks of the slab that step tt will publish
        const unsigned off = pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - tt) : tt) * TS : 0u);
#pragma unroll
        for (int g = 0; g < LG; ++g) pub_store<decltype(FASTC)::value != 0>(rs, off + (pk_ok ? (unsigned)(g * Hp) * 2u : 0u), sentinel);
    };
    if (a.self_fill && half == 0) {
        for (int k = 0; k < PK_R2_FILL_AHEAD && k < T; ++k) fill_slab(T - 1 - k, BoolC<0>());
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int* ltab = reinterpret_cast<int*>(smem + LTAB) + tid;
#pragma unroll
    for (int i = 0; i < NCH; ++i) ltab[i * 512] = clds[i];

    bool dead = false;
    const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    auto run = [&](auto HALFC) {
    constexpr int HALF = decltype(HALFC)::value;
    const bool fast = fast_rt;
    int it = 0;
    for (int t = T - 1; t >= 0; --t, ++it) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned char* At = smem;
        // this step's saved tensors (loaded a step ago) go to the pair's slots BEFORE the poll: their registers are free
        // while the poll holds its chunks (the slots were last read before barrier B of the previous step)
#pragma unroll
        for (int k = 0; k < NLD; ++k) *reinterpret_cast<f32x4*>(xin + (HALF * NLD + k) * 1024) = iv[k];
        if (t < T - 1) {
            const int nap = a.poll_delay + (HALF ? a.helper_delay : 0);
            for (int d = 0; d < nap; ++d) __builtin_amdgcn_s_sleep(1);
            int retries = 0;
            dead = fast ? poll_to_lds_tab<NCH, true>(rs, goff, ltab, At, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds_tab<NCH, false>(rs, goff, ltab, At, a.err, a.spin_limit, lane, dead, retries);
        }
        PK_BARRIER_LDS();  // A: the polled dgates_{t+1} tile and the pair's input slots are complete
        if constexpr (HALF == 0) {
            if (a.self_fill && t - PK_R2_FILL_AHEAD >= 0) {
                if (fast) fill_slab(t - PK_R2_FILL_AHEAD, BoolC<1>());
                else fill_slab(t - PK_R2_FILL_AHEAD, BoolC<0>());
            }
        }
        if (t < T - 1) {
            const unsigned char* Ar = At + (lane & 15) * (LDA * 2) + kq * 16;
#pragma unroll
            for (int gg = 0; gg < GW; ++gg)
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const bf16x8 df = *reinterpret_cast<const bf16x8*>(Ar + ((HALF * GW + gg) * KPAD + kk * 32) * 2);
                    if ((kk & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Uf[gg][kk], df, acc0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Uf[gg][kk], df, acc1, 0, 0, 0);
                }
        }
        // saved tensors of the next step: issued BEHIND the MFMA block (their registers are free during it; they have the
        // gate math, the publish and the cluster hand-over to land)
        if (t > 0) {
#define PK_L3SN(E) load_step_e(t - 1, E, HALFC)
            PK_EDGE_DISPATCH(PK_L3SN);
        }
        if constexpr (HALF == 1) *reinterpret_cast<f32x4*>(xch) = acc0 + acc1;
        // the first wave reads the pair's inputs before the barrier: they were complete at barrier A
        f32x4 sin[NIN];
        if constexpr (HALF == 0) {
#pragma unroll
            for (int k = 0; k < NIN; ++k) sin[k] = *reinterpret_cast<const f32x4*>(xin + k * 1024);
        }
        PK_BARRIER_LDS();  // B: the partial sums are handed over, and everybody is done reading the A tile
        if constexpr (HALF == 0) {
            const f32x4 part = *reinterpret_cast<const f32x4*>(xch);
            float dgv[LG][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s[LNS];
#pragma unroll
                for (int k = 0; k < LNS; ++k) s[k] = sin[k][r];
                const float hp = sin[LNS][r], dy = sin[LNS + 1][r], cp = sin[LNS + 2][r];
                const float dh = dy + acc0[r] + acc1[r] + part[r];
                float dg[LG], dhd, dcp;
                pk_cell_bwd<PK_CELL_LSTM>(act, s, hp, cp, msk[r], dh, dc_car[r], dg, dhd, dcp);
                dc_car[r] = ok4[r] ? dcp : 0.f;
#pragma unroll
                for (int g = 0; g < LG; ++g) dgv[g][r] = ok4[r] ? dg[g] : 0.f;
            }
            {
                const unsigned off = pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - t) : t) * TS : 0u);
#pragma unroll
                for (int g = 0; g < LG; ++g) {
                    const u32x4 o = pack_chunk_l(pack2_l(dgv[g][0], dgv[g][1]), pack2_l(dgv[g][2], dgv[g][3]));
                    const unsigned og = off + (pk_ok ? (unsigned)(g * Hp) * 2u : 0u);
                    if (fast) pub_store<true>(rs, og, o);
                    else pub_store<false>(rs, og, o);
                }
            }
            if (a.dP2 != nullptr) {  // fp32 gate gradients (only when the caller wants them): straight from registers
                const unsigned ts = (unsigned)(dir ? (T - 1 - t) : t);
#define PK_L3FO(E)                                                                                                         \
    _Pragma("unroll") for (int g = 0; g < LG; ++g)                                                                         \
        st4<decltype(E)::value>(a.dP2, vG0 + ts * vGs + g * H, nv, trash, f32x4{dgv[g][0], dgv[g][1], dgv[g][2], dgv[g][3]})
                PK_EDGE_DISPATCH(PK_L3FO);
            }
        }
        // next step's poll offsets, behind the publish (the empty asm keeps the nine selects in the loop)
        if (t < T - 1) {
            asm volatile("" : "+v"(dirbits));
#pragma unroll
            for (int i = 0; i < NCH; ++i) goff[i] += ((dirbits >> i) & 1u) ? TS : 0u - TS;
        }
    }
    };
    if (half == 0) run(BoolC<0>());
    else run(BoolC<1>());
}

typedef void (*Rec2Kernel)(R2Args);
Rec2Kernel pick_fwd(int act) {
    return act == PK_ACT_RELU ? rec2l_fwd_kernel<PK_ACT_RELU> : act == PK_ACT_TANH ? rec2l_fwd_kernel<PK_ACT_TANH> : rec2l_fwd_kernel<-1>;
}
Rec2Kernel pick_bwd(int act) {
    return act == PK_ACT_RELU ? rec2l_bwd_kernel<PK_ACT_RELU> : act == PK_ACT_TANH ? rec2l_bwd_kernel<PK_ACT_TANH> : rec2l_bwd_kernel<-1>;
}
Rec2Kernel pick_bwd3(int act) {
    return act == PK_ACT_RELU ? rec3l_bwd_kernel<PK_ACT_RELU> : act == PK_ACT_TANH ? rec3l_bwd_kernel<PK_ACT_TANH> : rec3l_bwd_kernel<-1>;
}
// PK_EXPERIMENT lstm_bwd_gen=2 keeps the second-generation backward kernel (A/B measurements)
inline bool bwd_gen3() {
    static int g = -1;
    if (g < 0) {
        const char* e = pk_experiment("lstm_bwd_gen");
        g = (e && e[0] == '2') ? 2 : 3;
    }
    return g == 3;
}
inline int act_slot(int act) { return act == PK_ACT_RELU ? 0 : act == PK_ACT_TANH ? 1 : 2; }

}  // namespace

// Which LSTM kernels pk_rec_{fwd,bwd}_bf16 launch: 8 = the ones in this file (default), 4 = the four-wave kernels of
// pk_rec_persist2.hip (what every other cell uses).  PK_EXPERIMENT lstm_waves=4 changes the default; pk_persist2_set_lstm_waves
// overrides it at run time (tests/test_gpu_lstm_waves.py compares the two).
namespace {
int g_lstm_waves = 0;  // 0: not decided yet
}
int pk_rec2l_enabled() {
    if (g_lstm_waves == 0) {
        const char* v = pk_experiment("lstm_waves");
        g_lstm_waves = (v && atoi(v) == 4) ? 4 : 8;
    }
    return g_lstm_waves == 8;
}
extern "C" void pk_persist2_set_lstm_waves(int waves) { g_lstm_waves = waves == 8 ? 8 : 4; }
extern "C" int pk_persist2_get_lstm_waves(void) { return pk_rec2l_enabled() ? 8 : 4; }
int pk_rec2l_helper_delay() {
    static int d = -1;
    if (d < 0) {
        const char* v = pk_experiment("lstm_helper_delay");
        d = v ? atoi(v) : 4;
        if (d < 0) d = 0;
    }
    return d;
}

int pk_rec2l_launch(hipStream_t st, R2Args& a, const Plan2& pl, int act, bool backward) {
    a.helper_delay = pk_rec2l_helper_delay();
    size_t lds;
    if (!backward) {
        lds = 2 * (size_t)RMAX * pk_r2_lda_bf16(KPAD) * 2 + NWV * ((size_t)(GW + 1 + 3) * 1024 + 512) + 4 * GW * 1024 + 16;
    } else {
        lds = (size_t)RMAX * pk_r2_lda_bf16(LG * KPAD) * 2 + 4 * ((size_t)(LNS + 3 + LG) * 1024 + LG * 512) + 4 * 1024 +
              (size_t)((RMAX * LG * (KPAD / 8) + 511) / 512) * 512 * 4 + 16;
    }
    const bool g3 = backward && bwd_gen3();
    if (g3)  // A tile | the pairs' input slots | partial sums | LDS table | trash
        lds = (size_t)RMAX * pk_r2_lda_bf16(LG * KPAD) * 2 + 4 * (size_t)(LNS + 3) * 1024 + 4 * 1024 +
              (size_t)((RMAX * LG * (KPAD / 8) + 511) / 512) * 512 * 4 + 32;
    Rec2Kernel k = backward ? (g3 ? pick_bwd3(act) : pick_bwd(act)) : pick_fwd(act);
    {
        static size_t granted[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // hipFuncSetAttribute is slow (milliseconds): once per kernel and size
        size_t& g = granted[backward ? (g3 ? 2 : 1) : 0][act_slot(act)];
        if (g < lds) {
            PK_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            g = lds;
        }
    }
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        int rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        dim3 grid(pl.C * pl.Pn), block(512);
        rc = pk_rec2_check_residency((const void*)k, 512, lds, pl.C * pl.Pn, "pk_rec_*_bf16 (LSTM, eight waves)");
        if (rc) return rc;
        const int help = pk_rec_helper_wanted(backward, pl.launches, PK_CELL_LSTM);
        if (help && (rc = pk_rec_helper_fork(st)) != 0) return rc;
        hipLaunchKernelGGL(k, grid, block, lds, st, a);
        PK_LAUNCH_CHECK();
        if (help && (rc = pk_rec_helper_launch(st, a, pl, 4, pk_cell_saved(PK_CELL_LSTM), backward, true, help)) != 0) return rc;
    }
    return 0;
}
