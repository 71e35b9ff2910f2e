// pk_rec_persist3.hip - third generation of the perf-mode (bf16 MFMA operands) persistent
// recurrent time loops for liGRU / RNN: the MFMA operands are SWAPPED.
//
// Same cluster / exchange protocol as pk_rec_persist2.hip (read its header first): clusters of Pn
// workgroups own 16 rows (sequences of both directions), a wave owns 16 hidden units and keeps its
// slice of the recurrent matrix in registers for all T steps, h_t (forward) / dgates_t (backward) is
// exchanged through L2 in bf16, 16 bytes at a time, the data being the flag.  Replaces the reference's
// python time loops and their autograd (neural_networks.py:1130-1141 liGRU, :1438-1447 RNN).
//
// What changed.  The second generation multiplies  D[row][unit] = h[row][:] . U[unit][:]  with the
// polled h tile as the MFMA A operand and the U slice as B.  The C/D layout of
// v_mfma_f32_16x16x32_bf16 then gives a lane FOUR ROWS of ONE unit, which is the transpose of every
// tensor the step touches in HBM (P, Y, S, dY, the exchange buffers: unit-contiguous rows): each
// step moved its projections, outputs, saved gates and the bf16 publish chunk through wave-private
// LDS "patches" - 26 LDS instructions and two lgkmcnt drains per step on the dependency chain
// (~550 of 5 200 clocks, profiles/r02_rec_step_floor.json).  Here the SAME registers are passed in
// the other order:  D^T[unit][row] = U[unit][:] . h[row][:]  (the A-operand layout of the U slice is
// the B-operand layout it already had, and vice versa for the h fragments read from LDS), and the
// C/D layout gives a lane FOUR CONSECUTIVE UNITS of ONE row (row = lane & 15, units 4*(lane >> 4)..+3):
//   * P / S / Y / dY are read and written with ONE 16-byte access per lane straight from / to
//     registers (no patchP / patchY / patchS / patchI / patchG);
//   * the 8-unit bf16 publish chunk is assembled from two lanes 16 apart with two
//     v_permlane16_swap_b32 (gfx950) - no patchB, no LDS at all behind the MFMA block;
//   * the fp32 outputs of step t stay in registers until the poll of step t+1 has landed (same
//     place in the step as before: they must not be young stores in front of the next poll's
//     vmcnt(0)).
// LDS holds the two A tiles only (42 KB forward, 84 KB backward).
#include <stdlib.h>

#include "pk_rec2_common.h"

namespace {

// The 16-byte publish chunk (8 consecutive units of one row) from the 8 bytes (4 units) every lane
// holds: lanes l and l + 16 (kq even / odd) own the two halves.  v_permlane16_swap_b32 swaps the odd
// 16-lane rows of its first operand with the even rows of its second; with both operands = x the
// pair (first, second) reads (x of lane l, x of lane l + 16) in the lanes of rows 0 and 2.
__device__ __forceinline__ u32x4 pack_chunk(unsigned lo, unsigned hi) {
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return u32x4{a[0], b[0], a[1], b[1]};
}
// transposer patch of one fp32 tensor tile [16 rows][16 units], row pitch 20 floats (80 B: the eight lanes one
// ds_write_b128 phase serves land on eight different 16-byte bank groups)
constexpr int PK3_PROW = 20, PK3_PATCH_F = 16 * PK3_PROW;
__device__ __forceinline__ unsigned pack2(float x, float y) {
    return (unsigned)to_bf_pub(x) | ((unsigned)to_bf_pub(y) << 16);
}

// ============================================================================
// forward
// ============================================================================
template <int CELL, int ACT, bool TR, bool COAL>
__global__ __launch_bounds__(256, 1) void rec3_fwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr int LDA = pk_r2_lda_bf16(KPAD);
    constexpr int ATILE = RMAX * LDA * 2;
    constexpr int NCH = (RMAX * (KPAD / 8) + 255) / 256;
    constexpr int LDS_TRASH = 2 * ATILE;
    constexpr int NPATCH = G + 1 + NS;  // COAL: P of every gate | Y | S slots, one transposer patch each per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][ATILE] | trash | COAL: 4 x NPATCH patches

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int ubase = p * 64 + wave * 16;
    const int kq = lane >> 4;
    const int frag_unit = ubase + (lane & 15);  // the unit whose U row this lane holds as MFMA A fragments
    const bool frag_ok = frag_unit < H;

    // ---- recurrent weights of my 16 units -> registers (once): A[m = unit][k] = U_g[unit][k]
    bf16x8 Uf[G][KSTEPS];
    {
        const unsigned szU = (unsigned)((size_t)G * H * H * 4);
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            u32x4 raw[KSTEPS][2];
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const int k0 = kk * 32 + kq * 8;
                const unsigned off = (unsigned)(((g * H + frag_unit) * H + k0) * 4);
                raw[kk][0] = __builtin_amdgcn_raw_buffer_load_b128(rsU, (frag_ok && k0 < H) ? off : szU, 0, 0);
                raw[kk][1] = __builtin_amdgcn_raw_buffer_load_b128(rsU, (frag_ok && k0 + 4 < H) ? off + 16 : szU, 0, 0);
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
                Uf[g][kk] = f;
            }
        }
    }
    for (int i = tid; i < (LDS_TRASH + 16) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // ---- poll descriptors (as in the second generation): chunk ci = (row, col) of the cluster's [nrows][Hp/8] block
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
    // ---- my (row, 4 units) of the step: row = lane & 15, units u0 .. u0 + 3
    const int row = lane & 15, u0 = ubase + kq * 4;
    const bool row_ok = row < nrows;
    const int n = n_base + (row_ok ? row : 0);
    const int dir = n >= B ? 1 : 0, bb = n - dir * B;
    int nv = H - u0;
    nv = nv > 4 ? 4 : (nv < 0 ? 0 : nv);
    // wave-uniform: 0 = my 16 units do not straddle H, 1 = they do and H is even, 2 = H is odd
    const int edge = __any(nv > 0 && nv < 4) != 0 ? ((H & 1) ? 2 : 1) : 0;
    nv = row_ok ? nv : 0;
    float psc[G][4], psh[G][4], msk[4], hprev[4], cprev[4];
    bool ok4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ok4[r] = r < nv;
        const int u = ok4[r] ? u0 + r : 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            psc[g][r] = ok4[r] ? a.pscale[g * H + u] : 0.f;
            psh[g][r] = ok4[r] ? a.pshift[g * H + u] : 0.f;
        }
        msk[r] = (a.mask != nullptr && ok4[r]) ? a.mask[(long)n * H + u] : a.mask_scalar;
        hprev[r] = 0.f;
        cprev[r] = 0.f;
    }
    // ---- HBM access layout of the fp32 tensors (P, Y, S).  COAL: the "vector" layout (row = lane >> 2, four
    // adjacent lanes cover 64 contiguous bytes of that row): every wave instruction touches 16 x 64-byte pieces.  The
    // gate layout itself (row = lane & 15) puts the 16 lanes of a 16-lane row on 16 DIFFERENT tensor rows: 64 separate
    // 16-byte requests per instruction - four times the address-unit time of the step's 5-6 such instructions, in
    // front of the publish store (measured: 19.9 vs 18.3 ms per training step against the second generation).  The
    // two layouts are exchanged through wave-private LDS patches with ONE 16-byte access per lane on either side.
    const int arow = COAL ? (lane >> 2) : row, au0 = ubase + (COAL ? (lane & 3) : kq) * 4;
    const int an = n_base + (arow < nrows ? arow : 0);
    const int adir = an >= B ? 1 : 0, ab = an - adir * B;
    int anv = H - au0;
    anv = anv > 4 ? 4 : (anv < 0 ? 0 : anv);
    anv = arow < nrows ? anv : 0;
    // element offsets of my 4 units at storage time 0 / per unit of storage time, for P, Y and S
    const unsigned vP0 = ((unsigned)ab * GH + au0), vPs = (unsigned)B * GH;
    const unsigned vY0 = ((unsigned)ab * a.YH + adir * H + au0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)adir * T * B + ab) * (NS * H) + au0), vSs = (unsigned)B * NS * H;
    // ---- publish: the lanes of the even 16-lane rows store one 16-byte chunk (my row, 8 units from pu0)
    const int pu0 = ubase + (kq >> 1) * 8;
    const bool pk_ok = (kq & 1) == 0 && row_ok && pu0 < Hp;
    const unsigned pbase = pk_ok ? ((unsigned)bb * a.Ypitch + dir * Hp + pu0) * 2u : szYb;  // out of range: dropped
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.Yb, szYb);
    float* trash = a.trash + (tid & 63) * 4;
    // transposer patches: [16 rows][PK3_PROW floats]; gate side: (lane & 15, kq), access side: (lane >> 2, lane & 3)
    float* wp = reinterpret_cast<float*>(smem + 2 * ATILE + 32) + wave * (NPATCH * PK3_PATCH_F);
    float* wp_g = wp + row * PK3_PROW + kq * 4;
    float* wp_a = wp + (lane >> 2) * PK3_PROW + (lane & 3) * 4;

    f32x4 pv[G], pnext[G];        // projections of this step (gate layout) / of the next one (access layout, a step ahead)
    f32x4 yout, sout[NS];         // !COAL: fp32 outputs of the previous step, stored behind this step's poll
    auto load_proj = [&](f32x4 (&dst)[G], int tt, auto E) {
        const unsigned ts = (unsigned)(adir ? (T - 1 - tt) : tt);
#pragma unroll
        for (int g = 0; g < G; ++g) dst[g] = ld4<decltype(E)::value>(a.P, vP0 + ts * vPs + g * H, anv);
    };
    auto flush_outputs = [&](int tt, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(adir ? (T - 1 - tt) : tt);
        if (COAL) {
            yout = *reinterpret_cast<const f32x4*>(wp_a + G * PK3_PATCH_F);
#pragma unroll
            for (int k = 0; k < NS; ++k) sout[k] = *reinterpret_cast<const f32x4*>(wp_a + (G + 1 + k) * PK3_PATCH_F);
        }
        st4<EE>(a.Y, vY0 + ts * vYs, anv, trash, yout);
#pragma unroll
        for (int k = 0; k < NS; ++k) st4<EE>(a.S, vS0 + ts * vSs + k * H, anv, trash, sout[k]);
    };
#define PK3_LP0(E) load_proj(pnext, 0, E)
    PK_EDGE_DISPATCH(PK3_LP0);
#pragma unroll
    for (int g = 0; g < G; ++g) pv[g] = pnext[g];
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (a.self_fill) {  // my chunks of the first slabs, visible everywhere before the handshake lets anyone poll
        for (int tt = 0; tt < PK_R2_FILL_AHEAD && tt < T; ++tt)
            pub_store<false>(rs, pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - tt) : tt) * TS : 0u), sentinel);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bool dead = false;
    const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    const bool flush_late = __builtin_amdgcn_readfirstlane(a.flush_late) != 0;
    // Nothing may be in flight when the time loop is entered.  The compiler's wait-count pass merges the state of the
    // loop entry (the set-up loads above) with the state of the back edge; a load that is still pending on ONE of the
    // two paths becomes a counted vmcnt wait inside the loop - with the count of the entry path - and on the back-edge
    // path that count makes the first MFMAs wait for the step's output stores (2 000 clocks per step, measured).  The
    // builtin form is visible to that pass (inline asm is not).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    auto run = [&](auto FASTC, auto SEC) {
    constexpr bool fast = decltype(FASTC)::value != 0;
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
            for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
            int retries = 0;
            dead = fast ? poll_to_lds<NCH, true>(rs, goff, clds, At, a.err, a.spin_limit, lane, dead, retries)
                        : poll_to_lds<NCH, false>(rs, goff, clds, At, a.err, a.spin_limit, lane, dead, retries);
            if (a.trace != nullptr && blockIdx.x == 0 && tid == 0) a.trace[(long)step_idx * 8 + 6] = (unsigned long long)retries;
        }
        PK_TRACE(1);
        if (COAL) {  // this step's projections (loaded a step ago, access layout) -> my patches, next to the A-tile stores
#pragma unroll
            for (int g = 0; g < G; ++g) *reinterpret_cast<f32x4*>(wp_a + g * PK3_PATCH_F) = pnext[g];
        }
        if (t > 0) PK_BARRIER_LDS();
        else if (COAL) PK_LDS_ORDER();
        PK_TRACE(2);
        // the h fragments come from LDS PKD k-steps ahead of the MFMAs that use them (a rolling prefetch pinned with
        // scheduling groups: left alone the compiler reads two fragments, waits, multiplies, and re-uses the same two
        // registers - nine exposed LDS round trips per step); the first PKD reads go out right behind the barrier
        const bool empty = TR && a.empty_step != 0;
        const bool mm = t > 0 && !empty;
        const unsigned char* Ar = At + (lane & 15) * (LDA * 2) + kq * 16;
        constexpr int PKD = 4;
        bf16x8 hf[PKD];
        if (mm) {
#pragma unroll
            for (int kk = 0; kk < PKD; ++kk) hf[kk] = *reinterpret_cast<const bf16x8*>(Ar + kk * 64);
        }
        if (COAL) {
#pragma unroll
            for (int g = 0; g < G; ++g) pv[g] = *reinterpret_cast<const f32x4*>(wp_g + g * PK3_PATCH_F);
        }
        // off the dependency chain: fp32 outputs of the previous step, projections of the next one, the fill pattern
        // PK_R2_FILL_AHEAD steps ahead - in front of the MFMA block, or (flush_late) behind it
        auto side_traffic = [&]() {
            // LOADS FIRST: vmcnt counts loads and stores in issue order, so a wait for one of these loads (the compiler
            // places one wherever it cannot prove the destination registers idle) must not have the output stores - and
            // their ~1 us HBM acknowledge - in front of it
            if (t + 1 < T) {
#define PK3_LP1(E) load_proj(pnext, t + 1, E)
                PK_EDGE_DISPATCH_S(PK3_LP1);
            }
            if (t > 0) {
#define PK3_FO(E) flush_outputs(t - 1, E)
                PK_EDGE_DISPATCH_S(PK3_FO);
            }
            if (a.self_fill && t + PK_R2_FILL_AHEAD < T) {
                const unsigned off = pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - (t + PK_R2_FILL_AHEAD)) : (t + PK_R2_FILL_AHEAD)) * TS : 0u);
                if (fast) pub_store<true>(rs, off, sentinel);
                else pub_store<false>(rs, off, sentinel);
            }
        };
        if (!flush_late) side_traffic();
        __builtin_amdgcn_sched_barrier(0);
        if (mm) {
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                const bf16x8 cur = hf[kk % PKD];
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Uf[g][kk], cur, acc[g], 0, 0, 0);
                if (kk + PKD < KSTEPS) hf[kk % PKD] = *reinterpret_cast<const bf16x8*>(Ar + (kk + PKD) * 64);
            }
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
                __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
                if (kk + PKD < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (flush_late) side_traffic();
        PK_TRACE(3);
        // ---- gate math for my row, units u0 .. u0 + 3 (acc[g][r]: unit u0 + r)
        float hv[4], sv[NS][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float pr[G];
#pragma unroll
            for (int g = 0; g < G; ++g) pr[g] = pv[g][r] * psc[g][r] + psh[g][r] + acc[g][r];
            float h, cc, s[NS];
            if (empty) {
                h = 0.25f;
                cc = 0.f;
#pragma unroll
                for (int k = 0; k < NS; ++k) s[k] = pr[0];
            } else {
                pk_cell_fwd<CELL>(act, pr, hprev[r], cprev[r], msk[r], h, cc, s);
            }
            h = ok4[r] ? h : 0.f;  // rows / units outside the layer carry exact zeros (published as padding)
            cc = ok4[r] ? cc : 0.f;
            hprev[r] = h;
            cprev[r] = cc;
            hv[r] = h;
#pragma unroll
            for (int k = 0; k < NS; ++k) sv[k][r] = s[k];
        }
        PK_TRACE(4);
        // ---- publish h_t: what the other workgroups of the cluster wait for
        {
            const u32x4 o = pack_chunk(pack2(hv[0], hv[1]), pack2(hv[2], hv[3]));
            const unsigned off = pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - t) : t) * TS : 0u);
            if (fast) pub_store<true>(rs, off, o);
            else pub_store<false>(rs, off, o);
        }
        // The registers the output stores of this step took their data from (flush_outputs) stay untouched until here:
        // the compiler guards a store's data registers with a vmcnt wait in front of whatever overwrites them, and with
        // the registers recycled for the first MFMA fragments that wait put the stores' ~1 us HBM acknowledge in front
        // of the MFMA block (MFMA phase 1 450 -> 3 300 clocks, profiles/r03_rec_generations.json).
        if (COAL) {
            asm volatile("" ::"v"(yout));
#pragma unroll
            for (int k = 0; k < NS; ++k) asm volatile("" ::"v"(sout[k]));
        }
        // ---- the fp32 outputs wait (COAL: in my patches; else in registers) until the next step's poll has landed
        if (COAL) {
            *reinterpret_cast<f32x4*>(wp_g + G * PK3_PATCH_F) = f32x4{hv[0], hv[1], hv[2], hv[3]};
#pragma unroll
            for (int k = 0; k < NS; ++k)
                *reinterpret_cast<f32x4*>(wp_g + (G + 1 + k) * PK3_PATCH_F) = f32x4{sv[k][0], sv[k][1], sv[k][2], sv[k][3]};
        } else {
            yout = f32x4{hv[0], hv[1], hv[2], hv[3]};
#pragma unroll
            for (int k = 0; k < NS; ++k) sout[k] = f32x4{sv[k][0], sv[k][1], sv[k][2], sv[k][3]};
#pragma unroll
            for (int g = 0; g < G; ++g) pv[g] = pnext[g];
        }
        PK_TRACE(5);
    }
    };
    PK_RUN_SPECIALISED(run, fast_rt);
    if (COAL) PK_LDS_ORDER();
#define PK3_FOL(E) flush_outputs(T - 1, E)
    PK_EDGE_DISPATCH(PK3_FOL);
}

// ============================================================================
// backward: dL/dh_{t-1} = direct + [dgates_t] . [U_0; U_1; ...]
// ============================================================================
template <int CELL, int ACT, bool TR, bool COAL>
__global__ __launch_bounds__(256, 1) void rec3_bwd_kernel(R2Args a) {
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr int LDA = pk_r2_lda_bf16(G * KPAD);
    constexpr int ATILE = RMAX * LDA * 2;
    constexpr int NCH = (RMAX * G * (KPAD / 8) + 255) / 256;
    constexpr int NIN = NS + 2;                              // saved gates, h_{t-1}, dY
    constexpr int LDS_TRASH = 2 * ATILE;
    constexpr int NPATCH = NIN + G;                          // COAL: transposer patches of the inputs | fp32 gate gradients
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
    const int kq = lane >> 4;
    const int frag_unit = ubase + (lane & 15);
    const bool frag_ok = frag_unit < H;

    // A[m = unit][kidx = (g, j)] = U_g[j][unit]
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

    // ---- poll descriptors: chunk ci = (row, gate, col) of the cluster's dgates_{t+1} block
    const int CPR = Hp >> 3;
    const unsigned TS = (unsigned)B * a.Gpitch * 2u;
    const unsigned ndir = (unsigned)(a.R / B);
    const unsigned szGb = ndir * (unsigned)T * TS;
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
        cbase[i] = ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)b * a.Gpitch + g * Hp + col * 8) * 2u +
                            (unsigned)(dir ? 0 : (T - 1)) * TS
                      : szGb;
        cstep[i] = ok ? (dir ? TS : 0u - TS) : 0u;
        clds[i] = ok ? row * (LDA * 2) + (g * KPAD + col * 8) * 2 : LDS_TRASH;
    }
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
    // HBM access layout of the fp32 tensors (see the forward kernel): COAL = row lane >> 2, four adjacent lanes per 64 bytes
    const int arow = COAL ? (lane >> 2) : row, au0 = ubase + (COAL ? (lane & 3) : kq) * 4;
    const int an = n_base + (arow < nrows ? arow : 0);
    const int adir = an >= B ? 1 : 0, ab = an - adir * B;
    int anv = H - au0;
    anv = anv > 4 ? 4 : (anv < 0 ? 0 : anv);
    anv = arow < nrows ? anv : 0;
    const unsigned vY0 = ((unsigned)ab * a.YH + adir * H + au0), vYs = (unsigned)B * a.YH;
    const unsigned vS0 = (((unsigned)adir * TB + ab) * (NS * H) + au0), vSs = (unsigned)B * NS * H;
    const unsigned vG0 = (((unsigned)adir * TB + ab) * GH + au0), vGs = (unsigned)B * GH;
    const int pu0 = ubase + (kq >> 1) * 8;
    const bool pk_ok = (kq & 1) == 0 && row_ok && pu0 < Hp;
    const unsigned pbase = pk_ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)bb * a.Gpitch + pu0) * 2u : szGb;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dGb, szGb);
    float* trash = a.trash + (tid & 63) * 4;
    float* wp = reinterpret_cast<float*>(smem + 2 * ATILE + 32) + wave * (NPATCH * PK3_PATCH_F);
    float* wp_g = wp + row * PK3_PROW + kq * 4;
    float* wp_a = wp + (lane >> 2) * PK3_PROW + (lane & 3) * 4;

    // saved tensors of a step, one 16-byte access each: [0..NS) gates, NS = h_{t-1}, NS+1 = dY
    f32x4 iv[NIN], inext[NIN];  // this step (gate layout) / the next one (access layout, loaded a step ahead)
    auto load_step_e = [&](f32x4 (&dst)[NIN], int t, auto E) {
        constexpr int EE = decltype(E)::value;
        const unsigned ts = (unsigned)(adir ? (T - 1 - t) : t);
        const unsigned tp = t > 0 ? (adir ? ts + 1 : ts - 1) : ts;  // storage time of step t-1 (any valid row when t == 0)
        const int nvp = t > 0 ? anv : 0;
#pragma unroll
        for (int k = 0; k < NS; ++k) dst[k] = ld4<EE>(a.S, vS0 + ts * vSs + k * H, anv);
        dst[NS] = ld4<EE>(a.Y, vY0 + tp * vYs, nvp);
        dst[NS + 1] = ld4<EE>(a.dY, vY0 + ts * vYs, anv);
        if (t == 0) dst[NS] = f32x4{0.f, 0.f, 0.f, 0.f};  // h_{-1} = 0
    };
    f32x4 gout[G];  // fp32 gate gradients of the previous step (only when the caller wants them: dP2 != null)
    auto flush_outputs_e = [&](int tt, auto E) {
        const unsigned ts = (unsigned)(adir ? (T - 1 - tt) : tt);
        if (COAL) {
#pragma unroll
            for (int g = 0; g < G; ++g) gout[g] = *reinterpret_cast<const f32x4*>(wp_a + (NIN + g) * PK3_PATCH_F);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) st4<decltype(E)::value>(a.dP2, vG0 + ts * vGs + g * H, anv, trash, gout[g]);
    };
#define PK3_LS0(E) load_step_e(inext, T - 1, E)
    PK_EDGE_DISPATCH(PK3_LS0);
#pragma unroll
    for (int k = 0; k < NIN; ++k) iv[k] = inext[k];
#pragma unroll
    for (int g = 0; g < G; ++g) gout[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    auto fill_slab = [&](int tt, auto FASTC) {
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
    const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
    const bool flush_late = __builtin_amdgcn_readfirstlane(a.flush_late) != 0;
    // Nothing may be in flight when the time loop is entered.  The compiler's wait-count pass merges the state of the
    // loop entry (the set-up loads above) with the state of the back edge; a load that is still pending on ONE of the
    // two paths becomes a counted vmcnt wait inside the loop - with the count of the entry path - and on the back-edge
    // path that count makes the first MFMAs wait for the step's output stores (2 000 clocks per step, measured).  The
    // builtin form is visible to that pass (inline asm is not).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    auto run = [&](auto FASTC, auto SEC) {
    constexpr bool fast = decltype(FASTC)::value != 0;
    constexpr int SE = decltype(SEC)::value;
    int it = 0;
    for (int t = T - 1; t >= 0; --t, ++it) {
        const int step_idx = it;
        PK_TRACE(0);
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned char* At = smem + (it & 1) * ATILE;
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
        if (COAL) {  // this step's saved tensors (loaded a step ago, access layout) -> my patches
#pragma unroll
            for (int k = 0; k < NIN; ++k) *reinterpret_cast<f32x4*>(wp_a + k * PK3_PATCH_F) = inext[k];
        }
        if (t < T - 1) PK_BARRIER_LDS();
        else if (COAL) PK_LDS_ORDER();
        PK_TRACE(2);
        // rolling prefetch of the dgate fragments, PKD fragments ahead of their MFMA (see the forward kernel)
        const bool empty = TR && a.empty_step != 0;
        const bool mm = t < T - 1 && !empty;
        const unsigned char* Ar = At + (lane & 15) * (LDA * 2) + kq * 16;
        constexpr int PKD = 4, NF = G * KSTEPS;
        bf16x8 df[PKD];
        if (mm) {
#pragma unroll
            for (int f = 0; f < PKD; ++f) df[f] = *reinterpret_cast<const bf16x8*>(Ar + ((f / KSTEPS) * KPAD + (f % KSTEPS) * 32) * 2);
        }
        if (COAL) {
#pragma unroll
            for (int k = 0; k < NIN; ++k) iv[k] = *reinterpret_cast<const f32x4*>(wp_g + k * PK3_PATCH_F);
        }
        // off the dependency chain: fp32 gate gradients of the previous step (if wanted), the saved tensors of the next
        // one, the fill pattern ahead - in front of the MFMA block, or (flush_late) behind it
        auto side_traffic = [&]() {
            if (t > 0) {  // (loads first: see the forward kernel)
#define PK3_LS1(E) load_step_e(inext, t - 1, E)
                PK_EDGE_DISPATCH_S(PK3_LS1);
            }
            if (t < T - 1 && a.dP2 != nullptr) {
#define PK3_FOB(E) flush_outputs_e(t + 1, E)
                PK_EDGE_DISPATCH_S(PK3_FOB);
            }
            if (a.self_fill && t - PK_R2_FILL_AHEAD >= 0) {
                if (fast) fill_slab(t - PK_R2_FILL_AHEAD, BoolC<1>());
                else fill_slab(t - PK_R2_FILL_AHEAD, BoolC<0>());
            }
        };
        if (!flush_late) side_traffic();
        __builtin_amdgcn_sched_barrier(0);
        if (mm) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int g = f / KSTEPS, kk = f % KSTEPS;
                const bf16x8 cur = df[f % PKD];
                if ((f & 1) == 0) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Uf[g][kk], cur, acc0, 0, 0, 0);
                else acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Uf[g][kk], cur, acc1, 0, 0, 0);
                if (f + PKD < NF) {
                    const int f2 = f + PKD;
                    df[f % PKD] = *reinterpret_cast<const bf16x8*>(Ar + ((f2 / KSTEPS) * KPAD + (f2 % KSTEPS) * 32) * 2);
                }
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (f + PKD < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (flush_late) side_traffic();
        PK_TRACE(3);
        float dgv[G][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) s[k] = iv[k][r];
            const float hp = iv[NS][r], dy = iv[NS + 1][r];
            const float dh = dy + dh_dir[r] + acc0[r] + acc1[r];
            float dg[G], dhd, dcp, dc0 = 0.f;
            if (empty) {
                dhd = 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) dg[g] = 0.125f;
            } else {
                pk_cell_bwd<CELL>(act, s, hp, 0.f, msk[r], dh, dc0, dg, dhd, dcp);
            }
            // rows / units outside the layer: exact zeros (select, not multiply: their inputs are arbitrary)
            dh_dir[r] = ok4[r] ? dhd : 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) dgv[g][r] = ok4[r] ? dg[g] : 0.f;
        }
        PK_TRACE(4);
        {
            const unsigned off = pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - t) : t) * TS : 0u);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const u32x4 o = pack_chunk(pack2(dgv[g][0], dgv[g][1]), pack2(dgv[g][2], dgv[g][3]));
                const unsigned og = off + (pk_ok ? (unsigned)(g * Hp) * 2u : 0u);
                if (fast) pub_store<true>(rs, og, o);
                else pub_store<false>(rs, og, o);
            }
        }
        if (COAL) {
            if (a.dP2 != nullptr) {
#pragma unroll
                for (int g = 0; g < G; ++g)
                    *reinterpret_cast<f32x4*>(wp_g + (NIN + g) * PK3_PATCH_F) = f32x4{dgv[g][0], dgv[g][1], dgv[g][2], dgv[g][3]};
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) gout[g] = f32x4{dgv[g][0], dgv[g][1], dgv[g][2], dgv[g][3]};
#pragma unroll
            for (int k = 0; k < NIN; ++k) iv[k] = inext[k];
        }
        PK_TRACE(5);
    }
    };
    PK_RUN_SPECIALISED(run, fast_rt);
    if (COAL) PK_LDS_ORDER();
    if (a.dP2 != nullptr) {
#define PK3_FOBL(E) flush_outputs_e(0, E)
        PK_EDGE_DISPATCH(PK3_FOBL);
    }
}

typedef void (*Rec3Kernel)(R2Args);
template <int CELL, bool COAL>
Rec3Kernel pick3_fwd(int act, bool tr) {
    if (tr) return rec3_fwd_kernel<CELL, PK_ACT_RELU, true, COAL>;
    return act == PK_ACT_RELU ? rec3_fwd_kernel<CELL, PK_ACT_RELU, false, COAL>
         : act == PK_ACT_TANH ? rec3_fwd_kernel<CELL, PK_ACT_TANH, false, COAL> : rec3_fwd_kernel<CELL, -1, false, COAL>;
}
template <int CELL, bool COAL>
Rec3Kernel pick3_bwd(int act, bool tr) {
    if (tr) return rec3_bwd_kernel<CELL, PK_ACT_RELU, true, COAL>;
    return act == PK_ACT_RELU ? rec3_bwd_kernel<CELL, PK_ACT_RELU, false, COAL>
         : act == PK_ACT_TANH ? rec3_bwd_kernel<CELL, PK_ACT_TANH, false, COAL> : rec3_bwd_kernel<CELL, -1, false, COAL>;
}
int g3_gen[2] = {-1, -1};  // per pass (forward, backward): 2 = second generation; 3 = third, HBM accesses through transposer patches; 4 = third, direct

}  // namespace

// Which generation runs a pass of this cell?  Measured at the benchmarked geometry (profiles/r03_rec_generations.json,
// shader clocks per step, second generation / third with patches / third direct): forward 5 213 / 6 666 / 6 727,
// backward 6 243 / 6 820 / 5 517.  The backward pass only LOADS fp32 tensors besides its exchange (S, h_{t-1}, dY): taken
// straight into the gate layout they cost nothing on the dependency chain and the whole patch traffic disappears
// (-12 %).  The forward pass also STORES three fp32 tensors per step (Y, S); in the third-generation loop those stores
// end up in front of a counted vmcnt wait inside the MFMA block and their ~1 us HBM acknowledge lands on the dependency
// chain (MFMA phase 1 450 -> 3 300 clocks; issued behind the MFMA block instead they delay the next poll by as much),
// so forward stays on the second generation.  PK_EXPERIMENT rec_gen=2 / 3 / 4 forces one generation for both passes,
// PK_EXPERIMENT rec_gen_fwd / PK_EXPERIMENT rec_gen_bwd for one pass (A/B measurements).
int pk_rec3_covers(int cell, int backward) {
    if (g3_gen[0] < 0) {
        const char* both = pk_experiment("rec_gen");
        const char* ef = pk_experiment("rec_gen_fwd");
        const char* eb = pk_experiment("rec_gen_bwd");
        auto parse = [](const char* e, int dflt) { return (e && e[0] >= '2' && e[0] <= '4') ? e[0] - '0' : dflt; };
        g3_gen[0] = parse(ef, parse(both, 2));
        g3_gen[1] = parse(eb, parse(both, 4));
    }
    return g3_gen[backward ? 1 : 0] != 2 && (cell == PK_CELL_LIGRU || cell == PK_CELL_RNN);
}

// Launch loop of the third-generation kernels; `a` and `pl` are prepared by pk_rec_fwd_bf16 / pk_rec_bwd_bf16
// (pk_rec_persist2.hip).  traced: the phase-trace instantiation (Li-GRU / relu only).
int pk_rec3_launch(hipStream_t st, R2Args& a, const Plan2& pl, int cell, int act, bool backward, bool traced) {
    const int G = pk_cell_gates(cell), NS = pk_cell_saved(cell);
    const bool coal = g3_gen[backward ? 1 : 0] != 4;
    const size_t atile = (size_t)RMAX * pk_r2_lda_bf16(backward ? G * KPAD : KPAD) * 2;
    const int npatch = backward ? NS + 2 + G : G + 1 + NS;
    const size_t lds = 2 * atile + 32 + (coal ? (size_t)4 * npatch * PK3_PATCH_F * 4 : 0);
    Rec3Kernel k;
    if (cell == PK_CELL_LIGRU) {
        if (coal) k = backward ? pick3_bwd<PK_CELL_LIGRU, true>(act, traced) : pick3_fwd<PK_CELL_LIGRU, true>(act, traced);
        else k = backward ? pick3_bwd<PK_CELL_LIGRU, false>(act, traced) : pick3_fwd<PK_CELL_LIGRU, false>(act, traced);
    } else {
        if (coal) k = backward ? pick3_bwd<PK_CELL_RNN, true>(act, false) : pick3_fwd<PK_CELL_RNN, true>(act, false);
        else k = backward ? pick3_bwd<PK_CELL_RNN, false>(act, false) : pick3_fwd<PK_CELL_RNN, false>(act, false);
    }
    {   // dynamic LDS above the 64 KB default needs the opt-in; hipFuncSetAttribute is slow: once per kernel
        static const void* granted[64];
        static int n_granted = 0;
        bool have = false;
        for (int i = 0; i < n_granted; ++i) have = have || granted[i] == (const void*)k;
        if (!have) {
            PK_CHECK_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (n_granted < 64) granted[n_granted++] = (const void*)k;
        }
    }
    for (int l = 0; l < pl.launches; ++l) {
        a.row0 = l * pl.C * pl.rpc;
        int rc = pk_rec2_reset_handshake(st, a);
        if (rc) return rc;
        rc = pk_rec2_check_residency((const void*)k, 256, lds, pl.C * pl.Pn, backward ? "pk_rec_bwd_bf16" : "pk_rec_fwd_bf16");
        if (rc) return rc;
        const int help = pk_rec_helper_wanted(backward, pl.launches, cell);
        if (help && (rc = pk_rec_helper_fork(st)) != 0) return rc;
        hipLaunchKernelGGL(k, dim3(pl.C * pl.Pn), dim3(256), lds, st, a);
        PK_LAUNCH_CHECK();
        if (help && (rc = pk_rec_helper_launch(st, a, pl, G, NS, backward, true, help)) != 0) return rc;
    }
    return 0;
}
