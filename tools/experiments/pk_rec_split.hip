// pk_rec_split.hip - fifth generation of the perf-mode (bf16 MFMA operands) persistent recurrent
// time loops for liGRU / RNN: the waves of a workgroup are SPECIALISED.
//
// Same cluster / exchange protocol as pk_rec_persist2.hip / pk_rec_persist3.hip (read their headers
// first): clusters of Pn workgroups own 16 rows (sequences of both directions), a wave owns 16 hidden
// units and keeps its slice of the recurrent matrix in registers for all T steps, h_t (forward) /
// dgates_t (backward) is exchanged through L2 in bf16, 16 bytes at a time, the data being the flag;
// MFMA operands swapped as in the third generation (a lane holds four consecutive units of one row).
// Replaces the reference's python time loops and their autograd (neural_networks.py:1130-1141 liGRU,
// :1438-1447 RNN).
//
// What changed.  Up to the third generation every wave did everything: poll its quarter of the
// exchanged tile, prefetch the next step's fp32 operands from HBM, multiply, do the gate math,
// publish, store the step's fp32 outputs.  vmcnt is ONE in-order counter per wave for loads and
// stores (and loads return in issue order), so
//   * a poll can only be consumed behind s_waitcnt vmcnt(0), i.e. behind the acknowledge of every
//     HBM store and the return of every HBM prefetch the SAME wave issued before it - the measured
//     step of those kernels with the arithmetic removed is as long as the full step
//     (profiles/r03_rec_step_floor.json: structure_frac 1.0): the step is bound by that wave's own
//     memory queue, not by the hand-off and not by the 36 MFMAs;
//   * the six HBM instructions of a step are issued on the dependency chain (~570 clocks).
// Here a workgroup has EIGHT waves (two per SIMD, 256 registers each) with three roles:
//   * waves 0-3  COMPUTE: wait at the step's barrier, read the h fragments and the (already
//                BatchNorm-folded) projections from LDS, 36 MFMAs, gate math, publish h_t (their only
//                global-memory instructions: the 16-byte publish store and the fill pattern ahead),
//                leave the step's fp32 outputs in LDS;
//   * waves 4-6  POLL: nothing but the exchange - poll the 16 x 576 tile of h_{t-1} (one third
//                each) until no chunk holds the fill pattern, write it to the LDS A tile, barrier.
//                Their memory queue never holds anything but polls;
//   * wave 7     I/O: every HBM access of the step for all four compute waves, a step ahead /
//                behind, through LDS slots: projections P_{t+1} -> (scale, shift) -> LDS, outputs
//                Y_{t-1}, S_{t-1} LDS -> HBM.  Its HBM latencies meet nobody's dependency chain.
// One s_barrier per step orders everything (tile, projection slots and output slots are double
// buffered): T + 1 barriers per launch for every role.
#include <stdlib.h>

#include "pk_rec2_common.h"

// Defaults (measured: profiles/r04_rec_split.json): which passes run the role-split kernels when PK_REC_GEN* is unset,
// and the polling waves' idle time behind the barrier (the compute waves publish ~1 200 clocks behind it).
#ifndef PK_RECS_DEFAULT_FWD
#define PK_RECS_DEFAULT_FWD 0
#endif
#ifndef PK_RECS_DEFAULT_BWD
#define PK_RECS_DEFAULT_BWD 0
#endif
#define PK_RECS_DELAY_FWD 14
#define PK_RECS_DELAY_BWD 14

namespace {

// NP (template parameter) = number of polling waves: 3 (waves 4-6 poll, wave 7 does the I/O: 512 threads) or 0 (the
// compute waves poll their own quarter of the tile right behind their publish, as the earlier generations do, and wave 4
// does the I/O: 320 threads)
constexpr int S_PROW = 20, S_PATCH_F = 16 * S_PROW;  // LDS slot of one fp32 tensor tile [16 rows][16 units], row pitch 20 floats

__device__ __forceinline__ u32x4 s_pack_chunk(unsigned lo, unsigned hi) {  // see pk_rec_persist3.hip::pack_chunk
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return u32x4{a[0], b[0], a[1], b[1]};
}
__device__ __forceinline__ unsigned s_pack2(float x, float y) { return (unsigned)to_bf_pub(x) | ((unsigned)to_bf_pub(y) << 16); }

// 16-byte access of the I/O wave: EDGY = false: no wave of this workgroup straddles H (straight-line code);
// EDGY = true: the wave-uniform edge class e decides (0 = one 16-byte access, 1 = two 8-byte halves, 2 = element by element)
// (nt: streaming cache policy for tensors that are read / written exactly once - PK_REC_FLUSH_LATE bit 1 = stores, bit 2 = loads)
template <bool EDGY>
__device__ __forceinline__ f32x4 s_ld4(const float* base, unsigned off, int nv, int e, bool nt = false) {
    if constexpr (!EDGY) {
        if (nt) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (nv == 4 ? off : 0u)));
        return ld4<0>(base, off, nv);
    } else {
        if (e == 0) return ld4<0>(base, off, nv);
        if (e == 1) return ld4<1>(base, off, nv);
        return ld4<2>(base, off, nv);
    }
}
template <bool EDGY>
__device__ __forceinline__ void s_st4(float* base, unsigned off, int nv, int e, float* trash, f32x4 v, bool nt = false) {
    if constexpr (!EDGY) {
        if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(nv == 4 ? base + off : trash));
        else st4<0>(base, off, nv, trash, v);
    } else {
        if (e == 0) st4<0>(base, off, nv, trash, v);
        else if (e == 1) st4<1>(base, off, nv, trash, v);
        else st4<2>(base, off, nv, trash, v);
    }
}
// The polling waves start their poll when the workgroup's own compute waves have published (LDS flag words, one per
// compute wave, = number of steps published): a poll issued at that moment reaches L2 together with the stores of the
// other workgroups of the cluster, which run in lock step with this one - the hand-off then costs ONE round trip.  (A
// fixed idle time instead lets every workgroup's re-polls fall at a random phase of a ~1 800-clock round trip; the
// cluster then advances at the pace of the unluckiest workgroup: measured 6 480 instead of 5 440 clocks per step.)
// poll_to_lds (pk_rec2_common.h) with RUNNING offsets: polls the chunks at goff[], advances goff[] to the next step's slab
// while the loads are in flight (owned slots by +ts / -ts as upm says; slots not owned stay out of range), then checks,
// re-polls what still holds the fill pattern and stores the chunks to the LDS tile.
template <int NCH, bool FAST>
__device__ __forceinline__ bool poll_to_lds_adv(__amdgpu_buffer_rsrc_t rs, unsigned (&goff)[NCH], unsigned okm, unsigned upm, unsigned ts,
                                                const int (&loff)[NCH], unsigned char* tile, unsigned* err, int spin_limit, int lane,
                                                bool dead, int& retries) {
    u32x4 v[NCH];
    unsigned cur[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        cur[i] = goff[i];
        v[i] = poll_load<FAST>(rs, cur[i]);
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < NCH; ++i) bad = bad | has_sent16(v[i]);
    if (__any(bad) && !dead) {
        int spins = 0;
        while (true) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                if (has_sent16(v[i])) v[i] = poll_load<FAST>(rs, cur[i]);
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
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        *reinterpret_cast<u32x4*>(tile + loff[i]) = v[i];
        goff[i] = cur[i] + (((okm >> i) & 1u) ? (((upm >> i) & 1u) ? ts : 0u - ts) : 0u);
    }
    return dead;
}

// (the flag words are read with an inline-asm LDS load: a volatile C++ access through a generic pointer compiles to
// flat_load / flat_store + s_waitcnt vmcnt(0), i.e. it would wait for the wave's global-memory queue)
__device__ __forceinline__ void s_wait_published(const unsigned char* flags, unsigned want, int spin_limit) {
    const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)flags;
    for (int spins = 0; spins < spin_limit; ++spins) {
        u32x4 f;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f) : "v"(addr) : "memory");
        const unsigned m01 = f[0] < f[1] ? f[0] : f[1], m23 = f[2] < f[3] ? f[2] : f[3];
        if ((m01 < m23 ? m01 : m23) >= want) return;
        __builtin_amdgcn_s_sleep(1);
    }
}

// (the traced compute wave is a.helper_delay - PK_TRACE_WAVE, 0..3: the waves share their SIMDs with different helpers)
#define PKS_TRACE_AT(TID, slot)                                                              \
    do {                                                                                     \
        if (TR && a.trace != nullptr && blockIdx.x == 0 && tid == (TID) + 64 * a.helper_delay) a.trace[(long)step_idx * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// ============================================================================
// forward
// ============================================================================
template <int CELL, int ACT, bool TR, int NP>
__global__ __launch_bounds__((5 + NP) * 64) void recs_fwd_kernel(R2Args a) {
    constexpr int S_THREADS = (5 + NP) * 64, S_NP = NP > 0 ? NP : 1;
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr int LDA = pk_r2_lda_bf16(KPAD);
    constexpr int ATILE = RMAX * LDA * 2;
    constexpr int NCHP = (RMAX * (KPAD / 8) + 64 * S_NP - 1) / (64 * S_NP);  // 16-byte chunks per polling lane (6)
    constexpr int NCHC = (RMAX * (KPAD / 8) + 255) / 256;                    // ... per compute lane when the compute waves poll (5)
    constexpr int TSTRIDE = ATILE + 32;   // a tile is followed by its own 32-byte dump slot for chunk slots a lane does not own
    constexpr int LDS_TRASH = ATILE;      // (relative to the tile)
    constexpr int NOUT = 1 + NS;
    constexpr int PSLOT = 4 * G * S_PATCH_F, OSLOT = 4 * NOUT * S_PATCH_F;  // floats per step buffer (all four compute waves)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][tile | dump] | flags | [2] P slots | [2] output slots
    float* const pslots = reinterpret_cast<float*>(smem + 2 * TSTRIDE + 32);
    float* const oslots = pslots + 2 * PSLOT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int kq = lane >> 4;
    const unsigned TS = (unsigned)B * a.Ypitch * 2u;
    const unsigned szYb = (unsigned)T * TS;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.Yb, szYb);
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    unsigned char* const pub_flags = smem + 2 * TSTRIDE;  // unsigned [4]: steps published per compute wave (NP > 0)

    for (int i = tid; i < (2 * TSTRIDE + 32) / 4; i += S_THREADS) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    if (wave < 4) {
        // ===================================================================== COMPUTE
        const int ubase = p * 64 + wave * 16;
        const int frag_unit = ubase + (lane & 15);  // the unit whose U row this lane holds as MFMA A fragments
        const bool frag_ok = frag_unit < H;
        bf16x8 Uf[G][KSTEPS];
        {
            const unsigned szU = (unsigned)((size_t)G * H * H * 4);
            const __amdgpu_buffer_rsrc_t rsU = make_rsrc(a.U, szU);
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const int k0 = kk * 32 + kq * 8;
                    const unsigned off = (unsigned)(((g * H + frag_unit) * H + k0) * 4);
                    const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rsU, (frag_ok && k0 < H) ? off : szU, 0, 0);
                    const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rsU, (frag_ok && k0 + 4 < H) ? off + 16 : szU, 0, 0);
                    bf16x8 f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = k0 + e;
                        const float w = (k < H) ? __uint_as_float(e < 4 ? r0[e & 3] : r1[e & 3]) : 0.f;  // beyond H: the next row's data
                        f[e] = (short)pk_f2bf(w);
                    }
                    Uf[g][kk] = f;
                }
            }
        }
        // my (row, 4 units) of the step: row = lane & 15, units u0 .. u0 + 3
        const int row = lane & 15, u0 = ubase + kq * 4;
        const bool row_ok = row < nrows;
        const int n = n_base + (row_ok ? row : 0);
        const int dir = n >= B ? 1 : 0, bb = n - dir * B;
        int nv = H - u0;
        nv = nv > 4 ? 4 : (nv < 0 ? 0 : nv);
        nv = row_ok ? nv : 0;
        float msk[4], hprev[4], cprev[4];
        bool ok4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ok4[r] = r < nv;
            msk[r] = (a.mask != nullptr && ok4[r]) ? a.mask[(long)n * H + u0 + r] : a.mask_scalar;
            hprev[r] = 0.f;
            cprev[r] = 0.f;
        }
        // publish: the lanes of the even 16-lane rows store one 16-byte chunk (my row, 8 units from pu0)
        const int pu0 = ubase + (kq >> 1) * 8;
        const bool pk_ok = (kq & 1) == 0 && row_ok && pu0 < Hp;
        const unsigned pbase = pk_ok ? ((unsigned)bb * a.Ypitch + dir * Hp + pu0) * 2u : szYb;  // out of range: dropped
        // NP == 0: poll descriptors of my chunks (as in the earlier generations): chunk ci = (row, col) of the cluster's block
        // (kept as RUNNING offsets - advanced while the poll is in flight - plus two bit masks: the per-chunk base / step
        // arrays of the earlier generations cost 2 x NCHC registers more, which this 256-register wave does not have)
        unsigned gcur[NCHC], okm = 0u, upm = 0u;  // byte offset of the next poll; chunk slots I own; slots whose offset grows
        int clds[NCHC];
        if constexpr (NP == 0) {
            const int CPR = Hp >> 3;
#pragma unroll
            for (int i = 0; i < NCHC; ++i) {
                const int ci = tid + 256 * i;
                const bool ok = ci < nrows * CPR;
                const int crow = ok ? ci / CPR : 0, col = ok ? ci - crow * CPR : 0;
                const int cn = n_base + crow;
                const int cdir = cn >= B ? 1 : 0, cb = cn - cdir * B;
                // step t reads storage time (dir ? T-t : t-1); a slot I do not own stays out of range
                gcur[i] = ok ? ((unsigned)cb * a.Ypitch + cdir * Hp + col * 8) * 2u + (unsigned)(cdir ? (T - 1) : 0) * TS : szYb;
                okm |= ok ? (1u << i) : 0u;
                upm |= (ok && !cdir) ? (1u << i) : 0u;
                clds[i] = ok ? crow * (LDA * 2) + col * 16 : LDS_TRASH;
            }
        }
        const float* const my_p = pslots + wave * (G * S_PATCH_F) + row * S_PROW + kq * 4;
        float* const my_o = oslots + wave * (NOUT * S_PATCH_F) + row * S_PROW + kq * 4;
        if (a.self_fill) {  // my chunks of the first slabs, visible everywhere before the handshake lets anyone poll
            for (int tt = 0; tt < PK_R2_FILL_AHEAD && tt < T; ++tt)
                pub_store<false>(rs, pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - tt) : tt) * TS : 0u), sentinel);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        bool dead = false;
        const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
        auto run = [&](auto FASTC) {
            constexpr bool fast = decltype(FASTC)::value != 0;
            if ((a.flush_late & 1) == 0) __builtin_amdgcn_s_setprio(3);  // the helper wave on my SIMD takes the issue slots I leave (PK_REC_FLUSH_LATE=1: A/B)
            for (int t = 0; t < T; ++t) {
                const int step_idx = t;
                PKS_TRACE_AT(0, 0);
                if constexpr (NP == 0) {
                    if (t > 0) {
                        for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
                        int retries = 0;
                        dead = poll_to_lds_adv<NCHC, fast>(rs, gcur, okm, upm, TS, clds, smem + (t & 1) * TSTRIDE, a.err, a.spin_limit, lane, dead, retries);
                        if (TR && a.trace != nullptr && blockIdx.x == 0 && tid == 0) a.trace[(long)step_idx * 8 + 6] = (unsigned long long)retries;
                    }
                }
                PK_BARRIER_LDS();  // B(t): tile t, projections t in LDS; the I/O wave has taken the outputs of step t - 2
                PKS_TRACE_AT(0, 1);
                const unsigned char* At = smem + (t & 1) * TSTRIDE;
                f32x4 pv[G];
#pragma unroll
                for (int g = 0; g < G; ++g) pv[g] = *reinterpret_cast<const f32x4*>(my_p + (t & 1) * PSLOT + g * S_PATCH_F);
                f32x4 acc[G];
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool empty = TR && a.empty_step != 0 && a.empty_step < 6;  // (6, 7: full arithmetic, doctored I/O)
                const bool mm = t > 0 && !empty;
                const unsigned char* Ar = At + (lane & 15) * (LDA * 2) + kq * 16;
                constexpr int PKD = 4;
                bf16x8 hf[PKD];
                if (mm) {
#pragma unroll
                    for (int kk = 0; kk < PKD; ++kk) hf[kk] = *reinterpret_cast<const bf16x8*>(Ar + kk * 64);
                }
                PKS_TRACE_AT(0, 2);
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
                PKS_TRACE_AT(0, 3);
                float hv[4], sv[NS][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) pr[g] = pv[g][r] + acc[g][r];
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
                PKS_TRACE_AT(0, 4);
                {   // publish h_t: what the other workgroups of the cluster wait for; then the fill pattern ahead
                    const u32x4 o = s_pack_chunk(s_pack2(hv[0], hv[1]), s_pack2(hv[2], hv[3]));
                    const unsigned off = pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - t) : t) * TS : 0u);
                    pub_store<fast>(rs, off, o);
                    if (NP > 0 && lane == 0) reinterpret_cast<unsigned*>(smem + 2 * TSTRIDE)[wave] = (unsigned)(t + 1);  // the polling waves' cue
                    if (a.self_fill && t + PK_R2_FILL_AHEAD < T) {
                        const unsigned offf = pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - (t + PK_R2_FILL_AHEAD)) : (t + PK_R2_FILL_AHEAD)) * TS : 0u);
                        pub_store<fast>(rs, offf, sentinel);
                    }
                }
                // the fp32 outputs of the step: LDS, for the I/O wave (behind the next barrier)
                float* const o = my_o + (t & 1) * OSLOT;
                *reinterpret_cast<f32x4*>(o) = f32x4{hv[0], hv[1], hv[2], hv[3]};
#pragma unroll
                for (int k = 0; k < NS; ++k) *reinterpret_cast<f32x4*>(o + (1 + k) * S_PATCH_F) = f32x4{sv[k][0], sv[k][1], sv[k][2], sv[k][3]};
                PKS_TRACE_AT(0, 5);
            }
            PK_BARRIER_LDS();  // B(T): the outputs of the last step are in LDS
        };
        if (fast_rt) run(BoolC<1>());
        else run(BoolC<0>());
    } else if (NP > 0 && wave < 4 + NP) {
        // ===================================================================== POLL
        const int ptid = (wave - 4) * 64 + lane;
        const int CPR = Hp >> 3;
        unsigned cbase[NCHP], cstep[NCHP];
        int clds[NCHP];  // LDS byte offset inside a tile; chunk slots I do not own: the trash slot behind both tiles
#pragma unroll
        for (int i = 0; i < NCHP; ++i) {
            const int ci = ptid + 64 * S_NP * i;
            const bool ok = ci < nrows * CPR;
            const int row = ok ? ci / CPR : 0, col = ok ? ci - row * CPR : 0;
            const int n = n_base + row;
            const int dir = n >= B ? 1 : 0, b = n - dir * B;
            // step t reads storage time (dir ? T-t : t-1); a slot I do not own stays out of range
            cbase[i] = ok ? ((unsigned)b * a.Ypitch + dir * Hp + col * 8) * 2u + (unsigned)(dir ? (T - 1) : 0) * TS : szYb;
            cstep[i] = ok ? (dir ? 0u - TS : TS) : 0u;
            clds[i] = ok ? row * (LDA * 2) + col * 16 : LDS_TRASH;
        }
        __syncthreads();
        bool dead = false;
        const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
        auto run = [&](auto FASTC) {
            constexpr bool fast = decltype(FASTC)::value != 0;
            PK_BARRIER_LDS();  // B(0)
            for (int t = 1; t < T; ++t) {
                const int step_idx = t;
                unsigned goff[NCHP];
#pragma unroll
                for (int i = 0; i < NCHP; ++i) goff[i] = cbase[i] + (unsigned)(t - 1) * cstep[i];
                // the compute waves publish h_{t-1} a whole MFMA + gate phase behind the barrier: sleep through most of it,
                // then wait for their cue
                for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
                if (!dead) s_wait_published(pub_flags, (unsigned)t, a.spin_limit);
                int retries = 0;
                dead = poll_to_lds<NCHP, fast>(rs, goff, clds, smem + (t & 1) * TSTRIDE, a.err, a.spin_limit, lane, dead, retries);
                if (TR && a.trace != nullptr && blockIdx.x == 0 && tid == 256) {
                    a.trace[(long)step_idx * 8 + 6] = (unsigned long long)retries;
                    a.trace[(long)step_idx * 8 + 7] = __builtin_amdgcn_s_memtime();
                }
                PK_BARRIER_LDS();  // B(t)
            }
            PK_BARRIER_LDS();  // B(T)
        };
        if (fast_rt) run(BoolC<1>());
        else run(BoolC<0>());
    } else {
        // ===================================================================== I/O
        // access layout of the fp32 tensors: row = lane >> 2, four adjacent lanes cover 64 contiguous bytes of that row
        const int arow = lane >> 2;
        const int an = n_base + (arow < nrows ? arow : 0);
        const int adir = an >= B ? 1 : 0, ab = an - adir * B;
        int anv[4], edge[4];
        float psc[4][G][4], psh[4][G][4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int au0 = p * 64 + w * 16 + (lane & 3) * 4;
            int v = H - au0;
            v = v > 4 ? 4 : (v < 0 ? 0 : v);
            // wave-uniform: 0 = these 16 units do not straddle H, 1 = they do and H is even, 2 = H is odd
            edge[w] = __builtin_amdgcn_readfirstlane(__any(v > 0 && v < 4) != 0 ? ((H & 1) ? 2 : 1) : 0);
            anv[w] = arow < nrows ? v : 0;
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = r < anv[w];
                    psc[w][g][r] = ok ? a.pscale[g * H + au0 + r] : 0.f;
                    psh[w][g][r] = ok ? a.pshift[g * H + au0 + r] : 0.f;
                }
        }
        const int au00 = p * 64 + (lane & 3) * 4;  // wave w: + 16 w
        // element offsets of the units at storage time 0 / per unit of storage time, for P, Y and S
        const unsigned vP0 = ((unsigned)ab * GH + au00), vPs = (unsigned)B * GH;
        const unsigned vY0 = ((unsigned)ab * a.YH + adir * H + au00), vYs = (unsigned)B * a.YH;
        const unsigned vS0 = (((unsigned)adir * T * B + ab) * (NS * H) + au00), vSs = (unsigned)B * NS * H;
        float* trash = a.trash + lane * 4;
        const int aoff = arow * S_PROW + (lane & 3) * 4;  // my 16 bytes of a slot, access side
        __syncthreads();
        bool dead = false;
        (void)cluster_on_one_xcd(a, c, p, tid, dead);  // (takes part in the handshake's workgroup vote only)
        auto io = [&](auto EDGYC) {
            constexpr bool EDGY = decltype(EDGYC)::value != 0;
            // two register sets: the projections of even / odd steps, loaded TWO steps ahead of their use (an HBM load of
            // this access pattern can take longer than a whole step)
            f32x4 pn0[4][G], pn1[4][G];
            const bool nt_st = (a.flush_late & 2) != 0, nt_ld = (a.flush_late & 4) != 0;
            auto load_proj = [&](f32x4 (&pn)[4][G], int tt) {
                if (TR && a.empty_step >= 5) tt = tt & 3;  // diagnostics: the same few rows again and again (cache hits)
                const unsigned ts = (unsigned)(adir ? (T - 1 - tt) : tt);
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int g = 0; g < G; ++g) pn[w][g] = s_ld4<EDGY>(a.P, vP0 + ts * vPs + g * H + w * 16, anv[w], edge[w], nt_ld);
            };
            auto stage_proj = [&](const f32x4 (&pn)[4][G], int slot) {  // BatchNorm affine folded into the projection on the way: p * scale + shift
                float* const d = pslots + slot * PSLOT + aoff;
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(pn[w][g][r], psc[w][g][r], psh[w][g][r]);
                        *reinterpret_cast<f32x4*>(d + (w * G + g) * S_PATCH_F) = v;
                    }
            };
            auto flush_outputs = [&](int tt) {  // layer output and saved gates of step tt: LDS slots -> HBM, 16 bytes per lane
                const unsigned ts = (unsigned)(adir ? (T - 1 - tt) : tt);
                const float* const sl = oslots + (tt & 1) * OSLOT + aoff;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    f32x4 v[NOUT];
#pragma unroll
                    for (int k = 0; k < NOUT; ++k) v[k] = *reinterpret_cast<const f32x4*>(sl + (w * NOUT + k) * S_PATCH_F);
                    s_st4<EDGY>(a.Y, vY0 + ts * vYs + w * 16, anv[w], edge[w], trash, v[0], nt_st);
#pragma unroll
                    for (int k = 0; k < NS; ++k) s_st4<EDGY>(a.S, vS0 + ts * vSs + k * H + w * 16, anv[w], edge[w], trash, v[1 + k], nt_st);
                }
            };
            load_proj(pn0, 0);
            stage_proj(pn0, 0);
            if (T > 1) load_proj(pn1, 1);
            if (T > 2) load_proj(pn0, 2);
            PK_BARRIER_LDS();  // B(0)
            // diagnostics (results are garbage): EMPTY=2 no HBM traffic at all from this CU, 3 = loads only, 4 = stores only
            const bool no_ld = TR && (a.empty_step == 2 || a.empty_step == 4), no_st = TR && (a.empty_step == 2 || a.empty_step == 3 || a.empty_step == 7);
            auto iter = [&](int t, f32x4 (&pn)[4][G]) {  // while the compute waves work on step t; pn: the set of step t + 1
                if (t + 1 < T) {
                    stage_proj(pn, (t + 1) & 1);  // (the loads are two steps old)
                    if (t + 3 < T && !no_ld) load_proj(pn, t + 3);
                }
                if (t > 0 && !no_st) flush_outputs(t - 1);
                if (TR && NP == 0 && a.trace != nullptr && blockIdx.x == 0 && lane == 0 && t + 1 < T)
                    a.trace[(long)(t + 1) * 8 + 7] = __builtin_amdgcn_s_memtime();  // my arrival at B(t + 1)
                PK_BARRIER_LDS();  // B(t + 1)
            };
            for (int t = 0; t < T; t += 2) {
                iter(t, pn1);
                if (t + 1 < T) iter(t + 1, pn0);
            }
            flush_outputs(T - 1);
        };
        if ((edge[0] | edge[1] | edge[2] | edge[3]) != 0) io(BoolC<1>());
        else io(BoolC<0>());
    }
}

// ============================================================================
// backward: dL/dh_{t-1} = direct + [dgates_t] . [U_0; U_1; ...]
// ============================================================================
template <int CELL, int ACT, bool TR, int NP>
__global__ __launch_bounds__((5 + NP) * 64) void recs_bwd_kernel(R2Args a) {
    constexpr int S_THREADS = (5 + NP) * 64, S_NP = NP > 0 ? NP : 1;
    const int act = ACT >= 0 ? ACT : a.act;
    constexpr int G = pk_cell_gates(CELL), NS = pk_cell_saved(CELL);
    constexpr int LDA = pk_r2_lda_bf16(G * KPAD);
    constexpr int ATILE = RMAX * LDA * 2;
    constexpr int NCHP = (RMAX * G * (KPAD / 8) + 64 * S_NP - 1) / (64 * S_NP);  // 12 (liGRU) / 6 (RNN)
    constexpr int NCHC = (RMAX * G * (KPAD / 8) + 255) / 256;                    // ... per compute lane when the compute waves poll (9 / 5)
    constexpr int NIN = NS + 2;  // saved gates, h_{t-1}, dY
    constexpr int TSTRIDE = ATILE + 32, LDS_TRASH = ATILE;
    constexpr int ISLOT = 4 * NIN * S_PATCH_F, GSLOT = 4 * G * S_PATCH_F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][tile | dump] | flags | [2] input slots | [2] fp32 gate-gradient slots
    float* const islots = reinterpret_cast<float*>(smem + 2 * TSTRIDE + 32);
    float* const gslots = islots + 2 * ISLOT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x % a.C, p = blockIdx.x / a.C;
    const int H = a.H, Hp = a.Hp, B = a.B, T = a.T, GH = G * H;
    const unsigned TB = (unsigned)T * B;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int kq = lane >> 4;
    const unsigned TS = (unsigned)B * a.Gpitch * 2u;
    const unsigned ndir = (unsigned)(a.R / B);
    const unsigned szGb = ndir * (unsigned)T * TS;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dGb, szGb);
    const u32x4 sentinel = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const bool want_dp2 = a.dP2 != nullptr;  // fp32 gate gradients wanted (frozen-BatchNorm path); perf mode works from the bf16 copy
    unsigned char* const pub_flags = smem + 2 * TSTRIDE;  // unsigned [4]: iterations published per compute wave (NP > 0)

    for (int i = tid; i < (2 * TSTRIDE + 32) / 4; i += S_THREADS) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    if (wave < 4) {
        // ===================================================================== COMPUTE
        const int ubase = p * 64 + wave * 16;
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
        const int row = lane & 15, u0 = ubase + kq * 4;
        const bool row_ok = row < nrows;
        const int n = n_base + (row_ok ? row : 0);
        const int dir = n >= B ? 1 : 0, bb = n - dir * B;
        int nv = H - u0;
        nv = nv > 4 ? 4 : (nv < 0 ? 0 : nv);
        nv = row_ok ? nv : 0;
        float msk[4], dh_dir[4];
        bool ok4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ok4[r] = r < nv;
            msk[r] = (a.mask != nullptr && ok4[r]) ? a.mask[(long)n * H + u0 + r] : a.mask_scalar;
            dh_dir[r] = 0.f;
        }
        const int pu0 = ubase + (kq >> 1) * 8;
        const bool pk_ok = (kq & 1) == 0 && row_ok && pu0 < Hp;
        const unsigned pbase = pk_ok ? (unsigned)dir * (unsigned)T * TS + ((unsigned)bb * a.Gpitch + pu0) * 2u : szGb;
        unsigned gcur[NCHC], okm = 0u, upm = 0u;  // running poll offsets + masks (see the forward kernel)
        int clds[NCHC];
        if constexpr (NP == 0) {  // poll descriptors: chunk ci = (row, gate, col) of the cluster's dgates block
            const int CPR = Hp >> 3;
#pragma unroll
            for (int i = 0; i < NCHC; ++i) {
                const int ci = tid + 256 * i;
                const bool ok = ci < nrows * G * CPR;
                const int crow = ok ? ci / (G * CPR) : 0;
                const int rem = ok ? ci - crow * (G * CPR) : 0;
                const int cg = rem / CPR, col = rem - cg * CPR;
                const int cn = n_base + crow;
                const int cdir = cn >= B ? 1 : 0, cb = cn - cdir * B;
                // iteration it (t = T-1-it, it >= 1) reads storage time (dir ? T-2-t : t+1) = (dir ? it-1 : T-it)
                gcur[i] = ok ? (unsigned)cdir * (unsigned)T * TS + ((unsigned)cb * a.Gpitch + cg * Hp + col * 8) * 2u +
                                   (unsigned)(cdir ? 0 : (T - 1)) * TS
                             : szGb;
                okm |= ok ? (1u << i) : 0u;
                upm |= (ok && cdir) ? (1u << i) : 0u;
                clds[i] = ok ? crow * (LDA * 2) + (cg * KPAD + col * 8) * 2 : LDS_TRASH;
            }
        }
        const float* const my_i = islots + wave * (NIN * S_PATCH_F) + row * S_PROW + kq * 4;
        float* const my_g = gslots + wave * (G * S_PATCH_F) + row * S_PROW + kq * 4;
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
        const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
        auto run = [&](auto FASTC) {
            constexpr bool fast = decltype(FASTC)::value != 0;
            if ((a.flush_late & 1) == 0) __builtin_amdgcn_s_setprio(3);
            int it = 0;
            for (int t = T - 1; t >= 0; --t, ++it) {
                const int step_idx = it;
                PKS_TRACE_AT(0, 0);
                if constexpr (NP == 0) {
                    if (it > 0) {
                        for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
                        int retries = 0;
                        dead = poll_to_lds_adv<NCHC, fast>(rs, gcur, okm, upm, TS, clds, smem + (it & 1) * TSTRIDE, a.err, a.spin_limit, lane, dead, retries);
                        if (TR && a.trace != nullptr && blockIdx.x == 0 && tid == 0) a.trace[(long)step_idx * 8 + 6] = (unsigned long long)retries;
                    }
                }
                PK_BARRIER_LDS();  // B(it)
                PKS_TRACE_AT(0, 1);
                const unsigned char* At = smem + (it & 1) * TSTRIDE;
                f32x4 iv[NIN];
#pragma unroll
                for (int k = 0; k < NIN; ++k) iv[k] = *reinterpret_cast<const f32x4*>(my_i + (it & 1) * ISLOT + k * S_PATCH_F);
                f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
                const bool empty = TR && a.empty_step != 0 && a.empty_step < 6;  // (6, 7: full arithmetic, doctored I/O)
                const bool mm = t < T - 1 && !empty;
                const unsigned char* Ar = At + (lane & 15) * (LDA * 2) + kq * 16;
                constexpr int PKD = 4, NF = G * KSTEPS;
                bf16x8 df[PKD];
                if (mm) {
#pragma unroll
                    for (int f = 0; f < PKD; ++f) df[f] = *reinterpret_cast<const bf16x8*>(Ar + ((f / KSTEPS) * KPAD + (f % KSTEPS) * 32) * 2);
                }
                PKS_TRACE_AT(0, 2);
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
                PKS_TRACE_AT(0, 3);
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
                PKS_TRACE_AT(0, 4);
                {
                    const unsigned off = pbase + (pk_ok ? (unsigned)(dir ? (T - 1 - t) : t) * TS : 0u);
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const u32x4 o = s_pack_chunk(s_pack2(dgv[g][0], dgv[g][1]), s_pack2(dgv[g][2], dgv[g][3]));
                        pub_store<fast>(rs, off + (pk_ok ? (unsigned)(g * Hp) * 2u : 0u), o);
                    }
                    if (NP > 0 && lane == 0) reinterpret_cast<unsigned*>(smem + 2 * TSTRIDE)[wave] = (unsigned)(it + 1);  // the polling waves' cue
                    if (a.self_fill && t - PK_R2_FILL_AHEAD >= 0) fill_slab(t - PK_R2_FILL_AHEAD, FASTC);
                }
                if (want_dp2) {
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        *reinterpret_cast<f32x4*>(my_g + (it & 1) * GSLOT + g * S_PATCH_F) = f32x4{dgv[g][0], dgv[g][1], dgv[g][2], dgv[g][3]};
                }
                PKS_TRACE_AT(0, 5);
            }
            PK_BARRIER_LDS();  // B(T)
        };
        if (fast_rt) run(BoolC<1>());
        else run(BoolC<0>());
    } else if (NP > 0 && wave < 4 + NP) {
        // ===================================================================== POLL
        const int ptid = (wave - 4) * 64 + lane;
        const int CPR = Hp >> 3;
        unsigned cbase[NCHP], cstep[NCHP];
        int clds[NCHP];  // LDS byte offset inside a tile; chunk slots I do not own: the trash slot behind both tiles
#pragma unroll
        for (int i = 0; i < NCHP; ++i) {
            const int ci = ptid + 64 * S_NP * i;
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
        __syncthreads();
        bool dead = false;
        const bool fast_rt = __builtin_amdgcn_readfirstlane((int)(cluster_on_one_xcd(a, c, p, tid, dead) && a.force_safe == 0)) != 0;
        auto run = [&](auto FASTC) {
            constexpr bool fast = decltype(FASTC)::value != 0;
            PK_BARRIER_LDS();  // B(0)
            for (int it = 1; it < T; ++it) {
                const int step_idx = it;
                unsigned goff[NCHP];
#pragma unroll
                for (int i = 0; i < NCHP; ++i) goff[i] = cbase[i] + (unsigned)(it - 1) * cstep[i];
                for (int d = 0; d < a.poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
                if (!dead) s_wait_published(pub_flags, (unsigned)it, a.spin_limit);
                int retries = 0;
                dead = poll_to_lds<NCHP, fast>(rs, goff, clds, smem + (it & 1) * TSTRIDE, a.err, a.spin_limit, lane, dead, retries);
                if (TR && a.trace != nullptr && blockIdx.x == 0 && tid == 256) {
                    a.trace[(long)step_idx * 8 + 6] = (unsigned long long)retries;
                    a.trace[(long)step_idx * 8 + 7] = __builtin_amdgcn_s_memtime();
                }
                PK_BARRIER_LDS();  // B(it)
            }
            PK_BARRIER_LDS();  // B(T)
        };
        if (fast_rt) run(BoolC<1>());
        else run(BoolC<0>());
    } else {
        // ===================================================================== I/O
        const int arow = lane >> 2;
        const int an = n_base + (arow < nrows ? arow : 0);
        const int adir = an >= B ? 1 : 0, ab = an - adir * B;
        int anv[4], edge[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int au0 = p * 64 + w * 16 + (lane & 3) * 4;
            int v = H - au0;
            v = v > 4 ? 4 : (v < 0 ? 0 : v);
            edge[w] = __builtin_amdgcn_readfirstlane(__any(v > 0 && v < 4) != 0 ? ((H & 1) ? 2 : 1) : 0);
            anv[w] = arow < nrows ? v : 0;
        }
        const int au00 = p * 64 + (lane & 3) * 4;
        const unsigned vY0 = ((unsigned)ab * a.YH + adir * H + au00), vYs = (unsigned)B * a.YH;
        const unsigned vS0 = (((unsigned)adir * TB + ab) * (NS * H) + au00), vSs = (unsigned)B * NS * H;
        const unsigned vG0 = (((unsigned)adir * TB + ab) * GH + au00), vGs = (unsigned)B * GH;
        float* trash = a.trash + lane * 4;
        const int aoff = arow * S_PROW + (lane & 3) * 4;
        __syncthreads();
        bool dead = false;
        (void)cluster_on_one_xcd(a, c, p, tid, dead);
        auto io = [&](auto EDGYC) {
            constexpr bool EDGY = decltype(EDGYC)::value != 0;
            // saved tensors of a step, one 16-byte access each: [0..NS) gates, NS = h_{t-1}, NS+1 = dY
            f32x4 in0[4][NIN], in1[4][NIN];  // the saved tensors of even / odd iterations, loaded two iterations ahead
            const bool nt_ld = (a.flush_late & 4) != 0;
            auto load_step = [&](f32x4 (&in)[4][NIN], int t) {
                if (TR && a.empty_step >= 5) t = 1 + (t & 3);  // diagnostics: the same few rows again and again (cache hits)
                const unsigned ts = (unsigned)(adir ? (T - 1 - t) : t);
                const unsigned tp = t > 0 ? (adir ? ts + 1 : ts - 1) : ts;  // storage time of step t-1 (any valid row when t == 0)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int nvp = t > 0 ? anv[w] : 0;
#pragma unroll
                    for (int k = 0; k < NS; ++k) in[w][k] = s_ld4<EDGY>(a.S, vS0 + ts * vSs + k * H + w * 16, anv[w], edge[w], nt_ld);
                    in[w][NS] = s_ld4<EDGY>(a.Y, vY0 + tp * vYs + w * 16, nvp, edge[w], nt_ld);
                    in[w][NS + 1] = s_ld4<EDGY>(a.dY, vY0 + ts * vYs + w * 16, anv[w], edge[w], nt_ld);
                    if (t == 0) in[w][NS] = f32x4{0.f, 0.f, 0.f, 0.f};  // h_{-1} = 0
                }
            };
            auto stage_step = [&](const f32x4 (&in)[4][NIN], int slot) {
                float* const d = islots + slot * ISLOT + aoff;
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int k = 0; k < NIN; ++k) *reinterpret_cast<f32x4*>(d + (w * NIN + k) * S_PATCH_F) = in[w][k];
            };
            auto flush_gates = [&](int it) {  // fp32 gate gradients of iteration it (step tt = T-1-it): LDS slots -> HBM
                const int tt = T - 1 - it;
                const unsigned ts = (unsigned)(adir ? (T - 1 - tt) : tt);
                const float* const sl = gslots + (it & 1) * GSLOT + aoff;
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        s_st4<EDGY>(a.dP2, vG0 + ts * vGs + g * H + w * 16, anv[w], edge[w], trash,
                                    *reinterpret_cast<const f32x4*>(sl + (w * G + g) * S_PATCH_F));
            };
            load_step(in0, T - 1);
            stage_step(in0, 0);
            if (T > 1) load_step(in1, T - 2);
            if (T > 2) load_step(in0, T - 3);
            PK_BARRIER_LDS();  // B(0)
            const bool no_io = TR && (a.empty_step == 2 || a.empty_step == 4);  // diagnostics (EMPTY=2 / 4): no HBM loads from this CU
            auto iter = [&](int it, f32x4 (&in)[4][NIN]) {  // while the compute waves work on iteration it; in: the set of iteration it + 1
                if (it + 1 < T) {
                    stage_step(in, (it + 1) & 1);
                    if (it + 3 < T && !no_io) load_step(in, T - 1 - (it + 3));
                }
                if (want_dp2 && it > 0) flush_gates(it - 1);
                if (TR && NP == 0 && a.trace != nullptr && blockIdx.x == 0 && lane == 0 && it + 1 < T)
                    a.trace[(long)(it + 1) * 8 + 7] = __builtin_amdgcn_s_memtime();  // my arrival at B(it + 1)
                PK_BARRIER_LDS();  // B(it + 1)
            };
            for (int it = 0; it < T; it += 2) {
                iter(it, in1);
                if (it + 1 < T) iter(it + 1, in0);
            }
            if (want_dp2) flush_gates(T - 1);
        };
        if ((edge[0] | edge[1] | edge[2] | edge[3]) != 0) io(BoolC<1>());
        else io(BoolC<0>());
    }
}

typedef void (*RecSKernel)(R2Args);
template <int CELL, int NP>
RecSKernel picks_fwd(int act, bool tr) {
    if (tr) return recs_fwd_kernel<CELL, PK_ACT_RELU, true, NP>;
    return act == PK_ACT_RELU ? recs_fwd_kernel<CELL, PK_ACT_RELU, false, NP>
         : act == PK_ACT_TANH ? recs_fwd_kernel<CELL, PK_ACT_TANH, false, NP> : recs_fwd_kernel<CELL, -1, false, NP>;
}
template <int CELL, int NP>
RecSKernel picks_bwd(int act, bool tr) {
    if (tr) return recs_bwd_kernel<CELL, PK_ACT_RELU, true, NP>;
    return act == PK_ACT_RELU ? recs_bwd_kernel<CELL, PK_ACT_RELU, false, NP>
         : act == PK_ACT_TANH ? recs_bwd_kernel<CELL, PK_ACT_TANH, false, NP> : recs_bwd_kernel<CELL, -1, false, NP>;
}
int gs_on[2] = {-1, -1};     // per pass: 1 = the role-split kernels run this pass
int gs_np[2] = {0, 0};       // per pass: polling waves (0 = the compute waves poll)
int gs_delay[2] = {-1, -1};  // NP > 0: idle time of the polling waves behind the barrier, s_sleep units of 64 clocks

}  // namespace

// Do the role-split kernels run this pass of this cell?  PK_REC_GEN = 5 (both passes) / PK_REC_GEN_FWD / PK_REC_GEN_BWD = 5
// select them, 2 / 3 / 4 the earlier generations (pk_rec_persist3.hip::pk_rec3_covers).
int pk_recs_covers(int cell, int backward) {
    if (gs_on[0] < 0) {
        const char* both = getenv("PK_REC_GEN");
        const char* ef = getenv("PK_REC_GEN_FWD");
        const char* eb = getenv("PK_REC_GEN_BWD");
        auto parse = [](const char* e, int dflt) { return (e && e[0] >= '2' && e[0] <= '5') ? (e[0] == '5' ? 1 : 0) : dflt; };
        gs_on[0] = parse(ef, parse(both, PK_RECS_DEFAULT_FWD));
        gs_on[1] = parse(eb, parse(both, PK_RECS_DEFAULT_BWD));
        const char* np = getenv("PK_SPLIT_POLLERS");  // "3" / "0", or "<fwd><bwd>" e.g. "03"
        if (np && np[0]) {
            gs_np[0] = np[0] == '3' ? 3 : 0;
            gs_np[1] = (np[1] ? np[1] : np[0]) == '3' ? 3 : 0;
        }
        const char* df = getenv("PK_SPLIT_POLL_DELAY_FWD");
        const char* db = getenv("PK_SPLIT_POLL_DELAY_BWD");
        gs_delay[0] = df ? atoi(df) : PK_RECS_DELAY_FWD;
        gs_delay[1] = db ? atoi(db) : PK_RECS_DELAY_BWD;
    }
    return gs_on[backward ? 1 : 0] != 0 && (cell == PK_CELL_LIGRU || cell == PK_CELL_RNN);
}

// Launch loop of the role-split kernels; `a` and `pl` are prepared by pk_rec_fwd_bf16 / pk_rec_bwd_bf16
// (pk_rec_persist2.hip).  traced: the phase-trace instantiation (Li-GRU / relu only).
int pk_recs_launch(hipStream_t st, R2Args& a, const Plan2& pl, int cell, int act, bool backward, bool traced, bool delay_forced) {
    const int G = pk_cell_gates(cell), NS = pk_cell_saved(cell);
    const size_t atile = (size_t)RMAX * pk_r2_lda_bf16(backward ? G * KPAD : KPAD) * 2;
    const int nslot = backward ? (NS + 2) + G : G + (1 + NS);
    const size_t lds = 2 * (atile + 32) + 32 + (size_t)2 * 4 * nslot * S_PATCH_F * 4;
    const int np = gs_np[backward ? 1 : 0];
    const int threads = (5 + np) * 64;
    if (np > 0 && !delay_forced) a.poll_delay = gs_delay[backward ? 1 : 0];  // (np == 0: the per-pass defaults of the earlier generations)
    RecSKernel k;
    if (cell == PK_CELL_LIGRU) {
        if (np > 0) k = backward ? picks_bwd<PK_CELL_LIGRU, 3>(act, traced) : picks_fwd<PK_CELL_LIGRU, 3>(act, traced);
        else k = backward ? picks_bwd<PK_CELL_LIGRU, 0>(act, traced) : picks_fwd<PK_CELL_LIGRU, 0>(act, traced);
    } else {
        if (np > 0) k = backward ? picks_bwd<PK_CELL_RNN, 3>(act, false) : picks_fwd<PK_CELL_RNN, 3>(act, false);
        else k = backward ? picks_bwd<PK_CELL_RNN, 0>(act, false) : picks_fwd<PK_CELL_RNN, 0>(act, false);
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
        rc = pk_rec2_check_residency((const void*)k, threads, lds, pl.C * pl.Pn, backward ? "pk_rec_bwd_bf16" : "pk_rec_fwd_bf16");
        if (rc) return rc;
        hipLaunchKernelGGL(k, dim3(pl.C * pl.Pn), dim3(threads), lds, st, a);
        PK_LAUNCH_CHECK();
    }
    return 0;
}
