// pk_rec2_common.h - shared device helpers of the perf-mode persistent recurrences
// (pk_rec_persist2.hip: liGRU / RNN / LSTM; pk_rec_persist2_gru.hip: GRU / minimalGRU).
// See pk_rec_persist2.hip for the design notes.
#pragma once
#ifndef PK_REC2_PRECISE       // (pk_rec_persist2_f32.hip: the exact-fp32 twin keeps the precise exp / division)
#define PK_CELL_FAST_MATH 1  // perf mode: hardware-rate exp / reciprocal in the gate math
#endif
#include "pk_cell.h"

// Self-filling exchange (liGRU / RNN / 4-wave LSTM kernels of pk_rec_persist2.hip): the lane that publishes a chunk of
// step t also stores the 0xFF pattern into the same chunk of step t + PK_R2_FILL_AHEAD (same lane, same address, program
// order: the pattern can never overtake the data), and the first PK_R2_FILL_AHEAD slabs before the placement handshake
// (write-through, drained with vmcnt(0): every member's pattern is in place before any member polls).  Replaces the
// whole-buffer fill (141 + 295 MB per layer and step) - the pattern now meets its data in L2 a few microseconds later.
#define PK_R2_FILL_AHEAD 4
struct R2Args {
    int T, B, R, H, Hp, YH, act;
    int C, Pn, rpc, row0;  // clusters, workgroups per cluster, rows per cluster, first row of this launch
    const float *P, *pscale, *pshift, *U, *mask;
    float mask_scalar;
    float* Y;
    float* S;
    unsigned short* Yb;
    unsigned short* Xb;  // two-phase cells: bf16 r*h (GRU) / z*h (minimalGRU), same layout as Yb
    int Ypitch;  // elements per (t,b) row of Yb; direction d starts at d*Hp
    const float* dY;
    float* dP2;
    unsigned short* dGb;
    int Gpitch;  // elements per row of dGb; gate g starts at g*Hp
    float* Yx;   // exact-fp32 kernels (pk_rec_persist2_f32.hip): fp32 exchange buffers with the same geometry as Yb / dGb
    float* dGx;  //   (pitches in floats; a chunk = 16 bytes = 4 units)
    float* Xx;   // exact-fp32 two-phase cells (pk_rec_persist4_f32.hip): r*h (GRU) / z*h (minimalGRU), same layout as Yx
    unsigned* err;
    int spin_limit;
    float* trash;               // >= 64 bytes per lane-group of write-only scratch for masked-off stores
    int poll_delay;             // s_sleep units (64 clocks) between the publish and the first poll of the next step
    int force_safe;             // 1 = always use the placement-independent write-through exchange
    unsigned* xcd_tab;          // [C][16] placement handshake words: (hs_gen << 4) | XCD of the member that wrote it
    unsigned hs_gen;            // this launch's handshake generation (words of earlier launches are simply not current)
    unsigned long long* trace;  // optional [T][8] phase time stamps of (cluster 0, member 0, wave 0); null = off
    int helper_delay;           // eight-wave LSTM kernels: extra s_sleep units before the second wave of a pair polls
    int self_fill;              // 1 = the kernel writes the "not written yet" pattern itself, PK_R2_FILL_AHEAD steps ahead of its publishes
    int flush_late;             // third generation: 1 = the step's off-chain HBM traffic is issued behind the MFMA block
    int empty_step;             // diagnostics (traced kernels only): skip the MFMA block and the gate math - what is left
                                // of a step is the hand-off itself (poll, barrier, flush / prefetch issue, publish)
    // ---- per-step LayerNorm of h_t inside the time loop (LN kernels only; ln_gamma == null otherwise)
    const float *ln_gamma, *ln_beta;  // [H]
    float ln_eps;
    float* lnh;      // pre-LN h_t, same layout as Y ([T][B][YH], storage time)
    float* lnstat;   // [T][R][2]: mean, 1 / (std + eps) of (step t, row n)
    float* lnx;      // row-statistics exchange [T + 1][ln_ncg][4 row quads][4 * Pn slots][8 floats], 0xFF-filled before the launch
    float* lnpart;   // backward: [2][ln_ncg][KPAD] per-cluster partial sums of d gamma, d beta
    int ln_cg0;      // global index of this launch's first cluster
    int ln_ncg;      // clusters over all launches
};

struct Plan2 {
    int Pn, C, rpc, launches;
};

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KPAD = 576;    // K (hidden units) padded to 18 MFMA k-steps of 32
constexpr int KSTEPS = 18;
constexpr int RMAX = 16;     // rows per cluster (one MFMA M tile)

// Row stride of an A tile in LDS.  The MFMA A fragments are read with ds_read_b128, lane (row r = lane & 15, quarter
// q = lane >> 4) at r * stride + q * 16 bytes (+ a constant per k-step).  The hardware serves that instruction in four
// groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 - i.e. eight rows at quarter q
// together with the OTHER eight rows at quarter q + 1.  With the stride an odd multiple of 16 bytes (the first choice:
// 1168 B) seven of the sixteen 16-byte bank groups of such a lane group are hit twice (PMC: 490 / 780 conflict cycles
// per forward / backward step and CU); a stride of 2 (mod 16) sixteen-byte units is conflict-free for all four groups.
__host__ __device__ constexpr int pk_r2_lda_bf16(int k_elems) { return k_elems + 8 * ((((2 - k_elems / 8) % 16) + 16) % 16); }
__host__ __device__ constexpr int pk_r2_lda_f32(int k_elems) { return k_elems + 4 * ((((2 - k_elems / 4) % 16) + 16) % 16); }


#define PK_TRACE(slot)                                                                      \
    do {                                                                                    \
        if (TR && a.trace != nullptr && blockIdx.x == 0 && tid == 0) a.trace[(long)step_idx * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// Is any part of a polled 16-byte chunk still the fill pattern?  A chunk is written by ONE 16-byte store of one
// producer lane; its dwords are written atomically and a published dword never equals 0xFFFFFFFF (to_bf_pub cannot
// emit a 0xFFFF half), so the chunk is complete iff none of its four dwords is all ones: an unsigned max and one
// compare (the half-by-half test this replaces cost 16+ VALU operations per chunk, on the dependency chain).
__device__ __forceinline__ bool has_sent16(const u32x4 v) {
    const unsigned m01 = v[0] > v[1] ? v[0] : v[1], m23 = v[2] > v[3] ? v[2] : v[3];
    return (m01 > m23 ? m01 : m23) == 0xFFFFFFFFu;
}
// bf16 (round to nearest even: the gfx950 v_cvt_pk_bf16_f32) with the sentinel pattern excluded: the one NaN encoding
// that would read as "not written yet" becomes the canonical quiet NaN
__device__ __forceinline__ unsigned short to_bf_pub(float f) {
    const unsigned short u = __builtin_bit_cast(unsigned short, (__bf16)f);
    return u == (unsigned short)0xFFFF ? (unsigned short)0x7FC0 : u;
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
// FAST (every member of the cluster sits on one XCD, decided by the start-up handshake): the
// exchange stays inside that XCD's L2 - producers use plain stores (the line stays in L2), consumers
// poll with nt loads (bypass the per-CU L1, served by L2).  Otherwise: write-through (sc1) stores
// and agent-scope (sc1) loads, correct for any placement.  Placement only ever changes the speed.
//
// vmcnt is ONE counter for loads and stores on gfx9-class hardware and the two kinds retire out of order
// with respect to each other, so a counted wait cannot tell a landed poll from an acknowledged store:
// the poll data is only safe behind s_waitcnt vmcnt(0).  The step is therefore ordered so that everything
// still in flight at that point is old: the fp32 output stores of step t-1 and the prefetch loads of
// step t+1 are issued right AFTER the poll of step t has landed (a whole MFMA + gate phase before the
// next poll); only the 16-byte publish store is young.
template <bool FAST>
__device__ __forceinline__ u32x4 poll_load(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return FAST ? __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2) : __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
}
template <bool FAST>
__device__ __forceinline__ void pub_store(__amdgpu_buffer_rsrc_t rs, unsigned off, u32x4 v) {
    if (FAST) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
}

// Poll NCH 16-byte chunks per lane until none holds the sentinel, then store them to the LDS tile.
// Chunk slots a lane does not own alias one of the cluster's real chunks (harmless duplicate read) and
// land in an LDS trash slot, which keeps the code branch-free.  (The loads are compiler-visible on
// purpose: with inline-asm loads nothing stops the register allocator from copying a destination register
// before the s_waitcnt that makes it valid.)
template <int NCH, bool FAST>
__device__ __forceinline__ bool poll_to_lds(__amdgpu_buffer_rsrc_t rs, const unsigned (&goff)[NCH], const int (&loff)[NCH],
                                            unsigned char* tile, unsigned* err, int spin_limit, int lane, bool dead,
                                            int& retries) {
    // straight-line on purpose (a branch per slot makes hipcc wait for every load separately).  Chunk slots a
    // lane does not own carry an out-of-range offset: the bounds check answers with zeros without touching
    // memory, zeros are never the sentinel, and the value lands in an LDS trash slot.
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
#pragma unroll
    for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(tile + loff[i]) = v[i];
    return dead;
}

// ---- per-step LayerNorm (neural_networks.py:23-33 applied to h_t at :466-467, :1138-1139, :1444-1445): the row
// statistics need every unit of the row, i.e. one more exchange between the members of a cluster inside the step.
// Sum over the 16 lanes of a DPP row (= the 16 units a wave holds for one row quad), every lane ends with the total.
// Each stage adds the value of ONE partner group, so the result is bit-identical on all 16 lanes.
__device__ __forceinline__ float dpp_row_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}
__device__ __forceinline__ unsigned ln_bits(float f) {
    const unsigned u = __float_as_uint(f);
    return u == 0xFFFFFFFFu ? 0x7FC00000u : u;  // the one NaN encoding that reads as "not written yet"
}
// Two sums per row over ALL units of the cluster.  In: a[r], b[r] = this lane's terms for rows kq*4 + r (zero for
// units / rows outside the layer).  Every WAVE publishes its 16-unit partial sums (lane 0 of each DPP row: one 32-byte
// entry per row quad, slot = member * 4 + wave) and then polls all 4 * Pn entries of its row quads - lane u of a DPP row
// takes slots u, u + 16, u + 32 - with the data as the flag, like the h_t exchange itself.  No workgroup barrier, no
// LDS.  Fixed summation order: every lane, wave and member of the cluster ends with bit-identical totals.
//   pub_off: byte offset of my entry in this step's slab (lanes that do not publish: out of range, dropped)
//   poll_off[s]: byte offsets of the entries I poll (slots beyond 4 * Pn: out of range, answered with zeros)
// NP: 16-slot groups a lane polls (3 covers the 4 x 9 slots of the second-generation clusters, 5 the 4 x 18 of the fourth)
template <bool FAST, int NP = 3>
__device__ __forceinline__ bool ln_row_allreduce(__amdgpu_buffer_rsrc_t rsx, unsigned pub_off, const unsigned (&poll_off)[NP],
                                                 float (&a)[4], float (&b)[4], unsigned* err, int spin_limit, int lane, bool dead) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a[r] = dpp_row_sum(a[r]);
        b[r] = dpp_row_sum(b[r]);
    }
    pub_store<FAST>(rsx, pub_off, u32x4{ln_bits(a[0]), ln_bits(b[0]), ln_bits(a[1]), ln_bits(b[1])});
    pub_store<FAST>(rsx, pub_off + 16u, u32x4{ln_bits(a[2]), ln_bits(b[2]), ln_bits(a[3]), ln_bits(b[3])});
    u32x4 v[2 * NP];
#pragma unroll
    for (int i = 0; i < 2 * NP; ++i) v[i] = poll_load<FAST>(rsx, poll_off[i >> 1] + (unsigned)(i & 1) * 16u);
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 2 * NP; ++i) bad = bad | has_sent16(v[i]);
    if (__any(bad) && !dead) {
        int spins = 0;
        while (true) {
#pragma unroll
            for (int i = 0; i < 2 * NP; ++i)
                if (has_sent16(v[i])) v[i] = poll_load<FAST>(rsx, poll_off[i >> 1] + (unsigned)(i & 1) * 16u);
            bad = false;
#pragma unroll
            for (int i = 0; i < 2 * NP; ++i) bad = bad | has_sent16(v[i]);
            if (!__any(bad)) break;
            if (spin_check2(spins, spin_limit, err, lane)) {
                dead = true;
                break;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ch = r >> 1, e = (r & 1) * 2;
        float sa = __uint_as_float(v[ch][e]), sb = __uint_as_float(v[ch][e + 1]);
#pragma unroll
        for (int g = 1; g < NP; ++g) {
            sa += __uint_as_float(v[2 * g + ch][e]);
            sb += __uint_as_float(v[2 * g + ch][e + 1]);
        }
        a[r] = dpp_row_sum(sa);
        b[r] = dpp_row_sum(sb);
    }
    return dead;
}
// byte offsets into the row-statistics exchange for one lane at step 0; a slot this lane does not use is out of range
// (== size: stores are dropped, loads answer zeros - never the sentinel) at every step
template <int NP>
struct LnSlotsN {
    unsigned pub, poll[NP], slab, size;
    bool pub_ok, poll_ok[NP];
    __device__ __forceinline__ unsigned pub_at(int step) const { return pub_ok ? pub + (unsigned)step * slab : size; }
    __device__ __forceinline__ void poll_at(int step, unsigned (&o)[NP]) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) o[i] = poll_ok[i] ? poll[i] + (unsigned)step * slab : size;
    }
};
typedef LnSlotsN<3> LnSlots;
template <int NP>
__device__ __forceinline__ LnSlotsN<NP> ln_slots_n(const R2Args& a, int c, int p, int wave, int lane, int T) {
    LnSlotsN<NP> s;
    const unsigned nslot = 4u * (unsigned)a.Pn, kq = (unsigned)lane >> 4, u = (unsigned)lane & 15u;
    s.slab = (unsigned)a.ln_ncg * 4u * nslot * 32u;
    s.size = (unsigned)(T + 1) * s.slab;  // (slab T: the first step's extra exchange, forward kernels)
    const unsigned base = (((unsigned)(a.ln_cg0 + c) * 4u + kq) * nslot) * 32u;
    s.pub_ok = u == 0u;
    s.pub = base + ((unsigned)p * 4u + (unsigned)wave) * 32u;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const unsigned slot = u + 16u * (unsigned)i;
        s.poll_ok[i] = slot < nslot;
        s.poll[i] = base + slot * 32u;
    }
    return s;
}
__device__ __forceinline__ LnSlots ln_slots(const R2Args& a, int c, int p, int wave, int lane, int T) {
    return ln_slots_n<3>(a, c, p, wave, lane, T);
}

// One-time placement handshake: every member publishes the XCD it runs on (write-through) and
// reads all members' words (agent scope); all members see the same words, hence take the same
// decision.  Returns true when the whole cluster shares one XCD.  A word carries the launch's generation number, so the
// table needs no reset between launches (that was one hipMemsetAsync on the critical path in front of every recurrence).
__device__ __forceinline__ bool cluster_on_one_xcd(const R2Args& a, int c, int p, int tid, bool& dead) {
    const unsigned my = (a.hs_gen << 4) | (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu);  // HW_REG_XCC_ID
    unsigned* tab = a.xcd_tab + c * 16;
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

typedef float pk_f32x4_t __attribute__((ext_vector_type(4)));
template <int B>
struct BoolC;  // (defined below)
// compile-time loop: f(BoolC<0>()), f(BoolC<1>()), ... f(BoolC<N - 1>()) - for loop indices that select a template argument
template <int I, int N, typename F>
__device__ __forceinline__ void pk_static_for(F&& f) {
    if constexpr (I < N) {
        f(BoolC<I>());
        pk_static_for<I + 1, N>(f);
    }
}
// One v_mfma_f32_16x16x4_f32.  An exact-fp32 LSTM or Li-GRU wave holds 288 B-operand registers (GRU: 216) next to 36 operand, 32 accumulator
// and ~60 other registers: more than the 256 architectural VGPRs.  Left to itself the compiler parks the excess in AGPRs and
// copies every such value into ONE temporary VGPR in front of its MFMA (v_accvgpr_read + wait states: 56 % of the LSTM
// kernels' MFMAs, 44 instead of 32 clocks per MFMA measured - tools/ubench/mfma_f32_rate.hip shows the pipe itself sustains
// 32).  gfx90a and later read SrcB straight from an AGPR: the B fragments of the SECOND unit tile are bound to AGPRs for the
// whole kernel by passing them to the instruction through an "a" constraint (BA = true), which takes inline assembly - and
// with it the wait states the compiler's hazard recogniser would have inserted: two in front of every such MFMA (a VALU
// write of a source / accumulator register just before it), pk4_mfma_settle() behind the last one of a block (an 8-pass
// MFMA's result may be read 18 wait states after issue).  PK4_ASM_B = false gives the builtin form everywhere (A/B).
constexpr bool PK4_ASM_B = true;
template <bool BA>
__device__ __forceinline__ void pk4_mfma(pk_f32x4_t& acc, unsigned a_bits, float b) {
    if constexpr (BA && PK4_ASM_B) {
        // (not volatile: the result depends on the operands only, and a volatile asm would pin every LDS / global load of the
        // block in place - the fragment reads of the second-generation kernels landed right in front of their MFMAs)
        asm("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a_bits), "a"(b));
    } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a_bits), b, acc, 0, 0, 0);
    }
}
// behind the last pk4_mfma<true> on these accumulators, before anything reads them (one statement: every accumulator is
// an operand, so none of them can be read in front of the wait)
__device__ __forceinline__ void pk4_mfma_settle(pk_f32x4_t& x0) {
    if constexpr (PK4_ASM_B) asm("s_nop 15\n\ts_nop 7" : "+a"(x0));
}
__device__ __forceinline__ void pk4_mfma_settle(pk_f32x4_t& x0, pk_f32x4_t& x1) {
    if constexpr (PK4_ASM_B) asm("s_nop 15\n\ts_nop 7" : "+a"(x0), "+a"(x1));
}
__device__ __forceinline__ void pk4_mfma_settle(pk_f32x4_t& x0, pk_f32x4_t& x1, pk_f32x4_t& x2) {
    if constexpr (PK4_ASM_B) asm("s_nop 15\n\ts_nop 7" : "+a"(x0), "+a"(x1), "+a"(x2));
}
__device__ __forceinline__ void pk4_mfma_settle(pk_f32x4_t& x0, pk_f32x4_t& x1, pk_f32x4_t& x2, pk_f32x4_t& x3) {
    if constexpr (PK4_ASM_B) asm("s_nop 15\n\ts_nop 7" : "+a"(x0), "+a"(x1), "+a"(x2), "+a"(x3));
}
template <int N>
__device__ __forceinline__ void pk4_mfma_settle(pk_f32x4_t (&acc)[N]) {
    static_assert(N >= 1 && N <= 4, "pk4_mfma_settle: up to four accumulators");
    if constexpr (N == 1) pk4_mfma_settle(acc[0]);
    else if constexpr (N == 2) pk4_mfma_settle(acc[0], acc[1]);
    else if constexpr (N == 3) pk4_mfma_settle(acc[0], acc[1], acc[2]);
    else pk4_mfma_settle(acc[0], acc[1], acc[2], acc[3]);
}

// workgroup barrier that orders LDS only: __syncthreads() also drains vmcnt, i.e. it would wait for
// the prefetch loads issued just before it (HBM latency on the dependency chain)
#define PK_BARRIER_LDS()                               \
    do {                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_s_barrier();                  \
        asm volatile("" ::: "memory");                 \
    } while (0)

// ---- wave-private fp32 patches [16 rows][16 units] in LDS: the transposer between the MFMA C/D
// layout (lane -> rows kq*4+r, unit lane&15: what the gate math works in) and the "vector" layout
// (lane -> row lane>>2, 4 consecutive units: one 16-byte global access per lane, 1 KB per wave
// instruction).  4-byte-per-lane global accesses cost the same issue slot as 16-byte ones and were
// 60 % of a step; LDS round trips are an order of magnitude cheaper.
__device__ __forceinline__ void patch_put_cd(float* patch, int kq, int lane, const float (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) patch[(kq * 4 + r) * 16 + (lane & 15)] = v[r];
}
__device__ __forceinline__ void patch_get_cd(const float* patch, int kq, int lane, float (&v)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = patch[(kq * 4 + r) * 16 + (lane & 15)];
}
__device__ __forceinline__ f32x4 patch_get_vec(const float* patch, int lane) {
    return *reinterpret_cast<const f32x4*>(patch + (lane >> 2) * 16 + (lane & 3) * 4);
}
__device__ __forceinline__ void patch_put_vec(float* patch, int lane, f32x4 v) {
    *reinterpret_cast<f32x4*>(patch + (lane >> 2) * 16 + (lane & 3) * 4) = v;
}
#define PK_LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// 16-byte (4 unit) global access of the vector layout; nv = number of valid units (edge of H).
// Straight-line code: invalid lanes / elements are redirected (loads: to the tensor base, result
// unused; stores: to a library-owned trash page) instead of being branched around, so that the compiler
// can count the outstanding operations exactly - a divergent store makes it fall back to
// s_waitcnt vmcnt(0), which would put the store acknowledgements back on the dependency chain.
// Plain 64-bit addressing (no buffer descriptors: four of them per kernel exhaust the SGPRs and the
// compiler then wraps every access in a waterfall loop).
// EDGE is wave-uniform: true only for the one wave whose 16 units straddle H when H % 4 != 0; it uses
// four 4-byte accesses per lane, every other wave a single 16-byte access.
template <int EDGE>
__device__ __forceinline__ f32x4 ld4(const float* base, unsigned off, int nv) {
    if (EDGE == 0) {
        return *reinterpret_cast<const f32x4*>(base + (nv == 4 ? off : 0u));
    } else if (EDGE == 1) {  // even H: nv is 0, 2 or 4 - two 8-byte halves
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 lo = *reinterpret_cast<const f32x2*>(base + (nv >= 2 ? off : 0u));
        const f32x2 hi = *reinterpret_cast<const f32x2*>(base + (nv == 4 ? off + 2 : 0u));
        return f32x4{lo[0], lo[1], hi[0], hi[1]};
    } else {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = base[e < nv ? off + e : 0u];
        return v;
    }
}
template <int EDGE>
__device__ __forceinline__ void st4(float* base, unsigned off, int nv, float* trash, f32x4 v) {
    if (EDGE == 0) {
        *reinterpret_cast<f32x4*>(nv == 4 ? base + off : trash) = v;
    } else if (EDGE == 1) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2*>(nv >= 2 ? base + off : trash) = f32x2{v[0], v[1]};
        *reinterpret_cast<f32x2*>(nv == 4 ? base + off + 2 : trash + 2) = f32x2{v[2], v[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) *(e < nv ? base + off + e : trash + e) = v[e];
    }
}
template <int B>
struct BoolC {  // (an int: 0 = no edge, 1 = even-H edge in 8-byte halves, 2 = odd-H edge element by element)
    static constexpr int value = B;
};


#define PK_EDGE_DISPATCH(CALL)                 \
    do {                                        \
        if (edge == 0) CALL(BoolC<0>());        \
        else if (edge == 1) CALL(BoolC<1>());   \
        else CALL(BoolC<2>());                  \
    } while (0)
// inside a time loop that was instantiated for a wave whose units do not straddle H (SE == 0) the dispatch is static;
// SE < 0 keeps the run-time one.  (The scalar branches of the run-time forms - this one, the XCD fast/safe select and
// the trace hooks - cost ~4 % of a step when they sit in the loop.)
#define PK_EDGE_DISPATCH_S(CALL)               \
    do {                                        \
        if constexpr (SE == 0) {                \
            CALL(BoolC<0>());                   \
        } else {                                \
            PK_EDGE_DISPATCH(CALL);             \
        }                                       \
    } while (0)
#define PK_RUN_SPECIALISED(RUN, FAST_RT)                                              \
    do {                                                                              \
        const int edge_u = __builtin_amdgcn_readfirstlane(edge);                      \
        if (FAST_RT) {                                                                \
            if (edge_u == 0) RUN(BoolC<1>(), BoolC<0>()); else RUN(BoolC<1>(), BoolC<-1>()); \
        } else {                                                                      \
            if (edge_u == 0) RUN(BoolC<0>(), BoolC<0>()); else RUN(BoolC<0>(), BoolC<-1>()); \
        }                                                                             \
    } while (0)

}  // namespace

// host-side plumbing shared by both translation units (defined in pk_rec_persist2.hip)
int pk_rec2_make_plan(int R, int H, Plan2& pl);
// per-step LayerNorm (PkLnHost: pk_cell.h)
int pk_rec2_ln_setup(hipStream_t st, R2Args& a, const Plan2& pl, const PkLnHost* ln, bool backward);  // fills a.ln_*, fills the exchange
int pk_rec2_ln_finish(hipStream_t st, const R2Args& a, const PkLnHost* ln);  // backward: per-cluster partial sums -> d gamma, d beta
int pk_rec2_check(const char* who, int cell_ok, int cell, int T, int B, int bidir, int H);
int pk_rec2_host_setup(R2Args& a, bool backward, int cell);                 // error word, trash page, handshake table, tuning knobs
int pk_rec2_reset_handshake(hipStream_t st, R2Args& a);  // before every launch: a fresh handshake generation
// The clusters of a persistent launch exchange h_t with each other every step: every workgroup of the grid has to be
// resident at the same time.  Checks the grid against what the device can hold (occupancy query for this kernel, block
// size and dynamic LDS x CU count, cached per kernel) - a grid that cannot be co-resident is refused here instead of
// ending in bounded-spin time-outs.  (Kernels of OTHER streams or processes can still delay a workgroup's start; that
// only costs time: every spin is bounded at ~10 ms and reported.)
int pk_rec2_check_residency(const void* kernel, int threads, size_t lds, int grid, const char* who);
// third generation (pk_rec_persist3.hip: swapped MFMA operands): the backward pass of liGRU / RNN by default (PK_EXPERIMENT rec_gen*)
int pk_rec3_covers(int cell, int backward);
int pk_rec3_launch(hipStream_t st, R2Args& a, const Plan2& pl, int cell, int act, bool backward, bool traced);
// L2 run-ahead helpers on the idle CUs (pk_rec_helper.hip; PK_REC_HELPER): fork before the recurrence is launched,
// launch behind it.  They only load - results never depend on them.
int pk_rec_helper_wanted(bool backward, int launches, int cell);  // -> the mode bits this pass takes (0: none)
int pk_rec_helper_fork(hipStream_t st);
int pk_rec_helper_launch(hipStream_t st, const R2Args& a, const Plan2& pl, int G, int NS, bool backward, bool s_layout_ok,
                         int mode);
// eight-wave LSTM kernels (pk_rec_persist2_lstm.hip): on unless PK_EXPERIMENT lstm_waves=4; the launch loop over pl.launches
int pk_rec2l_enabled();
int pk_rec2l_launch(hipStream_t st, R2Args& a, const Plan2& pl, int act, bool backward);
