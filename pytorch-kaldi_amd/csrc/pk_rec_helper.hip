// pk_rec_helper.hip - L2 run-ahead helper of the persistent bf16 recurrences (round 5).
//
// What round 4 measured (profiles/r04_rec_io_diagnosis.json): a step of the persistent time loops
// (neural_networks.py:457-469, :629-641, :1130-1141) is bounded by hand-off + the HBM round trip of the step's own fp32
// traffic, because an HBM-latency load or store in flight on a CU delays that CU's polls of the exchange.  The 144 CUs
// that hold a recurrent workgroup leave 112 idle.  This kernel runs on some of those, next to the recurrence, and
// touches the lines the recurrent CUs are about to use a few steps AHEAD of them, so that the recurrent CUs' own
// accesses are served by their XCD's L2 instead of HBM:
//   forward:  the projections P(t + lead) (read by load_proj) and - PK_REC_HELPER bit 1 - the lines of Y / S that
//             flush_outputs is about to write (a store to a line that is already resident does not wait for HBM);
//   backward: the saved gates S, the layer output Y and the incoming gradient dY of the next steps (bit 2).
// It only LOADS: results never depend on it (a helper that is late, early, on the wrong XCD or absent changes the speed
// of the recurrence, nothing else).  Pacing needs no change in the recurrent kernels: the helpers of cluster c watch
// the chunk of the exchange buffer that (member 0, wave 0, lane 0) publishes each step - the data is the flag there
// already - after waiting for that member's word of the placement handshake (before it, a recycled exchange buffer may
// still hold valid-looking data of an earlier launch; behind it the first PK_R2_FILL_AHEAD slabs carry the pattern and
// every later slab is patterned four steps before it is published).  A helper workgroup serves a cluster that sits on
// ITS OWN XCD (HW_REG_XCC_ID against the handshake word), because L2s are per XCD; one CU sustains ~25 GB/s of misses,
// a cluster's step is ~190 KB per 2.3 us, hence several helper workgroups per cluster (PK_EXPERIMENT helper_wgs).  A helper
// asks for 96 KB of LDS it never uses, so that it cannot be placed on a CU that holds a recurrent workgroup.
// Every wait is bounded in time.
#include <stdlib.h>

#include "pk_rec2_common.h"

namespace {

struct HelpRange {           // one contiguous byte range per (row, step): a row of P, Y (own direction), S or dY
    const char* base;        // tensor base
    long long row_bytes;     // bytes per (storage time, batch) row
    long long dir_rows;      // rows skipped per direction (S: T * B; others 0)
    int dir_bytes;           // byte offset of direction 1 inside a row (Y, dY: H * 4)
    int len;                 // bytes of the range
    int lead;                // steps ahead of the published step
};
struct HelpArgs {
    int T, B, R, C, rpc, row0, backward, hpc;
    int delay;               // s_sleep units (64 clocks) between seeing a step published and touching

    int nranges;
    HelpRange r[4];
    const char* xbase;       // exchange buffer (Yb forward, dGb backward)
    long long x_dir_bytes;   // offset of direction 1 (forward: Hp * 2 inside a row; backward: T * TS)
    long long x_row_bytes;   // bytes per batch row of a slab
    long long x_ts;          // bytes per time slab
    const unsigned* xcd_tab;
    unsigned hs_gen;
    unsigned* tickets;       // [8], zero before the launch
    unsigned* sink;
    unsigned limit_clocks;   // bound of every wait, shader clocks
};

constexpr int MAXJOBS = 8;      // 128-byte lines per toucher thread and step
constexpr int HELPER_LDS = 96 * 1024;
constexpr int TOUCHERS = 192;   // threads of a helper workgroup that touch lines (waves 1-3); wave 0 paces

__device__ __forceinline__ bool timed_out(unsigned long long t0, unsigned limit) {
    return (unsigned long long)__builtin_readcyclecounter() - t0 > (unsigned long long)limit;
}
// a load that is never waited for (the wave keeps no count of it: inline asm is invisible to the wait-count pass);
// `sink` stays allocated for the whole kernel ("+v" here, consumed behind the final s_waitcnt), so a late return can
// only ever overwrite itself
__device__ __forceinline__ void touch_line(const char* p, unsigned& sink) {
    asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p) : "memory");
}

// First round-5 version: every wave polled the exchange, issued its loads and consumed them in the same iteration - one
// L2 round trip + one HBM round trip per step, serial: ~3 us against the 2.3 us step of a Li-GRU recurrence, so the
// helpers fell behind and the launch (which joins them) went 1.09 -> 1.45 ms, while the LSTM forward pass (3.6 us per
// step) gained 15 %.  Now the roles are split by WAVE, because loads return in order within a wave: wave 0 only polls
// (its vmcnt never holds an HBM load) and posts the newest published step in LDS; waves 1-3 read that word (lgkmcnt)
// and issue their touches without ever waiting for one.
__global__ __launch_bounds__(256) void rec_helper_kernel(HelpArgs a) {
    extern __shared__ unsigned char unused_lds[];  // placement only (see the header)
    __shared__ int s_cluster, s_part;
    __shared__ volatile int s_progress;            // newest published step seen by wave 0 (-1: none; T: stop)
    const int tid = threadIdx.x;
    const unsigned my_xcd = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;  // HW_REG_XCC_ID
    if (tid == 0) {
        int mine = -1, part = 0;
        bool ok = true;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int c = 0; c < a.C && ok; ++c) {  // every cluster's member 0 has started (and patterned its first slabs)
            while ((__hip_atomic_load(a.xcd_tab + c * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 4) != a.hs_gen) {
                if (timed_out(t0, a.limit_clocks)) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        }
        if (ok) {
            const unsigned rank = atomicAdd(a.tickets + my_xcd, 1u);
            unsigned here = 0;  // clusters on my XCD
            for (int c = 0; c < a.C; ++c)
                here += ((__hip_atomic_load(a.xcd_tab + c * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFu) == my_xcd) ? 1u : 0u;
            if (here > 0 && rank < here * (unsigned)a.hpc) {
                const unsigned which = rank % here;
                part = (int)(rank / here);
                unsigned seen = 0;
                for (int c = 0; c < a.C; ++c) {
                    const unsigned w = __hip_atomic_load(a.xcd_tab + c * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((w & 0xFu) == my_xcd) {
                        if (seen == which) mine = c;
                        ++seen;
                    }
                }
            }
        }
        s_cluster = mine;
        s_part = part;
        s_progress = -1;
    }
    __syncthreads();
    const int c = s_cluster, part = s_part;
    if (c < 0) return;
    const int T = a.T, B = a.B;
    const int n_base = a.row0 + c * a.rpc;
    int nrows = a.R - n_base;
    nrows = nrows < a.rpc ? nrows : a.rpc;
    if (nrows <= 0) return;
    const int rev = a.backward;

    if (tid < 64) {
        // ---- wave 0: the pace.  The chunk (row n_base, units 0..7) of the exchange buffer is published by (member 0,
        // wave 0, lane 0) each step; a chunk is ONE 16-byte store and a published dword is never all ones.
        const int dir0 = n_base >= B ? 1 : 0, b0 = n_base - dir0 * B;
        const char* xrow = a.xbase + (long long)dir0 * a.x_dir_bytes + (long long)b0 * a.x_row_bytes;
        unsigned long long tlast = __builtin_readcyclecounter();
        for (int q = 0; q < T; ++q) {
            const long long ts = (dir0 ^ rev) ? (T - 1 - q) : q;
            const unsigned* px = reinterpret_cast<const unsigned*>(xrow + ts * a.x_ts);
            while (__hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0xFFFFFFFFu) {
                if (timed_out(tlast, a.limit_clocks)) {
                    if (tid == 0) s_progress = T;  // give up: the touchers stop too
                    return;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            tlast = __builtin_readcyclecounter();
            if (tid == 0) s_progress = q;
        }
        if (tid == 0) s_progress = T;
        return;
    }

    // ---- waves 1-3: my line jobs: (range, row, line) -> byte offset at storage time 0 + per-time stride
    const char* jb[MAXJOBS];
    long long jstride[MAXJOBS], joff[MAXJOBS];
    int jdir[MAXJOBS], jlead[MAXJOBS];
    unsigned sink[MAXJOBS];
    int njobs = 0;
#pragma unroll
    for (int i = 0; i < MAXJOBS; ++i) sink[i] = 0u;
    {
        const int stride = TOUCHERS * a.hpc;
        int j = (tid - 64) + TOUCHERS * part, acc = 0;  // global job index walks ranges x rows x lines
        for (int k = 0; k < a.nranges; ++k) {
            const HelpRange& r = a.r[k];
            const int lines = (r.len + 127) / 128 + 1;  // (+1: a range that does not start on a line boundary)
            const int total = nrows * lines;
            while (j - acc < total && njobs < MAXJOBS) {
                const int q = j - acc, row = q / lines, ln = q - row * lines;
                const int n = n_base + row, dir = n >= B ? 1 : 0, b = n - dir * B;
                const long long start = ((long long)dir * r.dir_rows + b) * r.row_bytes + (long long)dir * r.dir_bytes;
                long long o = (start & ~127ll) + (long long)ln * 128;
                if (o > start + r.len - 4) o = start + r.len - 4;  // the extra line of an aligned range: its last word again
                jb[njobs] = r.base;
                joff[njobs] = o;
                jstride[njobs] = (long long)B * r.row_bytes;
                jdir[njobs] = dir;
                jlead[njobs] = r.lead;
                ++njobs;
                j += stride;
            }
            acc += total;
        }
    }
    // published step q -> touch step q + lead of every range; before anything is published: the steps 0 .. lead - 1
    int done = -2;  // last q handled (-1 = the head start)
    unsigned long long t0 = __builtin_readcyclecounter();  // of the last progress seen: the bound is per wait, not per launch
    while (true) {
        int p = s_progress;
        if (done == -2) p = -1;
        else if (p <= done) {
            if (p >= T || timed_out(t0, 4u * a.limit_clocks)) break;
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        t0 = __builtin_readcyclecounter();
        if (p >= T) break;
        // (a toucher that fell behind skips to the newest published step: old steps are history)
        const int q = p;
        // The publish of step q is the START of the cluster's hand-off (every member now polls the exchange through the
        // same L2): a burst of misses issued right then queues in front of those polls.  The recurrent kernels issue
        // their own traffic behind the poll for the same reason; the touchers wait out the hand-off too.
        if (q >= 0)
            for (int d = 0; d < a.delay; ++d) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < MAXJOBS; ++i) {
            if (i < njobs) {
                const int first = q < 0 ? 0 : q + jlead[i];
                const int last = q < 0 ? jlead[i] - 1 : q + jlead[i];
                for (int s = first; s <= last && s < T; ++s) {
                    const long long ts = (jdir[i] ^ rev) ? (T - 1 - s) : s;
                    touch_line(jb[i] + joff[i] + ts * jstride[i], sink[i]);
                }
            }
        }
        done = q;
        if (q >= T - 1) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < MAXJOBS; ++i) acc ^= sink[i];
    if (acc == 0x9E3779B9u && a.sink != nullptr) a.sink[0] = acc + unused_lds[0];  // (keeps the sinks alive)
}

hipStream_t g_hstream = nullptr;
hipEvent_t g_fork = nullptr, g_join = nullptr;
unsigned* g_tickets = nullptr;  // [64][8]: one row of tickets per launch, recycled
unsigned g_ticket_row = 0;
int g_mode = -1, g_lead_p = 3, g_lead_o = 1, g_lead_b = 2, g_hpc = 4, g_delay = 0;

// -1 = PK_REC_HELPER unset: the per-cell default (pk_rec_helper_wanted)
int helper_mode() {
    static bool read = false;
    if (!read) {
        read = true;
        const char* e = getenv("PK_REC_HELPER");
        g_mode = e ? (atoi(e) & 7) : -1;
        const char* l = pk_experiment("helper_lead");  // "p:o:b": steps ahead for P / the output lines / the backward loads
        if (l) {
            int p = 0, o = 0, b = 0;
            const int n = sscanf(l, "%d:%d:%d", &p, &o, &b);
            if (n >= 1 && p > 0) g_lead_p = p;
            if (n >= 2 && o > 0) g_lead_o = o;
            if (n >= 3 && b > 0) g_lead_b = b;
        }
        const char* w = pk_experiment("helper_wgs");  // helper workgroups per cluster
        if (w && atoi(w) > 0) g_hpc = atoi(w) > 6 ? 6 : atoi(w);
        const char* d = pk_experiment("helper_delay");  // units of 64 clocks
        if (d && atoi(d) >= 0) g_delay = atoi(d);
    }
    return g_mode;
}
int ensure_helper() {
    if (g_hstream) return 0;
    PK_CHECK_HIP(hipStreamCreateWithFlags(&g_hstream, hipStreamNonBlocking));
    PK_CHECK_HIP(hipEventCreateWithFlags(&g_fork, hipEventDisableTiming));
    PK_CHECK_HIP(hipEventCreateWithFlags(&g_join, hipEventDisableTiming));
    PK_CHECK_HIP(hipMalloc((void**)&g_tickets, 64 * 8 * sizeof(unsigned) + 64));
    PK_CHECK_HIP(hipFuncSetAttribute((const void*)rec_helper_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HELPER_LDS));
    return 0;
}

}  // namespace

// mode: bits 0-2 as PK_REC_HELPER; optional tuning fields (0 = keep): bits 8-11 lead of P, 12-15 lead of the output lines,
// 16-19 lead of the backward loads, 20-23 helper workgroups per cluster, 24-29 delay behind a publish (units of 256 clocks)
extern "C" void pk_rec_helper_set_mode(int mode) {
    helper_mode();  // (reads the environment once)
    if (mode < 0) {  // back to the per-cell default
        g_mode = -1;
        return;
    }
    g_mode = mode & 7;
    if ((mode >> 8) & 15) g_lead_p = (mode >> 8) & 15;
    if ((mode >> 12) & 15) g_lead_o = (mode >> 12) & 15;
    if ((mode >> 16) & 15) g_lead_b = (mode >> 16) & 15;
    if ((mode >> 20) & 15) g_hpc = ((mode >> 20) & 15) > 6 ? 6 : ((mode >> 20) & 15);
    g_delay = ((mode >> 24) & 63) * 4;  // bits 24-29: units of 256 clocks
}
extern "C" int pk_rec_helper_get_mode(void) { return helper_mode(); }  // -1: the per-cell default

// Which helpers does this pass take?  (bit 0: forward P, bit 1: forward output lines, bit 2: backward loads)
// Default (PK_REC_HELPER unset), from the round-5 sweeps on one layer at the BASELINE geometry and the recipes' steps
// (profiles/r05_rec_helper.json): the eight-wave LSTM forward pass gains 15-18 % per launch with P and the output lines
// run ahead (timit_lstm 24.98 -> 23.76 ms per step); its backward pass gains 7 % alone but nothing in the training step
// (the helpers then compete with the side-stream weight-gradient GEMMs for the idle CUs); the Li-GRU forward pass - the
// shortest step, 2.2 us - LOSES 20-40 % whatever the lead, helper count or delay behind the publish, its backward pass
// gains 1-3 % alone and nothing in the step; GRU does not move.  So: LSTM forward only.
int pk_rec_helper_wanted(bool backward, int launches, int cell) {
    int m = helper_mode();
    if (m < 0) m = (cell == PK_CELL_LSTM && !backward) ? 3 : 0;
    if (launches != 1) return 0;
    return backward ? (m & 4) : (m & 3);
}
// Before the recurrence is launched on st: the helper stream joins st here, i.e. it starts together with the recurrence.
int pk_rec_helper_fork(hipStream_t st) {
    int rc = ensure_helper();
    if (rc) return rc;
    PK_CHECK_HIP(hipEventRecord(g_fork, st));
    PK_CHECK_HIP(hipStreamWaitEvent(g_hstream, g_fork, 0));
    return 0;
}
// Behind the launch of the recurrence (a carries its arguments, handshake generation included): launch the helpers on
// their own stream and make st wait for their exit (they end a few steps before the recurrence does).
// s_layout_ok: S is [ndir][T * B][NS * H] fp32 (liGRU / RNN / LSTM).
int pk_rec_helper_launch(hipStream_t st, const R2Args& a, const Plan2& pl, int G, int NS, bool backward, bool s_layout_ok,
                         int m) {
    HelpArgs h;
    h.T = a.T; h.B = a.B; h.R = a.R; h.C = pl.C; h.rpc = pl.rpc; h.row0 = a.row0; h.backward = backward ? 1 : 0;
    h.hpc = g_hpc;
    h.delay = g_delay;
    const long long TB = (long long)a.T * a.B, H4 = (long long)a.H * 4;
    int n = 0;
    auto add = [&](const void* base, long long row_bytes, long long dir_rows, int dir_bytes, int len, int lead) {
        if (base == nullptr || n >= 4) return;
        h.r[n].base = (const char*)base; h.r[n].row_bytes = row_bytes; h.r[n].dir_rows = dir_rows;
        h.r[n].dir_bytes = dir_bytes; h.r[n].len = len; h.r[n].lead = lead;
        ++n;
    };
    if (!backward) {
        if (m & 1) add(a.P, G * H4, 0, 0, (int)(G * H4), g_lead_p);
        if (m & 2) {
            add(a.Y, (long long)a.YH * 4, 0, (int)H4, (int)H4, g_lead_o);
            if (s_layout_ok) add(a.S, NS * H4, TB, 0, (int)(NS * H4), g_lead_o);
        }
        h.xbase = (const char*)a.Yb;
        h.x_dir_bytes = (long long)a.Hp * 2;
        h.x_row_bytes = (long long)a.Ypitch * 2;
        h.x_ts = (long long)a.B * a.Ypitch * 2;
    } else {
        if (s_layout_ok) add(a.S, NS * H4, TB, 0, (int)(NS * H4), g_lead_b);
        add(a.Y, (long long)a.YH * 4, 0, (int)H4, (int)H4, g_lead_b + 1);  // h_{t-1}: the row of the next iteration
        add(a.dY, (long long)a.YH * 4, 0, (int)H4, (int)H4, g_lead_b);
        h.x_ts = (long long)a.B * a.Gpitch * 2;
        h.xbase = (const char*)a.dGb;
        h.x_dir_bytes = (long long)a.T * h.x_ts;
        h.x_row_bytes = (long long)a.Gpitch * 2;
    }
    {   // every helper workgroup pins a whole CU (HELPER_LDS of dummy LDS) and the recurrence must stay co-resident:
        // only CUs the recurrence's own grid leaves free may be taken (pk_rec2_check_residency counts the recurrence alone)
        const int free_cus = pk_num_cu() - pl.C * pl.Pn;
        const int fit = pl.C > 0 ? free_cus / pl.C : 0;
        if (fit < 1) return 0;
        if (h.hpc > fit) h.hpc = fit;
    }
    h.nranges = n;
    h.xcd_tab = a.xcd_tab; h.hs_gen = a.hs_gen;
    h.limit_clocks = 48000000u;  // ~20 ms
    if (n == 0 || h.xbase == nullptr) return 0;
    {   // the job lists of a cluster's helpers must hold the lines of one step (speed only: no helper otherwise)
        long long lines = 0;
        for (int k = 0; k < n; ++k) lines += (long long)pl.rpc * ((h.r[k].len + 127) / 128 + 1);
        if (lines > (long long)TOUCHERS * MAXJOBS * h.hpc) return 0;
    }
    g_ticket_row = (g_ticket_row + 1) & 63;
    h.tickets = g_tickets + g_ticket_row * 8;
    h.sink = g_tickets + 64 * 8;
    PK_CHECK_HIP(hipMemsetAsync(h.tickets, 0, 8 * sizeof(unsigned), g_hstream));
    hipLaunchKernelGGL(rec_helper_kernel, dim3(pl.C * h.hpc), dim3(256), HELPER_LDS, g_hstream, h);
    PK_LAUNCH_CHECK();
    PK_CHECK_HIP(hipEventRecord(g_join, g_hstream));
    PK_CHECK_HIP(hipStreamWaitEvent(st, g_join, 0));
    return 0;
}
