// pk_conv.hip - valid 1-D convolution (stride 1) fused with max_pool1d
// (kernel = stride = pool, floor), replacing F.conv1d + F.max_pool1d of the
// reference CNN / SincNet stacks (neural_networks.py:1546-1552, :1655-1661,
// :1805-1813).  The un-pooled conv output (201 MB for SincNet layer 1 at
// batch 128) is never written in forward.
//
//   x [B,Cin,L]  w [Cout,Cin,K]  y [B,Cout,Lp]  Lp = (L-K+1)/pool
//
// fp32 direct convolutions are VALU work on CDNA4 (the fp32 MFMA rate equals the packed-FMA VALU rate), so
// all three passes are register-tiled FMA kernels whose only per-tap traffic is one LDS word:
//
//   forward / data gradient   conv_tile_kernel: a thread owns CT output channels x LT consecutive positions
//       (CT*LT accumulators).  The input window of a chunk of input channels lives in LDS (one pad word per 64
//       so that the stride-LT reads of a wave hit 64 different banks) and slides through LT registers, one new
//       LDS word per tap; the weights come from a [channel tile][in channel][k][CT] packed copy through the
//       scalar cache, so a tap costs 1 ds_read_b32 + CT*LT v_fmac.  Forward adds bias, reduces the pooling
//       windows in registers and stores the pooled value + arg-max position; the data gradient runs the same
//       loop over dz (dy routed to the arg-max positions) with the taps reversed.
//   filter gradient           conv_bwd_filter_tile_kernel: a thread owns COT output channels x KT taps of one
//       input channel; a block walks (batch, position-tile) pairs, rebuilding the dz tile in LDS from dy + argmax
//       (never from HBM), and the x window slides through KT registers.  The reduction over batch x positions is
//       split over up to 256 blocks whose partial sums are added in a fixed order (deterministic).
#include "pk_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// weight packing: wt[tile][ic][k][c]
//   mode 0 (forward):        oc = co, ic = ci   wt = w[tile*CT + c][ic][k]
//   mode 1 (data gradient):  oc = ci, ic = co   wt = w[ic][tile*CT + c][k]
__global__ void conv_w_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int K, int CT, int mode,
                                   float* __restrict__ wt) {
    const int NOC = mode == 0 ? Cout : Cin, NIC = mode == 0 ? Cin : Cout;
    const int ntile = (NOC + CT - 1) / CT;
    const long n = (long)ntile * NIC * K * CT;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CT);
        long r = i / CT;
        const int k = (int)(r % K);
        r /= K;
        const int ic = (int)(r % NIC), tile = (int)(r / NIC);
        const int oc = tile * CT + c;
        float v = 0.f;
        if (oc < NOC) v = mode == 0 ? w[((long)oc * Cin + ic) * K + k] : w[((long)ic * Cin + oc) * K + k];
        wt[i] = v;
    }
}

__host__ __device__ inline int lds_pad(int a) { return a + (a >> 6); }

// out[b,oc,p] = sum_ic sum_k wt[oc][ic][k] * x[b,ic,p + k]         (FWD,  p over conv positions, + bias, max-pool)
// out[b,oc,p] = sum_ic sum_k wt[oc][ic][k] * dz[b,ic,p - k]        (!FWD, p over input positions; dz = dy routed to the
//                                                                    arg-max positions, rebuilt in LDS from dy + argmax)
// A block is 4 waves = OW output-channel tiles x PW position groups of 64*LT positions (OW*PW = 4): the waves of
// one position group share the staged window.  With a single channel tile the four waves instead split the input
// channels (IW = 4) and their partial sums are added in a fixed order through LDS.
// grid = (position tiles, B, ceil(ntile / OW)).
template <int CT, int LT, bool FWD, int POOL>
__global__ __launch_bounds__(256) void conv_tile_kernel(const float* __restrict__ in, const int* __restrict__ amax_in,
                                                        const float* __restrict__ wt, const float* __restrict__ bias, int NIC,
                                                        int Lin, int NOC, int K, int Lout, int ICC, int OW, int IW, int pool,
                                                        float* __restrict__ out, int* __restrict__ argmax, int Lp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int PW = 4 / (OW * IW);
    const int ow = wave % OW, iw = (wave / OW) % IW, pw = wave / (OW * IW);
    const int ntile = (NOC + CT - 1) / CT;
    const int tile = blockIdx.z * OW + ow;
    const int TL = PW * 64 * LT;
    const int span = TL + K - 1;
    const int spanp = (lds_pad(span) + 4) & ~3;
    const int b = blockIdx.y, p0 = blockIdx.x * TL;
    const int tp = LT * (pw * 64 + lane);      // first position of this thread inside the tile
    const int in0 = FWD ? p0 : p0 - (K - 1);  // first input position of the window
    constexpr int UNR = CT == 1 ? 16 : 4;  // taps per unrolled body (loads of a body are issued together)
    float acc[CT][LT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int j = 0; j < LT; ++j) acc[c][j] = 0.f;
    for (int ic0 = 0; ic0 < NIC; ic0 += ICC) {
        const int nic = min(ICC, NIC - ic0);
        __syncthreads();
        if (FWD) {
            // four rows per pass so that four loads are in flight per thread
            for (int ic = 0; ic < nic; ic += 4) {
                const float* src = in + ((long)b * NIC + ic0 + ic) * Lin;
                for (int s = tid; s < span; s += 256) {
                    const int l = in0 + s;
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = (ic + q < nic && l < Lin) ? src[(long)q * Lin + l] : 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (ic + q < nic) smem[(ic + q) * spanp + lds_pad(s)] = v[q];
                }
            }
        } else {
            f32x4* z4 = reinterpret_cast<f32x4*>(smem);
            const int n4 = nic * spanp / 4;
            for (int i = tid; i < n4; i += 256) z4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            __syncthreads();
            // pooled outputs whose window [lp*pool, lp*pool+pool) can intersect [in0, in0+span)
            const int lpa = max(in0, 0) / pool;
            const int lpb = min(Lp - 1, (in0 + span - 1) / pool);
            for (int ic = 0; ic < nic; ic += 4) {
                const long o = ((long)b * NIC + ic0 + ic) * Lp;
                for (int lp = lpa + tid; lp <= lpb; lp += 256) {
                    float v[4];
                    int ps[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool ok = ic + q < nic;
                        v[q] = ok ? in[o + (long)q * Lp + lp] : 0.f;
                        ps[q] = ok ? amax_in[o + (long)q * Lp + lp] - in0 : -1;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (ps[q] >= 0 && ps[q] < span) smem[(ic + q) * spanp + lds_pad(ps[q])] = v[q];
                }
            }
        }
        __syncthreads();
        if (tile < ntile) {
            for (int ic = iw; ic < nic; ic += IW) {
                const float* dr = smem + ic * spanp;
                const float* wr = wt + ((long)tile * NIC + ic0 + ic) * K * CT;
                float win[LT];
                if (FWD) {
#pragma unroll
                    for (int j = 0; j < LT - 1; ++j) win[j] = dr[lds_pad(tp + j)];
                } else {
#pragma unroll
                    for (int j = 1; j < LT; ++j) win[j] = dr[lds_pad(tp + j + K - 1)];
                }
#pragma unroll UNR
                for (int k = 0; k < K; ++k) {
                    if (FWD) win[LT - 1] = dr[lds_pad(tp + LT - 1 + k)];
                    else win[0] = dr[lds_pad(tp + K - 1 - k)];
#pragma unroll
                    for (int c = 0; c < CT; ++c) {
                        const float wv = wr[k * CT + c];
#pragma unroll
                        for (int j = 0; j < LT; ++j) acc[c][j] = fmaf(wv, win[j], acc[c][j]);
                    }
                    if (FWD) {
#pragma unroll
                        for (int j = 0; j < LT - 1; ++j) win[j] = win[j + 1];
                    } else {
#pragma unroll
                        for (int j = LT - 1; j > 0; --j) win[j] = win[j - 1];
                    }
                }
            }
        }
    }
    if (IW > 1) {
        // IW == 4 implies one channel tile and one position group: wave iw holds a partial sum of every output
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int j = 0; j < LT; ++j) smem[((iw * CT + c) * LT + j) * 64 + lane] = acc[c][j];
        __syncthreads();
        if (iw != 0) return;
        for (int q = 1; q < IW; ++q)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int j = 0; j < LT; ++j) acc[c][j] += smem[((q * CT + c) * LT + j) * 64 + lane];
    }
    if (tile >= ntile) return;
    if (FWD) {
        // bias, then max over the POOL-wide windows this thread owns; first maximum wins, as torch's max_pool1d
        const int lpb = (p0 + tp) / POOL;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int oc = tile * CT + c;
            if (oc >= NOC) break;
            const float bv = bias ? bias[oc] : 0.f;
#pragma unroll
            for (int g = 0; g < LT / POOL; ++g) {
                const int lp = lpb + g;
                float best = acc[c][g * POOL] + bv;
                int bi = 0;
#pragma unroll
                for (int q = 1; q < POOL; ++q) {
                    const float v = acc[c][g * POOL + q] + bv;
                    if (v > best) {
                        best = v;
                        bi = q;
                    }
                }
                if (lp < Lp) {
                    const long o = ((long)b * NOC + oc) * Lp + lp;
                    out[o] = best;
                    argmax[o] = lp * POOL + bi;
                }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int oc = tile * CT + c;
            if (oc >= NOC) break;
#pragma unroll
            for (int j = 0; j < LT; ++j) {
                const int l = p0 + tp + j;
                if (l < Lout) out[((long)b * NOC + oc) * Lout + l] = acc[c][j];
            }
        }
    }
}

// FWD: in = x [B,Cin,L];  !FWD: in = dy [B,Cout,Lp] + amax (positions in [0, Lc))
template <int CT, int LT, bool FWD, int POOL>
int launch_conv_tile(hipStream_t st, const float* in, const int* amax_in, const float* w, float* wt, const float* bias, int B,
                     int Cin, int Cout, int K, int Lin, int Lout, int pool, float* out, int* argmax, int Lp) {
    const int NOC = FWD ? Cout : Cin, NIC = FWD ? Cin : Cout;
    const int ntile = (NOC + CT - 1) / CT;
    {
        const long n = (long)ntile * NIC * K * CT;
        long blocks = (n + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(conv_w_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, Cout, Cin, K, CT, FWD ? 0 : 1, wt);
        PK_LAUNCH_CHECK();
    }
    // waves of a block: as many channel tiles as there are (up to 4), the rest along the position axis
    const int OW = ntile >= 4 ? 4 : (ntile >= 2 ? 2 : 1);
    const int IW = (ntile == 1 && NIC >= 4) ? 4 : 1;
    const int PW = 4 / (OW * IW);
    const int TL = PW * 64 * LT;
    const int spanp = (lds_pad(TL + K - 1) + 4) & ~3;
    int ICC = (40 * 1024) / (spanp * 4);
    if (ICC < 1) ICC = 1;
    if (ICC > NIC) ICC = NIC;
    size_t lds = sizeof(float) * (size_t)spanp * ICC;
    if (IW > 1 && lds < sizeof(float) * IW * CT * LT * 64) lds = sizeof(float) * IW * CT * LT * 64;
    PK_REQUIRE(lds <= 160 * 1024, "pk_conv1d_pool: filter length %d too large for the LDS window", K);
    static bool attr = false;
    if (!attr) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)conv_tile_kernel<CT, LT, FWD, POOL>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    dim3 grid((Lout + TL - 1) / TL, B, (ntile + OW - 1) / OW);
    hipLaunchKernelGGL((conv_tile_kernel<CT, LT, FWD, POOL>), grid, dim3(256), lds, st, in, amax_in, wt, bias, NIC, Lin, NOC, K,
                       Lout, ICC, OW, IW, pool, out, argmax, Lp);
    PK_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// generic forward (any pool width): one block = (b, 16 output channels, 128 pooled positions); weights and
// the x window in LDS.  Only used when pool is not 1, 2 or 3 (no shipped recipe).
constexpr int CO_TILE = 16;
constexpr int LP_TILE = 128;
constexpr int CI_CHUNK = 8;

__global__ __launch_bounds__(LP_TILE) void conv_pool_fwd_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                        const float* __restrict__ bias, int B, int Cin, int L,
                                                                        int Cout, int K, int pool, int Lp,
                                                                        float* __restrict__ y, int* __restrict__ argmax) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int span = LP_TILE * pool + K - 1;
    float* xs = smem;                   // [CI_CHUNK][span]
    float* ws = smem + CI_CHUNK * span; // [CO_TILE][CI_CHUNK][K]
    const int b = blockIdx.z, co0 = blockIdx.y * CO_TILE, lp0 = blockIdx.x * LP_TILE;
    const int tid = threadIdx.x;
    const int lp = lp0 + tid;
    const int l0 = lp0 * pool;
    for (int q0 = 0; q0 < pool; q0 += 4) {
        float acc[CO_TILE][4];
#pragma unroll
        for (int c = 0; c < CO_TILE; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[c][q] = 0.f;
        for (int ci0 = 0; ci0 < Cin; ci0 += CI_CHUNK) {
            const int nci = min(CI_CHUNK, Cin - ci0);
            __syncthreads();
            for (int i = tid; i < nci * span; i += LP_TILE) {
                const int ci = i / span, s = i - ci * span;
                const int l = l0 + s;
                xs[ci * span + s] = (l < L) ? x[((long)b * Cin + ci0 + ci) * L + l] : 0.f;
            }
            for (int i = tid; i < CO_TILE * nci * K; i += LP_TILE) {
                const int c = i / (nci * K), r = i - c * (nci * K);
                const int ci = r / K, k = r - ci * K;
                ws[(c * CI_CHUNK + ci) * K + k] = (co0 + c < Cout) ? w[((long)(co0 + c) * Cin + ci0 + ci) * K + k] : 0.f;
            }
            __syncthreads();
            for (int ci = 0; ci < nci; ++ci) {
                const float* xr = xs + ci * span + tid * pool + q0;
                for (int k = 0; k < K; ++k) {
                    float xv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) xv[q] = (q0 + q < pool) ? xr[k + q] : 0.f;
#pragma unroll
                    for (int c = 0; c < CO_TILE; ++c) {
                        const float wv = ws[(c * CI_CHUNK + ci) * K + k];
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[c][q] = fmaf(wv, xv[q], acc[c][q]);
                    }
                }
            }
        }
        if (lp < Lp) {
#pragma unroll
            for (int c = 0; c < CO_TILE; ++c) {
                if (co0 + c >= Cout) continue;
                const long o = ((long)b * Cout + co0 + c) * Lp + lp;
                float best = (q0 == 0) ? -INFINITY : y[o];
                int bi = (q0 == 0) ? 0 : argmax[o];
                const float bv = bias ? bias[co0 + c] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q0 + q < pool) {
                        const float v = acc[c][q] + bv;
                        if (v > best) {
                            best = v;
                            bi = lp * pool + q0 + q;
                        }
                    }
                }
                y[o] = best;
                argmax[o] = bi;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// dw[co,ci,k] = sum_b sum_lc dz[b,co,lc] * x[b,ci,lc+k],  dbias[co] = sum dz
// thread tile = (co tile of COT, ci, k tile of KT); lanes are ordered co-tile fastest so that a wave reads
// 64/ncot distinct x words (broadcast) and ncot distinct 16-byte dz groups per position.
constexpr int FILTER_PF = 12;  // dy / argmax words a thread prefetches per (batch, tile) pair
constexpr int FILTER_XF = 6;   // x words

template <int KT, int COT>
__global__ __launch_bounds__(256, 2) void conv_bwd_filter_tile_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   const int* __restrict__ argmax, int B, int Cin, int L,
                                                                   int Cout, int K, int Lp, int pool, int lsh, int ntile,
                                                                   float* __restrict__ part, float* __restrict__ part_b) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nthr = blockDim.x, tid = threadIdx.x;
    const int LPT = 1 << lsh;                  // pooled positions per tile
    const int LCT = LPT * pool;
    const int nkt = (K + KT - 1) / KT, ncot = (Cout + COT - 1) / COT;
    const int xw = LCT + nkt * KT - 1;         // x window a thread may touch
    const int xpitch = xw + 1;
    const int dpitch = ncot * COT + 4;         // dz tile row: [lc][co]
    float* dzs = smem;                         // [LCT][dpitch]
    float* xs = smem + LCT * dpitch;           // [Cin][xpitch]
    const int NTT = ncot * nkt * Cin;
    const int tt = blockIdx.x * nthr + tid;
    const bool active = tt < NTT;
    const int cot = tt % ncot, r = tt / ncot;
    const int kt = r % nkt, ci = active ? r / nkt : 0;
    const int k0 = kt * KT, co0 = cot * COT;
    const bool bias_thread = active && kt == 0 && ci == 0;
    const int ci_lo = (blockIdx.x * nthr) / (ncot * nkt);
    const int ci_hi = min(Cin - 1, (blockIdx.x * nthr + nthr - 1) / (ncot * nkt));
    const int z = blockIdx.y, R = gridDim.y;
    const int npairs = B * ntile;
    float acc[COT][KT], accb[COT];
#pragma unroll
    for (int i = 0; i < COT; ++i) {
        accb[i] = 0.f;
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = 0.f;
    }
    // The dy / argmax / x words of the NEXT (batch, tile) pair are fetched into registers before the FMA loop
    // of the current one, so the HBM latency of the staging is hidden behind compute.
    float pv[FILTER_PF], px[FILTER_XF];
    int pp[FILTER_PF];
    const int lmask = LPT - 1;
    const int ndz = Cout << lsh;
    const int nx = (ci_hi - ci_lo + 1) * xw;
    auto prefetch = [&](int pair) {
        const int b = pair / ntile, lt = pair - b * ntile;
        const int lp0 = lt * LPT, lc0 = lp0 * pool;
        const int nlp = min(LPT, Lp - lp0);
#pragma unroll
        for (int i = 0; i < FILTER_PF; ++i) {
            const int e = tid + i * nthr;
            const int co = e >> lsh, j = e & lmask;
            const bool ok = e < ndz && j < nlp;
            const long o = ((long)b * Cout + co) * Lp + lp0 + j;
            pv[i] = ok ? dy[o] : 0.f;
            pp[i] = ok ? argmax[o] - lc0 : -1;
        }
#pragma unroll
        for (int i = 0; i < FILTER_XF; ++i) {
            const int e = tid + i * nthr;
            const int row = e / xw, sx = e - row * xw;
            const int l = lc0 + sx;
            px[i] = (e < nx && l < L) ? x[((long)b * Cin + ci_lo + row) * L + l] : 0.f;
        }
    };
    if (z < npairs) prefetch(z);
    for (int pair = z; pair < npairs; pair += R) {
        const int lt = pair % ntile;
        const int nlp = min(LPT, Lp - lt * LPT);
        __syncthreads();
        {
            f32x4* d4 = reinterpret_cast<f32x4*>(dzs);
            const int n4 = LCT * dpitch / 4;
            for (int i = tid; i < n4; i += nthr) d4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < FILTER_XF; ++i) {
            const int e = tid + i * nthr;
            const int row = e / xw, sx = e - row * xw;
            if (e < nx) xs[(ci_lo + row) * xpitch + sx] = px[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < FILTER_PF; ++i) {
            const int co = (tid + i * nthr) >> lsh;
            if (pp[i] >= 0) dzs[pp[i] * dpitch + co] = pv[i];
        }
        __syncthreads();
        if (pair + R < npairs) prefetch(pair + R);
        if (active) {
            const float* xr = xs + ci * xpitch + k0;
            float win[KT];
#pragma unroll
            for (int j = 0; j < KT - 1; ++j) win[j] = xr[j];
            const int nlc = nlp * pool;
#pragma unroll 2
            for (int lc = 0; lc < nlc; ++lc) {
                win[KT - 1] = xr[lc + KT - 1];
                const f32x4* d4 = reinterpret_cast<const f32x4*>(dzs + lc * dpitch + co0);
                float d[COT];
#pragma unroll
                for (int q = 0; q < COT / 4; ++q) {
                    const f32x4 v = d4[q];
                    d[4 * q] = v[0];
                    d[4 * q + 1] = v[1];
                    d[4 * q + 2] = v[2];
                    d[4 * q + 3] = v[3];
                }
#pragma unroll
                for (int i = 0; i < COT; ++i)
#pragma unroll
                    for (int j = 0; j < KT; ++j) acc[i][j] = fmaf(d[i], win[j], acc[i][j]);
                if (bias_thread) {
#pragma unroll
                    for (int i = 0; i < COT; ++i) accb[i] += d[i];
                }
#pragma unroll
                for (int j = 0; j < KT - 1; ++j) win[j] = win[j + 1];
            }
        }
    }
    if (active) {
        const long npair_out = (long)Cin * K;
#pragma unroll
        for (int i = 0; i < COT; ++i) {
            const int co = co0 + i;
            if (co >= Cout) break;
#pragma unroll
            for (int j = 0; j < KT; ++j)
                if (k0 + j < K) part[((long)z * Cout + co) * npair_out + (long)ci * K + k0 + j] = acc[i][j];
            if (bias_thread) part_b[(long)z * Cout + co] = accb[i];
        }
    }
}

__global__ __launch_bounds__(64) void conv_bwd_filter_sum_kernel(const float* __restrict__ part, const float* __restrict__ part_b,
                                                                 int nb, long n, int Cout, float* __restrict__ dw,
                                                                 float* __restrict__ dbias) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) {
        float s = 0.f;
#pragma unroll 16
        for (int z = 0; z < nb; ++z) s += part[(long)z * n + i];
        dw[i] = s;
    }
    if (dbias != nullptr && i < Cout) {
        float s = 0.f;
#pragma unroll 16
        for (int z = 0; z < nb; ++z) s += part_b[(long)z * Cout + i];
        dbias[i] = s;
    }
}

constexpr int FILTER_SLICES = 256;  // upper bound of the reduction split (scratch is sized for it)

template <int KT, int COT>
int launch_bwd_filter(hipStream_t st, const float* x, const float* dy, const int* argmax, int B, int Cin, int L, int Cout, int K,
                      int pool, float* fpart, float* dw, float* dbias) {
    const int Lc = L - K + 1, Lp = Lc / pool;
    const int nkt = (K + KT - 1) / KT, ncot = (Cout + COT - 1) / COT;
    const int NTT = ncot * nkt * Cin;
    const int nblk = (NTT + 255) / 256;
    int nthr = ((NTT + nblk - 1) / nblk + 63) / 64 * 64;
    if (nthr > 256) nthr = 256;
    // pooled positions per tile (a power of two <= 32): the dz tile + x windows fit ~64 KB of LDS and the words of
    // one tile fit the per-thread prefetch registers
    const int rows = (nthr + ncot * nkt - 1) / (ncot * nkt) + 1;  // input channels one block can span
    int lsh = 5;
    size_t lds = 0;
    for (;; --lsh) {
        const int LCT = (1 << lsh) * pool;
        const int xw = LCT + nkt * KT - 1;
        lds = sizeof(float) * ((size_t)LCT * (ncot * COT + 4) + (size_t)Cin * (xw + 1));
        const bool fits = lds <= 64 * 1024 && ((long)Cout << lsh) <= (long)FILTER_PF * nthr &&
                          (long)(rows < Cin ? rows : Cin) * xw <= (long)FILTER_XF * nthr;
        if (fits || lsh == 0) {
            PK_REQUIRE(fits, "pk_conv1d_pool_bwd: Cin %d / Cout %d / K %d exceed the filter-gradient tile", Cin, Cout, K);
            break;
        }
    }
    const int LPT = 1 << lsh;
    const int ntile = (Lp + LPT - 1) / LPT;
    const long npairs = (long)B * ntile;
    long R = 1024 / nblk;
    if (R > FILTER_SLICES) R = FILTER_SLICES;
    if (R > npairs) R = npairs;
    if (R < 1) R = 1;
    const long n = (long)Cout * Cin * K;
    float* fpart_b = fpart + (size_t)R * n;
    static bool attr = false;
    if (!attr) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)conv_bwd_filter_tile_kernel<KT, COT>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL((conv_bwd_filter_tile_kernel<KT, COT>), dim3(nblk, (unsigned)R), dim3(nthr), lds, st, x, dy, argmax, B, Cin,
                       L, Cout, K, Lp, pool, lsh, ntile, fpart, fpart_b);
    PK_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv_bwd_filter_sum_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, fpart, fpart_b, (int)R, n,
                       Cout, dw, dbias);
    PK_LAUNCH_CHECK();
    return 0;
}

inline int64_t packed_w_floats(int Cin, int Cout, int K) {
    const int64_t a = (int64_t)((Cout + 15) / 16 * 16) * Cin * K;  // forward / CT = 16
    const int64_t b = (int64_t)((Cin + 15) / 16 * 16) * Cout * K;  // data gradient
    return a > b ? a : b;
}

}  // namespace

extern "C" int64_t pk_conv_fwd_work_floats(int Cin, int Cout, int K) { return packed_w_floats(Cin, Cout, K); }

extern "C" int pk_conv1d_pool_fwd(void* stream, const float* x, const float* w, const float* bias, int B, int Cin, int L,
                                  int Cout, int K, int pool, float* y, int32_t* argmax, float* work) {
    PK_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && K > 0 && pool > 0 && L >= K, "pk_conv1d_pool_fwd: bad geometry");
    const int Lc = L - K + 1, Lp = Lc / pool;
    PK_REQUIRE(Lp > 0, "pk_conv1d_pool_fwd: empty output");
    hipStream_t st = pk_stream(stream);
    if (pool <= 3) {
        PK_REQUIRE(work != nullptr, "pk_conv1d_pool_fwd: needs the packed-weight scratch (pk_conv_fwd_work_floats)");
        // positions [Lp*pool, Lc) are never pooled: compute Lp*pool of them
        const int Lout = Lp * pool;
        if (pool == 1) return launch_conv_tile<16, 6, true, 1>(st, x, nullptr, w, work, bias, B, Cin, Cout, K, L, Lout, pool, y, argmax, Lp);
        if (pool == 2) return launch_conv_tile<16, 6, true, 2>(st, x, nullptr, w, work, bias, B, Cin, Cout, K, L, Lout, pool, y, argmax, Lp);
        return launch_conv_tile<16, 6, true, 3>(st, x, nullptr, w, work, bias, B, Cin, Cout, K, L, Lout, pool, y, argmax, Lp);
    }
    const int span = LP_TILE * pool + K - 1;
    const size_t lds = sizeof(float) * ((size_t)CI_CHUNK * span + (size_t)CO_TILE * CI_CHUNK * K);
    PK_REQUIRE(lds <= 160 * 1024, "pk_conv1d_pool_fwd: filter length %d / pool %d exceed the LDS tile", K, pool);
    static bool attr_done = false;
    if (!attr_done) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)conv_pool_fwd_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
        attr_done = true;
    }
    dim3 grid((Lp + LP_TILE - 1) / LP_TILE, (Cout + CO_TILE - 1) / CO_TILE, B);
    hipLaunchKernelGGL(conv_pool_fwd_generic_kernel, grid, dim3(LP_TILE), lds, st, x, w, bias, B, Cin, L, Cout, K, pool, Lp, y,
                       argmax);
    PK_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t pk_conv_partial_floats(int B, int Cin, int L, int Cout, int K, int pool) {
    (void)pool;
    (void)B; (void)L;
    // the partial filter / bias sums of the reduction slices + packed weights (the un-pooled gradient never
    // exists in HBM: both backward kernels rebuild their window of it in LDS from dy + argmax)
    return (int64_t)FILTER_SLICES * Cout * ((int64_t)Cin * K + 1) + 64 + packed_w_floats(Cin, Cout, K);
}

extern "C" int pk_conv1d_pool_bwd(void* stream, const float* x, const float* w, const float* dy, const int32_t* argmax,
                                  int B, int Cin, int L, int Cout, int K, int pool, float* dw, float* dbias, float* dx,
                                  float* partial) {
    hipStream_t st = pk_stream(stream);
    PK_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && K > 0 && pool > 0 && L >= K, "pk_conv1d_pool_bwd: bad geometry");
    const int Lc = L - K + 1, Lp = Lc / pool;
    PK_REQUIRE(Lp > 0, "pk_conv1d_pool_bwd: empty output");
    PK_REQUIRE(partial != nullptr, "pk_conv1d_pool_bwd: needs the scratch buffer (pk_conv_partial_floats)");
    float* fpart = partial;
    float* wt = fpart + (size_t)FILTER_SLICES * Cout * ((size_t)Cin * K + 1) + 64;
    {
        int rc;
        if (K <= 5 || K == 10) rc = launch_bwd_filter<5, 12>(st, x, dy, argmax, B, Cin, L, Cout, K, pool, fpart, dw, dbias);
        else {
            // taps per thread: 8 or 9, whichever wastes fewer lanes (K = 129, Cout = 128: 17 x 16 = 272 thread
            // tiles = 5 waves of 64 accumulators, or 15 x 16 = 240 = 4 waves of 72)
            const int ncot = (Cout + 7) / 8;
            const long c8 = (((long)ncot * ((K + 7) / 8) * Cin + 63) / 64) * 8, c9 = (((long)ncot * ((K + 8) / 9) * Cin + 63) / 64) * 9;
            rc = c9 < c8 ? launch_bwd_filter<9, 8>(st, x, dy, argmax, B, Cin, L, Cout, K, pool, fpart, dw, dbias)
                         : launch_bwd_filter<8, 8>(st, x, dy, argmax, B, Cin, L, Cout, K, pool, fpart, dw, dbias);
        }
        if (rc) return rc;
    }
    if (dx != nullptr) {
        const int rc =
            Cin < 4 ? launch_conv_tile<1, 8, false, 1>(st, dy, argmax, w, wt, nullptr, B, Cin, Cout, K, Lc, L, pool, dx, nullptr, Lp)
                    : launch_conv_tile<16, 4, false, 1>(st, dy, argmax, w, wt, nullptr, B, Cin, Cout, K, Lc, L, pool, dx, nullptr, Lp);
        if (rc) return rc;
    }
    return 0;
}

// The data gradient alone (exact fp32): dx [B,Cin,L] from dy + argmax.  The perf-mode path (pk_conv_bf16.hip) uses it for
// layers with very few input channels (SincNet's first layer: one - a matrix-vector product per position, where an MFMA
// tile would be 15/16 padding).  work: >= pk_conv_fwd_work_floats(Cin, Cout, K) floats.
extern "C" int pk_conv1d_pool_dgrad(void* stream, const float* w, const float* dy, const int32_t* argmax, int B, int Cin, int L,
                                    int Cout, int K, int pool, float* dx, float* work) {
    hipStream_t st = pk_stream(stream);
    PK_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && K > 0 && pool > 0 && L >= K && dx && work, "pk_conv1d_pool_dgrad: bad argument");
    const int Lc = L - K + 1, Lp = Lc / pool;
    PK_REQUIRE(Lp > 0, "pk_conv1d_pool_dgrad: empty output");
    return Cin < 4 ? launch_conv_tile<1, 8, false, 1>(st, dy, argmax, w, work, nullptr, B, Cin, Cout, K, Lc, L, pool, dx, nullptr, Lp)
                   : launch_conv_tile<16, 4, false, 1>(st, dy, argmax, w, work, nullptr, B, Cin, Cout, K, Lc, L, pool, dx, nullptr, Lp);
}
