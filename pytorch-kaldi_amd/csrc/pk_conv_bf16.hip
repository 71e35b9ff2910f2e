// pk_conv_bf16.hip - perf-mode (bf16 MFMA operands, fp32 accumulate) valid 1-D convolution fused with max_pool1d:
// the SincNet / CNN stacks' F.conv1d + F.max_pool1d (neural_networks.py:1546-1552, :1655-1661, :1805-1813) on the
// matrix pipe.  pk_conv.hip holds the exact-fp32 (packed-FMA) kernels of the parity mode.
//
//   x [B,Cin,L] fp32   w [Cout,Cin,K] fp32   y [B,Cout,Lp] fp32   Lp = (L-K+1)/pool   argmax [B,Cout,Lp] (conv position)
//
// Implicit GEMM, no im2col in memory.  The reduction index is r = ci * Kt + k with the filter length padded to Kt (a
// multiple of 8, zero weights), so that the 8 consecutive r a lane feeds to v_mfma_f32_16x16x32_bf16 are 8 consecutive
// TAPS of one input channel - for an output position p that is the 8 consecutive input samples x[ci][p+k0 .. p+k0+7].
// Such a window starts at an arbitrary sample, but ds_read_b128 wants 16-byte alignment: the staged input window lives
// in LDS as EIGHT copies, copy c shifted by c samples (copy_c[j] = x[j + c]); the window at sample s is the aligned
// 16-byte word s - (s & 7) of copy s & 7.  (8 x 2 bytes per staged sample: 3-6 KB per input channel and tile.)
//
//   forward        out[p][co]  = sum_r  X[p][r] * W[co][r]              A = shifted copies of x,   B = packed weights
//   data gradient  dx[l][ci]   = sum_r' Z[l][r'] * Wt[ci][r']           the same kernel on dz (dy routed to the arg-max
//                                                                        positions, rebuilt in LDS) with the taps reversed
//   filter gradient dw[co][r]  = sum_p  dz[co][p] * x[ci][p + k]        reduction over positions: A = dz rows,
//                                                                        B = shifted copies of x (rows r, 8 positions)
// Rounding: x, w, dz enter as bf16 (round to nearest even), products accumulate in fp32 - the oracle's bf16-operand model
// in the test tree does the same.
#include "pk_common.h"

namespace {

constexpr int WP = 48;           // conv positions per wave (3 MFMA row blocks; a multiple of the pool widths 1, 2, 3, 4, 6)
constexpr int TP = 4 * WP;       // positions per workgroup
constexpr int MAXNB = 8;         // output-channel blocks of 16 per workgroup (128 channels)

__device__ __forceinline__ unsigned short f2bf(float f) { return pk_f2bf(f); }

// The eight shifted copies of one staged row (copy_c[j] = x[j + c], rows of `xpitch` elements): work item T takes the
// samples 4T .. 4T+6 (its own four and the next three) and writes, for every copy, the 8-byte word that starts inside its
// own four: word m = T - (c >> 2) of copy c holds x[4T + (c & 3) .. + 3].  (8 ds_write_b64 per four samples.)
// ptr_of(s): a VALID address for sample s (clamped), ok_of(s): whether the sample exists (else zero).  All seven loads
// are issued before any is used (the empty asm consumes them together): written as `ok ? *p : 0` the compiler puts every
// load under its own branch with its own s_waitcnt vmcnt(0) - seven dependent round trips per work item.
template <typename P, typename V>
__device__ __forceinline__ void stage_row_copies(unsigned short* base, int xpitch, int T, P ptr_of, V ok_of) {
    float raw[7];
#pragma unroll
    for (int e = 0; e < 7; ++e) raw[e] = *ptr_of(4 * T + e);
    asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]));
    unsigned short h[7];
#pragma unroll
    for (int e = 0; e < 7; ++e) h[e] = f2bf(ok_of(4 * T + e) ? raw[e] : 0.f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int m = T - (c >> 2), d = c & 3;
        if (m >= 0) {
            uint2 v;
            v.x = (unsigned)h[d] | ((unsigned)h[d + 1] << 16);
            v.y = (unsigned)h[d + 2] | ((unsigned)h[d + 3] << 16);
            *reinterpret_cast<uint2*>(base + (size_t)c * xpitch + 4 * m) = v;
        }
    }
}

// packed weights: wb[oc][r], r = ic * Kt + k', zero for k' >= K, row pitch Rp (multiple of 32)
//   mode 0 (forward):        oc = co, ic = ci, k' = k           wb = w[oc][ic][k']
//   mode 1 (data gradient):  oc = ci, ic = co, k' = K-1-k       wb = w[ic][oc][K-1-k']
__global__ void conv_w_pack_bf16_kernel(const float* __restrict__ w, int Cout, int Cin, int K, int Kt, int Rp, int NOCp,
                                        int mode, unsigned short* __restrict__ wb) {
    const int NOC = mode == 0 ? Cout : Cin, NIC = mode == 0 ? Cin : Cout;
    const long n = (long)NOCp * Rp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int oc = (int)(i / Rp), r = (int)(i - (long)oc * Rp);
        const int ic = r / Kt, k = r - ic * Kt;
        float v = 0.f;
        if (oc < NOC && ic < NIC && k < K)
            v = mode == 0 ? w[((long)oc * Cin + ic) * K + k] : w[((long)ic * Cin + oc) * K + (K - 1 - k)];
        wb[i] = f2bf(v);
    }
}

struct ConvArgs {
    const float* in;        // FWD: x [B][NIC][Lin];  DGRAD: dy [B][NIC][Lp]
    const int* amax;        // DGRAD: arg-max conv positions of dy
    const unsigned short* wb;
    const float* bias;
    float* out;             // FWD: y [B][NOC][Lp];   DGRAD: dx [B][NOC][Lout]
    int* argmax;
    int NIC, NOC, NB;       // input channels, output channels, output-channel blocks of 16
    int Lin, Lout, Lp;      // FWD: input length, conv positions, pooled positions;  DGRAD: conv positions (of dz), input length (of dx), pooled length
    int K, Kt, Rp, ICC;     // filter length, padded, packed row pitch, input channels per LDS chunk
    int pool;
    int wlen, xpitch;       // staged window length (samples, multiple of 8) and elements per copy row
    int wpitch;             // elements per staged weight row (chunk of ICC * Kt, + 8 pad)
};

// LDS: [ICC][8 copies][xpitch] bf16 | [NB*16][wpitch] bf16 | 4 waves x [16][WP + 1] fp32 (epilogue patches)
template <bool FWD>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, p0 = blockIdx.x * TP;
    unsigned short* xs = reinterpret_cast<unsigned short*>(smem);
    unsigned short* ws = xs + (size_t)a.ICC * 8 * a.xpitch;
    float* patch = reinterpret_cast<float*>(ws + (size_t)a.NB * 16 * a.wpitch) + wave * (16 * (WP + 1));
    // FWD: output position p reads x[p + k];  DGRAD: output position l reads dz[l - (K-1) + k']
    const int in0 = FWD ? p0 : p0 - (a.K - 1);
    const int g = lane >> 4, lr = lane & 15;
    const int kt8 = a.Kt >> 3;

    f32x4 acc[3][MAXNB];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < MAXNB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int ic0 = 0; ic0 < a.NIC; ic0 += a.ICC) {
        const int nic = min(a.ICC, a.NIC - ic0);
        const int rc = a.ICC * a.Kt;  // reduction elements of a chunk (channels beyond nic: zero weights, zero input)
        __syncthreads();
        // ---- stage the input window of the chunk: eight shifted bf16 copies per channel
        if (FWD) {
            const int w4 = a.wlen >> 2;
#pragma unroll 2
            for (int i = tid; i < a.ICC * w4; i += 256) {
                const int ic = i / w4, T = i - ic * w4;
                const float* row = a.in + ((long)b * a.NIC + ic0 + (ic < nic ? ic : 0)) * a.Lin;
                const bool cok = ic < nic;
                stage_row_copies(xs + (size_t)ic * 8 * a.xpitch, a.xpitch, T,
                                 [&](int s) { return row + min(max(in0 + s, 0), a.Lin - 1); },
                                 [&](int s) { return cok && in0 + s >= 0 && in0 + s < a.Lin; });
            }
        } else {
            // dz = dy routed to the arg-max positions: zero the copies, then scatter
            unsigned* z = reinterpret_cast<unsigned*>(xs);
            for (int i = tid; i < a.ICC * 8 * a.xpitch / 2; i += 256) z[i] = 0u;
            __syncthreads();
            const int lpa = max(in0, 0) / a.pool;
            const int lpb = min(a.Lp - 1, (in0 + a.wlen - 1) / a.pool);
            const int nlp = lpb - lpa + 1;
            // (eight (dy, arg-max) pairs per thread in flight: one at a time the loop was a chain of global round trips)
            for (int i0 = tid; i0 < nic * nlp; i0 += 256 * 8) {
                float v[8];
                int sidx[8], icv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = min(i0 + e * 256, nic * nlp - 1);  // (clamped: the loads are unconditional, see stage_row_copies)
                    const int ic = i / nlp, lp = lpa + (i - ic * nlp);
                    const long o = ((long)b * a.NIC + ic0 + ic) * a.Lp + lp;
                    sidx[e] = a.amax[o];
                    v[e] = a.in[o];
                    icv[e] = ic;
                }
                asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
                for (int e = 0; e < 8; ++e) sidx[e] = (i0 + e * 256 < nic * nlp) ? sidx[e] - in0 : -1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int s = sidx[e];
                    if (s >= 0 && s < a.wlen) {
                        const unsigned short h = f2bf(v[e]);
                        unsigned short* base = xs + (size_t)icv[e] * 8 * a.xpitch;
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            if (s - c >= 0) base[c * a.xpitch + (s - c)] = h;
                    }
                }
            }
        }
        // ---- stage the chunk's weights: rows of rc elements (16-byte pieces)
        {
            const int pieces = rc >> 3, total = a.NB * 16 * pieces;
            for (int i0 = tid; i0 < total; i0 += 256 * 4) {  // four 16-byte pieces per thread in flight
                uint4 v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = min(i0 + e * 256, total - 1);
                    const int row = i / pieces, pc = i - row * pieces;
                    v[e] = *reinterpret_cast<const uint4*>(a.wb + (size_t)row * a.Rp + (size_t)ic0 * a.Kt + pc * 8);
                }
                asm volatile("" : "+v"(v[0].x), "+v"(v[1].x), "+v"(v[2].x), "+v"(v[3].x));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = i0 + e * 256;
                    if (i < total) {
                        const int row = i / pieces, pc = i - row * pieces;
                        *reinterpret_cast<uint4*>(ws + (size_t)row * a.wpitch + pc * 8) = v[e];
                    }
                }
            }
        }
        __syncthreads();
        // ---- MFMA: D[position][channel] += X[position][r] * W[channel][r]
        for (int ks = 0; ks < (rc >> 5); ++ks) {
            const int r8 = ks * 4 + g;            // group of 8 reduction elements this lane feeds
            const int ic = r8 / kt8, k0 = (r8 - ic * kt8) * 8;
            bf16x8 af[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int s = wave * WP + i * 16 + lr + k0;  // window-local sample of the first tap
                const int c = s & 7;
                af[i] = *reinterpret_cast<const bf16x8*>(xs + ((size_t)ic * 8 + c) * a.xpitch + (s - c));
            }
#pragma unroll
            for (int j = 0; j < MAXNB; ++j) {
                if (j < a.NB) {
                    const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(ws + (size_t)(j * 16 + lr) * a.wpitch + ks * 32 + g * 8);
#pragma unroll
                    for (int i = 0; i < 3; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue.  acc[i][j][r]: position wave*WP + i*16 + g*4 + r, channel j*16 + lr
#pragma unroll
    for (int j = 0; j < MAXNB; ++j) {
        if (j >= a.NB) break;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) patch[lr * (WP + 1) + i * 16 + g * 4 + r] = acc[i][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private patch: no barrier)
        if (FWD) {
            // max over the pool-wide windows; first maximum wins, as torch's max_pool1d.  Lane -> (window lane & 15 [+16],
            // channels g*4 .. g*4+3): 16 lanes store 16 consecutive pooled positions of one channel
            const int nwin = WP / a.pool;
            for (int w0 = 0; w0 < nwin; w0 += 16) {
                const int win = w0 + lr;
                const int lp = (p0 + wave * WP) / a.pool + win;
                if (win < nwin && lp < a.Lp) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int ch = g * 4 + cc, oc = j * 16 + ch;
                        if (oc < a.NOC) {
                            const float* pr = patch + ch * (WP + 1) + win * a.pool;
                            float best = pr[0];
                            int bi = 0;
                            for (int q = 1; q < a.pool; ++q) {
                                const float v = pr[q];
                                if (v > best) best = v, bi = q;
                            }
                            const long o = ((long)b * a.NOC + oc) * a.Lp + lp;
                            a.out[o] = best + (a.bias ? a.bias[oc] : 0.f);
                            a.argmax[o] = lp * a.pool + bi;
                        }
                    }
                }
            }
        } else {
            // dx[b][oc][l]: lane -> (position lane & 15 + 16 q, channels g*4 .. g*4+3)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int l = p0 + wave * WP + q * 16 + lr;
                if (l < a.Lout) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int ch = g * 4 + cc, oc = j * 16 + ch;
                        if (oc < a.NOC) a.out[((long)b * a.NOC + oc) * a.Lout + l] = patch[ch * (WP + 1) + q * 16 + lr];
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------------------
// filter gradient: dw[co][ci][k] = sum_b sum_p dz[b][co][p] * x[b][ci][p + k]
// A workgroup walks (batch, position tile) pairs; per pair the dz tile [Cout][TPW] (bf16, rebuilt from dy + argmax) and
// the shifted copies of the x window of a chunk of input channels are staged; wave w owns the output-channel blocks
// w, w + 4 (, ...) and ALL tap blocks of the chunk: acc[cb][tb].  Reduction elements = positions (32 per MFMA).
// The per-workgroup partial sums go to a workspace and are added in a fixed order (deterministic).
constexpr int TPW = 128;    // positions per staged tile (4 k-steps)
constexpr int WGRAD_WGS = 512;  // workgroups of the filter gradient (two per CU: one stages while the other multiplies)
constexpr int MAXCB = 2;    // output-channel blocks per wave (4 waves x 2 x 16 = 128 channels)
constexpr int MAXTB = 9;    // tap blocks of 16 per chunk (one input channel of K <= 144, or several short ones)

struct ConvWArgs {
    const float* x;         // [B][Cin][L]
    const float* dy;        // [B][Cout][Lp]
    const int* amax;
    float* part;            // [nwg][Cout][Cin*Kt16]  (Kt16: taps padded to the chunk's tap-block layout)
    int B, Cin, Cout, L, Lc, Lp, K, pool;
    int ICC;                // input channels per chunk; a chunk's rows are (ic, k) with k padded to Kq (multiple of 16 / ICC-dependent)
    int Kq;                 // taps per channel inside a chunk (multiple of 16 when ICC == 1, else divides 16 ... see host)
    int TB;                 // tap blocks per chunk = ICC * Kq / 16
    int xlen, xpitch;       // staged x window (TPW + Kq + 8 rounded up to 8) and elements per copy row
    int zpitch;             // elements per dz row (TPW + 8)
    int items, per_wg;      // (b, tile) pairs in total and per workgroup
    int ntiles;             // position tiles per batch row
};

// TB (tap blocks per chunk) is a template parameter: with a run-time bound the accumulator array went to scratch memory
template <int TB>
__global__ __launch_bounds__(256) void conv_mfma_wgrad_kernel(ConvWArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, lr = lane & 15;
    unsigned short* zs = reinterpret_cast<unsigned short*>(smem);            // [Coutp][zpitch]
    const int Coutp = (a.Cout + 15) & ~15;
    unsigned short* xs = zs + (size_t)Coutp * a.zpitch;                      // [ICC][8][xpitch]
    const int ncb = Coutp >> 4;
    const int nchunk = (a.Cin + a.ICC - 1) / a.ICC;
    const int first = blockIdx.x * a.per_wg;
    const int last = min(a.items, first + a.per_wg);
    for (int ch = 0; ch < nchunk; ++ch) {
        const int ic0 = ch * a.ICC;
        const int nic = min(a.ICC, a.Cin - ic0);
        f32x4 acc[MAXCB][TB];
#pragma unroll
        for (int c = 0; c < MAXCB; ++c)
#pragma unroll
            for (int t = 0; t < TB; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = first; it < last; ++it) {
            const int b = it / a.ntiles, p0 = (it - b * a.ntiles) * TPW;
            __syncthreads();
            // dz tile: zero, then scatter dy to its arg-max positions (rows = output channels, 8 trailing pad elements)
            {
                unsigned* z = reinterpret_cast<unsigned*>(zs);
                for (int i = tid; i < Coutp * a.zpitch / 2; i += 256) z[i] = 0u;
            }
            // x window copies of the chunk
            {
                const int w4 = a.xlen >> 2;
#pragma unroll 2
                for (int i = tid; i < a.ICC * w4; i += 256) {
                    const int ic = i / w4, T = i - ic * w4;
                    const float* row = a.x + ((long)b * a.Cin + ic0 + (ic < nic ? ic : 0)) * a.L;
                    const bool cok = ic < nic;
                    stage_row_copies(xs + (size_t)ic * 8 * a.xpitch, a.xpitch, T,
                                     [&](int s) { return row + min(p0 + s, a.L - 1); },
                                     [&](int s) { return cok && p0 + s < a.L; });
                }
            }
            __syncthreads();
            {
                const int lpa = p0 / a.pool;
                const int lpb = min(a.Lp - 1, (p0 + TPW - 1) / a.pool);
                const int nlp = lpb - lpa + 1;
                for (int i0 = tid; i0 < a.Cout * nlp; i0 += 256 * 8) {  // eight pairs per thread in flight (see conv_mfma_kernel)
                    float v[8];
                    int sidx[8], cov[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int i = min(i0 + e * 256, a.Cout * nlp - 1);
                        const int co = i / nlp, lp = lpa + (i - co * nlp);
                        const long o = ((long)b * a.Cout + co) * a.Lp + lp;
                        sidx[e] = a.amax[o];
                        v[e] = a.dy[o];
                        cov[e] = co;
                    }
                    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
                    for (int e = 0; e < 8; ++e) sidx[e] = (i0 + e * 256 < a.Cout * nlp) ? sidx[e] - p0 : -1;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (sidx[e] >= 0 && sidx[e] < TPW) zs[(size_t)cov[e] * a.zpitch + sidx[e]] = f2bf(v[e]);
                }
            }
            __syncthreads();
            // D[co][row r of the chunk] += dz[co][p] * x[ic][p + k]  over the tile's positions
#pragma unroll
            for (int ks = 0; ks < TPW / 32; ++ks) {
                const int pp = ks * 32 + g * 8;  // first of this lane's 8 positions
                bf16x8 zf[MAXCB];
#pragma unroll
                for (int c = 0; c < MAXCB; ++c) {
                    const int cb = wave + 4 * c;
                    zf[c] = cb < ncb ? *reinterpret_cast<const bf16x8*>(zs + (size_t)(cb * 16 + lr) * a.zpitch + pp)
                                     : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
#pragma unroll
                for (int t = 0; t < TB; ++t) {
                    const int row = t * 16 + lr;          // (ic, k) row of the chunk
                    const int ic = row / a.Kq, k = row - ic * a.Kq;
                    const int s = pp + k;
                    const int cpy = s & 7;
                    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xs + ((size_t)ic * 8 + cpy) * a.xpitch + (s - cpy));
#pragma unroll
                    for (int c = 0; c < MAXCB; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[c], xf, acc[c][t], 0, 0, 0);
                }
            }
        }
        // partial sums of this workgroup: part[wg][co][ic][k]  (acc[c][t][r]: co = cb*16 + g*4 + r, row = t*16 + lr)
#pragma unroll
        for (int c = 0; c < MAXCB; ++c) {
            const int cb = wave + 4 * c;
            if (cb >= ncb) continue;
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const int row = t * 16 + lr;
                const int ic = row / a.Kq, k = row - ic * a.Kq;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = cb * 16 + g * 4 + r;
                    if (co < a.Cout && ic < nic && k < a.K)
                        a.part[(((long)blockIdx.x * a.Cout + co) * a.Cin + ic0 + ic) * a.K + k] = acc[c][t][r];
                }
            }
        }
    }
}

// dw = sum over workgroups of part (fixed order); db = sum_b sum_lp dy (bias gradient)
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ part, int nwg, long n, float* __restrict__ dw) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        int w = 0;
        for (; w + 8 <= nwg; w += 8) {  // eight loads in flight; the order of the additions stays fixed
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = part[(long)(w + e) * n + i];
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e];
        }
        for (; w < nwg; ++w) s += part[(long)w * n + i];
        dw[i] = s;
    }
}

inline int round_up(int a, int m) { return (a + m - 1) / m * m; }

struct FwdGeom {
    int Kt, Rp, ICC, NB, wlen, xpitch, wpitch;
    size_t lds;
};
// geometry of conv_mfma_kernel for (NIC input channels, NOC output channels, filter length K)
inline size_t fwd_lds(const FwdGeom& q, int icc) {
    return (size_t)icc * 8 * q.xpitch * 2 + (size_t)q.NB * 16 * (icc * q.Kt + 8) * 2 + 4 * 16 * (WP + 1) * 4;
}
inline bool fwd_geom(int NIC, int NOC, int K, FwdGeom& q) {
    q.NB = (NOC + 15) / 16;
    if (q.NB > MAXNB) return false;
    // a chunk's reduction length ICC * Kt must be a multiple of 32 (one MFMA k-step): pad the taps to 8 and take 32 / gcd
    // channels per chunk - or, when the layer has fewer channels than that (SincNet's first layer: one), pad the taps to 32
    q.Kt = round_up(K, 8);
    int icc = 1;
    while ((icc * q.Kt) % 32 != 0) ++icc;  // 1, 2 or 4
    if (icc > NIC) {
        q.Kt = round_up(K, 32);
        icc = 1;
    }
    q.wlen = round_up(TP + q.Kt + 8, 8);
    q.xpitch = q.wlen + 8;
    // more channels per chunk (fewer barriers) while the reduction stays <= 512 elements and the LDS <= 96 KB
    while (icc * 2 * q.Kt <= 512 && icc * 2 <= round_up(NIC, icc) && fwd_lds(q, icc * 2) <= 96 * 1024) icc *= 2;
    q.ICC = icc;
    q.Rp = round_up(round_up(NIC, icc) * q.Kt, 32);
    q.wpitch = icc * q.Kt + 8;
    q.lds = fwd_lds(q, icc);
    return q.lds <= 150 * 1024;
}

}  // namespace

// Does the bf16 MFMA path cover this layer?  (pool widths that divide 48, up to 128 output channels, LDS budget)
extern "C" int pk_conv_bf16_covers(int Cin, int Cout, int K, int pool) {
    FwdGeom f, d;
    if (pool < 1 || (WP % pool) != 0 || Cin < 1 || Cout < 1 || K < 1) return 0;
    if (!fwd_geom(Cin, Cout, K, f) || !fwd_geom(Cout, Cin, K, d)) return 0;
    if (K > 8 && round_up(K, 16) > MAXTB * 16) return 0;  // filter gradient: one filter's taps in at most nine tap blocks
    if ((Cout + 15) / 16 > 4 * MAXCB) return 0;
    return 1;
}

// scratch of one call (bf16 packed weights of the forward / data-gradient pass, the filter gradient's partial sums), floats
extern "C" int64_t pk_conv_bf16_work_floats(int B, int Cin, int L, int Cout, int K, int pool, int backward) {
    (void)B; (void)L;
    FwdGeom f, d;
    if (!pk_conv_bf16_covers(Cin, Cout, K, pool)) return 0;
    fwd_geom(Cin, Cout, K, f);
    fwd_geom(Cout, Cin, K, d);
    const int64_t packed = ((int64_t)f.NB * 16 * f.Rp + (int64_t)d.NB * 16 * d.Rp) / 2 + 64;
    const int64_t part = backward ? (int64_t)WGRAD_WGS * Cout * Cin * K : 0;
    return packed + part + 64;
}

extern "C" int pk_conv1d_pool_fwd_bf16(void* stream, const float* x, const float* w, const float* bias, int B, int Cin, int L,
                                       int Cout, int K, int pool, float* y, int32_t* argmax, float* work) {
    PK_REQUIRE(pk_conv_bf16_covers(Cin, Cout, K, pool), "pk_conv1d_pool_fwd_bf16: layer not covered (Cin %d Cout %d K %d pool %d)", Cin, Cout, K, pool);
    PK_REQUIRE(L >= K, "pk_conv1d_pool_fwd_bf16: input shorter than the filter");
    hipStream_t st = pk_stream(stream);
    FwdGeom q;
    fwd_geom(Cin, Cout, K, q);
    const int Lc = L - K + 1, Lp = Lc / pool;
    if (Lp <= 0 || B <= 0) return 0;
    unsigned short* wb = reinterpret_cast<unsigned short*>(work);
    {
        const long n = (long)q.NB * 16 * q.Rp;
        long blocks = (n + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(conv_w_pack_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, Cout, Cin, K, q.Kt, q.Rp, q.NB * 16, 0, wb);
        PK_LAUNCH_CHECK();
    }
    ConvArgs a;
    a.in = x; a.amax = nullptr; a.wb = wb; a.bias = bias; a.out = y; a.argmax = argmax;
    a.NIC = Cin; a.NOC = Cout; a.NB = q.NB; a.Lin = L; a.Lout = Lc; a.Lp = Lp;
    a.K = K; a.Kt = q.Kt; a.Rp = q.Rp; a.ICC = q.ICC; a.pool = pool;
    a.wlen = q.wlen; a.xpitch = q.xpitch; a.wpitch = q.wpitch;
    static size_t granted = 0;
    if (granted < q.lds) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)conv_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        granted = 150 * 1024;
    }
    // only pooled positions are produced: the tiles cover Lp * pool conv positions
    dim3 grid((unsigned)((Lp * pool + TP - 1) / TP), (unsigned)B);
    hipLaunchKernelGGL(conv_mfma_kernel<true>, grid, dim3(256), q.lds, st, a);
    PK_LAUNCH_CHECK();
    return 0;
}

// dw (and dx when non-null) from dy [B,Cout,Lp] + argmax; db is NOT produced here (the caller's column sum of dy)
extern "C" int pk_conv1d_pool_bwd_bf16(void* stream, const float* x, const float* w, const float* dy, const int32_t* argmax,
                                       int B, int Cin, int L, int Cout, int K, int pool, float* dw, float* dx, float* work) {
    PK_REQUIRE(pk_conv_bf16_covers(Cin, Cout, K, pool), "pk_conv1d_pool_bwd_bf16: layer not covered");
    hipStream_t st = pk_stream(stream);
    const int Lc = L - K + 1, Lp = Lc / pool;
    if (Lp <= 0 || B <= 0) return 0;
    FwdGeom f, d;
    fwd_geom(Cin, Cout, K, f);
    fwd_geom(Cout, Cin, K, d);
    unsigned short* wbd = reinterpret_cast<unsigned short*>(work) + (size_t)f.NB * 16 * f.Rp;
    float* part = work + ((size_t)f.NB * 16 * f.Rp + (size_t)d.NB * 16 * d.Rp) / 2 + 64;
    if (dx != nullptr) {
        {
            const long n = (long)d.NB * 16 * d.Rp;
            long blocks = (n + 255) / 256;
            if (blocks > 1024) blocks = 1024;
            hipLaunchKernelGGL(conv_w_pack_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, Cout, Cin, K, d.Kt, d.Rp, d.NB * 16, 1, wbd);
            PK_LAUNCH_CHECK();
        }
        ConvArgs a;
        a.in = dy; a.amax = argmax; a.wb = wbd; a.bias = nullptr; a.out = dx; a.argmax = nullptr;
        a.NIC = Cout; a.NOC = Cin; a.NB = d.NB; a.Lin = Lc; a.Lout = L; a.Lp = Lp;
        a.K = K; a.Kt = d.Kt; a.Rp = d.Rp; a.ICC = d.ICC; a.pool = pool;
        a.wlen = d.wlen; a.xpitch = d.xpitch; a.wpitch = d.wpitch;
        static size_t granted = 0;
        if (granted < d.lds) {
            PK_CHECK_HIP(hipFuncSetAttribute((const void*)conv_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            granted = 150 * 1024;
        }
        dim3 grid((unsigned)((L + TP - 1) / TP), (unsigned)B);
        hipLaunchKernelGGL(conv_mfma_kernel<false>, grid, dim3(256), d.lds, st, a);
        PK_LAUNCH_CHECK();
    }
    // ---- filter gradient
    ConvWArgs q;
    q.x = x; q.dy = dy; q.amax = argmax; q.part = part;
    q.B = B; q.Cin = Cin; q.Cout = Cout; q.L = L; q.Lc = Lc; q.Lp = Lp; q.K = K; q.pool = pool;
    // rows of a chunk: (ic, k), k padded to Kq.  One long filter per chunk (Kq = K rounded up to 16, up to 144 taps), or
    // several short ones (Kq = 8 or 16: 16 / Kq ... channels per tap block)
    if (K > 8) {
        q.Kq = round_up(K, 16);
        q.ICC = 1;
        while ((q.ICC + 1) * q.Kq <= MAXTB * 16 && q.ICC + 1 <= Cin) ++q.ICC;
    } else {
        q.Kq = 8;
        q.ICC = 2;
        while ((q.ICC + 2) * q.Kq <= MAXTB * 16 && q.ICC + 2 <= round_up(Cin, 2)) q.ICC += 2;
    }
    q.TB = q.ICC * q.Kq / 16;
    PK_REQUIRE(q.TB >= 1 && q.TB <= MAXTB && (Cout + 15) / 16 <= 4 * MAXCB, "pk_conv1d_pool_bwd_bf16: filter gradient geometry not covered (K %d Cout %d)", K, Cout);
    q.xlen = round_up(TPW + q.Kq + 8, 8);
    q.xpitch = q.xlen + 8;
    q.zpitch = TPW + 8;
    q.ntiles = (Lp * pool + TPW - 1) / TPW;
    q.items = B * q.ntiles;
    int nwg = q.items < WGRAD_WGS ? q.items : WGRAD_WGS;
    q.per_wg = (q.items + nwg - 1) / nwg;
    nwg = (q.items + q.per_wg - 1) / q.per_wg;
    const size_t lds = (size_t)((Cout + 15) & ~15) * q.zpitch * 2 + (size_t)q.ICC * 8 * q.xpitch * 2;
    PK_REQUIRE(lds <= 150 * 1024, "pk_conv1d_pool_bwd_bf16: LDS budget");
    typedef void (*WKernel)(ConvWArgs);
    static const WKernel wk[MAXTB] = {conv_mfma_wgrad_kernel<1>, conv_mfma_wgrad_kernel<2>, conv_mfma_wgrad_kernel<3>,
                                      conv_mfma_wgrad_kernel<4>, conv_mfma_wgrad_kernel<5>, conv_mfma_wgrad_kernel<6>,
                                      conv_mfma_wgrad_kernel<7>, conv_mfma_wgrad_kernel<8>, conv_mfma_wgrad_kernel<9>};
    static bool grantedw[MAXTB] = {false, false, false, false, false, false, false, false, false};
    if (!grantedw[q.TB - 1]) {
        PK_CHECK_HIP(hipFuncSetAttribute((const void*)wk[q.TB - 1], hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        grantedw[q.TB - 1] = true;
    }
    hipLaunchKernelGGL(wk[q.TB - 1], dim3((unsigned)nwg), dim3(256), lds, st, q);
    PK_LAUNCH_CHECK();
    const long n = (long)Cout * Cin * K;
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, part, nwg, n, dw);
    PK_LAUNCH_CHECK();
    return 0;
}
