/* pk_amd.h - C ABI of libpk_amd.so: the MI355X (gfx950) engine behind the
 * PyTorch-Kaldi `neural_networks.py` hot path.
 *
 * Boundary rules:
 *  - plain pointers + sizes only; no torch types.  Every pointer is a DEVICE
 *    pointer unless its name starts with `h_`.  The caller (the Python host
 *    layer, or any other FFI) owns all memory; the library borrows it for the
 *    duration of the work it enqueues on `stream`.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream()).
 *  - every entry point returns 0 on success, non-zero on failure and never
 *    throws; pk_last_error() gives the message of the calling thread's last
 *    failure.
 *  - all matrices are fp32 row-major.  `prec` selects the MFMA operand type of
 *    the GEMM-shaped work: PK_PREC_F32 = exact fp32 (v_mfma_f32_32x32x2_f32 /
 *    16x16x4_f32), PK_PREC_BF16 = bf16 operands with fp32 accumulation.
 *
 * Each group cites the reference code it replaces (paths relative to the
 * mravanelli/pytorch-kaldi checkout).
 */
#ifndef PK_AMD_H
#define PK_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PK_PREC_F32 0
#define PK_PREC_BF16 1

/* activations: neural_networks.py:36-57 (act_fun) */
#define PK_ACT_LINEAR 0
#define PK_ACT_RELU 1
#define PK_ACT_TANH 2
#define PK_ACT_SIGMOID 3
#define PK_ACT_LEAKY_RELU 4 /* slope 0.2 */
#define PK_ACT_ELU 5

/* recurrent cells: gate order of the concatenated [G*H] axis */
#define PK_CELL_LIGRU 0  /* [z, a]        neural_networks.py:1133-1136 */
#define PK_CELL_RNN 1    /* [a]           neural_networks.py:1441-1442 */
#define PK_CELL_LSTM 2   /* [f, i, o, c]  neural_networks.py:460-464   */
#define PK_CELL_GRU 3    /* [z, r, a]     neural_networks.py:632-636   */
#define PK_CELL_MINGRU 4 /* [z, a]        neural_networks.py:1294-1297 */

/* recurrence algorithms */
#define PK_REC_STEPWISE 0   /* one GEMM + one gate kernel per time step */
#define PK_REC_PERSISTENT 1 /* one launch per layer, CU clusters, h_t exchanged through L2 */

/* ---- library ---------------------------------------------------------- */
int pk_version(void);
const char* pk_last_error(void);
/* number of CUs of the current device (cached) */
int pk_num_cu(void);

/* ---- GEMM: replaces nn.Linear forward and its autograd (F.linear / addmm /
 * mm), neural_networks.py:111,139-148 (MLP), :432-435, :609-611, :1114-1115
 * (input projections) and the per-step recurrent Linear.
 *   C[M,N] = alpha * sum_k A(m,k) * B(k,n) + beta * C + bias[n]
 * A(m,k) = A[m*a_rs + k*a_cs], B(k,n) = B[k*b_rs + n*b_cs] (one stride of each
 * must be 1), C row-major with leading dimension ldc.  bias may be NULL.
 * splitk > 1 partitions K over gridDim.z through `workspace`
 * (>= splitk*M*N floats, deterministic reduction); splitk <= 1 ignores it. */
int pk_gemm(void* stream, int prec, int M, int N, int K, float alpha, const float* A, int64_t a_rs,
            int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs, float beta, float* C, int64_t ldc,
            const float* bias, int splitk, float* workspace);
/* tests / tools: 1 = the register-staged first form of the exact-fp32 kernel for every shape, 0 = automatic (the LDS-DMA
 * form wherever the operands are 16-byte aligned; also PK_EXPERIMENT f32_dma=0).  Both forms compute every output element
 * as one fmaf chain over ascending k: bit-identical results. */
void pk_gemm_f32_set_form(int form);

/* ---- perf-mode GEMM on bf16 operands resident in HBM (fp32 accumulate and output); same reference
 * call sites as pk_gemm.  a_kc != 0: A is stored [M][lda] (k contiguous), else [K][lda] (m
 * contiguous - the dW / dU shapes whose reduction runs over the T*B rows); likewise B with N / ldb.
 * Bases 16-byte aligned, pitches multiples of 8 elements; elements between K and the next multiple of
 * 8 inside a k-contiguous row must be zero (pk_cvt_bf16 writes them so).  splitk as pk_gemm. */
/* rows of the block tile of the k-contiguous shapes (128); the k-major x k-major weight-gradient shapes with at least
 * 1024 rows and columns take a 256 x 256 tile (eight waves, eight phases per pair of k-tiles) */
int pk_gemm_bf16_tile_m(int M);
/* the split-K factor the library recommends for a k-major x k-major product of this shape (1 for short reductions) */
int pk_gemm_bf16_auto_splitk(int M, int N, int K);
/* tests / tools: 128 or 256 forces that block tile for every shape, 0 = automatic (also PK_EXPERIMENT gemm_tile) */
/* split-K factor for a k-major x k-major product that will only get `cus` CUs (0 = the whole device) */
int pk_gemm_bf16_auto_splitk_cus(int M, int N, int K, int cus);
void pk_gemm_bf16_set_tile(int tile);
int pk_gemm_bf16(void* stream, int M, int N, int K, float alpha, const uint16_t* A, int64_t lda, int a_kc,
                 const uint16_t* B, int64_t ldb, int b_kc, float beta, float* C, int64_t ldc, const float* bias,
                 int splitk, float* workspace);
/* The projection of a BatchNorm-ed layer (neural_networks.py:1114-1124: bn_wh(wh(x)), bn_wz(wz(x)) over all T*B
 * rows): C = alpha * A.B + bias with the per-column statistics of C taken in the GEMM epilogue instead of by a second
 * pass over C.  Shapes that run on the 256-tile produce them: *row_blocks = number of 256-row tiles and
 * stats[row_blocks][N][3] = (rows, mean, M2); fold with pk_bn_stats_merge.  Other shapes: *row_blocks = 0, plain GEMM,
 * the caller runs pk_bn_stats.  stats: pk_gemm_bf16_stats_floats(M, N) floats. */
int64_t pk_gemm_bf16_stats_floats(int M, int N);
int pk_gemm_bf16_stats(void* stream, int M, int N, int K, float alpha, const uint16_t* A, int64_t lda, int a_kc,
                       const uint16_t* B, int64_t ldb, int b_kc, float* C, int64_t ldc, const float* bias, float* stats,
                       int* row_blocks);
/* fp32 [rows][ld_src] -> bf16 [rows][ld_dst].  The source columns are nseg segments of seglen values;
 * segment s lands at destination column s*segpad (e.g. the two direction halves of a layer output,
 * 550 -> 576, so that each half starts 16-byte aligned); all other destination elements are zero. */
int pk_cvt_bf16(void* stream, const float* src, int64_t ld_src, int64_t rows, int nseg, int seglen, int segpad,
                uint16_t* dst, int64_t ld_dst);

/* ---- column statistics / BatchNorm: replaces nn.BatchNorm1d(momentum=0.05)
 * at neural_networks.py:85,105,142-145 (MLP) and :438-450, :614-623,
 * :1118-1124 (per-gate BN over the T*rows projection rows).
 * pk_bn_stats: per column mean and biased variance of x[M,N] (+ optional
 * second operand x2 added element-wise first); robust pairwise/Chan merge.
 * `partial` is scratch of >= pk_bn_partial_floats(M,N) floats. */
int64_t pk_bn_partial_floats(int64_t M, int64_t N);
int pk_bn_stats(void* stream, const float* x, int64_t ldx, int64_t M, int64_t N, float* partial, float* mean,
                float* var);
/* the second half of pk_bn_stats on its own: fold rb rows of (rows, mean, M2) partials per column */
int pk_bn_stats_merge(void* stream, const float* partial, int rb, int64_t N, float* mean, float* var);
/* scale = gamma * rsqrt(var+eps), shift = beta - mean*scale (gamma/beta NULL = 1/0);
 * if running_mean != NULL also updates running stats with `momentum` and the
 * unbiased factor count/(count-1) (count = rows that the reference would have
 * normalised over, e.g. 2*T*B for a bidirectional layer). */
int pk_bn_finalize(void* stream, int64_t N, const float* mean, const float* var, const float* gamma,
                   const float* beta, float eps, float* scale, float* shift, float* running_mean,
                   float* running_var, float momentum, double count);
/* the same over the G <= 4 concatenated gates of a recurrent layer (columns [g*H, (g+1)*H) belong to gate g) whose
 * running statistics sit in one BatchNorm1d per gate (neural_networks.py:1052-1055): running_mean / running_var /
 * num_batches are host arrays of G device pointers (num_batches or its entries may be NULL); always updates. */
int pk_bn_finalize_gates(void* stream, int G, int H, const float* mean, const float* var, const float* gamma,
                         const float* beta, float eps, float* scale, float* shift, float* const* running_mean,
                         float* const* running_var, int64_t* const* num_batches, float momentum, double count);
/* pk_bn_stats_merge followed by pk_bn_finalize_gates, as one launch (the [rb][G*H][3] partials pk_gemm_bf16_stats wrote). */
int pk_bn_stats_merge_finalize_gates(void* stream, const float* partial, int rb, int G, int H, float* mean, float* var,
                                     const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                     float* const* running_mean, float* const* running_var, int64_t* const* num_batches,
                                     float momentum, double count);
/* y = dropmask * act(x*scale[n] + shift[n]); scale/shift NULL = identity;
 * mask NULL = no dropout (mask holds 0 or 1/(1-p)).  In-place allowed. */
int pk_affine_act_fwd(void* stream, const float* x, int64_t ldx, int64_t M, int64_t N, const float* scale,
                      const float* shift, int act, const float* mask, float* y, int64_t ldy);
/* g = dy * mask * act'(y_saved) where y_saved is the pre-dropout activation
 * output `a` (act' is evaluated from the output).  In-place allowed. */
int pk_act_bwd(void* stream, const float* dy, const float* a, const float* mask, int act, int64_t n, float* g);
/* BatchNorm backward over g (+ optional g2 added element-wise, used for the
 * two time directions): sum_g[n] = sum_m g, sum_gx[n] = sum_m g * xhat,
 * xhat = (x - mean) * invstd.  partial >= pk_bn_partial_floats floats. */
int pk_bn_bwd_reduce(void* stream, const float* g, const float* g2, int64_t ldg, const float* x, int64_t ldx,
                     int64_t M, int64_t N, const float* mean, const float* var, float eps, float* partial,
                     float* sum_g, float* sum_gx);
/* dx = gamma*invstd * (g - sum_g/count - xhat*sum_gx/count)  (g = g + g2). */
int pk_bn_bwd_apply(void* stream, const float* g, const float* g2, int64_t ldg, const float* x, int64_t ldx,
                    int64_t M, int64_t N, const float* mean, const float* var, float eps, const float* gamma,
                    const float* sum_g, const float* sum_gx, double count, float* dx, int64_t lddx);
/* Perf mode: the whole BatchNorm backward of a recurrent layer's projections from the bf16 gate
 * gradients pk_rec_bwd_bf16 publishes (g0 / g1: one slab per direction, [M][g_pitch], gate g at
 * column g*Hp; g1 may be NULL).  Writes sum_g / sum_gx [G*H] (= dbeta / dgamma; sum_g = the bias
 * gradient when mean == NULL, i.e. no BatchNorm: then dx = g0 + g1) and the projection gradient as
 * bf16 in the plain layout out[M][out_pitch] (column g*H + j, pad columns zeroed) that the dX / dW
 * GEMMs (pk_gemm_bf16) read.  partial: >= pk_bn_partial_floats(M, G*H) floats.
 * acc_beta / acc_gamma (NULL or [G*H]): sum_g / sum_gx are also ADDED to these in place (the flat gradient of the
 * BatchNorm shifts / scales of the layer's gates - or of its biases without BatchNorm: acc_beta - when they lie
 * back to back). */
int pk_bn_bwd_bf16(void* stream, const uint16_t* g0, const uint16_t* g1, int64_t g_pitch, int G, int H, const float* x,
                   int64_t ldx, int64_t M, const float* mean, const float* var, float eps, const float* gamma,
                   double count, float* partial, float* sum_g, float* sum_gx, uint16_t* out, int64_t out_pitch,
                   float* acc_beta, float* acc_gamma);
/* Backward of drop(act(bn(z))) for a batch of up to 128 rows in ONE launch (an MLP layer at the recipes' batch size;
 * neural_networks.py:139-148 backwards): g = dy * mask * act'(a), sum_g / sum_gx [N] = the BatchNorm reductions
 * (d beta, d gamma), dz = gamma * invstd * (g - sum_g / M - xhat * sum_gx / M) as bf16 [M][ldb] (pad columns zero) and,
 * when dz != NULL, fp32 [M][N].  acc_beta / acc_gamma (NULL or [N]): the sums are also added to these in place.
 * db / acc_bias (NULL or [N]): the column sums of dz (gradient of the Linear bias in front of the BatchNorm,
 * neural_networks.py:120) written / accumulated in place. */
int pk_bn_act_bwd_small_covers(int64_t M, int64_t N);
int pk_bn_act_bwd_small(void* stream, const float* dy, const float* a, const float* mask, int act, const float* z,
                        const float* mean, const float* var, float eps, const float* gamma, int64_t M, int64_t N,
                        uint16_t* dzb, int64_t ldb, float* dz, float* sum_g, float* sum_gx, float* acc_beta,
                        float* acc_gamma, float* db, float* acc_bias);
/* column sums of g (+g2): bias gradient when there is no BatchNorm. */
int pk_colsum(void* stream, const float* g, const float* g2, int64_t ldg, int64_t M, int64_t N, float* partial,
              float* out);
/* out = a + b (element-wise, n floats) */
int pk_add(void* stream, const float* a, const float* b, int64_t n, float* out);

/* ---- the reference's drop-mask stream on the device.  neural_networks.py:1102-1107 (and :430-441, :604-615, :1266-1277,
 * :1411-1422) draw a layer's mask with torch.bernoulli(torch.Tensor(rows, H).fill_(1 - p)) on the global CPU generator: one
 * 32-bit mt19937 output per element, u = (y & 0xFFFFFF) * 2^-24, mask = u < 1 - p.  state [626] (device) = the engine's 624
 * words, `left`, `next` (the fields of the engine behind the CPU generator: bytes 24.., 8 and 16 of its saved state); the call draws n elements from it
 * exactly as that loop would and leaves the advanced state behind.  out: 1.0 / 0.0. */
int pk_mt19937_bernoulli(void* stream, uint32_t* state, int64_t n, float keep, float* out);

/* ---- SincNet's band-pass bank (neural_networks.py:1789-1800: SincConv.forward up to `self.filters`): N filters of K (odd)
 * taps from 2 x N parameters, forward and backward as one launch each.  n_ [K] and window [K] are the module's buffers
 * (n_ = (k - (K-1)/2) / sample_rate; Hamming window), min_low = min_low_hz / sample_rate, min_band likewise.  Forward
 * follows the reference operation by operation in fp32 (the right half of a filter repeats the left values, the maximum
 * takes the first index on ties); it saves the maxima mx [N] and their taps kstar [N] for backward, which is analytic. */
int pk_sinc_bank_fwd(void* stream, const float* low_hz, const float* band_hz, const float* n_, const float* window, int N, int K,
                     float sample_rate, float min_low, float min_band, float* filt, float* mx, int32_t* kstar);
int pk_sinc_bank_bwd(void* stream, const float* g, const float* low_hz, const float* band_hz, const float* n_, const float* window,
                     const float* mx, const int32_t* kstar, int N, int K, float sample_rate, float min_low, float min_band,
                     float* dlow, float* dband);

/* ---- the tail of a conv layer in one launch: drop(act(LayerNorm(z))) with the CNN / SincNet flavour of the reference's
 * LayerNorm (features [C, L], statistics over the last dim: neural_networks.py:1510-1512, 1546-1552, 1639-1641,
 * 1655-1661).  z, a, y, mask: [B, C, L]; gamma, beta: [C, L]; mean, rinv: [B * C] (saved for backward).
 * a = act(LN(z)) is always written (backward takes the activation's derivative from it); y = a * mask only with a mask
 * (both NULL otherwise).  Backward: dz, and pg [B][2][C][L] = (g * xhat, g) with g = dy * mask * act'(a): pk_colsum over the
 * B rows of pg gives (d gamma, d beta). */
int pk_ln_last_act_drop_fwd(void* stream, const float* z, int64_t B, int C, int L, const float* gamma, const float* beta,
                            float eps, int act, const float* mask, float* a, float* y, float* mean, float* rinv);
int pk_ln_last_act_drop_bwd(void* stream, const float* dy, const float* z, const float* a, const float* mask, int64_t B, int C,
                            int L, const float* gamma, const float* mean, const float* rinv, float eps, int act, float* dz,
                            float* pg);

/* ---- LayerNorm: neural_networks.py:23-33 (unbiased std, eps added to std).
 * Rows of length F; saves mean and 1/(std+eps) per row for backward. */
int pk_layernorm_fwd(void* stream, const float* x, int64_t rows, int64_t F, const float* gamma, const float* beta,
                     float eps, float* y, float* mean, float* rinv);
/* dx; dgamma/dbeta are produced as per-row-block partials reduced by pk_colsum
 * on the caller side: dgamma_rows[r,f] = dy*xhat, dbeta = dy (the kernel writes
 * xhat*dy into `dgx` so that colsum(dgx) = dgamma, colsum(dy) = dbeta). */
int pk_layernorm_bwd(void* stream, const float* dy, const float* x, int64_t rows, int64_t F, const float* gamma,
                     const float* mean, const float* rinv, float eps, float* dx, float* dgx);

/* ---- LogSoftmax(dim=1): neural_networks.py:53-54 ('softmax' activation) */
int pk_logsoftmax_fwd(void* stream, const float* x, int64_t rows, int64_t N, float* y);
int pk_logsoftmax_bwd(void* stream, const float* dy, const float* y, int64_t rows, int64_t N, float* dx);
/* the same backward writing dx at a pitch of lddx floats (N rounded up to a multiple of 4: 16-byte aligned rows for the
 * dX / dW GEMMs of the Linear in front - 1938 senones; pad columns zeroed); N <= 2048 */
int pk_logsoftmax_bwd_ld(void* stream, const float* dy, const float* y, int64_t rows, int64_t N, float* dx, int64_t lddx);
/* the same forward over an input whose rows sit at a pitch of ldx floats (the padded output of the head's GEMM) */
int pk_logsoftmax_fwd_ld(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t N, float* y);
/* Perf mode, Linear -> LogSoftmax heads (neural_networks.py:139-148 with dnn_act = softmax; the cost the reference
 * back-propagates through it: utils.py:2361): dz = dy - exp(y) * rowsum(dy) written once as the bf16 operand of the
 * dX / dW GEMMs (pitch ldb elements, a multiple of 8, pad columns zero) together with its fp32 column sums (the
 * bias gradient).  N <= 2048.  partial: pk_logsoftmax_bwd_bf16_partial_floats(rows, N) floats of workspace. */
int64_t pk_logsoftmax_bwd_bf16_partial_floats(int64_t rows, int64_t N);
int pk_logsoftmax_bwd_bf16(void* stream, const float* dy, const float* y, int64_t rows, int64_t N, uint16_t* dxb,
                           int64_t ldb, float* partial, float* colsum);
/* The cost lines on such a head (utils.py:2361-2367: loss = NLLLoss()(out, lab), err = mean(argmax(out, 1) != lab))
 * in one pass over the log-posteriors y [rows][N], N <= 2048, labels int64 on the device.
 * out4 (device) = { mean of -y[r][lab[r]] over the rows whose label is not ignore_index, error rate over all rows,
 * number of counted rows, number of labels outside [0, N) (the caller raises on it) }.
 * partial: pk_nll_err_partial_floats(rows) floats.  * loss_out (NULL or one float): a second copy of out4[0]; bad_acc (NULL or one float): += out4[3] in place. */
int64_t pk_nll_err_partial_floats(int64_t rows);
int pk_nll_err_fwd(void* stream, const float* y, const int64_t* lab, int64_t ignore_index, int64_t rows, int64_t N,
                   float* partial, float* out4, float* loss_out, float* bad_acc);
/* The same output together with the arg-max position of every row (first index on ties; a NaN row: column 0), and the
 * cost of such an output: cost_nll / cost_err (utils.py:2361-2367) as ONE gathered load and one compare per row instead of
 * a second pass over the [rows][N] log-posteriors.  amax: [rows] int32.  N <= 2048. */
int pk_logsoftmax_fwd_ld_argmax(void* stream, const float* x, int64_t ldx, int64_t rows, int64_t N, float* y, int32_t* amax);
int pk_nll_err_fwd_argmax(void* stream, const float* y, const int64_t* lab, const int32_t* amax, int64_t ignore_index,
                          int64_t rows, int64_t N, float* partial, float* out4, float* loss_out, float* bad_acc);
/* ... and its backward joined with the LogSoftmax backward: the one-hot gradient of the mean NLL is never written;
 * dz = (dloss / count) * (exp(y) - onehot(lab)) as bf16 plus its column sums.  dloss, count: device scalars. */
int pk_nll_logsoftmax_bwd_bf16(void* stream, const float* y, const int64_t* lab, const float* dloss, const float* count,
                               int64_t ignore_index, int64_t rows, int64_t N, uint16_t* dxb, int64_t ldb, float* partial,
                               float* colsum);
/* The two bf16 backward passes above writing into a COLUMN SLICE of a wider buffer (ldb columns per row: values + zero
 * padding; rows `pitch` elements apart).  Several output layers on one input (the senone and the monophone head of every
 * shipped recipe: cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg, [model] out_dnn2 / out_dnn3 on out_dnn1) then share ONE
 * concatenated operand and their input gradients are ONE GEMM over the concatenated reduction instead of one GEMM per
 * head accumulating into the same 282 MB (neural_networks.py:139-148 run backward by autograd, which adds the heads'
 * input gradients element-wise). */
int pk_logsoftmax_bwd_bf16_p(void* stream, const float* dy, const float* y, int64_t rows, int64_t N, uint16_t* dxb,
                             int64_t ldb, int64_t pitch, float* partial, float* colsum);
int pk_nll_logsoftmax_bwd_bf16_p(void* stream, const float* y, const int64_t* lab, const float* dloss, const float* count,
                                 int64_t ignore_index, int64_t rows, int64_t N, uint16_t* dxb, int64_t ldb, int64_t pitch,
                                 float* partial, float* colsum);

/* ---- recurrent layers (LSTM / GRU / liGRU / minimalGRU / RNN time loops):
 * neural_networks.py:457-469, 629-641, 1130-1141, 1291-1302, 1438-1447, with
 * the bidirectional pack/unpack of :415-417/:475-478 (cat + flip) folded into
 * the indexing.
 *
 * Geometry: T steps, B sequences, `bidir` (0/1): R = B*(1+bidir) rows; rows
 * >= B are the time-reversed copies.  H units, G gates (cell-dependent).
 *   P      [T*B, G*H]  raw input projections of the NON-duplicated batch
 *                      (row t*B+b); the kernel applies pscale/pshift [G*H]
 *                      (BatchNorm folded, or 1/bias) while loading, and reads
 *                      row (T-1-t)*B+b for the reversed half.
 *   U      [G*H, H]    recurrent weights, gate-major (nn.Linear layout).
 *   mask   [R, H] or NULL with mask_scalar (test mode: scalar 1-p).
 *   Y      [T, B, (1+bidir)*H]  output in the reference's unpacked layout:
 *                      Y[t,b,0:H] forward half, Y[t,b,H:2H] reversed half
 *                      stored at its ORIGINAL time index.
 *   S      [T, R, NS*H] saved per-step tensors for backward (cell-dependent:
 *                      liGRU z,a; RNN a; LSTM f,i,o,g,c; GRU z,r,a; minGRU z,a)
 *   ln_gamma/ln_beta [H] or NULL: per-step LayerNorm of h_t (neural_networks.py:23-33 applied at :466-467,
 *                      :638-639, :1138-1139, :1299-1300, :1444-1445), eps 1e-6.
 *   LNS    >= pk_rec_ln_saved_floats() floats when LayerNorm is on: LN statistics + pre-LN h, saved for backward
 *                      (opaque: the step-wise and the persistent algorithm lay it out differently).
 * work: scratch, >= pk_rec_work_floats() floats; PK_REC_PERSISTENT with LayerNorm: >= pk_rec_work_floats() +
 *                      pk_rec_ln_work_floats() floats (the row-statistics exchange sits behind the base scratch).
 * algo: PK_REC_STEPWISE handles everything; PK_REC_PERSISTENT handles liGRU/RNN/LSTM (per-step LayerNorm inside
 * the time loop: liGRU / RNN here in fp32, liGRU / RNN / LSTM in bf16 through pk_rec_fwd_bf16_ln) and returns an error
 * otherwise. */
int pk_rec_num_saved(int cell);
int pk_rec_num_gates(int cell);
int64_t pk_rec_work_floats(int cell, int T, int B, int bidir, int H);
int64_t pk_rec_ln_saved_floats(int T, int B, int bidir, int H);
int64_t pk_rec_ln_work_floats(int T, int B, int bidir, int H);
int pk_rec_fwd(void* stream, int algo, int prec, int cell, int act, int T, int B, int bidir, int H,
               const float* P, const float* pscale, const float* pshift, const float* U, const float* mask,
               float mask_scalar, const float* ln_gamma, const float* ln_beta, float* Y, float* S, float* LNS,
               float* work);
/* Backward through time.  dY [T,B,(1+bidir)*H] is the gradient of Y.
 * Outputs: dP2 [1+bidir][T*B][G*H]: gradient w.r.t. the (scaled+shifted)
 * projections, one slab per direction, both indexed by ORIGINAL time; the
 * caller adds the slabs (pk_bn_bwd_* take g and g2).  dU [G*H, H] (overwritten; NULL = skip the deferred
 * dU GEMMs, the caller forms dU = sum_t dgate_t^T . h_{t-1} itself, e.g. with pk_gemm_bf16).
 * dln_gamma/dln_beta [H] (overwritten) when LayerNorm is on.
 * Hprev for dU and the carry come from Y / LNS. */
int pk_rec_bwd(void* stream, int algo, int prec, int cell, int act, int T, int B, int bidir, int H,
               const float* U, const float* mask, float mask_scalar, const float* ln_gamma, const float* Y,
               const float* S, const float* LNS, const float* dY, float* dP2, float* dU, float* dln_gamma,
               float* dln_beta, float* work);

/* ---- perf-mode persistent recurrences (liGRU / RNN / LSTM; same reference loops as pk_rec_fwd/bwd):
 * bf16 MFMA operands, fp32 gate math / state / outputs.  The in-kernel exchange buffers are also
 * outputs: Yb [T*B][y_pitch] bf16 copy of Y (direction d at column d*Hp, Hp = H rounded up to 8, zero
 * padded) and dGb [ndir*T*B][g_pitch] bf16 gate gradients (gate g at column g*Hp) are laid out as the
 * k-major operands pk_gemm_bf16 needs for dU / dW.  Columns beyond ndir*Hp (G*Hp) of a row are left
 * undefined.  Pitches are multiples of 8 elements; H <= 576.  dP2 may be NULL (the fp32 gate-gradient
 * slabs are then not written: pk_bn_bwd_bf16 works from dGb).
 * Forward-only chunks (torch.no_grad: validation / forward, core.py:644-671): S may be NULL - nothing is saved for a
 * backward pass - and, with it, Y may be NULL for an inner layer of a stack whose output is consumed as Yb alone.
 */
/* prefilled (both entry points below): 1 = the caller stored the 0xFF "not written yet" pattern in the whole exchange
 * buffer, 0 = the library does it in front of the launch, 2 = the kernel does it on the way where it can
 * (pk_rec_self_fill(cell) == 1: every chunk is patterned by the lane that later publishes it, a few steps ahead; other
 * cells: as 0). */
/* CUs the bf16 persistent recurrence over R = B * (1 + bidir) rows of H units occupies (0: not covered) */
int pk_rec_plan_cus(int R, int H);
int pk_rec_self_fill(int cell);
int pk_rec_fwd_bf16(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                    const float* pscale, const float* pshift, const float* U, const float* mask, float mask_scalar,
                    float* Y, float* S, uint16_t* Yb, int64_t y_pitch, int prefilled);
int pk_rec_bwd_bf16(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U, const float* mask,
                    float mask_scalar, const float* Y, const float* S, const float* dY, float* dP2, uint16_t* dGb,
                    int64_t g_pitch, int prefilled);
/* ... with per-step LayerNorm of h_t inside the persistent time loop (liGRU / RNN / LSTM; the reference's
 * `if self.*_use_laynorm[i]: ht = self.ln[i](ht)`, neural_networks.py:466-467, :1138-1139, :1444-1445): every step
 * exchanges the rows' partial sums between the workgroups of a cluster a second time (fp32, 32 bytes per wave and row
 * quad) before h_t is normalised, stored, published and fed back.  LNS >= pk_rec_ln_saved_floats() floats (forward
 * writes, backward reads), lnwork >= pk_rec_ln_work_floats() floats of scratch per call, dln_gamma / dln_beta [H]
 * (overwritten).  Y / Yb hold the normalised h_t. */
int pk_rec_fwd_bf16_ln(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                       const float* pscale, const float* pshift, const float* U, const float* mask, float mask_scalar,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, float* Y, float* S, float* LNS,
                       uint16_t* Yb, int64_t y_pitch, int prefilled, float* lnwork);
int pk_rec_bwd_bf16_ln(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U, const float* mask,
                       float mask_scalar, const float* ln_gamma, float ln_eps, const float* Y, const float* S,
                       const float* LNS, const float* dY, float* dP2, uint16_t* dGb, int64_t g_pitch, int prefilled,
                       float* lnwork, float* dln_gamma, float* dln_beta);
/* diagnostics: when non-NULL, cluster 0 / member 0 / wave 0 of every following launch writes
 * T x 8 shader-clock stamps (phase boundaries of each step) to this device buffer. */
void pk_persist2_set_trace(void* dev_buf);
/* diagnostics, traced (Li-GRU / relu) kernels only: 1 = every step skips its MFMA block and gate math, so that what is
 * timed is the hand-off alone - the latency floor of a step (results of such a launch are meaningless). */
void pk_persist2_set_empty_step(int on);
/* 0 (default): clusters whose workgroups all run on one XCD exchange through that XCD's L2 (plain
 * stores + nt loads), others use write-through stores + agent-scope loads; 1: always the latter. */
void pk_persist2_set_mode(int force_safe);
/* tuning: idle time (units of 64 clocks) between a workgroup's publish and its first poll of the next step */
void pk_persist2_set_poll_delay(int units);
/* LSTM only: 4 = four waves per workgroup (the kernels every cell uses), 8 = eight waves, two per group of 16 hidden
 * units, each holding half of the recurrent-matrix fragments (pk_rec_persist2_lstm.hip).  Default: PK_EXPERIMENT lstm_waves. */
void pk_persist2_set_lstm_waves(int waves);
int pk_persist2_get_lstm_waves(void);
/* L2 run-ahead helpers of the persistent bf16 recurrences (pk_rec_helper.hip; default: PK_REC_HELPER): workgroups on the
 * CUs a recurrence leaves idle touch the lines its time loop (neural_networks.py:457-469, :629-641, :1130-1141) is about to
 * use a few steps ahead of it, paced by the exchange buffer itself.  bit 0: the projections of the forward pass, bit 1:
 * the Y / S lines the forward pass is about to write, bit 2: the saved tensors the backward pass reads; bits 8-29: tuning
 * fields (pk_rec_helper.hip).  -1 = the per-cell default (what an unset PK_REC_HELPER means: the LSTM forward pass takes
 * bits 0 and 1, everything else none - profiles/r05_rec_helper.json).  The helpers only load: results never depend on the
 * mode. */
void pk_rec_helper_set_mode(int mode);
int pk_rec_helper_get_mode(void);
/* two-phase cells (GRU :629-641, minimalGRU :1291-1302): the candidate GEMM consumes a gate of the same
 * step, so every step is two cluster-wide exchanges.  Xb [T*B][y_pitch] (bf16 r*h / z*h, laid out like Yb)
 * is the second exchange buffer and the k-major operand of the dU_h GEMM; of S only the z(,r),a slots are
 * written.  The backward leaves the fp32 gate gradients unwritten (dGb only, gates [z,(r,)a] at g*Hp). */
int pk_rec2p_fwd_bf16(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                      const float* pscale, const float* pshift, const float* U, const float* mask, float mask_scalar,
                      float* Y, float* S, uint16_t* Yb, uint16_t* Xb, int64_t y_pitch, int prefilled);
int pk_rec2p_bwd_bf16(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U, const float* mask,
                      float mask_scalar, const float* Y, const float* S, const float* dY, uint16_t* dGb, int64_t g_pitch,
                      int prefilled);
/* ... with per-step LayerNorm of h_t (GRU neural_networks.py:638-639, minimalGRU :1299-1300): a third exchange in the
 * step, of the rows' partial sums; arguments as pk_rec_fwd_bf16_ln / pk_rec_bwd_bf16_ln. */
int pk_rec2p_fwd_bf16_ln(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* P,
                         const float* pscale, const float* pshift, const float* U, const float* mask, float mask_scalar,
                         const float* ln_gamma, const float* ln_beta, float ln_eps, float* Y, float* S, float* LNS,
                         uint16_t* Yb, uint16_t* Xb, int64_t y_pitch, int prefilled, float* lnwork);
int pk_rec2p_bwd_bf16_ln(void* stream, int cell, int act, int T, int B, int bidir, int H, const float* U, const float* mask,
                         float mask_scalar, const float* ln_gamma, float ln_eps, const float* Y, const float* S,
                         const float* LNS, const float* dY, uint16_t* dGb, int64_t g_pitch, int prefilled, float* lnwork,
                         float* dln_gamma, float* dln_beta);
unsigned pk_persist2_error_count(void);
void pk_persist2_error_reset(void);

/* ---- conv1d (valid, stride 1) fused with max_pool1d(kernel=stride=pool):
 * replaces F.conv1d + F.max_pool1d at neural_networks.py:1546-1552,
 * :1655-1661, :1805-1813.  x [B,Cin,L], w [Cout,Cin,K], bias [Cout] or NULL,
 * y [B,Cout,Lp] with Lp = (L-K+1)/pool, argmax [B,Cout,Lp] (int32 position in
 * the un-pooled conv output) for backward. */
int pk_conv1d_pool_fwd(void* stream, const float* x, const float* w, const float* bias, int B, int Cin, int L,
                       int Cout, int K, int pool, float* y, int32_t* argmax, float* work);
/* work: scratch >= pk_conv_fwd_work_floats() floats (the weights re-packed per 16-channel tile so that the
 * kernel reads them through the scalar cache). */
int64_t pk_conv_fwd_work_floats(int Cin, int Cout, int K);
/* dw [Cout,Cin,K], dbias [Cout] (may be NULL), dx [B,Cin,L] (may be NULL, e.g.
 * first layer).  partial: scratch >= pk_conv_partial_floats(). */
int64_t pk_conv_partial_floats(int B, int Cin, int L, int Cout, int K, int pool);
int pk_conv1d_pool_bwd(void* stream, const float* x, const float* w, const float* dy, const int32_t* argmax, int B,
                       int Cin, int L, int Cout, int K, int pool, float* dw, float* dbias, float* dx,
                       float* partial);

/* ---- fused optimizers on flat parameter buckets (next row):
 * torch.optim.RMSprop / SGD as utils.optimizer_init configures them
 * (utils.py:2106-2164).  p, g, state: n floats. */
int pk_rmsprop_step(void* stream, float* p, const float* g, float* square_avg, int64_t n, float lr, float alpha,
                    float eps, float weight_decay);
int pk_sgd_step(void* stream, float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                float weight_decay, int first_step);
/* torch.optim.Adam as utils.py:2130-2146 builds it; step counts from 1; max_exp_avg_sq NULL unless amsgrad. */
int pk_adam_step(void* stream, float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                 int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step);
/* The same three steps (kind: 0 RMSprop - h0 alpha, h1 eps; 1 SGD - h0 momentum, s0 its buffer or NULL; 2 Adam - h0 / h1 betas,
 * h2 eps, s0 / s1 moments, s2 amsgrad maximum or NULL; step counts from 1) on a flat bucket whose length and pointers are
 * multiples of 4 elements, with two things done on the way out: zero_grad != 0 leaves g zeroed (the zero_grad() of the next
 * step has nothing left to do), and the bf16 copies of the 2-D weights that the perf-mode GEMMs read are refreshed - segs
 * (device, [nseg][5] int64, ascending: first element in the bucket, rows, columns (multiple of 4), pitch of the copy, first
 * element of the copy in `shadow`), or NULL / 0. */
int pk_fused_step(void* stream, int kind, float* p, float* g, float* s0, float* s1, float* s2, int64_t n, float lr, float h0,
                  float h1, float h2, float weight_decay, int step, int zero_grad, const int64_t* segs, int nseg,
                  uint16_t* shadow);

/* persistent-recurrence health: number of spin time-outs since the last reset
 * (host-mapped counter, readable without a device sync). */
unsigned pk_persist_error_count(void);
void pk_persist_error_reset(void);

/* ---- self tests (tests/ only): MFMA fragment-layout check on the device.
 * Writes 0 to *h_bad_count if every layout assumption holds. */
/* One perf-mode MLP layer of a SMALL batch in one launch (neural_networks.py:139-148, the drop(act(bn(wx(x)))) order with
 * BatchNorm1d in training mode): z = x W^T + b on bf16 operands, batch statistics of z's columns, a = act(bn(z)),
 * y = a * mask, yb = bf16(y); running statistics updated (momentum, unbiased variance).  The whole batch (M <= 128 rows)
 * sits in one row tile, so the statistics are a reduction inside the workgroup that owns the columns.
 * xb [M][ldx], wb [N][ldw] bf16, k-contiguous; z, a, y fp32 [M][N] (y NULL when mask is NULL: the output is a);
 * yb bf16 [M][ldyb]; mean, var [N] = biased batch statistics (what pk_bn_bwd_* take).  pk_linear_bn_act_bf16_covers:
 * does the shape take this path (2 <= M <= 128, N a multiple of 8)? */
int pk_linear_bn_act_bf16_covers(int64_t M, int64_t N, int64_t K);
int pk_linear_bn_act_bf16(void* stream, int M, int N, int K, const uint16_t* xb, int64_t ldx, const uint16_t* wb, int64_t ldw,
                          const float* bias, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, int act, const float* mask, float* z, float* a,
                          float* y, uint16_t* yb, int64_t ldyb, float* mean, float* var);
/* The same layer in two launches that cover the chip (round 4: a launch of the one-launch form is as long as its
 * 320 KB per-workgroup footprint takes through ONE CU's memory pipe): the product split along K into fp32 slabs
 * ws[splitk][M][N] - splitk = pk_gemm_bf16_small_splitk(M, N, K), >= 2 - then the layer epilogue straight from the
 * slabs.  pk_gemm_bf16_small_splitk also sizes the split of any small-batch pk_gemm_bf16 call (M <= 128 rows,
 * k-contiguous A; 1 = not worth splitting). */
int pk_gemm_bf16_small_splitk(int M, int N, int K);
int pk_linear_bn_act_bf16_sk(void* stream, int M, int N, int K, const uint16_t* xb, int64_t ldx, const uint16_t* wb,
                             int64_t ldw, const float* bias, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var, int act, const float* mask,
                             float* z, float* a, float* y, uint16_t* yb, int64_t ldyb, float* mean, float* var,
                             int splitk, float* ws);
/* ---- perf-mode convolutions (pk_conv_bf16.hip): F.conv1d + F.max_pool1d of the SincNet / CNN stacks
 * (neural_networks.py:1546-1552, :1655-1661, :1805-1813) as an implicit GEMM on v_mfma_f32_16x16x32_bf16 - x, w and the
 * un-pooled output gradient enter as bf16, accumulation in fp32; same tensors and arg-max convention as
 * pk_conv1d_pool_fwd / _bwd (the exact-fp32 kernels of the parity mode).  pk_conv_bf16_covers: pool widths that divide
 * 48, up to 128 output channels.  work: pk_conv_bf16_work_floats(..., backward) floats of scratch per call.  The bias
 * gradient is not produced here (a column sum of dy). */
int pk_conv1d_pool_dgrad(void* stream, const float* w, const float* dy, const int32_t* argmax, int B, int Cin, int L, int Cout,
                         int K, int pool, float* dx, float* work); /* exact-fp32 data gradient alone; work >= pk_conv_fwd_work_floats */
int pk_conv_bf16_covers(int Cin, int Cout, int K, int pool);
int64_t pk_conv_bf16_work_floats(int B, int Cin, int L, int Cout, int K, int pool, int backward);
int pk_conv1d_pool_fwd_bf16(void* stream, const float* x, const float* w, const float* bias, int B, int Cin, int L, int Cout,
                            int K, int pool, float* y, int32_t* argmax, float* work);
int pk_conv1d_pool_bwd_bf16(void* stream, const float* x, const float* w, const float* dy, const int32_t* argmax, int B,
                            int Cin, int L, int Cout, int K, int pool, float* dw, float* dx, float* work);
int pk_selftest_mfma(void* stream, int* h_bad_count);
/* v_permlane16_swap_b32 lane mapping the third-generation persistent recurrences rely on when they assemble a 16-byte
 * publish chunk from two lanes (pk_rec_persist3.hip) */
int pk_selftest_permlane(void* stream, int* h_bad_count);
/* the DPP row sum (quad_perm x2, row_half_mirror, row_mirror) of the per-step LayerNorm exchange in the persistent
 * recurrences: every lane of a 16-lane row must end with the row's total */
int pk_selftest_dpp_row_sum(void* stream, int* h_bad_count);
/* test aid: `blocks` workgroups (one per CU: 64 KB of LDS each) that copy a private 64 KB slice of buf (blocks x 16384
 * floats) through LDS for `usec` microseconds on `stream` - a stand-in for a collective's ring kernel next to the persistent
 * recurrences. */
int pk_selftest_cu_hog(void* stream, int blocks, int usec, float* buf);

/* ---- chunk loader pieces (next row): binary Kaldi matrix tables and the whole-chunk transforms
 * of data_io.load_chunk.  Host memory; plain files (the reference reads through Kaldi pipes, which stay outside).
 * pk_ark_open: path + byte offset (0 for an ark read from its start; the offset of an scp entry otherwise).
 * pk_ark_next: key_expected = 1 reads "<key> " first (ark), 0 expects a bare matrix (scp entry).  Returns 1 and the
 *   dimensions, 0 at end of file, 2 on a malformed table.  FM / DM (data_io.py:1106-1131), CM (:1150-1198), CM2, CM3.
 * pk_ark_read / pk_ark_skip: consume the announced matrix (row-major float32 into dst). */
typedef struct pk_ark pk_ark;
pk_ark* pk_ark_open(const char* path, int64_t offset);
void pk_ark_close(pk_ark* a);
int pk_ark_next(pk_ark* a, int key_expected, char* key, int keycap, int64_t* rows, int64_t* cols);
int pk_ark_read(pk_ark* a, float* dst);
int pk_ark_skip(pk_ark* a);
/* integer-vector tables (alignments / pdf ids, data_io.py:790-838) on the same handle: 1 + *n / 0 end / 2 malformed */
int pk_ivec_next(pk_ark* a, char* key, int keycap, int64_t* n);
int pk_ivec_read(pk_ark* a, int32_t* dst);
/* data_io.py:228-241 context_window on the concatenated chunk: out [(rows-left-right)][cols*(left+right+1)] */
int pk_context_window(const float* x, int64_t rows, int64_t cols, int left, int right, float* out);
/* data_io.py:263: x <- (x - mean) / std per column (population std, double accumulation), in place */
int pk_mean_var_norm(float* x, int64_t rows, int64_t cols);

#ifdef __cplusplus
}
#endif
#endif /* PK_AMD_H */
