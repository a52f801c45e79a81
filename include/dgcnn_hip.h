/*
 * dgcnn_hip.h -- C ABI of libdgcnn_hip.so: the MI355X (gfx950) EdgeConv hot path of DGCNN.
 *
 * The reference (DeepLearnPhysics/dynamic-gcnn) has no FFI of its own: its hot path is Python
 * that composes stock TensorFlow-1 ops.  Each entry point below therefore names
 * the reference *call site* (file:line under /root/reference) whose TF op group it replaces; the
 * Python mirror of the reference's operator surface (dynamic-gcnn_amd/dgcnn/{ops,model,trainval}.py)
 * binds exactly these symbols through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success or a negative DGCNN_E* code; dgcnn_last_error() gives
 *     the message of the last failure on the calling thread;
 *   - all pointers are DEVICE pointers owned by the caller (the library allocates nothing and
 *     keeps no state), tensors are dense row-major fp32 unless a leading dimension `ld*`
 *     (in elements) is given; indices are batch-local int32;
 *   - every call is asynchronous on the caller-supplied hipStream_t (passed as void*);
 *   - `stats` buffers are double[slots][2][F] (sum, sum of squares), zeroed by the
 *     caller, accumulated by the producing kernel's epilogue, reduced by dgcnn_bn_finalize_f32;
 *     slots = DGCNN_STAT_SLOTS unless dgcnn_set_stat_slots() changed it.
 */
#ifndef DGCNN_HIP_H_
#define DGCNN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGCNN_OK 0
#define DGCNN_EINVAL (-1)   /* bad argument (e.g. k > N: tf.nn.top_k would raise, dgcnn/ops.py:18) */
#define DGCNN_ELAUNCH (-2)  /* HIP launch/runtime error */
#define DGCNN_ENOSPC (-3)   /* workspace too small */
#define DGCNN_EUNSUP (-4)   /* shape not supported by this build */

#define DGCNN_STAT_SLOTS 32
/* Slots of every stats / red buffer prepared by THE CALLING THREAD from now on (thread-local; default DGCNN_STAT_SLOTS; read on
 * the host at launch time and passed to the kernels as an argument, so calls of other threads / streams are not affected).  A producer workgroup adds its partial
 * sums -- themselves formed in a fixed order -- to slot (writer index mod slots) with a double atomic; with slots >= the number of
 * writers (the library then also caps the grids of its reduction kernels at `slots`) every slot has ONE writer, and the finalize
 * kernels add the slots in a fixed order: sums, hence whole training steps, are bit-reproducible run to run at the speed of the
 * default kernels (the host's DETERMINISTIC mode: 768 slots for 49152 rows).  Buffers must be sized for the value in force. */
int dgcnn_set_stat_slots(int n);
int dgcnn_get_stat_slots(void);

int dgcnn_version(void);
const char* dgcnn_last_error(void);

/* ---- K1: dgcnn/ops.py:8-19 k_nn --------------------------------------------------------
 * idx[b][i][0..k) = the k smallest D_ij = (s_i + s_j) - 2 <x_i,x_j> of row i (self included),
 * ascending, ties -> lower j.  Arithmetic order is normative (oracle/knn_oracle.c): bit-exact.
 * x: (B,N,C) with row stride ldx.  No (B,N,N) matrix is ever written to HBM.
 * ws: caller scratch (16-byte aligned) of dgcnn_knn_workspace_bytes(B,N,C,k) bytes: the s_i of ops.py:14 and, for raw
 *   coordinates (C <= 4, k <= 40), the sorted copy / cell table of the exact cell-grid search (csrc/knn_grid.hip: the points of
 *   a cloud are bucketed into a uniform grid and a row only meets the candidates of the cells around its own until its k-th
 *   distance provably beats everything further out -- same pairs' arithmetic, same (D, j) order, same indices); for feature-space
 *   rows (16 < C <= 64) the seed bounds and the per-row candidate buffers of the append-form scan of dgcnn_knn_seeded_f32 (one count
 *   + 256 ... 512 eight-byte entries per row).  A workspace that only holds the s_i (B*N floats rounded up to 256 bytes) selects the
 *   all-pairs, list-keeping kernels. */
int64_t dgcnn_knn_workspace_bytes(int B, int N, int C, int k);
/* 0 = all-pairs kernel for C <= 4 too, 1 (default) = cell grid where it pays (N >= 4096: $DGCNN_KNN_GRID_MIN_N), 2 = cell grid
 * whenever applicable (tests); returns the previous setting ($DGCNN_KNN_GRID=0 switches it off). */
int dgcnn_knn_grid(int mode);
/* A/B switch: 1 = distances by VALU fmaf chains for every C (exact by construction), 0 (default) =
 * v_mfma_f32_32x32x2_f32 for C > 4 (bit-identical on gfx950; the tests compare both).  Returns the
 * previous setting. */
int dgcnn_knn_force_valu(int on);
/* Large feature-space graphs (16 < C <= 64, C % 4 == 0, 16-byte aligned rows): approximate distances from the two leading bf16
 * terms of every operand on the bf16 matrix pipe with a rigorous error bound as a conservative filter, the normative fp32 fmaf
 * chain only for the survivors -- the same indices, bit for bit.  mode 0 = never, 1 = whenever applicable, 2 (default) = for
 * N >= 8192 and wherever a seed bound exists ($DGCNN_KNN_BF16F = 0 | 1 | auto).  Returns the previous mode. */
int dgcnn_knn_bf16_filter(int mode);
int dgcnn_knn_f32(const float* x, int B, int N, int C, int64_t ldx, int k, int32_t* idx,
                  void* ws, size_t ws_bytes, void* stream);
/* The same search with a per-row upper bound taken from `seed` (B, N, >= kseed int32, row stride ldseed): kseed >= k DISTINCT
 * candidates of every row -- in the EdgeConv stack the previous layer's graph (ops.py:95-96).  Their largest distance bounds the
 * row's k-th distance before the scan starts: feature-space rows (16 < C <= 64) then run the append-form scan (no sorted lists: a
 * candidate under the bound is appended to the row's buffer, one selection per row at the end), other shapes filter their list
 * inserts with it.  Identical result (csrc/knn.hip).  seed == NULL or kseed < k: plain dgcnn_knn_f32.  Workspace as
 * dgcnn_knn_workspace_bytes says. */
int dgcnn_knn_seeded_f32(const float* x, int B, int N, int C, int64_t ldx, int k, const int32_t* seed, int64_t ldseed,
                         int kseed, int32_t* idx, void* ws, size_t ws_bytes, void* stream);
/* smallest N for which the seeds are used; -1 = the library's rule (always where the append-form scan applies -- 16 < C <= 64,
 * C % 4 == 0 -- else from 4096 on: below, computing the bound costs the list-keeping kernels what it saves); n < -1 only queries.
 * Returns the previous value.  (tools / tests) */
int dgcnn_knn_seed_min_n(int n);
/* 1 (default) / 0: seeded searches of feature-space rows (16 < C <= 64) run the append-form scan (candidates under the seed bound are
 * appended to a per-row buffer, one selection per row at the end: csrc/knn.hip) / keep per-lane sorted lists; on < 0 only queries.
 * Identical indices either way.  Returns the previous setting.  (tools / tests; env DGCNN_KNN_APPEND) */
int dgcnn_knn_append(int on);
/* Products of the append-form scan's bf16 filter: 1 (default since round 6: plain bf16 operands, margin 2^-6 (s_i + s_j)) or 3 (two bf16
 * terms per operand, margin 2^-13 (s_i + s_j)); any other n only queries.  Identical indices either way (survivors of the filter get
 * their normative distance).  Returns the previous setting.  (tools / tests; env DGCNN_KNN_APPEND_NPR="<N < 8192>,<N >= 8192>") */
int dgcnn_knn_append_products(int n);
/* Raw-coordinate rows (C <= 4) below the cell grid's range: a histogram pass over every `stride`-th candidate (half-octave bins of the
 * normative distance) bounds every row's k-th distance before the scan starts, so that ~1.3 k candidates reach the sorted lists instead
 * of ~k (1 + ln(N / k)) per list.  stride 0 = off, 1 / 2 (default) / 4; any other value only queries.  Identical indices either way.
 * Returns the previous setting.  (tools / tests; env DGCNN_KNN_HIST) */
int dgcnn_knn_hist(int stride);

/* ---- K3 in its bf16-operand form (BASELINE configs[2] "bf16 edge-MLP MFMA"): conv0 of an EdgeConv layer, ops.py:21-52 ------
 * E[e] = [x_i, x_j - x_i] formed in fp32 and rounded to bf16 once (RNE), W0 (2C x F, row-major) rounded to bf16 once,
 * y = E W0 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; neither E nor y is written by the two forward passes:
 *   dgcnn_edge_mlp_bf16_stats      stats[slot][0][f] += sum_e y[e][f], stats[slot][1][f] += sum_e y[e][f]^2  (double[slots][2][F], zeroed)
 *   dgcnn_edge_mlp_bf16_bn_kreduce z = relu((y - mean) rstd + beta) recomputed; max / mean over the k edges of each point and the
 *                                  number of edges attaining the max (cnt, (B N, F) dense; may be null); pack_cnt = 1: cnt =
 *                                  #ties + 256 #(z > 0), the packing of the fp32 edge kernels (dgcnn_edge_bn_act_kreduce_f32), which
 *                                  dgcnn_edge_bn_bwd_reduce_points_f32 and dgcnn_edge_mlp_bf16_bwd read (k < 256)
 *   dgcnn_edge_mlp_bf16            y written out, (B N k, F) dense -- bit-identical to what the two passes saw
 *   dgcnn_edge_mlp_bf16_bwd        the layer's backward in ONE pass over the edges, y recomputed: red = the slots written by
 *                                  dgcnn_edge_bn_bwd_reduce_points_f32 (reduced here; dbeta = dbeta_beta * dbeta + sum dz);
 *                                  dY = rstd (dz - c1 - xhat c2) rounded to bf16 -> dYb (B N k, F) bf16 and dysum = sum_m dY (B N, F)
 *                                  (either may be null); dW0 (2C, F) += E^T dY on the matrix pipe.  Neither E nor y nor an fp32 dY
 *                                  is written.  ws >= dgcnn_edge_mlp_bf16_bwd_workspace_bytes.  8 <= k < 256 (C = 64 with F = 128: k >= 10, LDS);
 *                                  dgcnn_edge_mlp_bf16_bwd_supported tells.
 *   dgcnn_edge_gather_sum_bf16     dgcnn_edge_gather_sum_f32 over that bf16 dY (same additions in the same order)
 * Shapes: C <= 4 or C == 64 (x float4-loadable), F in {32, 64, 128}, k <= 128 (dgcnn_edge_mlp_bf16_supported; DGCNN_EUNSUP else). */
int dgcnn_edge_mlp_bf16_supported(int C, int k, int F);
int dgcnn_edge_mlp_bf16_bwd_supported(int C, int k, int F);
int64_t dgcnn_edge_mlp_bf16_bwd_workspace_bytes(int B, int N, int C, int k, int F);
int dgcnn_edge_mlp_bf16_bwd(const float* x, int64_t ldx, const int32_t* idx, const float* W0, int B, int N, int C, int k, int F,
                            const float* mean, const float* rstd, const float* beta, const float* mx, int64_t ldmx,
                            const float* cnt, const float* dmx, int64_t lddmx, const float* dmn, int64_t lddmn, double* red,
                            void* dYb, float* dysum, int64_t lddysum, float* dW0, float* dbeta, float dbeta_beta, void* ws,
                            size_t ws_bytes, void* stream);
int dgcnn_edge_gather_sum_bf16(const void* dY, const int32_t* off, const int32_t* rev, int64_t R, int F, float* S, int64_t lds,
                               void* stream);
/* dst[i] = src[i] rounded to the nearest bf16 value (ties to even), kept as fp32 (weights of the mode's point-level gradient products) */
int dgcnn_round_bf16_f32(const float* src, float* dst, int64_t n, void* stream);
int dgcnn_edge_mlp_bf16(const float* x, int64_t ldx, const int32_t* idx, const float* W0, int B, int N, int C, int k, int F,
                        float* Y, void* stream);
int dgcnn_edge_mlp_bf16_stats(const float* x, int64_t ldx, const int32_t* idx, const float* W0, int B, int N, int C, int k, int F,
                              double* stats, void* stream);
int dgcnn_edge_mlp_bf16_bn_kreduce(const float* x, int64_t ldx, const int32_t* idx, const float* W0, int B, int N, int C, int k,
                                   int F, const float* mean, const float* rstd, const float* beta, float* mx, int64_t ldmx,
                                   float* mn, int64_t ldmn, float* cnt, int pack_cnt, void* stream);

/* ---- K2: dgcnn/ops.py:21-40 edges (gather + tile + sub + concat) ------------------------
 * E[b][i][m][0..C) = x_i ; E[b][i][m][C..2C) = x_{idx[b][i][m]} - x_i.                    */
int dgcnn_edge_gather_f32(const float* x, int64_t ldx, const int32_t* idx, int B, int N, int C, int k,
                          float* E, void* stream);
/* transpose of the above (tf.gather^T = scatter-add, tf.tile^T = sum over k): dx += ...   */
int dgcnn_edge_gather_bwd_f32(const float* dE, const int32_t* idx, int B, int N, int C, int k,
                              float* dx, int64_t lddx, void* stream);

/* ---- K3: dgcnn/ops.py:47-52 slim.conv2d 1x1 on the edge tensor (conv0) --------------------
 * Y[(b,i,m)][f] = sum_c E[(b,i,m)][c] W0[c][f], E gathered on the fly from (x, idx) into the LDS
 * A-tile (never written to HBM); fp32 MFMA; epilogue accumulates per-channel sum / sum-of-squares
 * of Y for the batch-norm statistics (ops.py:53).                                           */
int dgcnn_edge_mlp_f32(const float* x, int64_t ldx, const int32_t* idx, const float* W0,
                       int B, int N, int C, int k, int F, float* Y, double* stats, void* stream);
/* dW0[2C][F] (+)= E^T dY  (split over the edge dimension; ws = workspace)                   */
int dgcnn_edge_mlp_wgrad_f32(const float* x, int64_t ldx, const int32_t* idx, const float* dY,
                             int B, int N, int C, int k, int F, float* dW0, float beta,
                             void* ws, size_t ws_bytes, void* stream);
/* dx[nbr] += dY W0[C:2C]^T (scatter, fp32 atomics); the centre part
 * dx[i] += (sum_m dY[i,m]) (W0[:C]-W0[C:])^T is a plain GEMM issued by the host.            */
int dgcnn_edge_mlp_dgrad_scatter_f32(const float* dY, const float* W0, const int32_t* idx,
                                     int B, int N, int C, int k, int F, float* dx, int64_t lddx,
                                     void* stream);

/* conv0 in FACTORED form (what dgcnn.ops.edge_conv issues): E W0 = x_i (W0[:C]-W0[C:]) + x_j W0[C:].
 * The centre term U[point] = X (Wa-Wb) is one per-point dgcnn_gemm_f32; per edge remains the
 * (B*N*k) x C GEMM of the gathered neighbour rows with Wb = W0[C:], with U added per group of k rows
 * in the epilogue (+ BN column statistics): half the per-edge flops, identical result up to fp32
 * rounding (tests compare both forms).                                                          */
int dgcnn_edge_nbr_gemm_f32(const float* x, int64_t ldx, const int32_t* idx, const float* Wb,
                            const float* U, int64_t ldu, int B, int N, int C, int k, int F,
                            float* Y, double* stats, void* stream);
/* dWb[C][F] (+)= sum_e x_j(e)^T dY[e]; the host adds X^T (sum_m dY) for the centre half.        */
int dgcnn_edge_nbr_wgrad_f32(const float* x, int64_t ldx, const int32_t* idx, const float* dY,
                             int B, int N, int C, int k, int F, float* dWb, float beta,
                             void* ws, size_t ws_bytes, void* stream);

/* Deterministic-shape alternative to the scatter: bucket the edges by target once (transposed
 * adjacency: off[B*N+1], rev[B*N*k]; cnt_ws = 2*B*N int32 scratch) ...                         */
int dgcnn_edge_csr_build(const int32_t* idx, int B, int N, int k, int32_t* cnt_ws, int32_t* off,
                         int32_t* rev, void* stream);
/* ... then S[j][:] = sum over incoming edges e of dY[e][:] (tf.gather^T as a gather); the host
 * finishes with a plain GEMM dx += S W0[C:2C]^T.                                              */
int dgcnn_edge_gather_sum_f32(const float* dY, const int32_t* off, const int32_t* rev, int64_t R, int F,
                              float* S, int64_t lds, void* stream);

/* ---- conv0 of an EdgeConv layer (ops.py:47-52) as point-level GEMM + per-edge gather-add ---------
 * The convolution is linear: [x_i, x_j - x_i] W0 = x_i (Wa - Wb) + x_j Wb with W0 = [Wa ; Wb].  So
 *   Wcat = [Wa - Wb | Wb]              (C x 2F)    dgcnn_edge_weight_split_f32
 *   [U | V] = X Wcat                   (B*N x 2F)  dgcnn_gemm_f32   (k times fewer MACs than the edge tensor product)
 *   Y[b,i,m,:] = V[b, idx[b,i,m], :] + U[b,i,:]    dgcnn_edge_gather_add_f32 (+ BN column sums of Y into stats)
 * backward: dU = sum_m dY (dgcnn_bn_bwd_apply_f32's dYsum), dV = dgcnn_edge_gather_sum_f32(dY);
 *   dWcat = X^T [dU | dV], dX += [dU | dV] Wcat^T (dgcnn_gemm_f32), then
 *   dW0[:C] += dWcat[:, :F] ; dW0[C:] += dWcat[:, F:] - dWcat[:, :F]     dgcnn_edge_wgrad_combine_f32
 * F %% 4 == 0, F <= 1024, 16-byte aligned rows.                                                  */
int dgcnn_edge_weight_split_f32(const float* W0, int C, int F, float* Wcat, void* stream);
int dgcnn_edge_gather_add_f32(const float* V, int64_t ldv, const float* U, int64_t ldu, const int32_t* idx,
                              int B, int N, int k, int F, float* Y, double* stats, void* stream);
int dgcnn_edge_wgrad_combine_f32(const float* dWcat, int C, int F, float* dW0, void* stream);
/* Y may be NULL in dgcnn_edge_gather_add_f32 (column sums only).  The three BatchNorm passes below are
 * dgcnn_bn_act_kreduce_f32 / _bwd_reduce_f32 / _bwd_apply_f32 with the k rows of point (b,i) RECOMPUTED as
 * V[b, idx[b,i,m], :] + U[b,i,:] instead of read from a (B*N*k, F) tensor: the forward then writes no edge
 * tensor to HBM at all and the backward only dY (which the transposed-adjacency sum needs).  Same argument
 * meaning as their dense counterparts (see below); dY is (B*N*k, F) dense.                           */
int dgcnn_edge_bn_act_kreduce_f32(const float* V, int64_t ldv, const float* U, int64_t ldu,
                                  const int32_t* idx, int B, int N, int k, int F,
                                  const float* mean, const float* rstd, const float* beta, int relu,
                                  float* max_out, int64_t ldmax, float* mean_out, int64_t ldmean,
                                  float* cnt_out, void* stream);
int dgcnn_edge_bn_bwd_reduce_f32(const float* V, int64_t ldv, const float* U, int64_t ldu,
                                 const int32_t* idx, int B, int N, int k, int F,
                                 const float* mean, const float* rstd, const float* beta, int relu,
                                 const float* dmax, int64_t lddmax, const float* dmean, int64_t lddmean,
                                 const float* mx_in, int64_t ldmx, const float* cnt_in,
                                 double* red, void* stream);
/* The backward BN reduction of a relu conv0 from per-point data only (no pass over the edges): cntpos is the
 * cnt_out of dgcnn_edge_bn_act_kreduce_f32, which for the edge variants holds (#ties of the max) + 256 * (#rows
 * with z > 0) (k < 256); mx / mn are that call's max_out / mean_out.  Fills red like _bwd_reduce_f32.            */
int dgcnn_edge_bn_bwd_reduce_points_f32(const float* mx, int64_t ldmx, const float* mn, int64_t ldmn,
                                        const float* cntpos, const float* dmax, int64_t lddmax,
                                        const float* dmean, int64_t lddmean, const float* beta,
                                        int64_t R, int k, int F, double* red, void* stream);
int dgcnn_edge_bn_bwd_apply_f32(const float* V, int64_t ldv, const float* U, int64_t ldu,
                                const int32_t* idx, int B, int N, int k, int F,
                                const float* mean, const float* rstd, const float* beta, int relu,
                                const float* dmax, int64_t lddmax, const float* dmean, int64_t lddmean,
                                const float* mx_in, int64_t ldmx, const float* cnt_in,
                                double* red, float* dY, float* dYsum, int64_t lddysum, float* dbeta,
                                float dbeta_beta, void* stream);

/* The last FC layer with tf.nn.dropout fused behind it (model.py:88-91): out = dropout(relu?(bn(T)), keep) with the counter-based
 * mask of dgcnn_dropout_dev_f32 (seed read from device memory), and its whole BatchNorm backward (sums, finalise + dbeta, dT in
 * place or not) reading dout = d(dropped output) through the same mask: the activated tensor is never stored undropped and the
 * separate dropout passes (forward and backward) disappear.  T, out, dT: (R, F) contiguous rows of F floats (out: leading dim ldo). */
int dgcnn_bn1_act_dropout_f32(const float* T, int64_t R, int F, const float* mean, const float* rstd, const float* beta,
                              int relu, float keep, const uint64_t* seed_dev, float* out, int64_t ldo, void* stream);
int dgcnn_bn1_bwd_dropout_f32(const float* T, int64_t R, int F, const float* mean, const float* rstd, const float* beta,
                              int relu, float keep, const uint64_t* seed_dev, const float* dout, int64_t lddo, double* red,
                              float* dT, float* dbeta, float dbeta_beta, void* stream);

/* conv0 backward of a layer whose input needs no gradient and has C <= 4 channels (the first EdgeConv layer: raw coordinates),
 * in ONE pass: dY is formed per edge exactly as in dgcnn_edge_bn_bwd_apply_f32 and consumed on the spot,
 * dW0[2C][F] += [x_i, x_j - x_i]^T dY (ops.py:39-52) -- no dY tensor, no transposed adjacency, no point-level GEMMs.
 * red: the sums of dgcnn_edge_bn_bwd_reduce(_points)_f32 (finalised here; dbeta as in the apply call); relu is implied (conv0).
 * ws >= (<= 1024 blocks) * 2C * F floats.                                                                                    */
int dgcnn_edge_bn_bwd_apply_wgrad_f32(const float* V, int64_t ldv, const float* U, int64_t ldu,
                                      const int32_t* idx, int B, int N, int k, int F,
                                      const float* mean, const float* rstd, const float* beta,
                                      const float* dmax, int64_t lddmax, const float* dmean, int64_t lddmean,
                                      const float* mx_in, int64_t ldmx, const float* cnt_in, double* red,
                                      const float* x, int64_t ldx, int C, float* dW0, float* dbeta, float dbeta_beta,
                                      void* ws, size_t ws_bytes, void* stream);

/* ---- plain fp32 MFMA GEMM: every other slim.conv2d 1x1 (ops.py:62-70,125-133,153-160;
 * model.py:46-53,65-72,94-101) and their dgrad / wgrad --------------------------------------
 * C[M][N] = op(A)[M][K] op(B)[K][N] (+ beta*C) (+ gbias[row / rows_per_group][n]);
 * transA: A stored [K][M]; transB: B stored [N][K].  stats (optional): BN sums of C's columns.
 * ws is needed when the library decides to split K (transA, tall reductions).                */
/* Arithmetic of dgcnn_gemm_f32 when its operands are float4-loadable (16-byte aligned, leading dimensions
 * and contiguous extents multiples of 4):  0 = v_mfma_f32_32x32x2_f32 (an fmaf chain, 157 TFLOP/s peak);
 * 6 / 9 = every fp32 operand is split EXACTLY into three bf16 terms and 6 (terms below 2^-23 |a b| dropped)
 * or all 9 partial products run on the bf16 matrix pipe with fp32 accumulation (fp32-class results, see
 * gemm_x3.hip).  1 = only the leading bf16 term of each operand: plain bf16 operands with fp32 accumulation (NOT fp32
 * class: the bf16 edge-MLP mode of BASELINE configs[2]; the host selects it around the EdgeConv conv0 / conv1 products).
 * Default 6, or $DGCNN_GEMM_ARITH = f32 | bf16x6 | bf16x9 | bf16x1.  Process-wide.                     */
int dgcnn_gemm_set_arith(int mode);
int dgcnn_gemm_get_arith(void);
/* Rows of the output tile the bf16-split GEMM picks for an (M,N,K) product: 64 | 128 | 256 (256 = the wave-specialised
 * gemm_x3w2_kernel).  For tools that name kernel instances (bench.py's per-kernel table); no effect on results. */
/* row tiles (= workgroups that add column sums into `stats`) of the dgcnn_gemm_f32 launch with these operands (transA = 0); 0 when
 * the launch sizes its grid from the slot count by itself.  The host picks the slot count of that buffer from it (deterministic mode). */
int dgcnn_gemm_stat_writers(int transB, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb);
int dgcnn_gemm_x3_tile_rows(int M, int N, int K);
/* 256 when the product runs on a 256-column kernel (gemm_x3q_kernel: 256 x 256 or 192 x 256 tiles, rows from the call above), else 0 */
int dgcnn_gemm_x3_tile_cols(int M, int N, int K);
/* tools: force the 128- / 256-row tile of the bf16-split kernels (0 = the automatic rule); returns the previous setting */
int dgcnn_gemm_x3_tile_override(int bm);
/* colmax_keys (optional, uint64[M / colmax_rows_per_group][N], zeroed): the per-group column maximum of C and its FIRST row
 * (model.py:76-77: max-pool over the points of a cloud, taken in the epilogue of the GEMM that produces the tensor), as packed
 * keys decoded by dgcnn_colmax_decode_f32 -> value[g][n], row-in-group[g][n].  rows_per_group % 256 == 0, no transA.       */
int dgcnn_gemm_f32(int transA, int transB, int M, int N, int K,
                   const float* A, int64_t lda, const float* B, int64_t ldb,
                   float* C, int64_t ldc, float beta,
                   const float* gbias, int64_t ldgbias, int rows_per_group,
                   double* stats, void* colmax_keys, int colmax_rows_per_group,
                   void* ws, size_t ws_bytes, void* stream);
int dgcnn_colmax_decode_f32(const void* keys, int64_t n, float* vals, int32_t* arg, void* stream);

/* ---- GEMM from PRE-SPLIT operand planes (gemm_pl.hip) -------------------------------------------
 * The same 1x1 convolutions and their dgrad / wgrad (ops.py:62-70,153-160; model.py:65-72) when the
 * operands were split into 16-bit planes ONCE by their producer instead of inside every GEMM tile.
 * Plane set of a tensor X (rows x cols):  element (row, c) of plane p at
 *     base + p * plane_stride + ((c / 8) * rows_alloc + row) * 16 + (c % 8) * 2        [bytes]
 * rows_alloc = rows rounded up to 64, pad rows are zero.
 *   DGCNN_PLANES_F16X2 : x * scale = h1 + h2 (2 fp16 planes; scale = a power of two read from scale_dev, chosen so that
 *                        max |x| * scale < 2^15: dgcnn_planes_scale_f32 or a producer's analytic bound), 3 partial products;
 *                        the GEMM divides its result by scale_A * scale_B (device scalars, exact).
 * dgcnn_split_planes_f32: plane element (row, c) = src[row * row_stride + c * col_stride]  (col_stride 1 = an activation
 *   view; row_stride 1 = the transpose of a row-major matrix, for weights).
 * dgcnn_gemm_planes_f32:  C[M][N] (+= beta C) = sum_k A(m,k) B(n,k)
 *   DGCNN_PL_KC: A has M rows, B has N rows, the reduction runs over the K channels of both (K % 32 == 0); A / B point at the
 *                first octet of the channel range;   DGCNN_PL_TR: A has M channels, B has N channels (M, N % 16 == 0), the
 *                reduction runs over the K rows of both (X^T dY; split over K into ws).  gbias / stats as dgcnn_gemm_f32 (KC only). */
/* (format 0, three bf16 planes / 6 partial products, existed in round 3: slower than the in-kernel split, removed) */
#define DGCNN_PLANES_F16X2 1
#define DGCNN_PL_KC 0
#define DGCNN_PL_TR 1
int dgcnn_planes_scale_f32(const float* src, int64_t ld, int64_t rows, int cols, float bound_mul, float* scale_dev,
                           void* ws, void* stream);
int dgcnn_split_planes_f32(const float* src, int64_t row_stride, int64_t col_stride, int64_t rows, int cols, int fmt,
                           const float* scale_dev, void* dst, int64_t plane_stride, int64_t rows_alloc, void* stream);
int dgcnn_gemm_planes_f32(int form, int fmt, int M, int N, int K,
                          const void* A, int64_t a_plane_stride, int64_t a_rows_alloc,
                          const void* B, int64_t b_plane_stride, int64_t b_rows_alloc,
                          const float* a_scale_dev, const float* b_scale_dev,
                          float* C, int64_t ldc, float beta,
                          const float* gbias, int64_t ldgbias, int rows_per_group,
                          double* stats, void* colmax_keys, int colmax_rows_per_group,
                          void* ws, size_t ws_bytes, void* stream);

/* ---- BatchNorm passes that WRITE operand planes (planes_bn.hip) ------------------------------------
 * The per-point conv + BN + ReLU layers of the head (model.py:65-72, ops.py:153-160): the pass that normalises a GEMM
 * output writes the planes the next plane GEMM reads; the backward-apply pass writes dT as planes.
 * dgcnn_param_scales_f32: scales[0] = power-of-two scale of every activation plane set of the step, from the bound
 *   |z| <= act_mul (sqrt(rows_max) + max |parameter|) that holds for any batch-normalised tensor; scales[1] = scale of the
 *   weight plane sets (max |parameter|).  ws = 4 bytes.
 * dgcnn_bn_act_planes_f32: z = relu?((T - mean) rstd + beta) -> planes (pad rows zero) and, optionally, fp32 copies.
 * dgcnn_bn1_bwd_reduce_max_f32: red[slot][0/1][F] += column sums of dz, dz xhat; maxbits[0/1][F] = max |dz|, max |xhat| (uint bits;
 *   maxbits has 2 F + 1 zeroed words, the last one receives the tensor-wide bound in the apply call).
 * dgcnn_bn1_bwd_apply_planes_f32: finalises red (slot 0 <- sums; dbeta; maxbits is overwritten by the two column means), bounds |dT| per column from the maxima, writes the
 *   plane set's power-of-two scale to scale_dev and dT = rstd (dz - c1 - xhat c2) as planes (+ optional fp32 dT, + optional
 *   per-group column sums gsum[row / rows_per_group][F], rows_per_group % 64 == 0). */
int dgcnn_param_scales_f32(const float* params, int64_t n, double rows_max, float act_mul, float* scales, void* ws, void* stream);
int dgcnn_bn_act_planes_f32(const float* T, int64_t ldt, int64_t R, int F, const float* mean, const float* rstd,
                            const float* beta, int relu, int fmt, const float* scale_dev, void* planes,
                            int64_t plane_stride, int64_t rows_alloc, float* out, int64_t ldo, float* out2, int64_t ldo2,
                            void* stream);
int dgcnn_bn1_bwd_reduce_max_f32(const float* T, int64_t R, int F, const float* mean, const float* rstd,
                                 const float* beta, int relu, const float* dout, int64_t lddo, double* red,
                                 void* maxbits, void* stream);
int dgcnn_bn1_bwd_apply_planes_f32(const float* T, int64_t R, int F, const float* mean, const float* rstd,
                                   const float* beta, int relu, const float* dout, int64_t lddo, double* red,
                                   void* maxbits, int fmt, float* scale_dev, void* planes, int64_t plane_stride,
                                   int64_t rows_alloc, float* dT, float* gsum, int64_t ldgsum, int rows_per_group,
                                   float* dbeta, float dbeta_beta, void* stream);

/* ---- K4/K5: slim.batch_norm (ops.py:53,68 ...) + activation + reduce_max/mean (ops.py:56-57)
 * mean/rstd from the stats slots: mean = S/count, var = Q/count - mean^2 (double), rstd=1/sqrt(var+eps) */
int dgcnn_bn_finalize_f32(const double* stats, int F, double count, float eps,
                          float* mean, float* rstd, void* stream);
/* z = (Y - mean)*rstd + beta ; relu? ; over the k rows of each of the R points:
 * max_out[r][f] = max_m z, mean_out[r][f] = (sum_m z)/k.  k = 1: out = act(bn(Y)) -> max_out
 * (mean_out may be NULL; out2 optionally receives a second copy of max_out; cnt_out (R,F), optional,
 * receives the number of rows attaining the max -- reduce_max's gradient is shared among ties).  */
int dgcnn_bn_act_kreduce_f32(const float* Y, int64_t R, int k, int F,
                             const float* mean, const float* rstd, const float* beta, int relu,
                             float* max_out, int64_t ldmax, float* mean_out, int64_t ldmean,
                             float* out2, int64_t ldout2, float* cnt_out, void* stream);
/* backward, pass 1: red[0][f] = sum dZ, red[1][f] = sum dZ*xhat over all R*k rows, where
 * dZ = relu'(z) * ( dmax*[z==max]/ties + dmean/k ).  red: double[DGCNN_STAT_SLOTS][2][F], zeroed. */
int dgcnn_bn_bwd_reduce_f32(const float* Y, int64_t R, int k, int F,
                            const float* mean, const float* rstd, const float* beta, int relu,
                            const float* dmax, int64_t lddmax, const float* dmean, int64_t lddmean,
                            const float* mx_in, int64_t ldmx, const float* cnt_in,   /* forward's max / #ties, or NULL: recompute */
                            double* red, void* stream);
/* backward, pass 2: dY = rstd*(dZ - red0/cnt - xhat*red1/cnt) written to dY (may alias Y);
 * dbeta[f] (+)= red0 ; dYsum[r][f] = sum_m dY (optional, feeds the centre dgrad).
 * `red` is reduced over its slots in place (slot 0 then holds the totals).
 * relu: bit 0 = the layer has a ReLU; bit 1 (value 2, k > 1 only) = dY is written as bf16 VALUES (nearest even, stored as fp32)
 * and dYsum adds those: the operand rounding of the bf16 edge-MLP's gradient products, done once where dY is formed. */
int dgcnn_bn_bwd_apply_f32(const float* Y, int64_t R, int k, int F,
                           const float* mean, const float* rstd, const float* beta, int relu,
                           const float* dmax, int64_t lddmax, const float* dmean, int64_t lddmean,
                           const float* mx_in, int64_t ldmx, const float* cnt_in,
                           double* red, float* dY, float* dYsum, int64_t lddysum, float* dbeta,
                           float dbeta_beta, void* stream);

/* ---- deterministic mode: fixed-order replacements of the three atomically accumulated steps -------------------------
 * (the fused epilogues sum the BatchNorm statistics with LDS / fp64 atomics and the transposed adjacency is filled through
 * LDS cursors: results are reproducible to ~1e-7 only, and a dynamic graph amplifies that into different neighbour lists).
 * Two-stage sums: DGCNN_DET blocks own contiguous row ranges and walk them in order, then one thread per column adds the
 * partials in block order into SLOT 0 of `stats` / `red` (caller-zeroed, same layout as above).  ws: dgcnn_det_workspace_bytes(F). */
int dgcnn_det_workspace_bytes(int F);
/* column sum / sum of squares of a materialised (rows, F) tensor (leading dimension ld): slim.batch_norm's statistics */
int dgcnn_colstats_det_f32(const float* Y, int64_t rows, int F, int64_t ld, double* stats, void* ws, size_t ws_bytes,
                           void* stream);
/* dgcnn_bn_bwd_reduce_f32 on a materialised Y (dmean == NULL: the k = 1 form) */
int dgcnn_bn_bwd_reduce_det_f32(const float* Y, int64_t R, int k, int F, const float* mean, const float* rstd,
                                const float* beta, int relu, const float* dmax, int64_t lddmax,
                                const float* dmean, int64_t lddmean, const float* mx_in, int64_t ldmx,
                                const float* cnt_in, double* red, void* ws, size_t ws_bytes, void* stream);
/* every bucket of the transposed adjacency (dgcnn_edge_csr_build: off, rev) in ascending edge order, out of place -> rev_sorted */
int dgcnn_edge_csr_sort(const int32_t* idx, int B, int N, int k, const int32_t* off, const int32_t* rev,
                        int32_t* rev_sorted, void* stream);

/* ---- head helpers (model.py:76-91) and residual add (ops.py:134) -------------------------- */
/* max_pool_v2 over the N points of each cloud: out[b][f] = max_i x[b][i][f], arg[b][f] = first i */
int dgcnn_global_max_f32(const float* x, int64_t ldx, int B, int N, int F, float* out, int32_t* arg,
                         void* stream);
/* its gradient: dx[b][arg[b][f]][f] += dout[b][f] */
int dgcnn_global_max_bwd_f32(const float* dout, const int32_t* arg, int B, int N, int F,
                             float* dx, int64_t lddx, void* stream);
/* out[g][f] = sum over the rows_per_group rows of group g (tf.tile^T, model.py:81) */
int dgcnn_group_colsum_f32(const float* x, int64_t ldx, int G, int rows_per_group, int F,
                           float* out, void* stream);
/* tf.tile (model.py:80-81): dst[g * rows_per_group + i][f] = src[g][f] -- only FC_LAYERS = 0 materialises the tiled global
 * feature (the dropout of model.py:90-91 then acts on it element by element); FC0 otherwise folds it into a per-cloud bias */
int dgcnn_tile_rows_f32(const float* src, int64_t lds, int G, int rows_per_group, int F, float* dst, int64_t ldd,
                        void* stream);
/* tf.nn.dropout(net, keep) (model.py:91): counter-based RNG keyed by (seed, element index) so the
 * backward regenerates the same mask.  y may alias x. */
int dgcnn_dropout_f32(const float* x, float* y, int64_t n, float keep, uint64_t seed, void* stream);
/* the same mask function with the seed read from DEVICE memory at execution time: the host advances *seed_dev between
 * replays of a captured HIP graph (a by-value seed would be frozen into the graph). */
int dgcnn_dropout_dev_f32(const float* x, float* y, int64_t n, float keep, const uint64_t* seed_dev, void* stream);
/* out = relu(a + b) (ops.py:134); bwd: d = dout * [out > 0] */
int dgcnn_add_relu_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t R, int F,
                       float* out, int64_t ldo, void* stream);
int dgcnn_relu_bwd_f32(const float* dout, int64_t lddo, const float* out, int64_t ldo, int64_t R, int F,
                       float* d, int64_t ldd, void* stream);
/* dst (R, Cp) dense <- [src (R, C) | zero columns]: C = 3 coordinates padded to float4 rows */
int dgcnn_pad_copy_f32(const float* src, int64_t lds, int C, float* dst, int Cp, int64_t R, void* stream);
/* strided 2-D copy / accumulate: dst (+)= src   (views of concatenated buffers, tf.concat) */
int dgcnn_copy2d_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t R, int F,
                     int accumulate, void* stream);

/* ---- dgcnn/trainval.py:39-52: softmax, accuracy, sparse softmax CE (x weight), mean --------
 * scal[0] += sum loss*w / rows ; scal[1] += #correct / rows (caller zeroes scal).
 * dlogits = (softmax - onehot) * w / rows (NULL to skip); softmax optional.                   */
int dgcnn_softmax_xent_f32(const float* logits, const int32_t* labels, const float* weight,
                           int64_t rows, int ncls, float* softmax, float* dlogits, float* scal,
                           void* stream);

/* ---- dgcnn/trainval.py:17,75-80: accumulators + tf.train.AdamOptimizer ---------------------- */
/* y = a*x + b*y over n elements */
int dgcnn_axpby_f32(const float* x, float a, float* y, float b, int64_t n, void* stream);
/* one Adam step on a flat parameter bucket; lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by caller */
int dgcnn_adam_f32(float* param, const float* grad, float* m, float* v, int64_t n,
                   float lr_t, float b1, float b2, float eps, void* stream);

/* ---- launch plans (csrc/plan.cc): the counterpart of replaying a static TF graph with sess.run (dgcnn/trainval.py:103-129) ----
 * dgcnn_plan_begin .. dgcnn_plan_end: every launch, memset, cross-stream wait and all-reduce the library issues in between is
 * executed as usual AND recorded (kernel, geometry, a copy of the argument values, streams); dgcnn_plan_replay re-issues the list
 * from one C loop on the same streams (the GPU sees the eager schedule; ~2 us of host time per launch).  The caller keeps every
 * device address of the recorded step valid and unchanged for as long as the plan is replayed, and keeps per-step host values
 * (dropout seed: device memory; Adam's step size: outside the plan) out of it.  One recorder per process.
 * dgcnn_stream_wait: `waiter` waits for everything issued so far on `signaller` (event record + wait; recordable).
 * dgcnn_memset_async: hipMemsetAsync (recordable): the zero-fills of the step go through it. */
int dgcnn_plan_begin(void);
int dgcnn_plan_end(void** plan_out);
int dgcnn_plan_abort(void);
int dgcnn_plan_replay(void* plan);
int dgcnn_plan_info(void* plan, int* kernels, int* memsets, int* waits, int* collectives);
int dgcnn_plan_destroy(void* plan);
int dgcnn_stream_wait(void* waiter, void* signaller);
int dgcnn_memset_async(void* ptr, int value, size_t bytes, void* stream);
/* A stream for work nothing on the critical path waits for (the reference has no counterpart: TF schedules its graph itself):
 * low_priority != 0 -> the device's least stream priority; reserve_cus > 0 -> created on a CU mask that leaves that many compute
 * units (evenly over the XCDs) to the other streams (then at the default priority: the masked entry point takes none).
 * The caller owns the stream (dgcnn_stream_destroy). */
int dgcnn_stream_create(int low_priority, int reserve_cus, void** stream_out);
int dgcnn_stream_destroy(void* stream);
int dgcnn_stream_priority_range(int* least, int* greatest);

/* ---- the gradient collective (trainval.py:64-73 mean over the towers; here: one process per GPU, RCCL over xGMI) ----
 * RCCL is dlopen'ed on first use (librccl.so, or $DGCNN_RCCL_LIB); no torch.distributed involved.
 * dgcnn_comm_unique_id: rank 0 fills 128 bytes; the host ships them to the other ranks (dgcnn/rccl.py).
 * dgcnn_comm_init: ncclCommInitRank for this process' current HIP device -> opaque communicator.
 * dgcnn_allreduce_f32: in-place SUM over the ranks on `stream`;  dgcnn_broadcast_f32: root's buffer to every rank. */
int dgcnn_comm_available(void);        /* dlopen + symbol lookup only: DGCNN_OK or DGCNN_EUNSUP with the reason in dgcnn_last_error */
int dgcnn_comm_unique_id(void* id128);
int dgcnn_comm_init(int world, int rank, const void* id128, void** comm_out);
int dgcnn_comm_destroy(void* comm);
/* ranks / this rank / HIP device of the communicator as RCCL reports them (ncclCommCount, ncclCommUserRank, ncclCommCuDevice) */
int dgcnn_comm_info(void* comm, int* nranks, int* rank, int* device);
int dgcnn_allreduce_f32(float* buf, int64_t count, void* comm, void* stream);
int dgcnn_broadcast_f32(float* buf, int64_t count, int root, void* comm, void* stream);
/* all-reduce calls / elements issued through the library so far, replayed launch plans included */
int dgcnn_comm_counters(int64_t* allreduce_calls, int64_t* allreduce_elems);

#ifdef __cplusplus
}
#endif
#endif /* DGCNN_HIP_H_ */
