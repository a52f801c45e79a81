// gemm.hip -- K3: fp32 MFMA (v_mfma_f32_32x32x2_f32) GEMM family for gfx950, the dgcnn_gemm_f32 dispatcher and
// the streaming kernels for the class dimension.  Since the bf16-split kernels (gemm_x3.hip) became the default
// arithmetic for float4-loadable operands, this family serves DGCNN_GEMM_ARITH=f32, unaligned / odd shapes and
// the edge-level conv0 forms that the model path keeps behind switches (A_EDGE*, E_SCATTER).
//
// One kernel template covers every 1x1 convolution of the path and its gradients
// (dgcnn/ops.py:47-52,62-70,125-133,153-160; dgcnn/model.py:46-53,65-72,94-101):
//   A source  A_ROW    A[m][k] row-major                      (forward, dgrad)
//             A_COL    A stored [k][m]                         (wgrad: X^T dY, split over k)
//             A_EDGE   rows are edges; E = [x_i, x_j - x_i] is gathered from (x, idx) straight
//                      into the LDS A tile -- the (B,N,k,2C) edge tensor of ops.py:21-40 is
//                      never written to HBM                    (conv0 forward)
//             A_EDGE_T E^T, reduction over edges               (conv0 wgrad)
//   B source  B_ROW    B[k][n] row-major;  B_COL  B stored [n][k]  (dgrad: dY W^T)
//   epilogue  E_STORE  C = acc (+ beta C) (+ per-cloud bias) (+ BN column statistics) or
//                      split-K partial;  E_SCATTER  dx[neighbour(row)][n] += acc (fp32 atomics)
//
// Tiling (wave64, 256 threads = 2x2 waves): block tile 128 x BN x 16, BN in {64,128};
// each wave owns (BM/2 x BN/2) as TM x TN tiles of 32x32 MFMA accumulators (16 VGPR each).  Both LDS tiles
// are k-major ([k][m] / [k][n]) so an MFMA operand read is one conflict-free ds_read_b32 of 32
// consecutive floats per half-wave; row-major sources are transposed on the LDS write with a +2
// row pad (4*(W+2) mod 32 = 8 -> the four k-quads of a half-wave hit disjoint bank octets).
// Global->register prefetch of tile t+1 overlaps the MFMAs of tile t; LDS is double buffered
// (one barrier per k-step).  fp32 MFMA == an fmaf chain in k order, so results are plain fp32.
#include "gemm_common.h"

namespace {

// VEC = every operand is float4-loadable (16-B aligned, leading dimensions and the contiguous
// extent multiples of 4).  Then each prefetch is ONE unconditional global_load_dwordx4 from a
// clamped (always valid) address, and the out-of-range predicate is applied when the value is
// written to LDS *after* the MFMAs of the current tile -- no control flow and no s_waitcnt between
// the loads and the MFMA block (the first version branched per element and drained vmcnt(0)
// between loads: rocprof showed the GEMMs at 40-50 % of the fp32 MFMA peak).  VEC = false is the
// generic (slow, fully predicated, scalar) path for odd shapes such as C = 3 or N = 2.
template <int ASRC, int BSRC, int EPI, int BM, int BN, bool VEC>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmP p) {
  constexpr bool A_TRANS = (ASRC == A_ROW || ASRC == A_EDGE);  // needs transposing LDS store
  constexpr bool B_TRANS = (BSRC == B_COL);
  constexpr int SA = BM + (A_TRANS ? 2 : 4);
  constexpr int SB = BN + (B_TRANS ? 2 : 4);
  constexpr int NVA = BM / 64;
  constexpr int NVB = BN / 64;
  constexpr int TM = BM / 64;
  constexpr int TN = BN / 64;

  __shared__ __attribute__((aligned(16))) float smem[2 * BK * SA + 2 * BK * SB];
  float* As = smem;
  float* Bs = smem + 2 * BK * SA;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = t >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // Tile order.  MI355X dispatches block b to XCD b % 8 (private L2 each).  When several column
  // blocks share one A row-panel (ntiles > 1, many row panels) they are given the same (id % 8) so
  // the panel is fetched into ONE L2; otherwise (few row panels, split-K) ids map 1:1 so that
  // consecutive blocks spread over all 8 XCDs.
  const int id = blockIdx.x;
  int mt, nt;
  if (p.xcd_group) {
    mt = ((id >> 3) / p.ntiles) * 8 + (id & 7);
    nt = (id >> 3) % p.ntiles;
  } else {
    mt = id / p.ntiles;
    nt = id % p.ntiles;
  }
  if (mt >= p.mtiles) return;
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  const int z = blockIdx.z;
  const int kbeg = z * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? (kbeg + p.kchunk) : p.K;

  // ---- per-thread source descriptors ----
  // transposing items: row = (t>>2) + 64*i, kq = t&3 ; direct items: kk = t/(W/4) + i*(1024/W), c4 = (t%(W/4))*4
  const float* a_ptr[NVA];
  const float* a_ptr2[NVA];   // EDGE: neighbour row
  bool a_ok[NVA];
#pragma unroll
  for (int i = 0; i < NVA; ++i) {
    a_ptr[i] = nullptr; a_ptr2[i] = nullptr; a_ok[i] = false;
    if (ASRC == A_ROW) {
      const int row = m0 + (t >> 2) + 64 * i;
      a_ok[i] = row < p.M;
      a_ptr[i] = p.A + (int64_t)imin(row, p.M - 1) * p.lda;
    } else if (ASRC == A_EDGE) {
      const int row = m0 + (t >> 2) + 64 * i;
      a_ok[i] = row < p.M;
      const int er = imin(row, p.M - 1);
      const int g = er / p.knn;
      const int nb = (g / p.npts) * p.npts + p.idx[er];
      a_ptr[i] = p.x + (int64_t)g * p.ldx;
      a_ptr2[i] = p.x + (int64_t)nb * p.ldx;
    }
  }

  bool oka[NVA], okb[NVB];    // predicates of the tile currently held in ra / rb

  auto fetch_a = [&](int i, int k0) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
      if (ASRC == A_ROW) {
        const int kc = k0 + 4 * (t & 3);
        oka[i] = a_ok[i] && (kc < kend);
        v = LD4(a_ptr[i] + imin(kc, kend - 4));
      } else if (ASRC == A_EDGE) {
        const int kc = k0 + 4 * (t & 3);
        const int C = p.cch;
        oka[i] = a_ok[i] && (kc < kend);
        const int kcc = imin(kc, kend - 4);
        if (p.edge_nbr) {                         // wave-uniform: A rows are the neighbour rows themselves
          v = LD4(a_ptr2[i] + kcc);
        } else if (k0 + BK <= C) {                // wave-uniform: the whole k-chunk is centre features
          v = LD4(a_ptr[i] + kcc);
        } else {
          const bool cen = kcc < C;
          const int col = cen ? kcc : kcc - C;
          const float4 xc = LD4(a_ptr[i] + col);
          const float4 xn = LD4((cen ? a_ptr[i] : a_ptr2[i]) + col);
          const float4 df = sub4(xn, xc);
          v = cen ? xc : df;
        }
      } else if (ASRC == A_COL) {
        const int kk = k0 + t / (BM / 4) + i * (1024 / BM);
        const int m = m0 + (t % (BM / 4)) * 4;
        oka[i] = (kk < kend) && (m < p.M);
        v = LD4(p.A + (int64_t)imin(kk, kend - 1) * p.lda + imin(m, p.M - 4));
      } else {  // A_EDGE_T: element (m = channel of E, kk = edge row)
        const int er = k0 + t / (BM / 4) + i * (1024 / BM);
        const int m = m0 + (t % (BM / 4)) * 4;
        const int C = p.cch;
        oka[i] = (er < kend) && (m < p.M);
        const int erc = imin(er, kend - 1);
        const int g = erc / p.knn;
        const int nb = (g / p.npts) * p.npts + p.idx[erc];
        const float* pc = p.x + (int64_t)g * p.ldx;
        const float* pn = p.x + (int64_t)nb * p.ldx;
        const int mc = imin(m, p.M - 4);
        if (p.edge_nbr) {
          v = LD4(pn + mc);
        } else {
          const bool cen = mc < C;
          const int col = cen ? mc : mc - C;
          const float4 xc = LD4(pc + col);
          const float4 xn = LD4((cen ? pc : pn) + col);
          const float4 df = sub4(xn, xc);
          v = cen ? xc : df;
        }
      }
      return v;
    }
    oka[i] = true;
    if (ASRC == A_ROW) {
      const int kc = k0 + 4 * (t & 3);
      if (a_ok[i]) {
        if (kc + 0 < kend) v.x = a_ptr[i][kc + 0];
        if (kc + 1 < kend) v.y = a_ptr[i][kc + 1];
        if (kc + 2 < kend) v.z = a_ptr[i][kc + 2];
        if (kc + 3 < kend) v.w = a_ptr[i][kc + 3];
      }
    } else if (ASRC == A_EDGE) {
      const int kc = k0 + 4 * (t & 3);
      const int C = p.cch;
      if (a_ok[i]) {
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = kc + q;
          e[q] = 0.f;
          if (c < kend) e[q] = p.edge_nbr ? a_ptr2[i][c] : ((c < C) ? a_ptr[i][c] : (a_ptr2[i][c - C] - a_ptr[i][c - C]));
        }
        v = make_float4(e[0], e[1], e[2], e[3]);
      }
    } else if (ASRC == A_COL) {
      const int kk = k0 + t / (BM / 4) + i * (1024 / BM);
      const int m = m0 + (t % (BM / 4)) * 4;
      if (kk < kend) {
        const float* s = p.A + (int64_t)kk * p.lda + m;
        if (m + 0 < p.M) v.x = s[0];
        if (m + 1 < p.M) v.y = s[1];
        if (m + 2 < p.M) v.z = s[2];
        if (m + 3 < p.M) v.w = s[3];
      }
    } else {
      const int er = k0 + t / (BM / 4) + i * (1024 / BM);
      const int m = m0 + (t % (BM / 4)) * 4;
      const int C = p.cch;
      if (er < kend) {
        const int g = er / p.knn;
        const int nb = (g / p.npts) * p.npts + p.idx[er];
        const float* pc = p.x + (int64_t)g * p.ldx;
        const float* pn = p.x + (int64_t)nb * p.ldx;
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = m + q;
          e[q] = 0.f;
          if (c < p.M) e[q] = p.edge_nbr ? pn[c] : ((c < C) ? pc[c] : (pn[c - C] - pc[c - C]));
        }
        v = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
    return v;
  };

  auto fetch_b = [&](int i, int k0) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
      if (BSRC == B_ROW) {
        const int kk = k0 + t / (BN / 4) + i * (1024 / BN);
        const int n = n0 + (t % (BN / 4)) * 4;
        okb[i] = (kk < kend) && (n < p.N);
        v = LD4(p.B + (int64_t)imin(kk, kend - 1) * p.ldb + imin(n, p.N - 4));
      } else {
        const int n = n0 + (t >> 2) + 64 * i;
        const int kc = k0 + 4 * (t & 3);
        okb[i] = (n < p.N) && (kc < kend);
        v = LD4(p.B + (int64_t)imin(n, p.N - 1) * p.ldb + imin(kc, kend - 4));
      }
      return v;
    }
    okb[i] = true;
    if (BSRC == B_ROW) {
      const int kk = k0 + t / (BN / 4) + i * (1024 / BN);
      const int n = n0 + (t % (BN / 4)) * 4;
      if (kk < kend) {
        const float* s = p.B + (int64_t)kk * p.ldb + n;
        if (n + 0 < p.N) v.x = s[0];
        if (n + 1 < p.N) v.y = s[1];
        if (n + 2 < p.N) v.z = s[2];
        if (n + 3 < p.N) v.w = s[3];
      }
    } else {
      const int n = n0 + (t >> 2) + 64 * i;
      const int kc = k0 + 4 * (t & 3);
      if (n < p.N) {
        const float* s = p.B + (int64_t)n * p.ldb + kc;
        if (kc + 0 < kend) v.x = s[0];
        if (kc + 1 < kend) v.y = s[1];
        if (kc + 2 < kend) v.z = s[2];
        if (kc + 3 < kend) v.w = s[3];
      }
    }
    return v;
  };

  auto store_a = [&](int buf, int i, float4 v) {
    float* d = As + buf * BK * SA;
    if (!oka[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (A_TRANS) {
      const int row = (t >> 2) + 64 * i, kq = t & 3;
      d[(4 * kq + 0) * SA + row] = v.x;
      d[(4 * kq + 1) * SA + row] = v.y;
      d[(4 * kq + 2) * SA + row] = v.z;
      d[(4 * kq + 3) * SA + row] = v.w;
    } else {
      const int kk = t / (BM / 4) + i * (1024 / BM), c4 = (t % (BM / 4)) * 4;
      *reinterpret_cast<float4*>(&d[kk * SA + c4]) = v;
    }
  };
  auto store_b = [&](int buf, int i, float4 v) {
    float* d = Bs + buf * BK * SB;
    if (!okb[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (B_TRANS) {
      const int row = (t >> 2) + 64 * i, kq = t & 3;
      d[(4 * kq + 0) * SB + row] = v.x;
      d[(4 * kq + 1) * SB + row] = v.y;
      d[(4 * kq + 2) * SB + row] = v.z;
      d[(4 * kq + 3) * SB + row] = v.w;
    } else {
      const int kk = t / (BN / 4) + i * (1024 / BN), c4 = (t % (BN / 4)) * 4;
      *reinterpret_cast<float4*>(&d[kk * SB + c4]) = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[NVA], rb[NVB];
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
#pragma unroll
    for (int i = 0; i < NVA; ++i) ra[i] = fetch_a(i, kbeg);
#pragma unroll
    for (int i = 0; i < NVB; ++i) rb[i] = fetch_b(i, kbeg);
#pragma unroll
    for (int i = 0; i < NVA; ++i) store_a(0, i, ra[i]);
#pragma unroll
    for (int i = 0; i < NVB; ++i) store_b(0, i, rb[i]);
  }
  __syncthreads();

  const int a_off = wr * (BM / 2) + l31;
  const int b_off = wc * (BN / 2) + l31;
  int knext = 0;
  auto prefetch_slot = [&](int sl, bool more) {     // one prefetch load per MFMA group (slots 0..NVA+NVB-1)
    // VEC: unconditional (addresses are clamped into the operand, so the extra fetch after the last
    // tile is harmless) -- a branch here would split the block and make hipcc drain vmcnt(0) per load
    if (!VEC && !more) return;
#pragma unroll
    for (int i = 0; i < NVA; ++i)
      if (sl == i) ra[i] = fetch_a(i, knext);
#pragma unroll
    for (int i = 0; i < NVB; ++i)
      if (sl == NVA + i) rb[i] = fetch_b(i, knext);
  };
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    knext = kbeg + (kt + 1) * BK;
    // The prefetch of tile kt+1 is NOT issued up front: its NVA+NVB loads (and their address VALU) are
    // slotted one per MFMA group below, so they issue in the shadow of this wave's own MFMAs
    // (measured: up-front prefetch costs 15 % of the MFMA rate, profiles/r01_gemm_ablation.txt).
    const float* as = As + buf * BK * SA + lh * SA + a_off;
    const float* bs = Bs + buf * BK * SB + lh * SB + b_off;
    // operand reads run one k-pair ahead of the MFMAs (two register sets, static indices); the
    // sched_barriers pin "issue next reads -> MFMAs of the current pair" so the LDS latency of pair
    // s+1 hides under the 8 MFMAs of pair s (hipcc otherwise sinks each read next to its use).
    float a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a0[i] = as[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b0[j] = bs[j * 32];
#pragma unroll
    for (int s = 0; s < BK / 2; s += 2) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a1[i] = as[2 * (s + 1) * SA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b1[j] = bs[2 * (s + 1) * SB + j * 32];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
      prefetch_slot(s, more);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < BK / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a0[i] = as[2 * (s + 2) * SA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[j] = bs[2 * (s + 2) * SB + j * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
      prefetch_slot(s + 1, more);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < NVA; ++i) store_a(buf ^ 1, i, ra[i]);
#pragma unroll
      for (int i = 0; i < NVB; ++i) store_b(buf ^ 1, i, rb[i]);
    }
    __syncthreads();
  }

  gemm_epilogue<EPI, BM, BN, VEC, TM, TN>(p, acc, smem, m0, n0, mt, z, t, wr, wc, l31, lh);
}

// C (+)= sum over splits of the partial tiles.  64 consecutive elements x 16 split-lanes per block
// (1024 threads): coalesced 256-B rows, 16-way split parallelism, 4 independent loads in flight per
// lane, fixed summation order (deterministic).
constexpr int RL = 16;
__global__ __launch_bounds__(64 * RL) void reduce_partials_kernel(const float* __restrict__ part, int splits, int M,
                                                                 int N, float* __restrict__ C, int64_t ldc, float beta) {
  __shared__ float sh[RL][64];
  const int64_t MN = (int64_t)M * N;
  const int64_t e = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int zl = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < MN) {
    int z = zl;
    for (; z + 3 * RL < splits; z += 4 * RL) {
      s0 += part[(int64_t)z * MN + e];
      s1 += part[(int64_t)(z + RL) * MN + e];
      s2 += part[(int64_t)(z + 2 * RL) * MN + e];
      s3 += part[(int64_t)(z + 3 * RL) * MN + e];
    }
    for (; z < splits; z += RL) s0 += part[(int64_t)z * MN + e];
  }
  sh[zl][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (zl == 0 && e < MN) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < RL; ++q) s += sh[q][threadIdx.x];
    const int row = (int)(e / N), col = (int)(e % N);
    float* c = C + (int64_t)row * ldc + col;
    *c = (beta != 0.f) ? (s + beta * *c) : s;
  }
}

// ---- skinny products (the class dimension: Final layer, N = NUM_CLASS <= 4).  An MFMA tile would be > 95 %
// padding and N = 2 is not float4-loadable, so these ran on the scalar generic kernel (~0.1 ms per step);
// they are plain streaming problems.
// NN:  C[M][N] = A[M][K] B[K][N] (+ beta C) (+ column sums), N <= 4, K % 4 == 0.  One wave per row at a time:
//      lanes own float4 chunks of k, wave-reduce the N partial dots.
template <int N>
__global__ __launch_bounds__(256) void skinny_nn_kernel(const float* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                        int64_t ldc, int M, int K, float beta, double* __restrict__ stats,
                                                        int nslots) {
  extern __shared__ float bs[];                  // [K][N] then [4 waves][2][N] for the statistics
  float* red = bs + (size_t)K * N;
  for (int e = threadIdx.x; e < K * N; e += 256) bs[e] = B[(int64_t)(e / N) * ldb + (e % N)];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nw = gridDim.x * 4;
  float cs[N], cq[N];
#pragma unroll
  for (int n = 0; n < N; ++n) { cs[n] = 0.f; cq[n] = 0.f; }
  for (int r = blockIdx.x * 4 + wv; r < M; r += nw) {
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = 0.f;
    const float* a = A + (int64_t)r * lda;
    for (int k4 = lane * 4; k4 < K; k4 += 256) {
      const float4 v = ld4(a + k4);
      const float* b = bs + k4 * N;
#pragma unroll
      for (int n = 0; n < N; ++n) acc[n] += v.x * b[n] + v.y * b[N + n] + v.z * b[2 * N + n] + v.w * b[3 * N + n];
    }
#pragma unroll
    for (int n = 0; n < N; ++n)
      for (int d = 32; d > 0; d >>= 1) acc[n] += __shfl_down(acc[n], d, 64);
    if (lane == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n) {
        float* c = C + (int64_t)r * ldc + n;
        const float v = (beta != 0.f) ? (acc[n] + beta * *c) : acc[n];
        *c = v;
        cs[n] += v;
        cq[n] += v * v;
      }
    }
  }
  if (stats) {
    if (lane == 0) {                             // the four waves' partial sums, added below in wave order (fixed)
#pragma unroll
      for (int n = 0; n < N; ++n) { red[wv * 2 * N + n] = cs[n]; red[wv * 2 * N + N + n] = cq[n]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * N) {
      const int which = threadIdx.x / N, n = threadIdx.x % N;
      const float a = ((red[threadIdx.x] + red[2 * N + threadIdx.x]) + red[4 * N + threadIdx.x]) + red[6 * N + threadIdx.x];
      atomicAdd(stats + ((int64_t)(blockIdx.x % nslots) * 2 + which) * N + n, (double)a);
    }
  }
}

// TN:  partial[z][M][N] = sum over the rows of block z of A[r][m] B[r][n]   (A stored [R][M], M % 4 == 0, N <= 4);
//      lanes own float4 chunks of m; combined by reduce_partials_kernel.
template <int N>
__global__ __launch_bounds__(256) void skinny_tn_kernel(const float* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ B, int64_t ldb, int M, int R,
                                                        float* __restrict__ part) {
  __shared__ float sh[4][64 * 4 * 4];            // per wave: 64 lanes x 4 m x N
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rows_per = (R + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per, r1 = (r0 + rows_per < R) ? (r0 + rows_per) : R;
  float* out = part + (int64_t)blockIdx.x * M * N;
  for (int mc = 0; mc < M; mc += 256) {             // uniform trip count (barriers inside)
    const int m4 = mc + lane * 4;
    const bool on = m4 < M;
    const int m4c = on ? m4 : 0;
    float acc[4][N];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n = 0; n < N; ++n) acc[q][n] = 0.f;
    int r = r0 + wv;
    for (; r + 12 < r1; r += 16) {                // 4 rows of this wave in flight
      float4 v[4];
      float b[4][N];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = ld4(A + (int64_t)(r + 4 * i) * lda + m4c);
#pragma unroll
        for (int n = 0; n < N; ++n) b[i][n] = B[(int64_t)(r + 4 * i) * ldb + n];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < N; ++n) {
          acc[0][n] += v[i].x * b[i][n]; acc[1][n] += v[i].y * b[i][n];
          acc[2][n] += v[i].z * b[i][n]; acc[3][n] += v[i].w * b[i][n];
        }
    }
    for (; r < r1; r += 4) {
      const float4 v = ld4(A + (int64_t)r * lda + m4c);
#pragma unroll
      for (int n = 0; n < N; ++n) {
        const float b = B[(int64_t)r * ldb + n];
        acc[0][n] += v.x * b; acc[1][n] += v.y * b; acc[2][n] += v.z * b; acc[3][n] += v.w * b;
      }
    }
    // combine the 4 waves (fixed order), write the block's partial tile
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n = 0; n < N; ++n) sh[wv][(lane * 4 + q) * 4 + n] = acc[q][n];
    __syncthreads();
    if (wv == 0 && on) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int n = 0; n < N; ++n) {
          const int e = (lane * 4 + q) * 4 + n;
          out[(int64_t)(m4 + q) * N + n] = (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
        }
    }
  }
}

// NT with a tiny reduction:  C[M][Nc] = A[M][K] B[Nc][K]^T (+ beta C),  K <= 4, Nc % 4 == 0: elementwise.
template <int K>
__global__ __launch_bounds__(256) void skinny_nt_kernel(const float* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                        int64_t ldc, int M, int Nc, float beta) {
  const int NV = Nc / 4;
  const int64_t items = (int64_t)M * NV;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < items; it += (int64_t)gridDim.x * 256) {
    const int r = (int)(it / NV), c4 = (int)(it % NV) * 4;
    float a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = A[(int64_t)r * lda + k];
    float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < K; ++k) o[q] += a[k] * B[(int64_t)(c4 + q) * ldb + k];
    float4* dst = reinterpret_cast<float4*>(C + (int64_t)r * ldc + c4);
    if (beta != 0.f) {
      const float4 old = *dst;
      o[0] += beta * old.x; o[1] += beta * old.y; o[2] += beta * old.z; o[3] += beta * old.w;
    }
    *dst = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// conv0 wgrad for raw point clouds (C <= 4, i.e. 2C <= 8 rows of dW0): the MFMA tile would be 95 %
// padding and its generic loader is scalar.  HBM-bound instead: stream dY once (coalesced 256-B
// rows), rebuild the 2C edge features of each edge from (x, idx) -- wave-uniform, so they sit in
// SGPRs -- and keep 2C x (F/64) partial sums per lane.  One partial tile per block, combined by
// reduce_partials_kernel (deterministic).
template <int CC>
__global__ __launch_bounds__(256) void edge_wgrad_smallc_kernel(const float* __restrict__ x, int64_t ldx,
                                                                const int32_t* __restrict__ idx,
                                                                const float* __restrict__ dY, int64_t Me, int npts,
                                                                int knn, int C, int F, int chunk, int nbr_only,
                                                                float* __restrict__ partial) {
  constexpr int FB = 4;                       // F <= 256
  __shared__ float sh[4][2 * CC * 64];
  const int t = threadIdx.x;
  const int fl = t & 63, sub = t >> 6;
  float acc[FB][2 * CC];
#pragma unroll
  for (int fb = 0; fb < FB; ++fb)
#pragma unroll
    for (int c = 0; c < 2 * CC; ++c) acc[fb][c] = 0.f;
  const int64_t e0 = (int64_t)blockIdx.x * chunk;
  const int64_t e1 = (e0 + chunk < Me) ? (e0 + chunk) : Me;
  // 4 edges in flight per lane-group: index, feature and dY loads of all four are issued before
  // the first use (the chain idx -> x -> fma is otherwise three exposed memory latencies per edge)
  for (int64_t eb = e0 + sub; eb < e1; eb += 16) {
    int64_t ee[4], gg[4], nn[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ok[u] = (eb + 4 * u) < e1;
      ee[u] = ok[u] ? (eb + 4 * u) : (e1 - 1);
      gg[u] = ee[u] / knn;
      nn[u] = (gg[u] / npts) * npts + idx[ee[u]];
    }
    float ev[4][2 * CC];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        const float xc = (c < C) ? x[gg[u] * ldx + c] : 0.f;
        const float xn = (c < C) ? x[nn[u] * ldx + c] : 0.f;
        ev[u][c] = nbr_only ? xn : xc;               // nbr_only: the C rows are x_j itself
        ev[u][CC + c] = nbr_only ? 0.f : xn - xc;
      }
    float dv[4][FB];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb) {
        const int f = fl + 64 * fb;
        dv[u][fb] = (f < F && ok[u]) ? dY[ee[u] * F + f] : 0.f;
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int fb = 0; fb < FB; ++fb)
#pragma unroll
        for (int c = 0; c < 2 * CC; ++c) acc[fb][c] = fmaf(ev[u][c], dv[u][fb], acc[fb][c]);
  }
  const int rows_out = nbr_only ? C : 2 * C;
  float* out = partial + (int64_t)blockIdx.x * rows_out * F;
#pragma unroll
  for (int fb = 0; fb < FB; ++fb) {
    if (64 * fb >= F) break;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2 * CC; ++c) sh[sub][c * 64 + fl] = acc[fb][c];
    __syncthreads();
    if (sub == 0) {
      const int f = fl + 64 * fb;
      if (f < F) {
#pragma unroll
        for (int c = 0; c < 2 * CC; ++c) {
          const float v = (sh[0][c * 64 + fl] + sh[1][c * 64 + fl]) + (sh[2][c * 64 + fl] + sh[3][c * 64 + fl]);
          const int cr = (c < CC) ? c : (C + (c - CC));     // row of dW0: centre rows 0..C-1, diff rows C..2C-1
          if ((c < CC ? c : c - CC) < C && cr < rows_out) out[(int64_t)cr * F + f] = v;
        }
      }
    }
  }
}

template <int ASRC, int BSRC, int EPI, int BM>
void launch_bm(GemmP& p, hipStream_t st, bool vec, int bn) {
  p.mtiles = (int)dg::cdiv(p.M, BM);
  p.ntiles = (int)dg::cdiv(p.N, bn);
  p.xcd_group = (p.ntiles > 1 && p.mtiles >= 16 && p.splits == 1) ? 1 : 0;
  const unsigned gx = p.xcd_group ? (unsigned)(dg::cdiv(p.mtiles, 8) * 8 * p.ntiles) : (unsigned)(p.mtiles * p.ntiles);
  dim3 grid(gx, 1, (unsigned)p.splits);
  if (bn == 64) {
    if (vec) dg::launch((gemm_kernel<ASRC, BSRC, EPI, BM, 64, true>), grid, dim3(NT), 0, st, p);
    else dg::launch((gemm_kernel<ASRC, BSRC, EPI, BM, 64, false>), grid, dim3(NT), 0, st, p);
  } else {
    if (vec) dg::launch((gemm_kernel<ASRC, BSRC, EPI, BM, 128, true>), grid, dim3(NT), 0, st, p);
    else dg::launch((gemm_kernel<ASRC, BSRC, EPI, BM, 128, false>), grid, dim3(NT), 0, st, p);
  }
}

inline int tile_m(int M, int N, int splits) {
  // Measured on MI355X in round 1 (profiles/r01_gemm_tile_height.txt): 128-row tiles (4 workgroups = 16 waves per CU) match
  // or beat 192 / 256 rows on every shape of the model for the native fp32-MFMA kernel; the taller instantiations and the
  // LDS-DMA staged variant (measured equal, profiles/r01_gemm_ablation.txt) were removed in round 2.
  (void)M; (void)N; (void)splits;
  return 128;
}

// column-tile width: 128, or 64 for narrow outputs and for problems too small to fill the chip with 128 x 128 tiles
inline int tile_n(int M, int N, int K) {
  return (N <= 64 || dg::cdiv(M, 128) * dg::cdiv(N, 128) * dg::cdiv(K, 256) < 256) ? 64 : 128;
}

// row tiles (= writer workgroups of the epilogue's column sums) of a product without transA, as launch<> below picks them
inline int stat_writers(const GemmP& p, bool b_col) {
  const bool a_ext = (p.K % 4 == 0 && p.K >= 4);
  const bool b_ext = b_col ? (p.K % 4 == 0 && p.K >= 4) : (p.N % 4 == 0 && p.N >= 4);
  const bool vec = p.avec && p.bvec && a_ext && b_ext;
  const int rows = (vec && dg::gemm_arith() != 0) ? dg::x3_tile_m(p.M, p.N, p.K) : 128;
  return (int)dg::cdiv(p.M, rows);
}

template <int ASRC, int BSRC, int EPI>
int launch(GemmP& p, hipStream_t st, const char* what) {
  int bn = tile_n(p.M, p.N, p.K);
  // float4 path: pointers / leading dimensions checked by the caller (avec, bvec); here the extents
  const bool a_ext = (ASRC == A_ROW || ASRC == A_EDGE) ? (p.K % 4 == 0 && p.K >= 4) : (p.M % 4 == 0 && p.M >= 4);
  const bool b_ext = (BSRC == B_ROW) ? (p.N % 4 == 0 && p.N >= 4) : (p.K % 4 == 0 && p.K >= 4);
  const bool vec = p.avec && p.bvec && a_ext && b_ext;
  if (EPI == E_STORE) {
    if (p.splits > 1) p.cvec = (p.N % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.partial) & 15) == 0);
    else p.cvec = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0);
  }
  constexpr bool plain = (ASRC == A_ROW || ASRC == A_COL) && EPI == E_STORE;
  if (plain && vec && dg::gemm_arith() != 0) {          // bf16-split kernel (gemm_x3.hip)
    p.bm = dg::x3_tile_m(p.M, p.N, p.K);
    if (dg::x3_tile_n(p.M, p.N, p.K) == 256) bn = 256;
    p.mtiles = (int)dg::cdiv(p.M, p.bm);
    p.ntiles = (int)dg::cdiv(p.N, bn);
    p.xcd_group = (p.ntiles > 1 && p.mtiles >= 16 && p.splits == 1) ? 1 : 0;
    // reproducible configuration (more slots than DGCNN_STAT_SLOTS): every row tile must own its slot -- refuse loudly instead of
    // letting two tiles share one (the host sizes the buffer from dgcnn_gemm_stat_writers)
    DG_REQUIRE(!(p.stats && p.splits == 1 && p.stat_slots > DGCNN_STAT_SLOTS && p.mtiles > p.stat_slots), DGCNN_EINVAL,
               "%s: %d row tiles write column sums into %d statistics slots (dgcnn_set_stat_slots)", what, p.mtiles, p.stat_slots);
    dg::launch_gemm_x3(ASRC, BSRC, &p, st, bn, dg::gemm_arith());
  } else {
    p.bm = 128;
    DG_REQUIRE(!(p.stats && p.splits == 1 && p.stat_slots > DGCNN_STAT_SLOTS && dg::cdiv(p.M, 128) > p.stat_slots), DGCNN_EINVAL,
               "%s: %d row tiles write column sums into %d statistics slots (dgcnn_set_stat_slots)", what, (int)dg::cdiv(p.M, 128), p.stat_slots);
    launch_bm<ASRC, BSRC, EPI, 128>(p, st, vec, bn);
  }
  int rc = dg::check_launch(what);
  if (rc) return rc;
  if (p.splits > 1) {
    const int64_t n = (int64_t)p.M * p.N;
    dg::launch(reduce_partials_kernel, dim3((unsigned)dg::cdiv(n, 64)), dim3(64 * RL), 0, st,
                       p.partial, p.splits, p.M, p.N, p.C, p.ldc, p.beta);
    rc = dg::check_launch(what);
  }
  return rc;
}

inline bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// choose a split of the reduction dimension
//   bf16-split kernels: every k-chunk is pinned to one XCD (gemm_common.h:block_tile), so the number of chunks is
//     8 x (chunks per XCD), sized so that the chunks of an XCD fill its 32 CUs in whole rounds;
//   native kernels: ~1024 workgroups in flight.
int plan_splits(GemmP& p, void* ws, size_t ws_bytes, const char* what) {
  int bn = tile_n(p.M, p.N, p.K);
  if (dg::gemm_arith() != 0 && p.avec && p.bvec && dg::x3_tile_n(p.M, p.N, p.K) == 256) bn = 256;
  p.bm = 128;
  p.zmajor = 0;
  const bool x3 = dg::gemm_arith() != 0 && p.avec && p.bvec;
  // the 256-row bf16-split kernel runs one 768-thread workgroup per CU: aim at 2 rounds of 256 workgroups
  const int tm = x3 ? dg::x3_tile_m(p.M, p.N, p.K) : 128;
  const bool big = tm >= 192;                  // (one workgroup per CU)
  const int64_t tiles = dg::cdiv(p.M, big ? tm : 128) * dg::cdiv(p.N, bn);
  int64_t s = dg::cdiv(big ? 512 : 1024, tiles);
  const int64_t maxs = p.K / 256 > 0 ? p.K / 256 : 1;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (x3 && maxs >= 8) {
    const int64_t cap = big ? 32 : 64;           // workgroups one XCD (32 CUs) runs at a time
    double best_eff = 0.0;
    int64_t best = 0;
    for (int64_t R = 1; R <= 3; ++R) {           // rounds of `cap` workgroups per XCD
      int64_t per = (cap * R) / tiles;           // k-chunks per XCD
      if (per < 1) per = 1;
      if (8 * per > maxs) per = maxs / 8;
      const double eff = (double)(tiles * per) / (double)(cap * dg::cdiv(tiles * per, cap));
      if (eff > best_eff + 0.03) { best_eff = eff; best = 8 * per; }
    }
    if (best >= 8) { s = best; p.zmajor = 1; }
  }
  int64_t chunk = dg::cdiv(dg::cdiv(p.K, s), 32) * 32;     // multiple of both kernels' k-slab
  s = dg::cdiv(p.K, chunk);
  p.splits = (int)s;
  p.kchunk = (int)chunk;
  p.partial = nullptr;
  if (s > 1) {
    const size_t need = (size_t)s * p.M * p.N * sizeof(float);
    if (!ws || ws_bytes < need) {
      dg::set_error("%s: workspace too small (%zu < %zu bytes)", what, ws_bytes, need);
      return DGCNN_ENOSPC;
    }
    p.partial = reinterpret_cast<float*>(ws);
  } else {
    p.zmajor = 0;
  }
  return DGCNN_OK;
}

}  // namespace

namespace dg {
void launch_reduce_partials(const float* part, int splits, int M, int N, float* C, int64_t ldc, float beta, hipStream_t st) {
  dg::launch(reduce_partials_kernel, dim3((unsigned)dg::cdiv((int64_t)M * N, 64)), dim3(64 * RL), 0, st, part, splits, M, N,
                     C, ldc, beta);
}
}  // namespace dg

__global__ void colmax_decode_kernel(const unsigned long long* __restrict__ keys, int64_t n, float* __restrict__ vals,
                                     int32_t* __restrict__ arg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  if (k == 0ull) {
    // no row ever beat the zero-initialised key: every value of the column compared false (NaN).  The pass this epilogue
    // replaces (dgcnn_global_max_f32) always reports an in-range row: NaN at row 0, so the backward scatters inside the cloud
    vals[i] = __builtin_nanf("");
    arg[i] = 0;
    return;
  }
  vals[i] = f32_from_ordered((unsigned)(k >> 32));
  arg[i] = (int32_t)(0xffffffffu - (unsigned)(k & 0xffffffffu));
}

extern "C" int dgcnn_colmax_decode_f32(const void* keys, int64_t n, float* vals, int32_t* arg, void* stream) {
  DG_REQUIRE(keys && vals && arg && n > 0, DGCNN_EINVAL, "dgcnn_colmax_decode_f32: bad args");
  dg::launch(colmax_decode_kernel, dim3((unsigned)dg::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned long long*)keys, n, vals, arg);
  return dg::check_launch("dgcnn_colmax_decode_f32");
}

extern "C" int dgcnn_gemm_f32(int transA, int transB, int M, int N, int K,
                              const float* A, int64_t lda, const float* B, int64_t ldb,
                              float* C, int64_t ldc, float beta,
                              const float* gbias, int64_t ldgbias, int rows_per_group,
                              double* stats, void* colmax_keys, int colmax_rows_per_group,
                              void* ws, size_t ws_bytes, void* stream) {
  DG_REQUIRE(A && B && C, DGCNN_EINVAL, "dgcnn_gemm_f32: null pointer");
  DG_REQUIRE(M > 0 && N > 0 && K > 0, DGCNN_EINVAL, "dgcnn_gemm_f32: bad shape %d %d %d", M, N, K);
  DG_REQUIRE(!(transA && transB), DGCNN_EUNSUP, "dgcnn_gemm_f32: transA && transB unsupported");
  DG_REQUIRE(!gbias || rows_per_group > 0, DGCNN_EINVAL, "dgcnn_gemm_f32: rows_per_group");
  GemmP p = {};
  p.stat_slots = dg::stat_slots();
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.beta = beta;
  p.gbias = gbias; p.ldgbias = ldgbias; p.rpg = rows_per_group > 0 ? rows_per_group : 1;
  p.stats = stats;
  p.splits = 1; p.kchunk = K;
  DG_REQUIRE(!colmax_keys || (colmax_rows_per_group > 0 && colmax_rows_per_group % 256 == 0 && !transA && N > 4), DGCNN_EUNSUP,
             "dgcnn_gemm_f32: the column-maximum epilogue needs rows_per_group %% 256 == 0 (a tile inside one group), no transA, N > 4");
  p.colmax = reinterpret_cast<unsigned long long*>(colmax_keys); p.colmax_rpg = colmax_rows_per_group;
  p.gbvec = gbias && (ldgbias % 4 == 0) && aligned16(gbias);
  p.avec = (lda % 4 == 0) && aligned16(A);
  p.bvec = (ldb % 4 == 0) && aligned16(B);
  hipStream_t st = (hipStream_t)stream;
  // ---- skinny shapes (class dimension): streaming kernels instead of 95 %-padded MFMA tiles
  if (!gbias && !transA && !transB && N <= 4 && K % 4 == 0 && K <= 4096 && p.avec) {
    const size_t sh = sizeof(float) * ((size_t)K * N + 8 * N);
    const unsigned g = (unsigned)dg::cap_writers(dg::cdiv(M, 4) < 2048 ? dg::cdiv(M, 4) : 2048);
#define DG_SK_NN(NN) dg::launch((skinny_nn_kernel<NN>), dim3(g), dim3(256), sh, st, A, lda, B, ldb, C, ldc, M, K, beta, stats, dg::stat_slots())
    if (N == 1) DG_SK_NN(1); else if (N == 2) DG_SK_NN(2); else if (N == 3) DG_SK_NN(3); else DG_SK_NN(4);
#undef DG_SK_NN
    return dg::check_launch("dgcnn_gemm_f32(NN skinny)");
  }
  if (!gbias && !stats && transA && N <= 4 && M % 4 == 0 && M <= 4096 && p.avec && ws) {
    int64_t nb = dg::cdiv(K, 64);                       // >= 64 rows of the reduction per block
    if (nb > 512) nb = 512;
    if (ws_bytes >= (size_t)nb * M * N * sizeof(float)) {
      float* part = reinterpret_cast<float*>(ws);
#define DG_SK_TN(NN) dg::launch((skinny_tn_kernel<NN>), dim3((unsigned)nb), dim3(256), 0, st, A, lda, B, ldb, M, K, part)
      if (N == 1) DG_SK_TN(1); else if (N == 2) DG_SK_TN(2); else if (N == 3) DG_SK_TN(3); else DG_SK_TN(4);
#undef DG_SK_TN
      dg::launch(reduce_partials_kernel, dim3((unsigned)dg::cdiv((int64_t)M * N, 64)), dim3(64 * RL), 0, st, part,
                         (int)nb, M, N, C, ldc, beta);
      return dg::check_launch("dgcnn_gemm_f32(TN skinny)");
    }
  }
  if (!gbias && !stats && transB && K <= 4 && N % 4 == 0 && ldc % 4 == 0 && aligned16(C)) {
    const int64_t items = (int64_t)M * (N / 4);
    const unsigned g = (unsigned)(dg::cdiv(items, 256) < 8192 ? dg::cdiv(items, 256) : 8192);
#define DG_SK_NT(KK) dg::launch((skinny_nt_kernel<KK>), dim3(g), dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, beta)
    if (K == 1) DG_SK_NT(1); else if (K == 2) DG_SK_NT(2); else if (K == 3) DG_SK_NT(3); else DG_SK_NT(4);
#undef DG_SK_NT
    return dg::check_launch("dgcnn_gemm_f32(NT skinny)");
  }
  if (transA) {
    DG_REQUIRE(!gbias && !stats, DGCNN_EUNSUP, "dgcnn_gemm_f32: bias/stats with transA unsupported");
    int rc = plan_splits(p, ws, ws_bytes, "dgcnn_gemm_f32");
    if (rc) return rc;
    return launch<A_COL, B_ROW, E_STORE>(p, st, "dgcnn_gemm_f32(TN)");
  }
  p.bm = tile_m(M, N, 1);
  // a handful of output tiles with a long reduction (per-cloud rows: M = B): split K so the chip is not idle
  if (!gbias && !stats && !colmax_keys && ws && K >= 512 && dg::cdiv(M, 128) * dg::cdiv(N, 128) <= 32) {
    int rc = plan_splits(p, ws, ws_bytes, "dgcnn_gemm_f32");
    if (rc) return rc;
    p.bm = 128;
  }
  if (transB) return launch<A_ROW, B_COL, E_STORE>(p, st, "dgcnn_gemm_f32(NT)");
  return launch<A_ROW, B_ROW, E_STORE>(p, st, "dgcnn_gemm_f32(NN)");
}

// Row tiles of the dgcnn_gemm_f32 launch with these operands (transA = 0), i.e. how many workgroups add column sums into a `stats`
// buffer: the host sizes that buffer's slot count from it in the reproducible configuration (one writer per slot).  0: the launch
// takes a kernel whose grid follows the slot count by itself (class-dimension products).
extern "C" int dgcnn_gemm_stat_writers(int transB, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  GemmP p = {};
  p.M = M; p.N = N; p.K = K;
  p.avec = (lda % 4 == 0) && aligned16(A);
  p.bvec = (ldb % 4 == 0) && aligned16(B);
  if (!transB && N <= 4 && K % 4 == 0 && K <= 4096 && p.avec) return 0;      // skinny_nn_kernel: grid = cap_writers(...)
  return stat_writers(p, transB != 0);
}

extern "C" int dgcnn_edge_mlp_f32(const float* x, int64_t ldx, const int32_t* idx, const float* W0,
                                  int B, int N, int C, int k, int F, float* Y, double* stats,
                                  void* stream) {
  DG_REQUIRE(x && idx && W0 && Y, DGCNN_EINVAL, "dgcnn_edge_mlp_f32: null pointer");
  DG_REQUIRE(B > 0 && N > 0 && C > 0 && k > 0 && F > 0, DGCNN_EINVAL, "dgcnn_edge_mlp_f32: bad shape");
  const int64_t Me = (int64_t)B * N * k;
  DG_REQUIRE(Me < (1ll << 31), DGCNN_EUNSUP, "dgcnn_edge_mlp_f32: B*N*k >= 2^31");
  GemmP p = {};
  p.stat_slots = dg::stat_slots();
  p.x = x; p.ldx = ldx; p.idx = idx; p.npts = N; p.cch = C; p.knn = k;
  p.B = W0; p.ldb = F; p.C = Y; p.ldc = F;
  p.M = (int)Me; p.N = F; p.K = 2 * C; p.beta = 0.f; p.rpg = 1;
  p.stats = stats; p.splits = 1; p.kchunk = p.K;
  p.avec = (C % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
  p.bvec = (F % 4 == 0) && aligned16(W0);
  p.bm = tile_m(p.M, p.N, 1);
  return launch<A_EDGE, B_ROW, E_STORE>(p, (hipStream_t)stream, "dgcnn_edge_mlp_f32");
}

extern "C" int dgcnn_edge_mlp_wgrad_f32(const float* x, int64_t ldx, const int32_t* idx, const float* dY,
                                        int B, int N, int C, int k, int F, float* dW0, float beta,
                                        void* ws, size_t ws_bytes, void* stream) {
  DG_REQUIRE(x && idx && dY && dW0, DGCNN_EINVAL, "dgcnn_edge_mlp_wgrad_f32: null pointer");
  const int64_t Me = (int64_t)B * N * k;
  DG_REQUIRE(Me < (1ll << 31), DGCNN_EUNSUP, "dgcnn_edge_mlp_wgrad_f32: B*N*k >= 2^31");
  GemmP p = {};
  p.stat_slots = dg::stat_slots();
  p.x = x; p.ldx = ldx; p.idx = idx; p.npts = N; p.cch = C; p.knn = k;
  p.B = dY; p.ldb = F; p.C = dW0; p.ldc = F;
  p.M = 2 * C; p.N = F; p.K = (int)Me; p.beta = beta; p.rpg = 1;
  p.avec = (C % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
  p.bvec = (F % 4 == 0) && aligned16(dY);
  if (C <= 4 && F <= 256) {
    const int nblk = (int)(Me < 1024 * 64 ? dg::cdiv(Me, 64) : 1024);
    const int chunk = (int)dg::cdiv(Me, nblk);
    const size_t need = (size_t)nblk * 2 * C * F * sizeof(float);
    DG_REQUIRE(ws && ws_bytes >= need, DGCNN_ENOSPC, "dgcnn_edge_mlp_wgrad_f32: workspace too small (%zu < %zu)", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    dg::launch((edge_wgrad_smallc_kernel<4>), dim3((unsigned)nblk), dim3(256), 0, st, x, ldx, idx, dY, Me, N, k,
                       C, F, chunk, 0, reinterpret_cast<float*>(ws));
    int rc0 = dg::check_launch("dgcnn_edge_mlp_wgrad_f32(small C)");
    if (rc0) return rc0;
    const int64_t n = (int64_t)2 * C * F;
    dg::launch(reduce_partials_kernel, dim3((unsigned)dg::cdiv(n, 64)), dim3(64 * RL), 0, st,
                       reinterpret_cast<const float*>(ws), nblk, 2 * C, F, dW0, (int64_t)F, beta);
    return dg::check_launch("dgcnn_edge_mlp_wgrad_f32(small C reduce)");
  }
  int rc = plan_splits(p, ws, ws_bytes, "dgcnn_edge_mlp_wgrad_f32");
  if (rc) return rc;
  return launch<A_EDGE_T, B_ROW, E_STORE>(p, (hipStream_t)stream, "dgcnn_edge_mlp_wgrad_f32");
}

extern "C" int dgcnn_edge_mlp_dgrad_scatter_f32(const float* dY, const float* W0, const int32_t* idx,
                                                int B, int N, int C, int k, int F, float* dx,
                                                int64_t lddx, void* stream) {
  DG_REQUIRE(dY && W0 && idx && dx, DGCNN_EINVAL, "dgcnn_edge_mlp_dgrad_scatter_f32: null pointer");
  const int64_t Me = (int64_t)B * N * k;
  DG_REQUIRE(Me < (1ll << 31), DGCNN_EUNSUP, "dgcnn_edge_mlp_dgrad_scatter_f32: B*N*k >= 2^31");
  GemmP p = {};
  p.stat_slots = dg::stat_slots();
  // G[e][c] = sum_f dY[e][f] * W0[C + c][f]  (B stored [n][k] = rows C..2C of W0), scattered to dx[nbr(e)]
  p.A = dY; p.lda = F; p.B = W0 + (int64_t)C * F; p.ldb = F;
  p.M = (int)Me; p.N = C; p.K = F; p.rpg = 1;
  p.idx = idx; p.npts = N; p.cch = C; p.knn = k; p.dx = dx; p.lddx = lddx;
  p.splits = 1; p.kchunk = F;
  p.avec = (F % 4 == 0) && aligned16(dY);
  p.bvec = (F % 4 == 0) && aligned16(p.B);
  p.bm = tile_m(p.M, p.N, 1);
  return launch<A_ROW, B_COL, E_SCATTER>(p, (hipStream_t)stream, "dgcnn_edge_mlp_dgrad_scatter_f32");
}

// ---- conv0 in factored form.  E W0 = x_i (W0[:C] - W0[C:]) + x_j W0[C:]: the centre term U = X (Wa - Wb)
// is a per-POINT GEMM the host issues once; what remains per EDGE is the (B*N*k) x C GEMM of the
// gathered neighbour rows with W0[C:], plus U[point] added in the epilogue as a per-group bias
// (rows_per_group = k).  Half the per-edge flops of the literal form and a pure gather in the loader.
extern "C" int dgcnn_edge_nbr_gemm_f32(const float* x, int64_t ldx, const int32_t* idx, const float* Wb,
                                       const float* U, int64_t ldu, int B, int N, int C, int k, int F,
                                       float* Y, double* stats, void* stream) {
  DG_REQUIRE(x && idx && Wb && U && Y, DGCNN_EINVAL, "dgcnn_edge_nbr_gemm_f32: null pointer");
  DG_REQUIRE(B > 0 && N > 0 && C > 0 && k > 0 && F > 0, DGCNN_EINVAL, "dgcnn_edge_nbr_gemm_f32: bad shape");
  const int64_t Me = (int64_t)B * N * k;
  DG_REQUIRE(Me < (1ll << 31), DGCNN_EUNSUP, "dgcnn_edge_nbr_gemm_f32: B*N*k >= 2^31");
  GemmP p = {};
  p.stat_slots = dg::stat_slots();
  p.x = x; p.ldx = ldx; p.idx = idx; p.npts = N; p.cch = C; p.knn = k; p.edge_nbr = 1;
  p.B = Wb; p.ldb = F; p.C = Y; p.ldc = F;
  p.M = (int)Me; p.N = F; p.K = C; p.beta = 0.f;
  p.gbias = U; p.ldgbias = ldu; p.rpg = k; p.gbvec = (ldu % 4 == 0) && aligned16(U);
  p.stats = stats; p.splits = 1; p.kchunk = p.K;
  p.avec = (C % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
  p.bvec = (F % 4 == 0) && aligned16(Wb);
  p.bm = tile_m(p.M, p.N, 1);
  return launch<A_EDGE, B_ROW, E_STORE>(p, (hipStream_t)stream, "dgcnn_edge_nbr_gemm_f32");
}

/* dWb[C][F] (+)= sum over edges of x_j^T dY (the per-edge half of conv0's wgrad in factored form) */
extern "C" int dgcnn_edge_nbr_wgrad_f32(const float* x, int64_t ldx, const int32_t* idx, const float* dY,
                                        int B, int N, int C, int k, int F, float* dWb, float beta,
                                        void* ws, size_t ws_bytes, void* stream) {
  DG_REQUIRE(x && idx && dY && dWb, DGCNN_EINVAL, "dgcnn_edge_nbr_wgrad_f32: null pointer");
  const int64_t Me = (int64_t)B * N * k;
  DG_REQUIRE(Me < (1ll << 31), DGCNN_EUNSUP, "dgcnn_edge_nbr_wgrad_f32: B*N*k >= 2^31");
  if (C <= 4 && F <= 256) {
    const int nblk = (int)(Me < 1024 * 64 ? dg::cdiv(Me, 64) : 1024);
    const int chunk = (int)dg::cdiv(Me, nblk);
    const size_t need = (size_t)nblk * C * F * sizeof(float);
    DG_REQUIRE(ws && ws_bytes >= need, DGCNN_ENOSPC, "dgcnn_edge_nbr_wgrad_f32: workspace too small (%zu < %zu)", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    dg::launch((edge_wgrad_smallc_kernel<4>), dim3((unsigned)nblk), dim3(256), 0, st, x, ldx, idx, dY, Me, N, k,
                       C, F, chunk, 1, reinterpret_cast<float*>(ws));
    int rc0 = dg::check_launch("dgcnn_edge_nbr_wgrad_f32(small C)");
    if (rc0) return rc0;
    const int64_t n = (int64_t)C * F;
    dg::launch(reduce_partials_kernel, dim3((unsigned)dg::cdiv(n, 64)), dim3(64 * RL), 0, st,
                       reinterpret_cast<const float*>(ws), nblk, C, F, dWb, (int64_t)F, beta);
    return dg::check_launch("dgcnn_edge_nbr_wgrad_f32(small C reduce)");
  }
  GemmP p = {};
  p.stat_slots = dg::stat_slots();
  p.x = x; p.ldx = ldx; p.idx = idx; p.npts = N; p.cch = C; p.knn = k; p.edge_nbr = 1;
  p.B = dY; p.ldb = F; p.C = dWb; p.ldc = F;
  p.M = C; p.N = F; p.K = (int)Me; p.beta = beta; p.rpg = 1;
  p.avec = (C % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
  p.bvec = (F % 4 == 0) && aligned16(dY);
  int rc = plan_splits(p, ws, ws_bytes, "dgcnn_edge_nbr_wgrad_f32");
  if (rc) return rc;
  return launch<A_EDGE_T, B_ROW, E_STORE>(p, (hipStream_t)stream, "dgcnn_edge_nbr_wgrad_f32");
}
