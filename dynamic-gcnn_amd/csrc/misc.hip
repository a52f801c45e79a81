// misc.hip -- the HBM / L2-bound kernels either side of the GEMMs: edge gather (the explicit dgcnn/ops.py:21-40
// tensor, API/tests only), the per-edge gather-add of conv0 (y = V[neighbour] + U[point]: statistics pass of the
// model path, ops.py:47-53), transposed adjacency (counting sort) + incoming-edge sums (tf.gather^T), global
// max-pool + its gradient (model.py:76-81), tf.tile^T column sums, dropout (model.py:91),
// residual add+relu (ops.py:134), strided copies (tf.concat), softmax / CE / accuracy
// (trainval.py:39-52), gradient accumulation and Adam (trainval.py:17,75-80).
#include "common.h"
#include <stdlib.h>

namespace {

inline unsigned grid1d(int64_t n, int bs = 256) {
  int64_t g = dg::cdiv(n, bs);
  if (g > 65536) g = 65536;
  if (g < 1) g = 1;
  return (unsigned)g;
}

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

__global__ void edge_gather_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ idx,
                                   int N, int C, int k, int64_t total, float* __restrict__ E) {
  GRID_STRIDE(i, total) {
    const int c = (int)(i % (2 * C));
    const int64_t e = i / (2 * C);
    const int64_t g = e / k;
    const float xc = x[g * ldx + (c < C ? c : c - C)];
    float v = xc;
    if (c >= C) {
      const int64_t nb = (g / N) * N + idx[e];
      v = x[nb * ldx + (c - C)] - xc;
    }
    E[i] = v;
  }
}

__global__ void edge_gather_bwd_kernel(const float* __restrict__ dE, const int32_t* __restrict__ idx, int N, int C,
                                       int k, int64_t total, float* __restrict__ dx, int64_t lddx) {
  GRID_STRIDE(i, total) {   // i over (edge, c<C)
    const int c = (int)(i % C);
    const int64_t e = i / C;
    const int64_t g = e / k;
    const float dc = dE[e * 2 * C + c];
    const float dn = dE[e * 2 * C + C + c];
    const int64_t nb = (g / N) * N + idx[e];
    atomicAdd(dx + g * lddx + c, dc - dn);
    atomicAdd(dx + nb * lddx + c, dn);
  }
}

// ---- transposed adjacency (who points at me?) of the k-NN graph: the backward of tf.gather
// (ops.py:34) is a scatter-add over neighbours; instead of 63 M fp32 atomics per layer the edges
// are bucketed by target once (counting sort: histogram, per-cloud exclusive scan -- a cloud owns
// exactly N*k edges -- and fill), and every point then SUMS its incoming dY rows (a gather).
__global__ void csr_count_kernel(const int32_t* __restrict__ idx, int N, int k, int64_t Me, int32_t* __restrict__ cnt) {
  GRID_STRIDE(e, Me) {                          // Me < 2^31: 32-bit divisions
    const unsigned g = (unsigned)e / (unsigned)k;
    atomicAdd(cnt + (g / (unsigned)N) * (unsigned)N + idx[e], 1);
  }
}

__global__ __launch_bounds__(1024) void csr_scan_kernel(const int32_t* __restrict__ cnt, int N, int k,
                                                        int32_t* __restrict__ off) {
  __shared__ int part[1024];
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int per = (N + 1023) / 1024;
  const int lo = t * per, hi = (lo + per < N) ? (lo + per) : N;
  int s = 0;
  for (int i = lo; i < hi; ++i) s += cnt[(int64_t)b * N + i];
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {          // Hillis-Steele inclusive scan of the 1024 partial sums
    const int v = (t >= d) ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = b * N * k + part[t] - s;              // exclusive prefix of this thread's slice
  for (int i = lo; i < hi; ++i) {
    off[(int64_t)b * N + i] = run;
    run += cnt[(int64_t)b * N + i];
  }
  if (b == gridDim.x - 1 && t == 1023) off[(int64_t)gridDim.x * N] = gridDim.x * N * k;
}

__global__ void csr_fill_kernel(const int32_t* __restrict__ idx, int N, int k, int64_t Me,
                                const int32_t* __restrict__ off, int32_t* __restrict__ cur, int32_t* __restrict__ rev) {
  GRID_STRIDE(e, Me) {
    const unsigned g = (unsigned)e / (unsigned)k;
    const unsigned tgt = (g / (unsigned)N) * (unsigned)N + idx[e];
    const int pos = off[tgt] + atomicAdd(cur + tgt, 1);
    rev[pos] = (int32_t)e;
  }
}

// The same counting sort with the histogram and the cursors in LDS (LDS atomics instead of ~2 N k global ones):
// block (g, b) owns the targets [g*T, (g+1)*T) of cloud b, scans all N*k neighbour indices of the cloud twice
// (they are L2-resident: 4 N k bytes) and keeps only its own range.  Used when T fits in LDS.
__global__ __launch_bounds__(1024) void csr_cloud_kernel(const int32_t* __restrict__ idx, int N, int k, int T,
                                                         int32_t* __restrict__ off, int32_t* __restrict__ rev) {
  extern __shared__ int sh[];          // [T] histogram -> cursors, [1024] scan partials, [1] edges below the range
  int* hist = sh;
  int* part = sh + T;
  int* below = part + 1024;
  const int b = blockIdx.y, g = blockIdx.x, t = threadIdx.x;
  const int t0 = g * T, t1 = (t0 + T < N) ? (t0 + T) : N;
  const int Me = N * k;
  const int32_t* ib = idx + (int64_t)b * Me;
  for (int i = t; i < T; i += 1024) hist[i] = 0;
  if (t == 0) *below = 0;
  __syncthreads();
  int lo = 0;
  for (int e = t; e < Me; e += 1024) {
    const int j = ib[e];
    if (j < t0) ++lo;
    else if (j < t1) atomicAdd(&hist[j - t0], 1);
  }
  for (int d = 32; d > 0; d >>= 1) lo += __shfl_down(lo, d, 64);
  if ((t & 63) == 0 && lo) atomicAdd(below, lo);
  __syncthreads();
  // exclusive scan of hist[0..T): per-thread slices, Hillis-Steele over the 1024 slice sums
  const int per = (T + 1023) / 1024;
  const int s0 = t * per, s1 = (s0 + per < T) ? (s0 + per) : T;
  int sum = 0;
  for (int i = s0; i < s1; ++i) sum += hist[i];
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = (t >= d) ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = *below + part[t] - sum;                 // position inside the cloud of this slice's first bucket
  const int64_t cb = (int64_t)b * Me;
  for (int i = s0; i < s1; ++i) {
    const int c = hist[i];
    if (t0 + i < N) off[(int64_t)b * N + t0 + i] = (int32_t)(cb + run);
    hist[i] = run;                                  // becomes the bucket's cursor
    run += c;
  }
  if (b == gridDim.y - 1 && g == gridDim.x - 1 && t == 1023) off[(int64_t)gridDim.y * N] = (int32_t)((int64_t)gridDim.y * Me);
  __syncthreads();
  for (int e = t; e < Me; e += 1024) {
    const int j = ib[e];
    if (j >= t0 && j < t1) rev[cb + atomicAdd(&hist[j - t0], 1)] = (int32_t)(cb + e);
  }
}

// S[j][:] = sum of dY[e][:] over the edges e that point at j.  One float4 channel-quad per lane
// (a wave covers whole 256-B rows), 4 rows in flight.
// every dY row is read exactly once: nontemporal loads (87 -> 76 us per launch at configs[1], profiles/r03 run16)
__device__ __forceinline__ float4 nt_ld4(const float* p) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__global__ __launch_bounds__(256) void csr_gather_sum_kernel(const float* __restrict__ dY, const int32_t* __restrict__ off,
                                                             const int32_t* __restrict__ rev, int64_t R, int F,
                                                             float* __restrict__ S, int64_t lds) {
  const int FV = F / 4;
  GRID_STRIDE(it, R * FV) {
    const int64_t j = it / FV;
    const int f = (int)(it % FV) * 4;
    const int p0 = off[j], p1 = off[j + 1];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = p0;
    for (; p + 3 < p1; p += 4) {
#define DG_LDY(ptr) nt_ld4(ptr)
      const float4 v0 = DG_LDY(dY + (int64_t)rev[p] * F + f);
      const float4 v1 = DG_LDY(dY + (int64_t)rev[p + 1] * F + f);
      const float4 v2 = DG_LDY(dY + (int64_t)rev[p + 2] * F + f);
      const float4 v3 = DG_LDY(dY + (int64_t)rev[p + 3] * F + f);
      a.x += (v0.x + v1.x) + (v2.x + v3.x);
      a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z);
      a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; p < p1; ++p) {
      const float4 v = DG_LDY(dY + (int64_t)rev[p] * F + f);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(S + j * lds + f) = a;
  }
}

// the same sum over dY stored as bf16 (edge_mlp_bf16.hip's backward): same values, same order of additions, half the bytes
__global__ __launch_bounds__(256) void csr_gather_sum_bf16_kernel(const uint16_t* __restrict__ dY, const int32_t* __restrict__ off,
                                                                  const int32_t* __restrict__ rev, int64_t R, int F,
                                                                  float* __restrict__ S, int64_t lds) {
  const int FV = F / 4;
  auto ld = [&](int64_t row, int f) {
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const u2v v = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(dY + row * F + f));
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                       __uint_as_float(v.y & 0xffff0000u));
  };
  GRID_STRIDE(it, R * FV) {
    const int64_t j = it / FV;
    const int f = (int)(it % FV) * 4;
    const int p0 = off[j], p1 = off[j + 1];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = p0;
    for (; p + 3 < p1; p += 4) {
      const float4 v0 = ld(rev[p], f), v1 = ld(rev[p + 1], f), v2 = ld(rev[p + 2], f), v3 = ld(rev[p + 3], f);
      a.x += (v0.x + v1.x) + (v2.x + v3.x);
      a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z);
      a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; p < p1; ++p) {
      const float4 v = ld(rev[p], f);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(S + j * lds + f) = a;
  }
}

// ---- conv0 of an EdgeConv layer without an edge-level GEMM.  The 1x1 convolution is linear, so for the
// edge (i, j):  [x_i, x_j - x_i] W0 = x_i (Wa - Wb) + x_j Wb = U[i] + V[j]  with  [U | V] = X [Wa-Wb | Wb]
// ONE point-level GEMM (k times fewer MACs than the edge-level product of dgcnn/ops.py:47-52).  What is
// left per edge is a gather-add, bound by the HBM write of Y (the V rows of a cloud live in L2):
//   Y[e][:] = V[cloud(e)*N + idx[e]][:] + U[e / k][:]   (+ BatchNorm column sums of Y)
// Lanes: F/4 float4 lanes per edge row, RP = 256/(F/4) rows per pass, 4 passes in flight.  XCD x (blockIdx % 8)
// sweeps the x-th eighth of the edge rows with all its blocks side by side, so the clouds an XCD's L2
// holds at any moment are few.
__global__ __launch_bounds__(256) void edge_gather_add_kernel(const float* __restrict__ V, int64_t ldv,
                                                              const float* __restrict__ U, int64_t ldu,
                                                              const int32_t* __restrict__ idx, unsigned pts,
                                                              unsigned npts, unsigned knn, int F,
                                                              float* __restrict__ Y, double* __restrict__ stats, int nslots) {
  // Point-major (round 3): a group of F/4 lanes owns a point, keeps its U quad in registers and walks the point's k neighbour
  // rows four at a time.  (The edge-major version spent ~60 VALU operations per gathered float4 on two integer divisions and
  // the per-edge U reload: it was issue-bound at 18 % of the L1/L2 gather bandwidth.)
  __shared__ float red[2 * 1024];
  const int FV = F >> 2;
  const unsigned RP = 256u / (unsigned)FV;
  const int t = threadIdx.x;
  const bool active = (unsigned)t < RP * (unsigned)FV;
  const unsigned r = (unsigned)t / (unsigned)FV;
  const int f = (t % FV) * 4;
  const unsigned xcd = blockIdx.x & 7, l = blockIdx.x >> 3, nl = gridDim.x >> 3;
  const unsigned per = (pts + 7) / 8;
  const unsigned xbeg = xcd * per;
  const unsigned xend = (xbeg + per < pts) ? (xbeg + per) : pts;
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const unsigned step = nl * RP;
    for (unsigned i = xbeg + l * RP + r; i < xend; i += step) {
      const unsigned b = i / npts;
      const float* vb = V + (int64_t)b * npts * ldv + f;
      const int32_t* ip = idx + (int64_t)i * knn;
      const float4 u = *reinterpret_cast<const float4*>(U + (int64_t)i * ldu + f);
      float* yp = Y ? (Y + (int64_t)i * knn * F + f) : nullptr;
      for (unsigned m = 0; m < knn; m += 4) {
        int row[4];
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) row[q] = ip[(m + q < knn) ? (m + q) : (knn - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(vb + __umul24((unsigned)row[q], (unsigned)ldv));   // (full-rate 24-bit multiply: N, ldv < 2^24, N * ldv < 2^32: host check)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (m + q < knn) {
            const float4 y = make_float4(v[q].x + u.x, v[q].y + u.y, v[q].z + u.z, v[q].w + u.w);
            if (yp) *reinterpret_cast<float4*>(yp + (int64_t)(m + q) * F) = y;
            cs[0] += y.x; cs[1] += y.y; cs[2] += y.z; cs[3] += y.w;
            cq[0] += y.x * y.x; cq[1] += y.y * y.y; cq[2] += y.z * y.z; cq[3] += y.w * y.w;
          }
      }
    }
  }
  if (!stats) return;
  // fixed order: the RP point groups park their partial quads ([r][2][F], RP * 2 * F = 2048 floats), one thread per (sum, column)
  // adds them in ascending r (LDS float atomics added them in whatever order the waves arrived)
  if (active) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      red[(r * 2 + 0) * F + f + q] = cs[q];
      red[(r * 2 + 1) * F + f + q] = cq[q];
    }
  }
  __syncthreads();
  const int slot = blockIdx.x % nslots;
  for (int e = t; e < 2 * F; e += 256) {
    const int which = e / F, c = e % F;
    float a = 0.f;
    for (unsigned g = 0; g < RP; ++g) a += red[(g * 2 + which) * F + c];
    atomicAdd(stats + ((int64_t)slot * 2 + which) * F + c, (double)a);
  }
}

// Wcat = [Wa - Wb | Wb] (C x 2F) from W0 = [Wa ; Wb] (2C x F), and the matching gradient fold
//   dWa += dWcat[:, :F] ;  dWb += dWcat[:, F:] - dWcat[:, :F]
__global__ void edge_weight_split_kernel(const float* __restrict__ W0, int C, int F, float* __restrict__ Wcat) {
  GRID_STRIDE(i, (int64_t)C * F) {
    const int c = (int)(i / F), f = (int)(i % F);
    const float wa = W0[(int64_t)c * F + f], wb = W0[(int64_t)(C + c) * F + f];
    Wcat[(int64_t)c * 2 * F + f] = wa - wb;
    Wcat[(int64_t)c * 2 * F + F + f] = wb;
  }
}

__global__ void edge_wgrad_combine_kernel(const float* __restrict__ dWcat, int C, int F, float* __restrict__ dW0) {
  GRID_STRIDE(i, (int64_t)C * F) {
    const int c = (int)(i / F), f = (int)(i % F);
    const float du = dWcat[(int64_t)c * 2 * F + f], dv = dWcat[(int64_t)c * 2 * F + F + f];
    dW0[(int64_t)c * F + f] += du;
    dW0[(int64_t)(C + c) * F + f] += dv - du;
  }
}

// block = 64 channels x RG row groups (1024 threads: 16 waves/block, B*F/64 blocks fill the chip);
// first arg-max on ties (tf max_pool gradient routing)
constexpr int RG = 16;
__global__ __launch_bounds__(64 * RG) void global_max_kernel(const float* __restrict__ x, int64_t ldx, int N, int F,
                                                             float* __restrict__ out, int32_t* __restrict__ arg) {
  __shared__ float sv[RG][64];
  __shared__ int si[RG][64];
  const int b = blockIdx.y;
  const int f = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (f < F) {
    const float* p = x + (int64_t)b * N * ldx + f;
    int i = rg;
    for (; i + 7 * RG < N; i += 8 * RG) {            // 8 independent row loads in flight per lane
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(i + q * RG) * ldx];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (v[q] > best) { best = v[q]; bi = i + q * RG; }
    }
    for (; i < N; i += RG) {
      const float v = p[(int64_t)i * ldx];
      if (v > best) { best = v; bi = i; }
    }
  }
  sv[rg][threadIdx.x & 63] = best;
  si[rg][threadIdx.x & 63] = bi;
  __syncthreads();
  if (rg == 0 && f < F) {
    for (int q = 1; q < RG; ++q) {
      const float v = sv[q][threadIdx.x];
      const int i = si[q][threadIdx.x];
      if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    out[(int64_t)b * F + f] = best;
    // a column that is all NaN / -inf never updates bi: keep the backward scatter inside the cloud (row 0, like
    // np.argmax) instead of leaving the 0x7fffffff sentinel for global_max_bwd_kernel to index with
    if (arg) arg[(int64_t)b * F + f] = bi < N ? bi : 0;
  }
}

__global__ void global_max_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg, int N, int F,
                                      int64_t total, float* __restrict__ dx, int64_t lddx) {
  GRID_STRIDE(i, total) {
    const int64_t b = i / F;
    const int f = (int)(i % F);
    dx[(b * N + arg[i]) * lddx + f] += dout[i];
  }
}


__global__ __launch_bounds__(64 * RG) void group_colsum_kernel(const float* __restrict__ x, int64_t ldx, int rows,
                                                               int F, float* __restrict__ out) {
  __shared__ float sv[RG][64];
  const int g = blockIdx.y;
  const int f = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  float s = 0.f;
  if (f < F) {
    const float* p = x + (int64_t)g * rows * ldx + f;
    int i = rg;
    for (; i + 7 * RG < rows; i += 8 * RG) {         // 8 independent row loads in flight per lane; same summation order
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = p[(int64_t)(i + q * RG) * ldx];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; i < rows; i += RG) s += p[(int64_t)i * ldx];
  }
  sv[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && f < F) {
    float t = 0.f;
    for (int q = 0; q < RG; ++q) t += sv[q][threadIdx.x];   // fixed order: deterministic
    out[(int64_t)g * F + f] = t;
  }
}

__global__ void dropout_kernel(const float* x, float* y, int64_t n, float keep, uint64_t seed) {
  const float scale = 1.0f / keep;
  const uint32_t thr = (keep >= 1.f) ? 0xffffffffu : (uint32_t)((double)keep * 4294967296.0);
  GRID_STRIDE(i, n) {
    const uint32_t r = mix32(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
    y[i] = (r < thr) ? x[i] * scale : 0.f;
  }
}

// the same with the seed read from device memory: a captured HIP graph replays with a new mask every step
__global__ void dropout_dev_kernel(const float* x, float* y, int64_t n, float keep, const uint64_t* __restrict__ seed_dev) {
  const float scale = 1.0f / keep;
  const uint32_t thr = (keep >= 1.f) ? 0xffffffffu : (uint32_t)((double)keep * 4294967296.0);
  const uint64_t seed = *seed_dev;
  GRID_STRIDE(i, n) {
    const uint32_t r = mix32(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
    y[i] = (r < thr) ? x[i] * scale : 0.f;
  }
}

__global__ void add_relu_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                int64_t R, int F, float* __restrict__ out, int64_t ldo) {
  GRID_STRIDE(i, R * F) {
    const int64_t r = i / F;
    const int f = (int)(i % F);
    out[r * ldo + f] = fmaxf(a[r * lda + f] + b[r * ldb + f], 0.f);
  }
}

__global__ void relu_bwd_kernel(const float* __restrict__ dout, int64_t lddo, const float* __restrict__ out,
                                int64_t ldo, int64_t R, int F, float* __restrict__ d, int64_t ldd) {
  GRID_STRIDE(i, R * F) {
    const int64_t r = i / F;
    const int f = (int)(i % F);
    d[r * ldd + f] = (out[r * ldo + f] > 0.f) ? dout[r * lddo + f] : 0.f;
  }
}

__global__ void copy2d_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd,
                              int64_t R, int F, int accumulate) {
  GRID_STRIDE(i, R * F) {
    const int64_t r = i / F;
    const int f = (int)(i % F);
    const float v = src[r * lds + f];
    if (accumulate) dst[r * ldd + f] += v;
    else dst[r * ldd + f] = v;
  }
}

// dst (R, Cp) <- [src (R, C) | zeros]: the raw coordinates (C = 3) padded to float4 rows for the point-level GEMM, padding written
// here (a separate memset of dst was a 6-us serial step on the main stream)
__global__ void pad_copy_kernel(const float* __restrict__ src, int64_t lds, int C, float* __restrict__ dst, int Cp, int64_t R) {
  GRID_STRIDE(i, R * Cp) {
    const int64_t r = i / Cp;
    const int f = (int)(i % Cp);
    dst[i] = f < C ? src[r * lds + f] : 0.f;
  }
}

// tf.tile of the per-cloud global feature over the points of its cloud (model.py:80-81): dst[g * rows + i][f] = src[g][f]
__global__ void tile_rows_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd,
                                 int64_t R, int rows, int F) {
  GRID_STRIDE(i, R * F) {
    const int64_t r = i / F;
    const int f = (int)(i % F);
    dst[r * ldd + f] = src[(r / rows) * lds + f];
  }
}

__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ logits,
                                                           const int32_t* __restrict__ labels,
                                                           const float* __restrict__ weight, int64_t rows, int ncls,
                                                           float* __restrict__ softmax, float* __restrict__ dlogits,
                                                           float* __restrict__ scal) {
  __shared__ float sl[256], sc[256];
  float loss = 0.f, corr = 0.f;
  const float inv_rows = 1.0f / (float)rows;
  GRID_STRIDE(r, rows) {
    const float* z = logits + r * ncls;
    float mx = z[0];
    int am = 0;
    for (int c = 1; c < ncls; ++c)
      if (z[c] > mx) { mx = z[c]; am = c; }
    float se = 0.f;
    for (int c = 0; c < ncls; ++c) se += expf(z[c] - mx);
    const float inv = 1.0f / se;
    const int lab = labels ? labels[r] : -1;
    const float w = weight ? weight[r] : 1.f;
    for (int c = 0; c < ncls; ++c) {
      const float p = expf(z[c] - mx) * inv;
      if (softmax) softmax[r * ncls + c] = p;
      if (dlogits) dlogits[r * ncls + c] = (p - (c == lab ? 1.f : 0.f)) * w * inv_rows;
    }
    if (lab >= 0) {
      loss += (logf(se) - (z[lab] - mx)) * w;
      corr += (am == lab) ? 1.f : 0.f;
    }
  }
  sl[threadIdx.x] = loss;
  sc[threadIdx.x] = corr;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sl[threadIdx.x] += sl[threadIdx.x + s]; sc[threadIdx.x] += sc[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0 && scal) {
    atomicAdd(scal + 0, sl[0] * inv_rows);
    atomicAdd(scal + 1, sc[0] * inv_rows);
  }
}

__global__ void axpby_kernel(const float* __restrict__ x, float a, float* __restrict__ y, float b, int64_t n) {
  GRID_STRIDE(i, n) y[i] = (b == 0.f) ? a * x[i] : a * x[i] + b * y[i];
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr_t, float b1, float b2, float eps) {
  GRID_STRIDE(i, n) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] + (gi * gi - v[i]) * (1.f - b2);
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

}  // namespace

#define ST ((hipStream_t)stream)


extern "C" int dgcnn_edge_gather_f32(const float* x, int64_t ldx, const int32_t* idx, int B, int N, int C, int k,
                                     float* E, void* stream) {
  DG_REQUIRE(x && idx && E && B > 0 && N > 0 && C > 0 && k > 0, DGCNN_EINVAL, "dgcnn_edge_gather_f32: bad args");
  const int64_t total = (int64_t)B * N * k * 2 * C;
  dg::launch(edge_gather_kernel, dim3(grid1d(total)), dim3(256), 0, ST, x, ldx, idx, N, C, k, total, E);
  return dg::check_launch("dgcnn_edge_gather_f32");
}

extern "C" int dgcnn_edge_gather_bwd_f32(const float* dE, const int32_t* idx, int B, int N, int C, int k,
                                         float* dx, int64_t lddx, void* stream) {
  DG_REQUIRE(dE && idx && dx && B > 0 && N > 0 && C > 0 && k > 0, DGCNN_EINVAL, "dgcnn_edge_gather_bwd_f32: bad args");
  const int64_t total = (int64_t)B * N * k * C;
  dg::launch(edge_gather_bwd_kernel, dim3(grid1d(total)), dim3(256), 0, ST, dE, idx, N, C, k, total, dx, lddx);
  return dg::check_launch("dgcnn_edge_gather_bwd_f32");
}

extern "C" int dgcnn_edge_csr_build(const int32_t* idx, int B, int N, int k, int32_t* cnt_ws, int32_t* off,
                                    int32_t* rev, void* stream) {
  DG_REQUIRE(idx && cnt_ws && off && rev && B > 0 && N > 0 && k > 0, DGCNN_EINVAL, "dgcnn_edge_csr_build: bad args");
  const int64_t Me = (int64_t)B * N * k;
  DG_REQUIRE(Me < (1ll << 31), DGCNN_EUNSUP, "dgcnn_edge_csr_build: B*N*k >= 2^31");
  {
    // LDS variant: split every cloud's targets over G blocks so that ~256 blocks run and a range fits in LDS
    int G = 1;
    while (G < 64 && ((int64_t)B * G < 256 || dg::cdiv(N, G) > 12288) && dg::cdiv(N, G * 2) >= 64) G *= 2;
    const int T = (int)dg::cdiv(N, G);
    if (T <= 12288) {     // <= 52 KB of dynamic LDS
      const size_t sh = sizeof(int) * ((size_t)T + 1024 + 1);
      dg::launch(csr_cloud_kernel, dim3((unsigned)dg::cdiv(N, T), (unsigned)B), dim3(1024), sh, ST, idx, N, k, T, off, rev);
      return dg::check_launch("dgcnn_edge_csr_build");
    }
  }
  (void)dg::memset_async(cnt_ws, 0, sizeof(int32_t) * 2 * (size_t)B * N, ST);      // [counts | cursors]
  dg::launch(csr_count_kernel, dim3(grid1d(Me)), dim3(256), 0, ST, idx, N, k, Me, cnt_ws);
  dg::launch(csr_scan_kernel, dim3((unsigned)B), dim3(1024), 0, ST, cnt_ws, N, k, off);
  dg::launch(csr_fill_kernel, dim3(grid1d(Me)), dim3(256), 0, ST, idx, N, k, Me, off, cnt_ws + (size_t)B * N, rev);
  return dg::check_launch("dgcnn_edge_csr_build");
}

extern "C" int dgcnn_edge_gather_sum_f32(const float* dY, const int32_t* off, const int32_t* rev, int64_t R, int F,
                                         float* S, int64_t lds, void* stream) {
  DG_REQUIRE(dY && off && rev && S && R > 0 && F > 0 && F % 4 == 0, DGCNN_EINVAL, "dgcnn_edge_gather_sum_f32: bad args");
  DG_REQUIRE(lds >= F && lds % 4 == 0 && (reinterpret_cast<uintptr_t>(S) & 15) == 0, DGCNN_EINVAL,
             "dgcnn_edge_gather_sum_f32: S must be 16-byte aligned with lds %% 4 == 0");
  unsigned g = grid1d(R * (F / 4));
  dg::launch(csr_gather_sum_kernel, dim3(g), dim3(256), 0, ST, dY, off, rev, R, F, S, lds);
  return dg::check_launch("dgcnn_edge_gather_sum_f32");
}

extern "C" int dgcnn_edge_gather_sum_bf16(const void* dY, const int32_t* off, const int32_t* rev, int64_t R, int F, float* S,
                                          int64_t lds, void* stream) {
  DG_REQUIRE(dY && off && rev && S && R > 0 && F > 0, DGCNN_EINVAL, "dgcnn_edge_gather_sum_bf16: bad args");
  DG_REQUIRE(F % 4 == 0 && (reinterpret_cast<uintptr_t>(dY) & 7) == 0, DGCNN_EUNSUP, "dgcnn_edge_gather_sum_bf16: F %% 4 == 0, dY 8-byte aligned");
  DG_REQUIRE(lds >= F && lds % 4 == 0 && (reinterpret_cast<uintptr_t>(S) & 15) == 0, DGCNN_EINVAL,
             "dgcnn_edge_gather_sum_bf16: S must be 16-byte aligned with lds %% 4 == 0");
  dg::launch(csr_gather_sum_bf16_kernel, dim3(grid1d(R * (F / 4))), dim3(256), 0, ST, reinterpret_cast<const uint16_t*>(dY), off,
                     rev, R, F, S, lds);
  return dg::check_launch("dgcnn_edge_gather_sum_bf16");
}

extern "C" int dgcnn_edge_gather_add_f32(const float* V, int64_t ldv, const float* U, int64_t ldu, const int32_t* idx,
                                         int B, int N, int k, int F, float* Y, double* stats, void* stream) {
  DG_REQUIRE(V && U && idx && (Y || stats) && B > 0 && N > 0 && k > 0 && F > 0, DGCNN_EINVAL, "dgcnn_edge_gather_add_f32: bad args");
  DG_REQUIRE(F % 4 == 0 && F <= 1024, DGCNN_EUNSUP, "dgcnn_edge_gather_add_f32: F must be a multiple of 4, <= 1024 (got %d)", F);
  const int64_t rows = (int64_t)B * N * k;
  DG_REQUIRE(rows < (1ll << 31), DGCNN_EUNSUP, "dgcnn_edge_gather_add_f32: B*N*k >= 2^31");
  DG_REQUIRE(N < (1 << 24) && ldv < (1 << 24) && (int64_t)N * ldv < (1ll << 32), DGCNN_EUNSUP,
             "dgcnn_edge_gather_add_f32: N * ldv must be < 2^32 elements (32-bit row offsets inside a cloud)");
  auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  DG_REQUIRE(a16(V) && a16(U) && (!Y || a16(Y)) && ldv % 4 == 0 && ldu % 4 == 0 && ldv >= F && ldu >= F, DGCNN_EINVAL,
             "dgcnn_edge_gather_add_f32: V, U, Y must be 16-byte aligned with leading dimensions %% 4 == 0");
  const unsigned rp = 256u / (unsigned)(F / 4);                   // points per block pass
  const int64_t pts = (int64_t)B * N;
  const int64_t passes_x = dg::cdiv(dg::cdiv(pts, 8), (int64_t)rp);      // block passes one XCD's eighth of the points needs
  const int64_t trips = dg::cdiv(passes_x, 256);                         // <= 256 blocks per XCD (8 per CU), every block the same trips
  int64_t g = dg::cdiv(passes_x, trips);
  g = dg::cap_writers(g * 8) / 8;                                         // (reproducible configuration: one writer per slot)
  if (g < 1) g = 1;
  dg::launch(edge_gather_add_kernel, dim3((unsigned)g * 8), dim3(256), 0, ST, V, ldv, U, ldu, idx, (unsigned)pts,
                     (unsigned)N, (unsigned)k, F, Y, stats, dg::stat_slots());
  return dg::check_launch("dgcnn_edge_gather_add_f32");
}

// dst = src rounded to bf16 values (nearest even), kept as fp32: the weight operand of the bf16 edge-MLP's point-level gradient products
__global__ void round_bf16_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  GRID_STRIDE(i, n) {
    const unsigned u = __float_as_uint(src[i]);
    dst[i] = __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
  }
}

extern "C" int dgcnn_round_bf16_f32(const float* src, float* dst, int64_t n, void* stream) {
  DG_REQUIRE(src && dst && n > 0, DGCNN_EINVAL, "dgcnn_round_bf16_f32: bad args");
  dg::launch(round_bf16_kernel, dim3(grid1d(n)), dim3(256), 0, ST, src, dst, n);
  return dg::check_launch("dgcnn_round_bf16_f32");
}

extern "C" int dgcnn_edge_weight_split_f32(const float* W0, int C, int F, float* Wcat, void* stream) {
  DG_REQUIRE(W0 && Wcat && C > 0 && F > 0, DGCNN_EINVAL, "dgcnn_edge_weight_split_f32: bad args");
  dg::launch(edge_weight_split_kernel, dim3(grid1d((int64_t)C * F)), dim3(256), 0, ST, W0, C, F, Wcat);
  return dg::check_launch("dgcnn_edge_weight_split_f32");
}

extern "C" int dgcnn_edge_wgrad_combine_f32(const float* dWcat, int C, int F, float* dW0, void* stream) {
  DG_REQUIRE(dWcat && dW0 && C > 0 && F > 0, DGCNN_EINVAL, "dgcnn_edge_wgrad_combine_f32: bad args");
  dg::launch(edge_wgrad_combine_kernel, dim3(grid1d((int64_t)C * F)), dim3(256), 0, ST, dWcat, C, F, dW0);
  return dg::check_launch("dgcnn_edge_wgrad_combine_f32");
}

extern "C" int dgcnn_global_max_f32(const float* x, int64_t ldx, int B, int N, int F, float* out, int32_t* arg,
                                    void* stream) {
  DG_REQUIRE(x && out && B > 0 && N > 0 && F > 0, DGCNN_EINVAL, "dgcnn_global_max_f32: bad args");
  dg::launch(global_max_kernel, dim3((unsigned)dg::cdiv(F, 64), (unsigned)B), dim3(64 * RG), 0, ST, x, ldx, N, F,
                     out, arg);
  return dg::check_launch("dgcnn_global_max_f32");
}

extern "C" int dgcnn_global_max_bwd_f32(const float* dout, const int32_t* arg, int B, int N, int F, float* dx,
                                        int64_t lddx, void* stream) {
  DG_REQUIRE(dout && arg && dx && B > 0 && N > 0 && F > 0, DGCNN_EINVAL, "dgcnn_global_max_bwd_f32: bad args");
  const int64_t total = (int64_t)B * F;
  dg::launch(global_max_bwd_kernel, dim3(grid1d(total)), dim3(256), 0, ST, dout, arg, N, F, total, dx, lddx);
  return dg::check_launch("dgcnn_global_max_bwd_f32");
}

extern "C" int dgcnn_group_colsum_f32(const float* x, int64_t ldx, int G, int rows_per_group, int F, float* out,
                                      void* stream) {
  DG_REQUIRE(x && out && G > 0 && rows_per_group > 0 && F > 0, DGCNN_EINVAL, "dgcnn_group_colsum_f32: bad args");
  dg::launch(group_colsum_kernel, dim3((unsigned)dg::cdiv(F, 64), (unsigned)G), dim3(64 * RG), 0, ST, x, ldx,
                     rows_per_group, F, out);
  return dg::check_launch("dgcnn_group_colsum_f32");
}

extern "C" int dgcnn_pad_copy_f32(const float* src, int64_t lds, int C, float* dst, int Cp, int64_t R, void* stream) {
  DG_REQUIRE(src && dst && R > 0 && C > 0 && Cp >= C, DGCNN_EINVAL, "dgcnn_pad_copy_f32: bad args");
  dg::launch(pad_copy_kernel, dim3(grid1d(R * Cp)), dim3(256), 0, ST, src, lds, C, dst, Cp, R);
  return dg::check_launch("dgcnn_pad_copy_f32");
}

extern "C" int dgcnn_tile_rows_f32(const float* src, int64_t lds, int G, int rows_per_group, int F, float* dst, int64_t ldd,
                                   void* stream) {
  DG_REQUIRE(src && dst && G > 0 && rows_per_group > 0 && F > 0, DGCNN_EINVAL, "dgcnn_tile_rows_f32: bad args");
  const int64_t R = (int64_t)G * rows_per_group;
  dg::launch(tile_rows_kernel, dim3(grid1d(R * F)), dim3(256), 0, ST, src, lds, dst, ldd, R, rows_per_group, F);
  return dg::check_launch("dgcnn_tile_rows_f32");
}

extern "C" int dgcnn_dropout_f32(const float* x, float* y, int64_t n, float keep, uint64_t seed, void* stream) {
  DG_REQUIRE(x && y && n > 0 && keep > 0.f && keep <= 1.f, DGCNN_EINVAL, "dgcnn_dropout_f32: bad args");
  dg::launch(dropout_kernel, dim3(grid1d(n)), dim3(256), 0, ST, x, y, n, keep, seed);
  return dg::check_launch("dgcnn_dropout_f32");
}

extern "C" int dgcnn_dropout_dev_f32(const float* x, float* y, int64_t n, float keep, const uint64_t* seed_dev,
                                     void* stream) {
  DG_REQUIRE(x && y && seed_dev && n > 0 && keep > 0.f && keep <= 1.f, DGCNN_EINVAL, "dgcnn_dropout_dev_f32: bad args");
  dg::launch(dropout_dev_kernel, dim3(grid1d(n)), dim3(256), 0, ST, x, y, n, keep, seed_dev);
  return dg::check_launch("dgcnn_dropout_dev_f32");
}

extern "C" int dgcnn_add_relu_f32(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t R, int F,
                                  float* out, int64_t ldo, void* stream) {
  DG_REQUIRE(a && b && out && R > 0 && F > 0, DGCNN_EINVAL, "dgcnn_add_relu_f32: bad args");
  dg::launch(add_relu_kernel, dim3(grid1d(R * F)), dim3(256), 0, ST, a, lda, b, ldb, R, F, out, ldo);
  return dg::check_launch("dgcnn_add_relu_f32");
}

extern "C" int dgcnn_relu_bwd_f32(const float* dout, int64_t lddo, const float* out, int64_t ldo, int64_t R, int F,
                                  float* d, int64_t ldd, void* stream) {
  DG_REQUIRE(dout && out && d && R > 0 && F > 0, DGCNN_EINVAL, "dgcnn_relu_bwd_f32: bad args");
  dg::launch(relu_bwd_kernel, dim3(grid1d(R * F)), dim3(256), 0, ST, dout, lddo, out, ldo, R, F, d, ldd);
  return dg::check_launch("dgcnn_relu_bwd_f32");
}

extern "C" int dgcnn_copy2d_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t R, int F,
                                int accumulate, void* stream) {
  DG_REQUIRE(src && dst && R > 0 && F > 0, DGCNN_EINVAL, "dgcnn_copy2d_f32: bad args");
  dg::launch(copy2d_kernel, dim3(grid1d(R * F)), dim3(256), 0, ST, src, lds, dst, ldd, R, F, accumulate);
  return dg::check_launch("dgcnn_copy2d_f32");
}

extern "C" int dgcnn_softmax_xent_f32(const float* logits, const int32_t* labels, const float* weight,
                                      int64_t rows, int ncls, float* softmax, float* dlogits, float* scal,
                                      void* stream) {
  DG_REQUIRE(logits && rows > 0 && ncls > 0, DGCNN_EINVAL, "dgcnn_softmax_xent_f32: bad args");
  DG_REQUIRE(!dlogits || labels, DGCNN_EINVAL, "dgcnn_softmax_xent_f32: dlogits needs labels");
  unsigned g = grid1d(rows);
  if (g > 1024) g = 1024;
  dg::launch(softmax_xent_kernel, dim3(g), dim3(256), 0, ST, logits, labels, weight, rows, ncls, softmax,
                     dlogits, scal);
  return dg::check_launch("dgcnn_softmax_xent_f32");
}

extern "C" int dgcnn_axpby_f32(const float* x, float a, float* y, float b, int64_t n, void* stream) {
  DG_REQUIRE(x && y && n > 0, DGCNN_EINVAL, "dgcnn_axpby_f32: bad args");
  dg::launch(axpby_kernel, dim3(grid1d(n)), dim3(256), 0, ST, x, a, y, b, n);
  return dg::check_launch("dgcnn_axpby_f32");
}

extern "C" int dgcnn_adam_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float b1,
                              float b2, float eps, void* stream) {
  DG_REQUIRE(param && grad && m && v && n > 0, DGCNN_EINVAL, "dgcnn_adam_f32: bad args");
  dg::launch(adam_kernel, dim3(grid1d(n)), dim3(256), 0, ST, param, grad, m, v, n, lr_t, b1, b2, eps);
  return dg::check_launch("dgcnn_adam_f32");
}
