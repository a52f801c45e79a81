// bn.hip -- K4/K5: slim.batch_norm (batch statistics, beta only, eps 1e-3: dgcnn/ops.py:53,68,
// model.py:52,71,100) + activation + reduce_max / reduce_mean over k (ops.py:56-57), and their
// backward.  All HBM-bound: one float4 column-quad per lane so that a
// wave reads whole 256-B feature rows, k rows per point streamed with k independent loads.
#include "common.h"
#include <stdlib.h>

namespace {

// Slot reduction shared by the two finalize kernels: 4 columns per workgroup, one WAVE per column; lane z adds slots z, z + 64, ...
// (ascending), the 64 partial sums are then combined by a butterfly of lane exchanges: one fixed order for a given slot count,
// whatever wrote the slots.  No LDS and no barrier on purpose: in the backward these small kernels run beside the weight-gradient
// GEMMs of the side stream, whose workgroups hold 154 KB of LDS on every CU (A/B in the step: no measurable difference, 4.75 vs
// 4.74 ms -- kept because it asks for less).
constexpr int FIN_COLS = 4, FIN_LANES = 64;
__device__ __forceinline__ bool slot_sums(const double* __restrict__ acc, int F, int nslots, int& f, double& s, double& q) {
  const int c = threadIdx.x / FIN_LANES, z = threadIdx.x % FIN_LANES;
  f = blockIdx.x * FIN_COLS + c;
  if (f >= F) return false;                    // wave-uniform
  double a = 0.0, b = 0.0;
  // four slots in flight per lane (slots z, z + 64, z + 128, z + 192 of every group of 256), added in that order
  int sl = z;
  for (; sl + 3 * FIN_LANES < nslots; sl += 4 * FIN_LANES) {
    double va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      va[u] = acc[((int64_t)(sl + u * FIN_LANES) * 2 + 0) * F + f];
      vb[u] = acc[((int64_t)(sl + u * FIN_LANES) * 2 + 1) * F + f];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { a += va[u]; b += vb[u]; }
  }
  for (; sl < nslots; sl += FIN_LANES) {
    a += acc[((int64_t)sl * 2 + 0) * F + f];
    b += acc[((int64_t)sl * 2 + 1) * F + f];
  }
#pragma unroll
  for (int o = 1; o < FIN_LANES; o <<= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  s = a;
  q = b;
  return z == 0;
}

__global__ __launch_bounds__(FIN_COLS * FIN_LANES) void bn_finalize_kernel(const double* __restrict__ stats, int F, int nslots,
                                                                           double count, float eps, float* __restrict__ mean,
                                                                           float* __restrict__ rstd) {
  int f;
  double s, q;
  if (!slot_sums(stats, F, nslots, f, s, q)) return;
  const double mu = s / count;
  double var = q / count - mu * mu;   // biased variance, in double: no fp32 cancellation
  if (var < 0.0) var = 0.0;
  mean[f] = (float)mu;
  rstd[f] = (float)(1.0 / sqrt(var + (double)eps));
}

// reduce the slots of a backward reduction in place into slot 0, and emit dbeta
__global__ __launch_bounds__(FIN_COLS * FIN_LANES) void bn_bwd_finalize_kernel(double* __restrict__ red, int F, int nslots,
                                                                               float* __restrict__ dbeta, float dbeta_beta) {
  int f;
  double s, q;
  if (!slot_sums(red, F, nslots, f, s, q)) return;
  red[f] = s;
  red[F + f] = q;
  if (dbeta) dbeta[f] = (dbeta_beta != 0.f) ? (float)s + dbeta_beta * dbeta[f] : (float)s;
}

// One definition of xhat and z for every kernel (file is built with -ffp-contract=off), so the
// max recomputed in the backward compares equal to the values it was taken over.
__device__ __forceinline__ float bn_z(float y, float mu, float rs, float be, int relu, float& xh) {
  xh = (y - mu) * rs;
  float z = xh + be;
  if (relu) z = fmaxf(z, 0.f);
  return z;
}

// item -> (row, first channel).  FV (channel quads per row) is a power of two for every width of the
// model, so the 64-bit division that used to dominate the k = 1 kernels becomes a shift.
__device__ __forceinline__ void split_item(int64_t it, int FV, int fv_shift, int64_t& r, int& fq) {
  if (fv_shift >= 0) {
    r = it >> fv_shift;
    fq = (int)(it & (FV - 1));
  } else if (it < 0x7fffffffll) {
    const unsigned u = (unsigned)it, d = (unsigned)FV;
    r = u / d;
    fq = (int)(u % d);
  } else {
    r = it / FV;
    fq = (int)(it % FV);
  }
}

template <int V> struct Vec;
template <> struct Vec<4> {
  using T = float4;
  static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
  static __device__ __forceinline__ void st(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct Vec<1> {
  static __device__ __forceinline__ void ld(const float* p, float (&o)[1]) { o[0] = *p; }
  static __device__ __forceinline__ void st(float* p, const float (&o)[1]) { *p = o[0]; }
};

// ---- where the k rows of a point come from ------------------------------------------------------------
// G = false: the materialised tensor Y[(r*k + m)*F + f].
// G = true : conv0 of an EdgeConv layer, never materialised: y = V[cloud(r)*N + idx[r*k + m]][f] + U[r][f]
//            (dgcnn_edge_gather_add_f32's formula; the same single fp32 add everywhere, so values recomputed
//            in different kernels compare equal).  V / U are point-level (k times smaller than Y) and are
//            served by L2 / MALL: the forward writes no edge tensor at all, the backward only dY.
struct Src {
  const float* Y;
  const float* V; int64_t ldv;
  const float* U; int64_t ldu;
  const int32_t* idx;
  unsigned npts;
};

#ifndef BN_NB
#define BN_NB 4
#endif
constexpr int CNT_POS = 256;   // edge variants pack (#ties of the max) + CNT_POS * (#rows with z > 0) into cnt_out (k <= 255)
constexpr int NB = BN_NB;    // rows in flight per lane

template <int V, bool G>
struct Rows {
  const float* base;
  const int32_t* ip;
  int64_t ld;
  float u[V];
  __device__ __forceinline__ void init(const Src& s, int64_t r, int k, int F, int f) {
    if (G) {
      const unsigned cloud = (unsigned)r / s.npts;
      base = s.V + (int64_t)cloud * s.npts * s.ldv + f;
      ip = s.idx + r * k;
      ld = s.ldv;
      Vec<V>::ld(s.U + r * s.ldu + f, u);
    } else {
      base = s.Y + (r * k) * F + f;
      ip = nullptr;
      ld = F;
    }
  }
  // rows m .. m+NB-1.  Gathered rows: no control flow, indices past k-1 re-read row k-1 (the caller ignores
  // them).  Dense rows (mostly k = 1): rows past k-1 are skipped (k, m are wave-uniform: scalar branches).
  __device__ __forceinline__ void load(int m, int k, float (&y)[NB][V]) const {
    if constexpr (G) {
      // row offset in ONE full-rate multiply (v_mul_u32_u24): index < N <= 2^24 and N * ld < 2^32 elements (check_edge); the
      // 64-bit form cost a v_mad_u64_u32 and two v_mul_lo_u32 -- quarter rate each -- per gathered row, a quarter of the vector
      // time of the forward max/mean pass (profiles/r03/kreduce_probe.txt)
      unsigned row[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) row[b] = __umul24((unsigned)ip[(m + b < k) ? (m + b) : (k - 1)], (unsigned)ld);
#pragma unroll
      for (int b = 0; b < NB; ++b) Vec<V>::ld(base + row[b], y[b]);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int v = 0; v < V; ++v) y[b][v] += u[v];
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (m + b < k) Vec<V>::ld(base + (int64_t)(m + b) * ld, y[b]);
    }
  }
};

// item = (row, channel quad).  G: XCD x (blockIdx % 8) owns the x-th eighth of the rows and its blocks sweep it
// side by side, so the clouds whose V rows an XCD's L2 holds at any moment are few.
struct Items { int64_t base, count, first, step; };
template <bool G>
__device__ __forceinline__ Items items_of(int64_t R, int FV) {
  Items it;
  if (G) {
    const int64_t per = (R + 7) / 8;
    const int64_t rb = (int64_t)(blockIdx.x & 7) * per;
    int64_t rows = R - rb;
    rows = rows < 0 ? 0 : (rows > per ? per : rows);
    it.base = rb * FV;
    it.count = rows * FV;
    it.first = (int64_t)(blockIdx.x >> 3) * blockDim.x + threadIdx.x;
    it.step = (int64_t)(gridDim.x >> 3) * blockDim.x;
  } else {
    it.base = 0;
    it.count = R * FV;
    it.first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    it.step = (int64_t)gridDim.x * blockDim.x;
  }
  return it;
}

// ---- forward: z = (y-mu)*rstd + beta; relu; max / mean over the k rows of each point ----
template <int V, bool G>
__global__ __launch_bounds__(256) void bn_act_kreduce_kernel(
    Src src, int64_t R, int k, int F, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ beta, int relu,
    float* __restrict__ max_out, int64_t ldmax, float* __restrict__ mean_out, int64_t ldmean,
    float* __restrict__ out2, int64_t ldout2, float* __restrict__ cnt_out, int fv_shift) {
  const int FV = F / V;
  const Items its = items_of<G>(R, FV);
  const float invk = 1.0f / (float)k;
  for (int64_t q = its.first; q < its.count; q += its.step) {
    int64_t r;
    int fq;
    split_item(its.base + q, FV, fv_shift, r, fq);
    const int f = fq * V;
    float mu[V], rs[V], be[V], mx[V], sm[V], cn[V], np[V];
    Vec<V>::ld(mean + f, mu); Vec<V>::ld(rstd + f, rs); Vec<V>::ld(beta + f, be);
#pragma unroll
    for (int v = 0; v < V; ++v) { mx[v] = -INFINITY; sm[v] = 0.f; cn[v] = 0.f; np[v] = 0.f; }
    Rows<V, G> rows;
    rows.init(src, r, k, F, f);
    for (int m = 0; m < k; m += NB) {
      float yv[NB][V];
      rows.load(m, k, yv);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (m + b < k) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            float xh;
            const float z = bn_z(yv[b][v], mu[v], rs[v], be[v], relu, xh);
            const bool gt = z > mx[v];
            cn[v] = gt ? 1.f : ((z == mx[v]) ? cn[v] + 1.f : cn[v]);   // ties share the max gradient (A.5)
            mx[v] = gt ? z : mx[v];
            sm[v] += z;
            if (G) np[v] += (z > 0.f) ? (float)CNT_POS : 0.f;
          }
        }
    }
    Vec<V>::st(max_out + r * ldmax + f, mx);
    if (out2) Vec<V>::st(out2 + r * ldout2 + f, mx);
    if (G) {                       // edge variant: ties + CNT_POS x (rows with z > 0), both exact small integers
#pragma unroll
      for (int v = 0; v < V; ++v) cn[v] += np[v];
    }
    if (cnt_out) Vec<V>::st(cnt_out + r * F + f, cn);
    if (mean_out) {
#pragma unroll
      for (int v = 0; v < V; ++v) sm[v] *= invk;
      Vec<V>::st(mean_out + r * ldmean + f, sm);
    }
  }
}

// dZ for the k rows of one (point, channel-quad); shared by reduce and apply
template <int V>
struct KState { float mx[V]; float cnt[V]; };

template <int V, bool G>
__device__ __forceinline__ void k_pass_max(const Rows<V, G>& rows, int k, const float (&mu)[V],
                                           const float (&rs)[V], const float (&be)[V], int relu,
                                           KState<V>& st) {
#pragma unroll
  for (int v = 0; v < V; ++v) { st.mx[v] = -INFINITY; st.cnt[v] = 0.f; }
  for (int m = 0; m < k; m += NB) {
    float yv[NB][V];
    rows.load(m, k, yv);
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (m + b < k) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float xh;
          const float z = bn_z(yv[b][v], mu[v], rs[v], be[v], relu, xh);
          if (z > st.mx[v]) { st.mx[v] = z; st.cnt[v] = 1.f; }
          else if (z == st.mx[v]) st.cnt[v] += 1.f;
        }
      }
  }
}

template <int V, bool G>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    Src src, int64_t R, int k, int F, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ beta, int relu,
    const float* __restrict__ dmax, int64_t lddmax, const float* __restrict__ dmean, int64_t lddmean,
    const float* __restrict__ mx_in, int64_t ldmx, const float* __restrict__ cnt_in, double* __restrict__ red,
    int fv_shift, int nslots) {
  extern __shared__ float lred[];  // [2][F]
  for (int e = threadIdx.x; e < 2 * F; e += blockDim.x) lred[e] = 0.f;
  __syncthreads();
  const int FV = F / V;
  const Items its = items_of<G>(R, FV);
  const float invk = 1.0f / (float)k;
  int curf = -1;
  float s0[V], s1[V];
#pragma unroll
  for (int v = 0; v < V; ++v) { s0[v] = 0.f; s1[v] = 0.f; }
  for (int64_t q = its.first; q < its.count; q += its.step) {
    int64_t r;
    int fq;
    split_item(its.base + q, FV, fv_shift, r, fq);
    const int f = fq * V;
    if (f != curf) {
      if (curf >= 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) { atomicAdd(&lred[curf + v], s0[v]); atomicAdd(&lred[F + curf + v], s1[v]); s0[v] = 0.f; s1[v] = 0.f; }
      }
      curf = f;
    }
    float mu[V], rs[V], be[V], dmx[V], dmn[V];
    Vec<V>::ld(mean + f, mu); Vec<V>::ld(rstd + f, rs); Vec<V>::ld(beta + f, be);
    Vec<V>::ld(dmax + r * lddmax + f, dmx);
    Rows<V, G> rows;
    rows.init(src, r, k, F, f);
    KState<V> st;
    if (dmean) {
      Vec<V>::ld(dmean + r * lddmean + f, dmn);
      if (mx_in) {                 // forward kept (max, #ties): the rows are read once
        Vec<V>::ld(mx_in + r * ldmx + f, st.mx);
        Vec<V>::ld(cnt_in + r * F + f, st.cnt);
        if (G) {
#pragma unroll
          for (int v = 0; v < V; ++v) st.cnt[v] -= (float)CNT_POS * floorf(st.cnt[v] * (1.0f / CNT_POS));
        }
      } else {
        k_pass_max<V, G>(rows, k, mu, rs, be, relu, st);
      }
    }
    for (int m = 0; m < k; m += NB) {
      float yv[NB][V];
      rows.load(m, k, yv);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (m + b < k) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            float xh;
            const float z = bn_z(yv[b][v], mu[v], rs[v], be[v], relu, xh);
            float dz;
            if (dmean) dz = ((z == st.mx[v]) ? dmx[v] / st.cnt[v] : 0.f) + dmn[v] * invk;
            else dz = dmx[v];
            if (relu && !(z > 0.f)) dz = 0.f;
            s0[v] += dz;
            s1[v] += dz * xh;
          }
        }
    }
  }
  if (curf >= 0) {
#pragma unroll
    for (int v = 0; v < V; ++v) { atomicAdd(&lred[curf + v], s0[v]); atomicAdd(&lred[F + curf + v], s1[v]); }
  }
  __syncthreads();
  const int slot = blockIdx.x % nslots;
  for (int e = threadIdx.x; e < 2 * F; e += blockDim.x)
    atomicAdd(red + (int64_t)slot * 2 * F + e, (double)lred[e]);
}

template <int V, bool G>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    Src src, int64_t R, int k, int F, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ beta, int relu,
    const float* __restrict__ dmax, int64_t lddmax, const float* __restrict__ dmean, int64_t lddmean,
    const float* __restrict__ mx_in, int64_t ldmx, const float* __restrict__ cnt_in,
    const double* __restrict__ red, float* dY, float* __restrict__ dYsum, int64_t lddysum, int fv_shift) {
  // relu: bit 0 = the layer has a ReLU; bit 1 = dY is written as bf16 VALUES (round to nearest even, stored as fp32) and dYsum adds
  // those -- the operand rounding of the bf16 edge-MLP's gradient products (oracle.conv_bn_act_bwd(operand_round)), done once here
  const bool round_dy = (relu & 2) != 0;
  relu &= 1;
  const int FV = F / V;
  const Items its = items_of<G>(R, FV);
  const float invk = 1.0f / (float)k;
  const double inv_cnt = 1.0 / ((double)R * (double)k);
  for (int64_t q = its.first; q < its.count; q += its.step) {
    int64_t r;
    int fq;
    split_item(its.base + q, FV, fv_shift, r, fq);
    const int f = fq * V;
    float mu[V], rs[V], be[V], dmx[V], dmn[V], c1[V], c2[V], acc[V];
    Vec<V>::ld(mean + f, mu); Vec<V>::ld(rstd + f, rs); Vec<V>::ld(beta + f, be);
    Vec<V>::ld(dmax + r * lddmax + f, dmx);
#pragma unroll
    for (int v = 0; v < V; ++v) {
      c1[v] = (float)(red[f + v] * inv_cnt);
      c2[v] = (float)(red[F + f + v] * inv_cnt);
      acc[v] = 0.f;
    }
    Rows<V, G> rows;
    rows.init(src, r, k, F, f);
    float* dy = dY + (r * k) * F + f;
    KState<V> st;
    if (dmean) {
      Vec<V>::ld(dmean + r * lddmean + f, dmn);
      if (mx_in) {                 // forward kept (max, #ties): the rows are read once
        Vec<V>::ld(mx_in + r * ldmx + f, st.mx);
        Vec<V>::ld(cnt_in + r * F + f, st.cnt);
        if (G) {
#pragma unroll
          for (int v = 0; v < V; ++v) st.cnt[v] -= (float)CNT_POS * floorf(st.cnt[v] * (1.0f / CNT_POS));
        }
      } else {
        k_pass_max<V, G>(rows, k, mu, rs, be, relu, st);
      }
    }
    for (int m = 0; m < k; m += NB) {
      float yv[NB][V];
      rows.load(m, k, yv);                 // all NB rows are in registers before the first store (dY may be Y)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (m + b < k) {
          float o[V];
#pragma unroll
          for (int v = 0; v < V; ++v) {
            float xh;
            const float z = bn_z(yv[b][v], mu[v], rs[v], be[v], relu, xh);
            float dz;
            if (dmean) dz = ((z == st.mx[v]) ? dmx[v] / st.cnt[v] : 0.f) + dmn[v] * invk;
            else dz = dmx[v];
            if (relu && !(z > 0.f)) dz = 0.f;
            o[v] = rs[v] * (dz - c1[v] - xh * c2[v]);
            if (round_dy) {
              const unsigned u = __float_as_uint(o[v]);
              o[v] = __uint_as_float((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u);
            }
            acc[v] += o[v];
          }
          Vec<V>::st(dy + (int64_t)(m + b) * F, o);
        }
    }
    if (dYsum) Vec<V>::st(dYsum + r * lddysum + f, acc);
  }
}


// ---- conv0 backward of an EdgeConv layer whose input needs NO gradient and has C <= 4 channels (the first layer: raw
// coordinates): the whole edge-level backward collapses into this one pass.  Per edge it forms dY exactly as
// bn_bwd_apply_kernel<4, true> does -- and consumes it on the spot: dW0 = E^T dY with E = [x_i, x_j - x_i] (ops.py:39-52),
//   dW0[c]     += x_i[c] * sum_m dY[i, m]          dW0[C + c] += sum_m (x_j[c] - x_i[c]) * dY[i, m]
// so no dY (252 MB at configs[1]) is written, no transposed adjacency is built or summed, and the two point-level GEMMs of
// the folded form are not needed.  A thread owns one channel quad for its whole life (items_of<true>: the item stride is a
// multiple of F / 4), accumulates 2 C x 4 partial sums in registers, the block reduces them through LDS into
// partial[block][2 C][F]; reduce_partials_kernel (gemm.hip) adds the blocks up in fixed order.
template <int CC>
__global__ __launch_bounds__(256) void edge_bwd_apply_wgrad_kernel(
    Src src, int64_t R, int k, int F, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ beta, const float* __restrict__ dmax, int64_t lddmax, const float* __restrict__ dmean,
    int64_t lddmean, const float* __restrict__ mx_in, int64_t ldmx, const float* __restrict__ cnt_in,
    const double* __restrict__ red, const float* __restrict__ x, int64_t ldx, int C, float* __restrict__ partial, int fv_shift) {
  constexpr int V = 4;
  extern __shared__ float sh[];                   // [256 / FV row lanes][2 CC][F]
  const int FV = F / V;
  const Items its = items_of<true>(R, FV);
  const float invk = 1.0f / (float)k;
  const double inv_cnt = 1.0 / ((double)R * (double)k);
  float wc[CC][V], wd[CC][V];
#pragma unroll
  for (int c = 0; c < CC; ++c)
#pragma unroll
    for (int v = 0; v < V; ++v) { wc[c][v] = 0.f; wd[c][v] = 0.f; }
  const int fq_fixed = (int)((its.base + its.first) & (FV - 1));          // FV is a power of two here (checked by the host)
  const int f = fq_fixed * V;
  float mu[V], rs[V], be[V], c1[V], c2[V];
  Vec<V>::ld(mean + f, mu); Vec<V>::ld(rstd + f, rs); Vec<V>::ld(beta + f, be);
#pragma unroll
  for (int v = 0; v < V; ++v) {
    c1[v] = (float)(red[f + v] * inv_cnt);
    c2[v] = (float)(red[F + f + v] * inv_cnt);
  }
  for (int64_t q = its.first; q < its.count; q += its.step) {
    const int64_t r = (its.base + q) >> fv_shift;
    float dmx[V], dmn[V];
    Vec<V>::ld(dmax + r * lddmax + f, dmx);
    Vec<V>::ld(dmean + r * lddmean + f, dmn);
    Rows<V, true> rows;
    rows.init(src, r, k, F, f);
    KState<V> st;
    Vec<V>::ld(mx_in + r * ldmx + f, st.mx);
    Vec<V>::ld(cnt_in + r * F + f, st.cnt);
#pragma unroll
    for (int v = 0; v < V; ++v) st.cnt[v] -= (float)CNT_POS * floorf(st.cnt[v] * (1.0f / CNT_POS));
    float xi[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) xi[c] = (c < C) ? x[r * ldx + c] : 0.f;
    const int64_t cloud0 = (int64_t)((unsigned)r / src.npts) * src.npts;
    float acc[V] = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < k; m += NB) {
      float yv[NB][V];
      rows.load(m, k, yv);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (m + b < k) {
          const float* xj = x + (cloud0 + rows.ip[m + b]) * ldx;
          float dx[CC];
#pragma unroll
          for (int c = 0; c < CC; ++c) dx[c] = (c < C) ? xj[c] - xi[c] : 0.f;
#pragma unroll
          for (int v = 0; v < V; ++v) {
            float xh;
            const float z = bn_z(yv[b][v], mu[v], rs[v], be[v], 1, xh);
            float dz = ((z == st.mx[v]) ? dmx[v] / st.cnt[v] : 0.f) + dmn[v] * invk;
            if (!(z > 0.f)) dz = 0.f;
            const float o = rs[v] * (dz - c1[v] - xh * c2[v]);
            acc[v] += o;
#pragma unroll
            for (int c = 0; c < CC; ++c) wd[c][v] = fmaf(dx[c], o, wd[c][v]);
          }
        }
    }
#pragma unroll
    for (int c = 0; c < CC; ++c)
#pragma unroll
      for (int v = 0; v < V; ++v) wc[c][v] = fmaf(xi[c], acc[v], wc[c][v]);
  }
  // block reduction: threads t and t + FV, t + 2 FV ... hold the same channel quad
  const int lanes = 256 / FV;
  const int rl = threadIdx.x / FV;
#pragma unroll
  for (int c = 0; c < CC; ++c)
#pragma unroll
    for (int v = 0; v < V; ++v) {
      sh[(rl * 2 * CC + c) * F + f + v] = wc[c][v];
      sh[(rl * 2 * CC + CC + c) * F + f + v] = wd[c][v];
    }
  __syncthreads();
  float* out = partial + (int64_t)blockIdx.x * 2 * C * F;
  for (int e = threadIdx.x; e < 2 * CC * F; e += 256) {
    const int cr = e / F, ff = e % F;
    const int c = cr < CC ? cr : cr - CC;
    if (c >= C) continue;
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += sh[(l * 2 * CC + cr) * F + ff];
    out[(int64_t)(cr < CC ? c : C + c) * F + ff] = s;
  }
}

// ---- k = 1 (per-point layers: conv1, shortcut, merged, FC, Final): column-fixed streaming kernels -------------
// A thread keeps ONE channel quad for its whole life (mean / rstd / beta / the backward constants live in
// registers) and walks rows with 4 independent row loads in flight; block = (256 / FVB rows) x FVB quads,
// blockIdx.y = chunk of 256 quads when F > 1024.  The generic kernels above reload the three parameter quads
// per item, which for k = 1 is most of their load instructions.
struct K1Map {
  int f; int64_t r0, rstep; bool on;
};
__device__ __forceinline__ K1Map k1_map(int F, int FVB, int RP) {   // RP = blockDim.x / FVB
  K1Map m;
  const int t = threadIdx.x;
  const int fq = blockIdx.y * 256 + (t % FVB);
  m.f = fq * 4;
  m.on = (t < RP * FVB) && (m.f < F);
  m.r0 = (int64_t)blockIdx.x * RP + t / FVB;
  m.rstep = (int64_t)gridDim.x * RP;
  return m;
}

// Column sums of a K1-mapped workgroup in a FIXED order: every (row group rp, column quad) thread parks its two partial quads in
// lpark[rp][2][4 FVB] (2048 floats), then one thread per (sum, column) adds the RP row groups in ascending order and hands the
// result to slot (blockIdx.x % nslots) -- no LDS float atomics, whose order changed from run to run.
constexpr int K1_PARK_FLOATS = 2048;
__device__ __forceinline__ void k1_block_sums(float* lpark, const K1Map& m, int F, int FVB, int RP, const float (&s0)[4],
                                              const float (&s1)[4], double* __restrict__ red, int nslots) {
  const int t = threadIdx.x;
  const int W = 4 * FVB;
  if (t < RP * FVB) {
    const int rp = t / FVB, cb = (t % FVB) * 4;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      lpark[(rp * 2 + 0) * W + cb + v] = m.on ? s0[v] : 0.f;
      lpark[(rp * 2 + 1) * W + cb + v] = m.on ? s1[v] : 0.f;
    }
  }
  __syncthreads();
  const int fbase = blockIdx.y * 1024;
  const int slot = blockIdx.x % nslots;
  for (int e = t; e < 2 * W; e += blockDim.x) {
    const int which = e / W, cb = e % W;
    if (fbase + cb < F) {
      float a = 0.f;
      for (int rp = 0; rp < RP; ++rp) a += lpark[(rp * 2 + which) * W + cb];
      atomicAdd(red + ((int64_t)slot * 2 + which) * F + fbase + cb, (double)a);
    }
  }
}

__global__ __launch_bounds__(256) void bn1_act_kernel(const float* __restrict__ T, int64_t R, int F, int FVB, int RP,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ beta, int relu,
                                                      float* __restrict__ out, int64_t ldo, float* __restrict__ out2,
                                                      int64_t ldo2, float* __restrict__ cnt_out,
                                                      const uint64_t* __restrict__ drop_seed, float drop_keep) {
  const K1Map m = k1_map(F, FVB, RP);
  if (!m.on) return;
  // optional dropout fused behind the activation (model.py:90-91): element (row, f) of the (R, F) output keeps its value x 1/keep
  // iff the counter-based mask of dropout_dev_kernel keeps flattened index row * F + f
  const uint64_t dseed = drop_seed ? *drop_seed : 0ull;
  const uint32_t dthr = dropout_threshold(drop_keep);
  const float dscale = drop_seed ? 1.0f / drop_keep : 1.0f;
  float mu[4], rs[4], be[4];
  Vec<4>::ld(mean + m.f, mu); Vec<4>::ld(rstd + m.f, rs); Vec<4>::ld(beta + m.f, be);
  const float one[4] = {1.f, 1.f, 1.f, 1.f};
  for (int64_t r = m.r0; r < R; r += 4 * m.rstep) {
    float y[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int64_t rr = r + b * m.rstep;
      if (rr < R) Vec<4>::ld(T + rr * F + m.f, y[b]);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int64_t rr = r + b * m.rstep;
      if (rr < R) {
        float z[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) { float xh; z[v] = bn_z(y[b][v], mu[v], rs[v], be[v], relu, xh); }
        if (drop_seed) {
#pragma unroll
          for (int v = 0; v < 4; ++v) z[v] = dropout_keeps(dseed, (uint64_t)(rr * F + m.f + v), dthr) ? z[v] * dscale : 0.f;
        }
        Vec<4>::st(out + rr * ldo + m.f, z);
        if (out2) Vec<4>::st(out2 + rr * ldo2 + m.f, z);
        if (cnt_out) Vec<4>::st(cnt_out + rr * F + m.f, one);
      }
    }
  }
}

// APPLY = false: column sums of dz and dz*xhat into red;  APPLY = true: dT = rstd (dz - c1 - xhat c2) (in place ok)
template <bool APPLY>
__global__ __launch_bounds__(256) void bn1_bwd_kernel(const float* T, int64_t R, int F, int FVB, int RP,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ beta, int relu,
                                                      const float* __restrict__ dout, int64_t lddo,
                                                      const float* __restrict__ dout2, int64_t lddo2,
                                                      double* __restrict__ red, float* dT, float* __restrict__ dsum,
                                                      int64_t lddsum, const uint64_t* __restrict__ drop_seed, float drop_keep,
                                                      int nslots) {
  extern __shared__ float lred[];   // reduce only: K1_PARK_FLOATS
  const uint64_t dseed = drop_seed ? *drop_seed : 0ull;      // dout is the gradient of the DROPPED output (fused dropout backward)
  const uint32_t dthr = dropout_threshold(drop_keep);
  const float dscale = drop_seed ? 1.0f / drop_keep : 1.0f;
  const K1Map m = k1_map(F, FVB, RP);
  float mu[4] = {0, 0, 0, 0}, rs[4] = {0, 0, 0, 0}, be[4] = {0, 0, 0, 0}, c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0};
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (m.on) {
    Vec<4>::ld(mean + m.f, mu); Vec<4>::ld(rstd + m.f, rs); Vec<4>::ld(beta + m.f, be);
    if (APPLY) {
      const double inv_cnt = 1.0 / (double)R;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        c1[v] = (float)(red[m.f + v] * inv_cnt);
        c2[v] = (float)(red[F + m.f + v] * inv_cnt);
      }
    }
    for (int64_t r = m.r0; r < R; r += 4 * m.rstep) {
      float y[4][4], d[4][4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int64_t rr = r + b * m.rstep;
        if (rr < R) {
          Vec<4>::ld(T + rr * F + m.f, y[b]); Vec<4>::ld(dout + rr * lddo + m.f, d[b]);
          if (dout2) {               // the output had two consumers (conv1: the concat slice and MergedEdgeConv's input): dz = d + d2
            float e[4];
            Vec<4>::ld(dout2 + rr * lddo2 + m.f, e);
#pragma unroll
            for (int v = 0; v < 4; ++v) d[b][v] += e[v];
          }
        }
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int64_t rr = r + b * m.rstep;
        if (rr < R) {
          float o[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            float xh;
            const float z = bn_z(y[b][v], mu[v], rs[v], be[v], relu, xh);
            float dz = d[b][v];
            if (drop_seed) dz = dropout_keeps(dseed, (uint64_t)(rr * F + m.f + v), dthr) ? dz * dscale : 0.f;
            if (relu && !(z > 0.f)) dz = 0.f;
            if (APPLY) o[v] = rs[v] * (dz - c1[v] - xh * c2[v]);
            else { s0[v] += dz; s1[v] += dz * xh; }
          }
          if (APPLY) {
            Vec<4>::st(dT + rr * F + m.f, o);
            if (dsum) Vec<4>::st(dsum + rr * lddsum + m.f, o);
          }
        }
      }
    }
  }
  if (!APPLY) k1_block_sums(lred, m, F, FVB, RP, s0, s1, red, nslots);
}

// Backward BN reduction of an EdgeConv conv0 WITHOUT touching the edges.  With relu and dz_e = [z_e = max_i]
// dmax_i / ties_i + dmean_i / k  (zero where z_e <= 0), summing over the k rows of point i gives
//   sum_m dz        = [max_i > 0] dmax_i + dmean_i npos_i / k
//   sum_m dz xhat   = [max_i > 0] dmax_i (max_i - beta) + (dmean_i / k) (k mean_i - beta npos_i)
// (xhat = z - beta where z > 0; the ties all equal max_i; k mean_i = sum of the positive z): only the forward's
// per-point outputs (max, mean, packed ties/positives) and the incoming gradients are needed.
__global__ __launch_bounds__(256) void edge_bwd_reduce_points_kernel(
    const float* __restrict__ mx, int64_t ldmx, const float* __restrict__ mn, int64_t ldmn,
    const float* __restrict__ cntpos, const float* __restrict__ dmax, int64_t lddmax,
    const float* __restrict__ dmean, int64_t lddmean, const float* __restrict__ beta, int64_t R, int k, int F,
    int FVB, int RP, double* __restrict__ red, int nslots) {
  extern __shared__ float lred[];          // K1_PARK_FLOATS
  const K1Map m = k1_map(F, FVB, RP);
  float s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (m.on) {
    float be[4];
    Vec<4>::ld(beta + m.f, be);
    const float invk = 1.0f / (float)k, kf = (float)k;
    for (int64_t r = m.r0; r < R; r += m.rstep) {
      float a[4], b[4], c[4], d[4], e[4];
      Vec<4>::ld(mx + r * ldmx + m.f, a); Vec<4>::ld(mn + r * ldmn + m.f, b); Vec<4>::ld(cntpos + r * F + m.f, c);
      Vec<4>::ld(dmax + r * lddmax + m.f, d); Vec<4>::ld(dmean + r * lddmean + m.f, e);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float npos = floorf(c[v] * (1.0f / CNT_POS));
        const float g1 = (a[v] > 0.f) ? d[v] : 0.f;
        const float g2 = e[v] * invk;
        s0[v] += g1 + g2 * npos;
        s1[v] += g1 * (a[v] - be[v]) + g2 * (kf * b[v] - be[v] * npos);
      }
    }
  }
  k1_block_sums(lred, m, F, FVB, RP, s0, s1, red, nslots);
}

struct K1Grid { dim3 grid; int FVB, RP; };
inline K1Grid k1_grid(int64_t R, int F, int max_blocks, int threads = 256) {
  K1Grid g;
  const int FV = F / 4;
  g.FVB = FV < 256 ? FV : 256;
  g.RP = threads / g.FVB;
  int64_t gx = dg::cdiv(R, (int64_t)g.RP * 4);             // >= 4 rows per thread
  if (gx > max_blocks) gx = max_blocks;
  if (gx < 1) gx = 1;
  g.grid = dim3((unsigned)gx, (unsigned)dg::cdiv(FV, 256));
  return g;
}

inline unsigned grid_for(int64_t items) {
  int64_t g = dg::cdiv(items, 256);
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int shift_of(int fv) {
  if (fv <= 0 || (fv & (fv - 1))) return -1;
  int s = 0;
  while ((1 << s) < fv) ++s;
  return s;
}

// the reduce kernel ends with 2F atomics per block: keep the grid small (4 blocks per CU)
inline unsigned grid_reduce(int64_t items) {
  int64_t g = dg::cdiv(items, 256);
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (unsigned)g;
}

inline Src dense_src(const float* Y) { Src s = {}; s.Y = Y; return s; }

inline Src edge_src(const float* V, int64_t ldv, const float* U, int64_t ldu, const int32_t* idx, int N) {
  Src s = {};
  s.V = V; s.ldv = ldv; s.U = U; s.ldu = ldu; s.idx = idx; s.npts = (unsigned)N;
  return s;
}

inline unsigned grid8(unsigned g) { return (g + 7u) & ~7u; }     // XCD-aware item mapping needs a multiple of 8

// shared argument checks of the gather-sourced (edge) variants
int check_edge(const char* what, const float* V, int64_t ldv, const float* U, int64_t ldu, const int32_t* idx,
               int B, int N, int k, int F) {
  DG_REQUIRE(V && U && idx, DGCNN_EINVAL, "%s: null pointer", what);
  DG_REQUIRE(B > 0 && N > 0 && k > 0 && F > 0, DGCNN_EINVAL, "%s: bad shape", what);
  DG_REQUIRE(F % 4 == 0, DGCNN_EUNSUP, "%s: F must be a multiple of 4 (got %d)", what, F);
  DG_REQUIRE((int64_t)B * N * k < (1ll << 31), DGCNN_EUNSUP, "%s: B*N*k >= 2^31", what);
  DG_REQUIRE(k < CNT_POS, DGCNN_EUNSUP, "%s: k must be < %d", what, CNT_POS);
  DG_REQUIRE(N < (1 << 24) && ldv < (1 << 24) && (int64_t)N * ldv < (1ll << 32), DGCNN_EUNSUP,
             "%s: N * ldv must be < 2^32 elements (32-bit row offsets inside a cloud)", what);
  DG_REQUIRE(a16(V) && a16(U) && ldv % 4 == 0 && ldu % 4 == 0 && ldv >= F && ldu >= F, DGCNN_EINVAL,
             "%s: V, U must be 16-byte aligned with leading dimensions %% 4 == 0", what);
  return DGCNN_OK;
}

template <bool G>
int launch_act_kreduce(const char* what, Src src, int64_t R, int k, int F, const float* mean, const float* rstd,
                       const float* beta, int relu, float* max_out, int64_t ldmax, float* mean_out, int64_t ldmean,
                       float* out2, int64_t ldout2, float* cnt_out, hipStream_t st) {
  DG_REQUIRE(mean && rstd && beta && max_out, DGCNN_EINVAL, "%s: null pointer", what);
  const bool vec = (F % 4 == 0) && (ldmax % 4 == 0) && (G || a16(src.Y)) && a16(max_out) && a16(mean) && a16(rstd) &&
                   a16(beta) && (!mean_out || ((ldmean % 4 == 0) && a16(mean_out))) &&
                   (!out2 || ((ldout2 % 4 == 0) && a16(out2))) && (!cnt_out || a16(cnt_out));
  if (G) {
    DG_REQUIRE(vec, DGCNN_EINVAL, "%s: outputs must be 16-byte aligned with leading dimensions %% 4 == 0", what);
    dg::launch((bn_act_kreduce_kernel<4, true>), dim3(grid8(grid_for(R * (F / 4)))), dim3(256), 0, st, src, R, k, F,
                       mean, rstd, beta, relu, max_out, ldmax, mean_out, ldmean, out2, ldout2, cnt_out, shift_of(F / 4));
  } else if (vec && k == 1 && !mean_out) {
    const K1Grid g = k1_grid(R, F, 4096);
    dg::launch(bn1_act_kernel, g.grid, dim3(256), 0, st, src.Y, R, F, g.FVB, g.RP, mean, rstd, beta, relu, max_out,
                       ldmax, out2, ldout2, cnt_out, (const uint64_t*)nullptr, 1.0f);
  } else if (vec) {
    dg::launch((bn_act_kreduce_kernel<4, false>), dim3(grid_for(R * (F / 4))), dim3(256), 0, st, src, R, k, F, mean,
                       rstd, beta, relu, max_out, ldmax, mean_out, ldmean, out2, ldout2, cnt_out, shift_of(F / 4));
  } else {
    dg::launch((bn_act_kreduce_kernel<1, false>), dim3(grid_for(R * F)), dim3(256), 0, st, src, R, k, F, mean, rstd,
                       beta, relu, max_out, ldmax, mean_out, ldmean, out2, ldout2, cnt_out, shift_of(F));
  }
  return dg::check_launch(what);
}

template <bool G>
int launch_bwd_reduce(const char* what, Src src, int64_t R, int k, int F, const float* mean, const float* rstd,
                      const float* beta, int relu, const float* dmax, int64_t lddmax, const float* dmean,
                      int64_t lddmean, const float* mx_in, int64_t ldmx, const float* cnt_in, double* red,
                      hipStream_t st) {
  DG_REQUIRE(mean && rstd && beta && dmax && red, DGCNN_EINVAL, "%s: null pointer", what);
  DG_REQUIRE(F <= 8192, DGCNN_EINVAL, "%s: F > 8192", what);
  DG_REQUIRE(!mx_in || cnt_in, DGCNN_EINVAL, "%s: mx_in needs cnt_in", what);
  const bool vec = (F % 4 == 0) && (lddmax % 4 == 0) && (G || a16(src.Y)) && a16(dmax) && a16(mean) && a16(rstd) &&
                   a16(beta) && (!dmean || ((lddmean % 4 == 0) && a16(dmean))) &&
                   (!mx_in || ((ldmx % 4 == 0) && a16(mx_in) && a16(cnt_in)));
  const size_t sh = (size_t)2 * F * sizeof(float);
  if (G) {
    DG_REQUIRE(vec, DGCNN_EINVAL, "%s: operands must be 16-byte aligned with leading dimensions %% 4 == 0", what);
    dg::launch((bn_bwd_reduce_kernel<4, true>), dim3(grid8(grid_reduce(R * (F / 4)))), dim3(256), sh, st, src, R, k,
                       F, mean, rstd, beta, relu, dmax, lddmax, dmean, lddmean, mx_in, ldmx, cnt_in, red, shift_of(F / 4), dg::stat_slots());
  } else if (vec && k == 1 && !mx_in) {          // k = 1: dz = dmax + dmean / 1 -- `dmean` is the gradient of a second copy of the output
    const int mb = 256;
    // the kernel ends with 2F double atomics per workgroup, which dominate above ~1 workgroup per CU
    // (F = 256: 19 us at 256 workgroups, 32 us at 2048; 1024-thread workgroups are slower: profiles/bn1_bench.py)
    const K1Grid g = k1_grid(R, F, mb);
    const size_t sh1 = sizeof(float) * K1_PARK_FLOATS;
    dg::launch((bn1_bwd_kernel<false>), g.grid, dim3(256), sh1, st, src.Y, R, F, g.FVB, g.RP, mean, rstd, beta, relu,
                       dmax, lddmax, dmean, lddmean, red, (float*)nullptr, (float*)nullptr, (int64_t)0, (const uint64_t*)nullptr, 1.0f,
                       dg::stat_slots());
  } else if (vec) {
    dg::launch((bn_bwd_reduce_kernel<4, false>), dim3(grid_reduce(R * (F / 4))), dim3(256), sh, st, src, R, k, F,
                       mean, rstd, beta, relu, dmax, lddmax, dmean, lddmean, mx_in, ldmx, cnt_in, red, shift_of(F / 4), dg::stat_slots());
  } else {
    dg::launch((bn_bwd_reduce_kernel<1, false>), dim3(grid_reduce(R * F)), dim3(256), sh, st, src, R, k, F, mean,
                       rstd, beta, relu, dmax, lddmax, dmean, lddmean, mx_in, ldmx, cnt_in, red, shift_of(F), dg::stat_slots());
  }
  return dg::check_launch(what);
}

template <bool G>
int launch_bwd_apply(const char* what, Src src, int64_t R, int k, int F, const float* mean, const float* rstd,
                     const float* beta, int relu, const float* dmax, int64_t lddmax, const float* dmean,
                     int64_t lddmean, const float* mx_in, int64_t ldmx, const float* cnt_in, double* red, float* dY,
                     float* dYsum, int64_t lddysum, float* dbeta, float dbeta_beta, hipStream_t st) {
  DG_REQUIRE(mean && rstd && beta && dmax && red && dY, DGCNN_EINVAL, "%s: null pointer", what);
  DG_REQUIRE(!dYsum || lddysum >= F, DGCNN_EINVAL, "%s: lddysum < F", what);
  dg::launch(bn_bwd_finalize_kernel, dim3((unsigned)dg::cdiv(F, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, st, red, F, dg::stat_slots(), dbeta,
                     dbeta_beta);
  const bool vec = (F % 4 == 0) && (lddmax % 4 == 0) && (G || a16(src.Y)) && a16(dY) && a16(dmax) && a16(mean) &&
                   a16(rstd) && a16(beta) && (!dmean || ((lddmean % 4 == 0) && a16(dmean))) &&
                   (!dYsum || (a16(dYsum) && lddysum % 4 == 0)) &&
                   (!mx_in || ((ldmx % 4 == 0) && a16(mx_in) && a16(cnt_in)));
  if (G) {
    DG_REQUIRE(vec, DGCNN_EINVAL, "%s: operands must be 16-byte aligned with leading dimensions %% 4 == 0", what);
    dg::launch((bn_bwd_apply_kernel<4, true>), dim3(grid8(grid_for(R * (F / 4)))), dim3(256), 0, st, src, R, k, F,
                       mean, rstd, beta, relu, dmax, lddmax, dmean, lddmean, mx_in, ldmx, cnt_in, red, dY, dYsum, lddysum,
                       shift_of(F / 4));
  } else if (vec && k == 1 && !mx_in) {
    const K1Grid g = k1_grid(R, F, 4096);
    dg::launch((bn1_bwd_kernel<true>), g.grid, dim3(256), 0, st, src.Y, R, F, g.FVB, g.RP, mean, rstd, beta, relu,
                       dmax, lddmax, dmean, lddmean, red, dY, dYsum, lddysum, (const uint64_t*)nullptr, 1.0f, dg::stat_slots());
  } else if (vec) {
    dg::launch((bn_bwd_apply_kernel<4, false>), dim3(grid_for(R * (F / 4))), dim3(256), 0, st, src, R, k, F, mean,
                       rstd, beta, relu, dmax, lddmax, dmean, lddmean, mx_in, ldmx, cnt_in, red, dY, dYsum, lddysum,
                       shift_of(F / 4));
  } else {
    dg::launch((bn_bwd_apply_kernel<1, false>), dim3(grid_for(R * F)), dim3(256), 0, st, src, R, k, F, mean, rstd,
                       beta, relu, dmax, lddmax, dmean, lddmean, mx_in, ldmx, cnt_in, red, dY, dYsum, lddysum, shift_of(F));
  }
  return dg::check_launch(what);
}

}  // namespace

extern "C" int dgcnn_bn_finalize_f32(const double* stats, int F, double count, float eps,
                                     float* mean, float* rstd, void* stream) {
  DG_REQUIRE(stats && mean && rstd && F > 0 && count > 0, DGCNN_EINVAL, "dgcnn_bn_finalize_f32: bad args");
  dg::launch(bn_finalize_kernel, dim3((unsigned)dg::cdiv(F, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, (hipStream_t)stream,
                     stats, F, dg::stat_slots(), count, eps, mean, rstd);
  return dg::check_launch("dgcnn_bn_finalize_f32");
}

extern "C" int dgcnn_bn_act_kreduce_f32(const float* Y, int64_t R, int k, int F,
                                        const float* mean, const float* rstd, const float* beta, int relu,
                                        float* max_out, int64_t ldmax, float* mean_out, int64_t ldmean,
                                        float* out2, int64_t ldout2, float* cnt_out, void* stream) {
  DG_REQUIRE(Y, DGCNN_EINVAL, "dgcnn_bn_act_kreduce_f32: null pointer");
  DG_REQUIRE(R > 0 && k > 0 && F > 0, DGCNN_EINVAL, "dgcnn_bn_act_kreduce_f32: bad shape");
  return launch_act_kreduce<false>("dgcnn_bn_act_kreduce_f32", dense_src(Y), R, k, F, mean, rstd, beta, relu, max_out,
                                   ldmax, mean_out, ldmean, out2, ldout2, cnt_out, (hipStream_t)stream);
}

extern "C" int dgcnn_bn_bwd_reduce_f32(const float* Y, int64_t R, int k, int F,
                                       const float* mean, const float* rstd, const float* beta, int relu,
                                       const float* dmax, int64_t lddmax, const float* dmean, int64_t lddmean,
                                       const float* mx_in, int64_t ldmx, const float* cnt_in,
                                       double* red, void* stream) {
  DG_REQUIRE(Y, DGCNN_EINVAL, "dgcnn_bn_bwd_reduce_f32: null pointer");
  DG_REQUIRE(R > 0 && k > 0 && F > 0, DGCNN_EINVAL, "dgcnn_bn_bwd_reduce_f32: bad shape");
  return launch_bwd_reduce<false>("dgcnn_bn_bwd_reduce_f32", dense_src(Y), R, k, F, mean, rstd, beta, relu, dmax, lddmax,
                                  dmean, lddmean, mx_in, ldmx, cnt_in, red, (hipStream_t)stream);
}

extern "C" int dgcnn_bn_bwd_apply_f32(const float* Y, int64_t R, int k, int F,
                                      const float* mean, const float* rstd, const float* beta, int relu,
                                      const float* dmax, int64_t lddmax, const float* dmean, int64_t lddmean,
                                      const float* mx_in, int64_t ldmx, const float* cnt_in,
                                      double* red, float* dY, float* dYsum, int64_t lddysum, float* dbeta,
                                      float dbeta_beta, void* stream) {
  DG_REQUIRE(Y, DGCNN_EINVAL, "dgcnn_bn_bwd_apply_f32: null pointer");
  DG_REQUIRE(R > 0 && k > 0 && F > 0, DGCNN_EINVAL, "dgcnn_bn_bwd_apply_f32: bad shape");
  return launch_bwd_apply<false>("dgcnn_bn_bwd_apply_f32", dense_src(Y), R, k, F, mean, rstd, beta, relu, dmax, lddmax,
                                 dmean, lddmean, mx_in, ldmx, cnt_in, red, dY, dYsum, lddysum, dbeta, dbeta_beta,
                                 (hipStream_t)stream);
}

// ---- the same three passes over the never-materialised conv0 output  y = V[neighbour] + U[point] ----
extern "C" int dgcnn_edge_bn_act_kreduce_f32(const float* V, int64_t ldv, const float* U, int64_t ldu,
                                             const int32_t* idx, int B, int N, int k, int F,
                                             const float* mean, const float* rstd, const float* beta, int relu,
                                             float* max_out, int64_t ldmax, float* mean_out, int64_t ldmean,
                                             float* cnt_out, void* stream) {
  int rc = check_edge("dgcnn_edge_bn_act_kreduce_f32", V, ldv, U, ldu, idx, B, N, k, F);
  if (rc) return rc;
  return launch_act_kreduce<true>("dgcnn_edge_bn_act_kreduce_f32", edge_src(V, ldv, U, ldu, idx, N), (int64_t)B * N, k, F,
                                  mean, rstd, beta, relu, max_out, ldmax, mean_out, ldmean, nullptr, 0, cnt_out,
                                  (hipStream_t)stream);
}

extern "C" int dgcnn_edge_bn_bwd_reduce_f32(const float* V, int64_t ldv, const float* U, int64_t ldu,
                                            const int32_t* idx, int B, int N, int k, int F,
                                            const float* mean, const float* rstd, const float* beta, int relu,
                                            const float* dmax, int64_t lddmax, const float* dmean, int64_t lddmean,
                                            const float* mx_in, int64_t ldmx, const float* cnt_in,
                                            double* red, void* stream) {
  int rc = check_edge("dgcnn_edge_bn_bwd_reduce_f32", V, ldv, U, ldu, idx, B, N, k, F);
  if (rc) return rc;
  return launch_bwd_reduce<true>("dgcnn_edge_bn_bwd_reduce_f32", edge_src(V, ldv, U, ldu, idx, N), (int64_t)B * N, k, F,
                                 mean, rstd, beta, relu, dmax, lddmax, dmean, lddmean, mx_in, ldmx, cnt_in, red,
                                 (hipStream_t)stream);
}

extern "C" int dgcnn_edge_bn_bwd_reduce_points_f32(const float* mx, int64_t ldmx, const float* mn, int64_t ldmn,
                                                   const float* cntpos, const float* dmax, int64_t lddmax,
                                                   const float* dmean, int64_t lddmean, const float* beta,
                                                   int64_t R, int k, int F, double* red, void* stream) {
  DG_REQUIRE(mx && mn && cntpos && dmax && dmean && beta && red, DGCNN_EINVAL, "dgcnn_edge_bn_bwd_reduce_points_f32: null pointer");
  DG_REQUIRE(R > 0 && k > 0 && k < CNT_POS && F > 0 && F % 4 == 0, DGCNN_EINVAL,
             "dgcnn_edge_bn_bwd_reduce_points_f32: bad shape (F %% 4 == 0, k < %d)", CNT_POS);
  DG_REQUIRE(a16(mx) && a16(mn) && a16(cntpos) && a16(dmax) && a16(dmean) && a16(beta) && ldmx % 4 == 0 && ldmn % 4 == 0 &&
                 lddmax % 4 == 0 && lddmean % 4 == 0, DGCNN_EINVAL,
             "dgcnn_edge_bn_bwd_reduce_points_f32: operands must be 16-byte aligned with leading dimensions %% 4 == 0");
  const K1Grid g = k1_grid(R, F, 256);
  const size_t sh = sizeof(float) * K1_PARK_FLOATS;
  dg::launch(edge_bwd_reduce_points_kernel, g.grid, dim3(256), sh, (hipStream_t)stream, mx, ldmx, mn, ldmn, cntpos,
                     dmax, lddmax, dmean, lddmean, beta, R, k, F, g.FVB, g.RP, red, dg::stat_slots());
  return dg::check_launch("dgcnn_edge_bn_bwd_reduce_points_f32");
}

extern "C" int dgcnn_edge_bn_bwd_apply_f32(const float* V, int64_t ldv, const float* U, int64_t ldu,
                                           const int32_t* idx, int B, int N, int k, int F,
                                           const float* mean, const float* rstd, const float* beta, int relu,
                                           const float* dmax, int64_t lddmax, const float* dmean, int64_t lddmean,
                                           const float* mx_in, int64_t ldmx, const float* cnt_in,
                                           double* red, float* dY, float* dYsum, int64_t lddysum, float* dbeta,
                                           float dbeta_beta, void* stream) {
  int rc = check_edge("dgcnn_edge_bn_bwd_apply_f32", V, ldv, U, ldu, idx, B, N, k, F);
  if (rc) return rc;
  return launch_bwd_apply<true>("dgcnn_edge_bn_bwd_apply_f32", edge_src(V, ldv, U, ldu, idx, N), (int64_t)B * N, k, F, mean,
                                rstd, beta, relu, dmax, lddmax, dmean, lddmean, mx_in, ldmx, cnt_in, red, dY, dYsum,
                                lddysum, dbeta, dbeta_beta, (hipStream_t)stream);
}

// ---- the last FC layer with tf.nn.dropout fused behind it (model.py:88-91): out = dropout(relu(bn(T))), and its backward
// reading the gradient of the dropped output through the same mask.  (R, F) contiguous, F % 4 == 0.
static int check_bn1_drop(const char* what, const float* T, int64_t R, int F, const float* mean, const float* rstd,
                          const float* beta, float keep, const uint64_t* seed_dev) {
  DG_REQUIRE(T && mean && rstd && beta && seed_dev && R > 0 && F > 0 && keep > 0.f && keep <= 1.f, DGCNN_EINVAL, "%s: bad args", what);
  DG_REQUIRE(F % 4 == 0 && a16(T) && a16(mean) && a16(rstd) && a16(beta), DGCNN_EUNSUP, "%s: F %% 4 == 0 and 16-byte aligned operands required", what);
  return DGCNN_OK;
}

extern "C" int dgcnn_bn1_act_dropout_f32(const float* T, int64_t R, int F, const float* mean, const float* rstd,
                                         const float* beta, int relu, float keep, const uint64_t* seed_dev, float* out,
                                         int64_t ldo, void* stream) {
  int rc = check_bn1_drop("dgcnn_bn1_act_dropout_f32", T, R, F, mean, rstd, beta, keep, seed_dev);
  if (rc) return rc;
  DG_REQUIRE(out && ldo % 4 == 0 && a16(out), DGCNN_EINVAL, "dgcnn_bn1_act_dropout_f32: out must be 16-byte aligned, ld %% 4 == 0");
  const K1Grid g = k1_grid(R, F, 4096);
  dg::launch(bn1_act_kernel, g.grid, dim3(256), 0, (hipStream_t)stream, T, R, F, g.FVB, g.RP, mean, rstd, beta, relu, out,
                     ldo, (float*)nullptr, (int64_t)0, (float*)nullptr, seed_dev, keep);
  return dg::check_launch("dgcnn_bn1_act_dropout_f32");
}

extern "C" int dgcnn_bn1_bwd_dropout_f32(const float* T, int64_t R, int F, const float* mean, const float* rstd,
                                         const float* beta, int relu, float keep, const uint64_t* seed_dev,
                                         const float* dout, int64_t lddo, double* red, float* dT, float* dbeta,
                                         float dbeta_beta, void* stream) {
  int rc = check_bn1_drop("dgcnn_bn1_bwd_dropout_f32", T, R, F, mean, rstd, beta, keep, seed_dev);
  if (rc) return rc;
  DG_REQUIRE(dout && red && dT && lddo % 4 == 0 && a16(dout) && a16(dT) && F <= 8192, DGCNN_EINVAL, "dgcnn_bn1_bwd_dropout_f32: bad args");
  hipStream_t st = (hipStream_t)stream;
  const K1Grid gr = k1_grid(R, F, 256);
  const size_t sh1 = sizeof(float) * K1_PARK_FLOATS;
  dg::launch((bn1_bwd_kernel<false>), gr.grid, dim3(256), sh1, st, T, R, F, gr.FVB, gr.RP, mean, rstd, beta, relu, dout, lddo,
                     (const float*)nullptr, (int64_t)0, red, (float*)nullptr, (float*)nullptr, (int64_t)0, seed_dev, keep, dg::stat_slots());
  dg::launch(bn_bwd_finalize_kernel, dim3((unsigned)dg::cdiv(F, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, st, red, F, dg::stat_slots(), dbeta, dbeta_beta);
  const K1Grid ga = k1_grid(R, F, 4096);
  dg::launch((bn1_bwd_kernel<true>), ga.grid, dim3(256), 0, st, T, R, F, ga.FVB, ga.RP, mean, rstd, beta, relu, dout, lddo,
                     (const float*)nullptr, (int64_t)0, red, dT, (float*)nullptr, (int64_t)0, seed_dev, keep, dg::stat_slots());
  return dg::check_launch("dgcnn_bn1_bwd_dropout_f32");
}

namespace dg {
void launch_reduce_partials(const float* part, int splits, int M, int N, float* C, int64_t ldc, float beta, hipStream_t st);
// slots of a backward reduction -> slot 0 (+ d(beta)); edge_mlp_bf16.hip
void launch_bn_bwd_finalize(double* red, int F, float* dbeta, float dbeta_beta, hipStream_t st) {
  dg::launch(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(F, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, st, red, F, stat_slots(), dbeta, dbeta_beta);
}
}

extern "C" int dgcnn_edge_bn_bwd_apply_wgrad_f32(const float* V, int64_t ldv, const float* U, int64_t ldu,
                                                 const int32_t* idx, int B, int N, int k, int F,
                                                 const float* mean, const float* rstd, const float* beta,
                                                 const float* dmax, int64_t lddmax, const float* dmean, int64_t lddmean,
                                                 const float* mx_in, int64_t ldmx, const float* cnt_in, double* red,
                                                 const float* x, int64_t ldx, int C, float* dW0, float* dbeta,
                                                 float dbeta_beta, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_edge("dgcnn_edge_bn_bwd_apply_wgrad_f32", V, ldv, U, ldu, idx, B, N, k, F);
  if (rc) return rc;
  DG_REQUIRE(mean && rstd && beta && dmax && dmean && mx_in && cnt_in && red && x && dW0 && ws, DGCNN_EINVAL,
             "dgcnn_edge_bn_bwd_apply_wgrad_f32: null pointer");
  DG_REQUIRE(C >= 1 && C <= 4, DGCNN_EUNSUP, "dgcnn_edge_bn_bwd_apply_wgrad_f32: C must be <= 4 (got %d)", C);
  const int FV = F / 4;
  DG_REQUIRE(shift_of(FV) >= 0 && FV <= 256 && lddmax % 4 == 0 && lddmean % 4 == 0 && ldmx % 4 == 0 && a16(dmax) && a16(dmean) &&
                 a16(mx_in) && a16(cnt_in) && a16(mean) && a16(rstd) && a16(beta), DGCNN_EUNSUP,
             "dgcnn_edge_bn_bwd_apply_wgrad_f32: F / 4 must be a power of two <= 256, operands 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t R = (int64_t)B * N;
  dg::launch(bn_bwd_finalize_kernel, dim3((unsigned)dg::cdiv(F, FIN_COLS)), dim3(FIN_COLS * FIN_LANES), 0, st, red, F, dg::stat_slots(), dbeta, dbeta_beta);
  const unsigned grid = grid8(grid_reduce(R * FV));                 // <= 1024 blocks: one partial tile each
  const size_t need = (size_t)grid * 2 * C * F * sizeof(float);
  DG_REQUIRE(ws_bytes >= need, DGCNN_ENOSPC, "dgcnn_edge_bn_bwd_apply_wgrad_f32: workspace too small (%zu < %zu)", ws_bytes, need);
  const size_t shb = (size_t)(256 / FV) * 2 * 4 * F * sizeof(float);
  dg::launch((edge_bwd_apply_wgrad_kernel<4>), dim3(grid), dim3(256), shb, st, edge_src(V, ldv, U, ldu, idx, N), R, k, F,
                     mean, rstd, beta, dmax, lddmax, dmean, lddmean, mx_in, ldmx, cnt_in, red, x, ldx, C,
                     reinterpret_cast<float*>(ws), shift_of(FV));
  rc = dg::check_launch("dgcnn_edge_bn_bwd_apply_wgrad_f32");
  if (rc) return rc;
  dg::launch_reduce_partials(reinterpret_cast<const float*>(ws), (int)grid, 2 * C, F, dW0, (int64_t)F, 1.0f, st);
  return dg::check_launch("dgcnn_edge_bn_bwd_apply_wgrad_f32(reduce)");
}
