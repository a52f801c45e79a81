// det.hip -- fixed-order (run-to-run reproducible) versions of the three steps of the path whose default kernels
// accumulate with atomics: BatchNorm column statistics, the two sums of the BatchNorm backward, and the fill order of the
// transposed adjacency.  Selected by the host in deterministic mode (dgcnn._engine.DETERMINISTIC / DGCNN_DETERMINISTIC=1);
// slower than the fused epilogues (one extra pass over the tensor, G blocks x sequential rows), same math:
//   stage 1  block g owns the contiguous row range g: every thread walks its columns down the rows in order, fp64 sums;
//   stage 2  one thread per column adds the G partials in g order -> slot 0 of the caller's (zeroed) stats buffer,
//            the layout dgcnn_bn_finalize_f32 / dgcnn_bn_bwd_apply_f32 read.
#include "common.h"
#include <math.h>

namespace {

constexpr int DET_G = 1024;    // row ranges (workgroups) of stage 1
constexpr int DET_C = 4;       // columns per thread (F <= 1024 with 256 threads)

// thread -> (column f, row lane): narrow tensors (F < 256) split a block's row range over 256 / F lanes, whose partials are
// added in lane order through LDS; wide ones give every thread up to DET_C columns.
struct DetMap {
  int lanes, lane, f0, nf;       // this thread's columns: f0 + 256 * c, c < nf  (lanes == 1)  or the single column f0
  bool on;
  __device__ __forceinline__ DetMap(int F) {
    lanes = (F < 256) ? 256 / F : 1;
    if (lanes > 1) { lane = threadIdx.x / F; f0 = threadIdx.x % F; nf = 1; on = lane < lanes; }
    else { lane = 0; f0 = threadIdx.x; nf = (F - threadIdx.x + 255) / 256; on = f0 < F; }
  }
};

__device__ __forceinline__ void det_store(const DetMap& m, int F, const double (&s)[DET_C], const double (&q)[DET_C],
                                          double* __restrict__ part, double* lds) {
  if (m.lanes > 1) {                                  // lds[lane][2][F]
    if (m.on) { lds[(m.lane * 2 + 0) * F + m.f0] = s[0]; lds[(m.lane * 2 + 1) * F + m.f0] = q[0]; }
    __syncthreads();
    if (m.on && m.lane == 0) {
      double ts = 0.0, tq = 0.0;
      for (int l = 0; l < m.lanes; ++l) { ts += lds[(l * 2 + 0) * F + m.f0]; tq += lds[(l * 2 + 1) * F + m.f0]; }
      part[((int64_t)blockIdx.x * 2 + 0) * F + m.f0] = ts;
      part[((int64_t)blockIdx.x * 2 + 1) * F + m.f0] = tq;
    }
  } else if (m.on) {
#pragma unroll
    for (int c = 0; c < DET_C; ++c)
      if (c < m.nf) {
        part[((int64_t)blockIdx.x * 2 + 0) * F + m.f0 + 256 * c] = s[c];
        part[((int64_t)blockIdx.x * 2 + 1) * F + m.f0 + 256 * c] = q[c];
      }
  }
}

// rows [r0, r1) of the block, then the sub-range of this thread's lane
__device__ __forceinline__ void det_range(const DetMap& m, int64_t rows, int64_t& r0, int64_t& r1) {
  const int64_t chunk = (rows + DET_G - 1) / DET_G;
  int64_t b0 = (int64_t)blockIdx.x * chunk;
  int64_t b1 = (b0 + chunk < rows) ? b0 + chunk : rows;
  if (b0 > rows) b0 = rows;
  const int64_t sub = (b1 - b0 + m.lanes - 1) / m.lanes;
  r0 = b0 + sub * m.lane;
  r1 = (r0 + sub < b1) ? r0 + sub : b1;
  if (r0 > b1) r0 = b1;
}

__global__ __launch_bounds__(256) void colstats_det_kernel(const float* __restrict__ Y, int64_t rows, int F, int64_t ld,
                                                           double* __restrict__ part) {
  __shared__ double lds[2 * 256];
  const DetMap m(F);
  int64_t r0, r1;
  det_range(m, rows, r0, r1);
  double s[DET_C], q[DET_C];
#pragma unroll
  for (int c = 0; c < DET_C; ++c) { s[c] = 0.0; q[c] = 0.0; }
  if (m.on) {
    for (int64_t r = r0; r < r1; ++r) {
#pragma unroll
      for (int c = 0; c < DET_C; ++c)
        if (c < m.nf) {
          const double v = (double)Y[r * ld + m.f0 + 256 * c];
          s[c] += v;
          q[c] += v * v;
        }
    }
  }
  det_store(m, F, s, q, part, lds);
}

// one wave per output (which, f): lane l adds the partials g = l, l + 64, ... in that order (DET_G / 64 independent loads in
// flight), then the 64 lane sums are added in lane order by a fixed butterfly -- a fixed order, run to run.  (Round 4: one
// THREAD per output walked the 1024 partials, 68 us for the two columns of the class dimension.)
__global__ __launch_bounds__(64) void det_stage2_kernel(const double* __restrict__ part, int F, double* __restrict__ out) {
  const int e = blockIdx.x;                                 // e in [0, 2F): which * F + f
  const int which = e / F, f = e % F;
  const int l = threadIdx.x;
  double v[DET_G / 64];
#pragma unroll
  for (int i = 0; i < DET_G / 64; ++i) v[i] = part[((int64_t)(l + 64 * i) * 2 + which) * F + f];
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < DET_G / 64; ++i) t += v[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
  if (l == 0) out[(int64_t)which * F + f] = t;              // slot 0
}

// sum dZ and sum dZ * xhat over all R*k rows (dgcnn_bn_bwd_reduce_f32's quantities, same formulas as bn.hip) from the
// materialised Y.  dmean == NULL: the k = 1 form (dZ = relu'(z) dout).
__global__ __launch_bounds__(256) void bn_bwd_reduce_det_kernel(
    const float* __restrict__ Y, int64_t R, int k, int F, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ beta, int relu, const float* __restrict__ dmax, int64_t lddmax,
    const float* __restrict__ dmean, int64_t lddmean, const float* __restrict__ mx_in, int64_t ldmx,
    const float* __restrict__ cnt_in, double* __restrict__ part) {
  __shared__ double lds[2 * 256];
  const DetMap m(F);
  int64_t r0, r1;
  det_range(m, R, r0, r1);
  const float invk = 1.0f / (float)k;
  double s[DET_C], q[DET_C];
  float mu[DET_C], rs[DET_C], be[DET_C];
#pragma unroll
  for (int c = 0; c < DET_C; ++c) {
    s[c] = 0.0; q[c] = 0.0;
    const int f = m.f0 + 256 * c;
    const bool ok = m.on && c < m.nf;
    mu[c] = ok ? mean[f] : 0.f; rs[c] = ok ? rstd[f] : 0.f; be[c] = ok ? beta[f] : 0.f;
  }
  if (m.on) {
    for (int64_t r = r0; r < r1; ++r) {
#pragma unroll
      for (int c = 0; c < DET_C; ++c) {
        if (c >= m.nf) continue;
        const int f = m.f0 + 256 * c;
        const float dmx = dmax[r * lddmax + f];
        float dmn = 0.f, mx = 0.f, cnt = 1.f;
        if (dmean) { dmn = dmean[r * lddmean + f]; mx = mx_in[r * ldmx + f]; cnt = cnt_in[r * F + f]; }
        for (int mm = 0; mm < k; ++mm) {
          const float y = Y[(r * k + mm) * F + f];
          const float xh = (y - mu[c]) * rs[c];
          float z = xh + be[c];
          if (relu) z = fmaxf(z, 0.f);
          float dz;
          if (dmean) dz = ((z == mx) ? dmx / cnt : 0.f) + dmn * invk;
          else dz = dmx;
          if (relu && !(z > 0.f)) dz = 0.f;
          s[c] += (double)dz;
          q[c] += (double)(dz * xh);
        }
      }
    }
  }
  det_store(m, F, s, q, part, lds);
}

// every bucket of the transposed adjacency in ascending edge order (the build fills buckets through LDS cursors: any order).
// Out of place, one thread per stored edge: its bucket is the edge's target point (from idx), its place the number of smaller
// edge ids in that bucket -- ~k reads of neighbouring words per thread, no dependent chain.  (Round 4: one thread per bucket,
// insertion sort in global memory, 35-65 us per layer.)
__global__ __launch_bounds__(256) void csr_sort_kernel(const int32_t* __restrict__ idx, int N, int k, const int32_t* __restrict__ off,
                                                       const int32_t* __restrict__ rev, int32_t* __restrict__ out, int64_t total) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const int32_t e = rev[p];                                  // edge id = point * k + m (global point numbering)
  const int64_t cloud = (int64_t)e / ((int64_t)N * k);
  const int64_t j = cloud * N + idx[e];
  const int b = off[j], n = off[j + 1] - b;
  int rank = 0;
  for (int q = 0; q < n; ++q) rank += (rev[b + q] < e) ? 1 : 0;
  out[b + rank] = e;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int dgcnn_det_workspace_bytes(int F) { return (int)(sizeof(double) * DET_G * 2 * (size_t)F); }

extern "C" int dgcnn_colstats_det_f32(const float* Y, int64_t rows, int F, int64_t ld, double* stats, void* ws,
                                      size_t ws_bytes, void* stream) {
  DG_REQUIRE(Y && stats && ws && rows > 0 && F > 0 && F <= 256 * DET_C, DGCNN_EINVAL, "dgcnn_colstats_det_f32: bad args (F <= %d)", 256 * DET_C);
  DG_REQUIRE(ws_bytes >= sizeof(double) * DET_G * 2 * (size_t)F, DGCNN_ENOSPC, "dgcnn_colstats_det_f32: workspace too small");
  double* part = reinterpret_cast<double*>(ws);
  dg::launch(colstats_det_kernel, dim3(DET_G), dim3(256), 0, ST, Y, rows, F, ld, part);
  dg::launch(det_stage2_kernel, dim3((unsigned)(2 * F)), dim3(64), 0, ST, part, F, stats);
  return dg::check_launch("dgcnn_colstats_det_f32");
}

extern "C" int dgcnn_bn_bwd_reduce_det_f32(const float* Y, int64_t R, int k, int F, const float* mean, const float* rstd,
                                           const float* beta, int relu, const float* dmax, int64_t lddmax,
                                           const float* dmean, int64_t lddmean, const float* mx_in, int64_t ldmx,
                                           const float* cnt_in, double* red, void* ws, size_t ws_bytes, void* stream) {
  DG_REQUIRE(Y && mean && rstd && beta && dmax && red && ws && R > 0 && k > 0 && F > 0 && F <= 256 * DET_C, DGCNN_EINVAL,
             "dgcnn_bn_bwd_reduce_det_f32: bad args");
  DG_REQUIRE(!dmean || (mx_in && cnt_in), DGCNN_EINVAL, "dgcnn_bn_bwd_reduce_det_f32: the k > 1 form needs the forward's max / tie counts");
  DG_REQUIRE(ws_bytes >= sizeof(double) * DET_G * 2 * (size_t)F, DGCNN_ENOSPC, "dgcnn_bn_bwd_reduce_det_f32: workspace too small");
  double* part = reinterpret_cast<double*>(ws);
  dg::launch(bn_bwd_reduce_det_kernel, dim3(DET_G), dim3(256), 0, ST, Y, R, k, F, mean, rstd, beta, relu, dmax, lddmax,
                     dmean, lddmean, mx_in, ldmx, cnt_in, part);
  dg::launch(det_stage2_kernel, dim3((unsigned)(2 * F)), dim3(64), 0, ST, part, F, red);
  return dg::check_launch("dgcnn_bn_bwd_reduce_det_f32");
}

extern "C" int dgcnn_edge_csr_sort(const int32_t* idx, int B, int N, int k, const int32_t* off, const int32_t* rev,
                                   int32_t* rev_sorted, void* stream) {
  DG_REQUIRE(idx && off && rev && rev_sorted && rev != rev_sorted && B > 0 && N > 0 && k > 0, DGCNN_EINVAL, "dgcnn_edge_csr_sort: bad args");
  const int64_t total = (int64_t)B * N * k;
  DG_REQUIRE(total < (int64_t)1 << 31, DGCNN_EINVAL, "dgcnn_edge_csr_sort: more than 2^31 edges");
  dg::launch(csr_sort_kernel, dim3((unsigned)dg::cdiv(total, 256)), dim3(256), 0, ST, idx, N, k, off, rev, rev_sorted, total);
  return dg::check_launch("dgcnn_edge_csr_sort");
}
