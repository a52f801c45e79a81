// planes_bn.hip -- the BatchNorm passes of the head layers (slim.conv2d 1x1 + slim.batch_norm + ReLU on per-point
// tensors: dgcnn/model.py:65-72 MergedEdgeConv, dgcnn/ops.py:153-160 FC%d) when their output feeds a plane GEMM
// (gemm_pl.hip): the pass that normalises the GEMM output anyway WRITES THE OPERAND PLANES of the next GEMM -- the
// split is VALU work inside an HBM-bound kernel, the planes of the 2-plane fp16 format are as many bytes as the fp32
// tensor they replace, and no GEMM tile ever splits an operand again.
//   forward   dgcnn_bn_act_planes_f32        z = relu((T - mean) rstd + beta)  ->  planes (+ optional fp32 copies)
//   backward  dgcnn_bn1_bwd_reduce_max_f32   column sums of dz, dz*xhat (as bn.hip) + column maxima of |dz|, |xhat|
//             dgcnn_bn1_bwd_apply_planes_f32 finalise the sums, bound |dT| per column, pick the tensor's power-of-two scale
//                                            from the bound, dT = rstd (dz - c1 - xhat c2) -> planes (+ per-cloud column sums)
//   scales    dgcnn_param_scales_f32         one pass over the parameter bucket: scale of every activation plane set
//                                            (|z| <= sqrt(rows - 1) + max |beta| for ANY batch-normalised tensor) and of the weights
// Tiles: 64 rows x 16 channel octets per 256-thread block, slots transposed through LDS (planes_common.h) so that every wave
// store is 1 KiB contiguous.
#include "planes_common.h"
#include <stdlib.h>

namespace {

constexpr int SLOTS = DGCNN_STAT_SLOTS;

__device__ __forceinline__ float bn_z(float y, float mu, float rs, float be, int relu, float& xh) {   // == bn.hip:bn_z
  xh = (y - mu) * rs;
  float z = xh + be;
  if (relu) z = fmaxf(z, 0.f);
  return z;
}

__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// ---------------------------------------------------------------------------------------------- scales
__global__ __launch_bounds__(256) void absmax_flat_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ out_bits) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out_bits, __float_as_uint(m));
}

// scales[0]: activations.  A batch-normalised value is (x - mean) rstd + beta with |x - mean| rstd <= sqrt(rows - 1) for
// ANY data (one sample cannot be further than that from the mean of its batch in units of the batch's standard deviation),
// so |z| <= sqrt(rows_max) + max |beta| <= sqrt(rows_max) + max |parameter|; act_mul covers sums of two such tensors
// (the residual add, ops.py:134).  scales[1]: the weights themselves (<= max |parameter|).
__global__ void param_scales_kernel(const unsigned* __restrict__ max_bits, double rows_max, float act_mul, float* __restrict__ scales) {
  const float pm = __uint_as_float(*max_bits);
  scales[0] = pow2_scale_for(act_mul * ((float)sqrt(rows_max) + pm));
  scales[1] = pow2_scale_for(pm);
}

// ---------------------------------------------------------------------------------------------- forward
// lane = row: a wave stores 64 consecutive 16-byte slots of one octet (1 KiB contiguous per plane).  No LDS, no barrier.
template <int FMT>
__device__ __forceinline__ void store_slot(char* dst, int64_t plane_stride, int64_t slot, const uint4 (&o)[3]) {
#pragma unroll
  for (int pl = 0; pl < PlaneFmt<FMT>::NPL; ++pl) *reinterpret_cast<uint4*>(dst + pl * plane_stride + slot * 16) = o[pl];
}

// Thread = one row x OPT = 4 adjacent channel octets (one full 128-byte line of every source row, used by nobody else -- with
// one octet per thread the four waves of a block shared each line through the 32 KB vector cache and thrashed it:
// 3.4 TB/s), lane = row.  The BatchNorm parameters of an octet are wave-uniform loads.
constexpr int OPT = 4;

__device__ __forceinline__ int wave_octet0() {                        // first octet of this wave (wave-uniform)
  return (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * OPT;
}

template <int FMT>
__global__ __launch_bounds__(256) void bn_act_planes_kernel(const float* __restrict__ T, int64_t ldt, int64_t R, int F,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ beta, int relu,
                                                            const float* __restrict__ scale_dev, char* __restrict__ dst,
                                                            int64_t plane_stride, int64_t rows_alloc,
                                                            float* __restrict__ out, int64_t ldo, float* __restrict__ out2,
                                                            int64_t ldo2) {
  const int lane = threadIdx.x & 63;
  const int oc0 = wave_octet0();
  const int noct = F >> 3;
  if (oc0 >= noct) return;
  const float scale = (FMT == DGCNN_PLANES_F16X2 && scale_dev) ? *scale_dev : 1.f;
  const int64_t step = (int64_t)gridDim.x * 64;
  for (int64_t r = (int64_t)blockIdx.x * 64 + lane; r < rows_alloc; r += step) {
    float y[OPT][8];
    if (r < R) {
#pragma unroll
      for (int u = 0; u < OPT; ++u)
        if (oc0 + u < noct) ld8(T + r * ldt + (oc0 + u) * 8, y[u]);
    }
#pragma unroll
    for (int u = 0; u < OPT; ++u) {
      const int oc = oc0 + u;
      if (oc >= noct) break;
      const int c = oc * 8;
      float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // pad rows hold zeros (the k-strided GEMM reduces over them)
      if (r < R) {
        float mu[8], rs[8], be[8];
        ld8(mean + c, mu); ld8(rstd + c, rs); ld8(beta + c, be);
#pragma unroll
        for (int e = 0; e < 8; ++e) { float xh; z[e] = bn_z(y[u][e], mu[e], rs[e], be[e], relu, xh); }
        if (out) st8(out + r * ldo + c, z);
        if (out2) st8(out2 + r * ldo2 + c, z);
      }
      uint4 o[3];
      split8<FMT>(z, scale, o);
      store_slot<FMT>(dst, plane_stride, (int64_t)oc * rows_alloc + r, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward: sums + maxima
// Column-fixed threads (parameters in registers), >= 4 rows in flight, one workgroup per CU: the layout of bn.hip's
// bn1_bwd_kernel<false>, plus max |dz| and max |xhat| per column (what bounds |dT| before a single dT exists).
__global__ __launch_bounds__(256) void bn1_bwd_reduce_max_kernel(const float* __restrict__ T, int64_t R, int F, int FVB, int RP,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ beta, int relu,
                                                                 const float* __restrict__ dout, int64_t lddo,
                                                                 double* __restrict__ red, unsigned* __restrict__ maxbits) {
  extern __shared__ float lred[];   // [2][fw] sums, then [2][fw] maxima (as uint bits)
  const int t = threadIdx.x;
  const int fbase = blockIdx.y * 1024;
  const int fw = (F - fbase < 1024) ? (F - fbase) : 1024;
  unsigned* lmax = reinterpret_cast<unsigned*>(lred + 2 * fw);
  for (int e = t; e < 2 * fw; e += blockDim.x) { lred[e] = 0.f; lmax[e] = 0u; }
  __syncthreads();
  const int fq = blockIdx.y * 256 + (t % FVB);
  const int f = fq * 4;
  const bool on = (t < RP * FVB) && (f < F);
  const int64_t rstep = (int64_t)gridDim.x * RP;
  if (on) {
    float mu[4], rs[4], be[4], s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    *reinterpret_cast<float4*>(mu) = *reinterpret_cast<const float4*>(mean + f);
    *reinterpret_cast<float4*>(rs) = *reinterpret_cast<const float4*>(rstd + f);
    *reinterpret_cast<float4*>(be) = *reinterpret_cast<const float4*>(beta + f);
    for (int64_t r = (int64_t)blockIdx.x * RP + t / FVB; r < R; r += 4 * rstep) {
      float4 y[4], d[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int64_t rr = r + b * rstep;
        if (rr < R) { y[b] = *reinterpret_cast<const float4*>(T + rr * F + f); d[b] = *reinterpret_cast<const float4*>(dout + rr * lddo + f); }
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int64_t rr = r + b * rstep;
        if (rr < R) {
          const float yy[4] = {y[b].x, y[b].y, y[b].z, y[b].w}, dd[4] = {d[b].x, d[b].y, d[b].z, d[b].w};
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            float xh;
            const float z = bn_z(yy[v], mu[v], rs[v], be[v], relu, xh);
            float dz = dd[v];
            if (relu && !(z > 0.f)) dz = 0.f;
            s0[v] += dz; s1[v] += dz * xh;
            a0[v] = fmaxf(a0[v], fabsf(dz)); a1[v] = fmaxf(a1[v], fabsf(xh));
          }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      atomicAdd(&lred[f - fbase + v], s0[v]); atomicAdd(&lred[fw + f - fbase + v], s1[v]);
      atomicMax(&lmax[f - fbase + v], __float_as_uint(a0[v])); atomicMax(&lmax[fw + f - fbase + v], __float_as_uint(a1[v]));
    }
  }
  __syncthreads();
  const int slot = blockIdx.x % SLOTS;
  for (int e = t; e < 2 * fw; e += blockDim.x) {
    const int which = e / fw, c = fbase + (e % fw);
    atomicAdd(red + ((int64_t)slot * 2 + which) * F + c, (double)lred[e]);
    if (lmax[e]) atomicMax(maxbits + (int64_t)which * F + c, lmax[e]);
  }
}

// Reduce the slots into red[0..2F) (+ dbeta, as bn.hip:bn_bwd_finalize_kernel), bound |dT| per column
//   |dT_rc| = rstd_c |dz_rc - m1_c - xhat_rc m2_c| <= rstd_c (max_r |dz| + |m1_c| + max_r |xhat| |m2_c|),  m1 = mean dz, m2 = mean dz xhat
// and leave the largest bound (as float bits) in maxbits[2F]; the two column means replace the maxima (the apply pass reads
// them as floats).
__global__ __launch_bounds__(128) void bn1_bwd_finalize_bound_kernel(double* __restrict__ red, unsigned* maxbits, int F, double count,
                                                                     const float* __restrict__ rstd, float* __restrict__ dbeta,
                                                                     float dbeta_beta) {
  float* cf = reinterpret_cast<float*>(maxbits);
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  float bound = 0.f;
  if (f < F) {
    double s = 0.0, q = 0.0;
    for (int sl = 0; sl < SLOTS; ++sl) {
      s += red[((int64_t)sl * 2 + 0) * F + f];
      q += red[((int64_t)sl * 2 + 1) * F + f];
    }
    const float m1 = (float)(s / count), m2 = (float)(q / count);
    const float A = __uint_as_float(maxbits[f]), Bx = __uint_as_float(maxbits[F + f]);
    bound = rstd[f] * (A + fabsf(m1) + Bx * fabsf(m2));
    red[f] = s;                          // (slot 0 of this thread's own column: read above, no other thread touches it)
    red[F + f] = q;
    cf[f] = m1;
    cf[F + f] = m2;
    if (dbeta) dbeta[f] = (dbeta_beta != 0.f) ? (float)s + dbeta_beta * dbeta[f] : (float)s;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o));
  if ((threadIdx.x & 63) == 0 && bound > 0.f) atomicMax(maxbits + 2 * F, __float_as_uint(bound * 1.0001f));   // (bound evaluated in fp32)
}

// dT = rstd (dz - c1 - xhat c2) -> planes; optional fp32 dT; optional per-group column sums (tf.tile^T of FC0's per-cloud
// bias: rows_per_group % 64 == 0, so the 64 rows of a wave lie inside one group).  Same thread layout as the forward.
template <int FMT>
__global__ __launch_bounds__(256) void bn1_bwd_apply_planes_kernel(const float* __restrict__ T, int64_t R, int F,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   const float* __restrict__ beta, int relu,
                                                                   const float* __restrict__ dout, int64_t lddo,
                                                                   const float* __restrict__ cf, float* __restrict__ scale_out,
                                                                   char* __restrict__ dst, int64_t plane_stride, int64_t rows_alloc,
                                                                   float* __restrict__ dT, float* __restrict__ gsum, int64_t ldg,
                                                                   int rpg, int chunk) {
  const int lane = threadIdx.x & 63;
  const int oc0 = wave_octet0();
  const int noct = F >> 3;
  if (oc0 >= noct) return;
  // the tensor's power-of-two scale, from the largest column bound (every thread derives it; one publishes it for the GEMMs)
  const float sc = pow2_scale_for(__uint_as_float(reinterpret_cast<const unsigned*>(cf)[2 * F]));
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *scale_out = sc;
  const float scale = (FMT == DGCNN_PLANES_F16X2) ? sc : 1.f;
  // a block walks ONE contiguous range of `chunk` rows (a multiple of 64 that divides rows_per_group when sums per group are wanted)
  const int64_t rbeg = (int64_t)blockIdx.x * chunk;
  float gacc[OPT][8];
#pragma unroll
  for (int u = 0; u < OPT; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) gacc[u][e] = 0.f;
  for (int64_t r = rbeg + lane; r < rbeg + chunk && r < rows_alloc; r += 64) {
    float y[OPT][8], d[OPT][8];
    if (r < R) {
#pragma unroll
      for (int u = 0; u < OPT; ++u)
        if (oc0 + u < noct) { ld8(T + r * F + (oc0 + u) * 8, y[u]); ld8(dout + r * lddo + (oc0 + u) * 8, d[u]); }
    }
#pragma unroll
    for (int u = 0; u < OPT; ++u) {
      const int oc = oc0 + u;
      if (oc >= noct) break;
      const int c = oc * 8;
      float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (r < R) {
        float mu[8], rs[8], be[8], c1[8], c2[8];
        ld8(mean + c, mu); ld8(rstd + c, rs); ld8(beta + c, be); ld8(cf + c, c1); ld8(cf + F + c, c2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float xh;
          const float z = bn_z(y[u][e], mu[e], rs[e], be[e], relu, xh);
          float dz = d[u][e];
          if (relu && !(z > 0.f)) dz = 0.f;
          o8[e] = rs[e] * (dz - c1[e] - xh * c2[e]);
          gacc[u][e] += o8[e];
        }
        if (dT) st8(dT + r * F + c, o8);
      }
      uint4 o[3];
      split8<FMT>(o8, scale, o);
      store_slot<FMT>(dst, plane_stride, (int64_t)oc * rows_alloc + r, o);
    }
  }
  if (gsum && rbeg < R) {                           // sum over the wave's rows, one atomic per channel and block
#pragma unroll
    for (int u = 0; u < OPT; ++u) {
      if (oc0 + u >= noct) break;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = gacc[u][e];
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
        if (lane == 0) atomicAdd(gsum + (rbeg / rpg) * ldg + (oc0 + u) * 8 + e, v);
      }
    }
  }
}

inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// forward: blocks of 4 waves x (64 rows x OPT octets); ~4096 blocks, each walks down the rows with a grid stride
inline dim3 plane_grid(int64_t rows_alloc, int F) {
  const int64_t gy = dg::cdiv(F / 8, 4 * OPT);
  int64_t gx = rows_alloc / 64;
  const int64_t cap = dg::cdiv(4096, gy);
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return dim3((unsigned)gx, (unsigned)gy);
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int dgcnn_param_scales_f32(const float* params, int64_t n, double rows_max, float act_mul, float* scales,
                                      void* ws, void* stream) {
  DG_REQUIRE(params && scales && ws && n > 0 && rows_max >= 1.0 && act_mul > 0.f, DGCNN_EINVAL, "dgcnn_param_scales_f32: bad args");
  (void)dg::memset_async(ws, 0, 4, ST);
  const unsigned g = (unsigned)(dg::cdiv(n, 256 * 16) < 1024 ? dg::cdiv(n, 256 * 16) : 1024);
  dg::launch(absmax_flat_kernel, dim3(g ? g : 1), dim3(256), 0, ST, params, n, (unsigned*)ws);
  dg::launch(param_scales_kernel, dim3(1), dim3(1), 0, ST, (const unsigned*)ws, rows_max, act_mul, scales);
  return dg::check_launch("dgcnn_param_scales_f32");
}

extern "C" int dgcnn_bn_act_planes_f32(const float* T, int64_t ldt, int64_t R, int F, const float* mean, const float* rstd,
                                       const float* beta, int relu, int fmt, const float* scale_dev, void* planes,
                                       int64_t plane_stride, int64_t rows_alloc, float* out, int64_t ldo, float* out2,
                                       int64_t ldo2, void* stream) {
  DG_REQUIRE(T && mean && rstd && beta && planes && R > 0 && F > 0, DGCNN_EINVAL, "dgcnn_bn_act_planes_f32: bad args");
  DG_REQUIRE(fmt == DGCNN_PLANES_F16X2, DGCNN_EINVAL, "dgcnn_bn_act_planes_f32: unknown format %d", fmt);
  DG_REQUIRE(F % 8 == 0 && ldt % 4 == 0 && a16(T) && a16(mean) && a16(rstd) && a16(beta) && a16(planes) && plane_stride % 16 == 0 &&
                 rows_alloc % 64 == 0 && rows_alloc >= R && (!out || (a16(out) && ldo % 4 == 0)) && (!out2 || (a16(out2) && ldo2 % 4 == 0)),
             DGCNN_EINVAL, "dgcnn_bn_act_planes_f32: F %% 8, 16-byte aligned operands, rows_alloc %% 64 required");
  dim3 grid = plane_grid(rows_alloc, F);
  dg::launch((bn_act_planes_kernel<DGCNN_PLANES_F16X2>), grid, dim3(256), 0, ST, T, ldt, R, F, mean, rstd, beta, relu,
                       scale_dev, (char*)planes, plane_stride, rows_alloc, out, ldo, out2, ldo2);
  return dg::check_launch("dgcnn_bn_act_planes_f32");
}

// red: double[SLOTS][2][F] (zeroed), maxbits: uint32[2 F + 1] (zeroed)
extern "C" int dgcnn_bn1_bwd_reduce_max_f32(const float* T, int64_t R, int F, const float* mean, const float* rstd,
                                            const float* beta, int relu, const float* dout, int64_t lddo, double* red,
                                            void* maxbits, void* stream) {
  DG_REQUIRE(T && mean && rstd && beta && dout && red && maxbits && R > 0 && F > 0, DGCNN_EINVAL, "dgcnn_bn1_bwd_reduce_max_f32: bad args");
  DG_REQUIRE(F % 4 == 0 && lddo % 4 == 0 && a16(T) && a16(dout) && a16(mean) && a16(rstd) && a16(beta), DGCNN_EINVAL,
             "dgcnn_bn1_bwd_reduce_max_f32: float4-loadable operands required");
  const int FV = F / 4;
  const int FVB = FV < 256 ? FV : 256;
  const int RP = 256 / FVB;
  int64_t gx = dg::cdiv(R, (int64_t)RP * 4);
  if (gx > 256) gx = 256;
  const size_t shb = sizeof(float) * 4 * (size_t)(F < 1024 ? F : 1024);
  dg::launch(bn1_bwd_reduce_max_kernel, dim3((unsigned)gx, (unsigned)dg::cdiv(FV, 256)), dim3(256), shb, ST, T, R, F, FVB, RP,
                     mean, rstd, beta, relu, dout, lddo, red, (unsigned*)maxbits);
  return dg::check_launch("dgcnn_bn1_bwd_reduce_max_f32");
}

extern "C" int dgcnn_bn1_bwd_apply_planes_f32(const float* T, int64_t R, int F, const float* mean, const float* rstd,
                                              const float* beta, int relu, const float* dout, int64_t lddo, double* red,
                                              void* maxbits, int fmt, float* scale_dev, void* planes, int64_t plane_stride,
                                              int64_t rows_alloc, float* dT, float* gsum, int64_t ldgsum, int rows_per_group,
                                              float* dbeta, float dbeta_beta, void* stream) {
  DG_REQUIRE(T && mean && rstd && beta && dout && red && maxbits && planes && scale_dev && R > 0 && F > 0, DGCNN_EINVAL,
             "dgcnn_bn1_bwd_apply_planes_f32: bad args");
  DG_REQUIRE(fmt == DGCNN_PLANES_F16X2, DGCNN_EINVAL, "dgcnn_bn1_bwd_apply_planes_f32: unknown format %d", fmt);
  DG_REQUIRE(F % 8 == 0 && lddo % 4 == 0 && a16(T) && a16(dout) && a16(planes) && plane_stride % 16 == 0 && rows_alloc % 64 == 0 &&
                 rows_alloc >= R && (!dT || a16(dT)), DGCNN_EINVAL, "dgcnn_bn1_bwd_apply_planes_f32: F %% 8, aligned operands required");
  DG_REQUIRE(!gsum || (rows_per_group > 0 && rows_per_group % 64 == 0), DGCNN_EUNSUP,
             "dgcnn_bn1_bwd_apply_planes_f32: per-group sums need rows_per_group %% 64 == 0 (got %d)", rows_per_group);
  dg::launch(bn1_bwd_finalize_bound_kernel, dim3((unsigned)dg::cdiv(F, 128)), dim3(128), 0, ST, red, (unsigned*)maxbits, F,
                     (double)R, rstd, dbeta, dbeta_beta);
  const float* cf = reinterpret_cast<const float*>(maxbits);
  // rows per block: a multiple of 64, <= 512, dividing rows_per_group when group sums are wanted
  int chunk = 256;
  if (gsum) {
    chunk = 64;
    for (int cnd = 512; cnd >= 64; cnd -= 64)
      if (rows_per_group % cnd == 0) { chunk = cnd; break; }
  }
  dim3 grid((unsigned)dg::cdiv(rows_alloc, chunk), (unsigned)dg::cdiv(F / 8, 4 * OPT));
  dg::launch((bn1_bwd_apply_planes_kernel<DGCNN_PLANES_F16X2>), grid, dim3(256), 0, ST, T, R, F, mean, rstd, beta, relu, dout,
                       lddo, cf, scale_dev, (char*)planes, plane_stride, rows_alloc, dT, gsum, ldgsum, rows_per_group, chunk);
  return dg::check_launch("dgcnn_bn1_bwd_apply_planes_f32");
}
