// edge_mlp_bf16.hip -- conv0 of an EdgeConv layer (dgcnn/ops.py:21-52) as the LITERAL edge-level product with bf16 operands:
// BASELINE.json configs[2] "bf16 edge-MLP MFMA" (the reference's production shape scripts/lsf/train_dgcnn.sh:8-9 with the operand
// precision BASELINE names; the reference itself computes conv0 in float32 -- the mode is defined by oracle.edge_conv(...,
// edge_mlp_dtype="bf16")):
//
//     E[e] = [x_i, x_j - x_i]   formed in fp32 (the difference BEFORE any rounding), rounded to bf16 once (RNE)
//     y[e] = E[e] W0            W0 rounded to bf16 once; v_mfma_f32_32x32x16_bf16, fp32 accumulation over the 2C channels in order
//
// The (B N k, 2C) edge tensor and the (B N k, F) product are never written in the forward -- the tile is recomputed by every pass:
//     PASS 0  BatchNorm statistics: column sums of y and y^2 over all edges -> double[slots][2][F]
//     PASS 1  z = relu((y - mean) rstd + beta), max / mean / #ties over the k edges of every point -> (B N, F) each
//     PASS 2  y written out (B N k, F): the backward's input (bit-identical to what passes 0 / 1 saw: same instruction sequence)
// One workgroup = 4 waves = 128 edge rows; wave w gathers and rounds ITS 32 rows into LDS (row stride 2K + 16 bytes: conflict-free
// ds_read_b128 of the MFMA A fragments) and multiplies them with all F columns; the bf16 weight fragments of the whole layer stay
// in registers for the kernel's lifetime (K x F <= 128 x 128: 128 VGPRs); workgroups are persistent over the tiles.
// PASS 1 tiles hold floor(128 / k) whole points (the max / mean over k must see all edges of a point); the others are dense.
//
// Backward (edge_mlp_bf16_bwd_kernel), same tiles of whole points, y recomputed once more by the same instruction sequence:
//     dz   = [z == max_i] dmax_i / ties_i + dmean_i / k, zero where z <= 0;  dY = rstd (dz - c1 - xhat c2), rounded to bf16 (RNE)
//            -- the expressions of bn.hip:bn_bwd_apply_kernel, c1 / c2 from the reduce pass over the POINT outputs
//            (bn.hip:edge_bwd_reduce_points_kernel: PASS 1 packs #ties + 256 #positives exactly as the fp32 edge kernels do);
//     dW0 += E^T dY   on the matrix pipe: both operands are read from their row-major LDS tiles with ds_read_b64_tr_b16 (the
//            reduction runs over the 128 edge rows of the tile), accumulated in registers over all tiles of the workgroup,
//            one partial (2C, F) per workgroup, added up in fixed order by reduce_partials_kernel;
//     dYsum_i = sum_m dY (fp32 sum of the rounded values, m ascending) -> (B N, F);  dY itself leaves as bf16 (B N k, F) -- half
//            the bytes, the same values -- for the transposed-adjacency sum (dgcnn_edge_gather_sum_bf16).
// Neither E (2.7 GB per layer at configs[2]) nor y nor an fp32 dY is written; the separate 5.2-M-row weight-gradient GEMM is gone.
#include "common.h"
#include <math.h>

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int RT = 128;                  // edge rows per tile

struct EdgeP {
  const float* x; int64_t ldx; const int32_t* idx; const float* W0;
  int B, N, C, k, F;
  float* Y;                              // PASS 2
  double* stats; int nslots;             // PASS 0
  const float* mean; const float* rstd; const float* beta;      // PASS 1
  float* mx; int64_t ldmx; float* mn; int64_t ldmn; float* cnt;
  int pack;                              // PASS 1: cnt = #ties + CNT_POS * #(z > 0) (what the fused backward reads) instead of #ties
};
constexpr int CNT_POS = 256;             // (bn.hip)

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// same expressions as bn.hip:bn_z (the library is built with -ffp-contract=off): the backward recomputes z from the materialised
// y and compares it with the maximum taken here
__device__ __forceinline__ float bn_z1(float y, float mu, float rs, float be) { return fmaxf((y - mu) * rs + be, 0.f); }

// Gather + round one tile of E into LDS (row stride S = 2 K + 16 bytes): wave w stages rows 32 w .. 32 w + 31.  POINTS: the tile
// holds P = floor(128 / k) whole points (rows past P k and points past the end are zero rows), else 128 consecutive edge rows.
template <int CK, bool POINTS>
__device__ __forceinline__ void gather_tile(const EdgeP& p, char* Es, int64_t tile, int P, int64_t R, int64_t Me, int t, int w,
                                            int l31, int lh) {
  constexpr int K = 16 * CK;
  constexpr int S = 2 * K + 16;
  const int C = p.C, k = p.k;
  const int r = (CK == 1) ? (32 * w + l31) : (t >> 1);            // (t >> 1 lies in [32 w, 32 w + 32))
  int64_t e, gp;
  bool valid;
  if (POINTS) {
    const int pi = r / k;
    gp = tile * P + pi;
    valid = pi < P && gp < R;
    e = gp * k + (r - pi * k);
  } else {
    e = tile * RT + r;
    valid = e < Me;
    gp = e / k;
  }
  char* dst = Es + r * S;
  if (CK == 1) {
    if (lh == 0) {
      float xi[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        const int64_t nb = (gp / p.N) * p.N + p.idx[e];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < C) {
            xi[c] = p.x[gp * p.ldx + c];
            d[c] = p.x[nb * p.ldx + c] - xi[c];
          }
      }
      const u32x4 a = {pk_bf16(xi[0], xi[1]), pk_bf16(xi[2], xi[3]), 0u, 0u};
      const u32x4 b = {pk_bf16(d[0], d[1]), pk_bf16(d[2], d[3]), 0u, 0u};
      *reinterpret_cast<u32x4*>(dst) = a;
      *reinterpret_cast<u32x4*>(dst + 16) = b;
    }
  } else {
    const int h = t & 1;                                            // channels 32 h .. 32 h + 31 of x_i and of x_j - x_i
    float4 xi[8], xj[8];
    if (valid) {
      const int64_t nb = (gp / p.N) * p.N + p.idx[e];
      const float4* pi4 = reinterpret_cast<const float4*>(p.x + gp * p.ldx + 32 * h);
      const float4* pj4 = reinterpret_cast<const float4*>(p.x + nb * p.ldx + 32 * h);
#pragma unroll
      for (int q = 0; q < 8; ++q) { xi[q] = pi4[q]; xj[q] = pj4[q]; }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) xi[q] = xj[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                   // 8 channels per 16-byte store
      const float4 a0 = xi[2 * q], a1 = xi[2 * q + 1], b0 = xj[2 * q], b1 = xj[2 * q + 1];
      const u32x4 ci = {pk_bf16(a0.x, a0.y), pk_bf16(a0.z, a0.w), pk_bf16(a1.x, a1.y), pk_bf16(a1.z, a1.w)};
      const u32x4 di = {pk_bf16(b0.x - a0.x, b0.y - a0.y), pk_bf16(b0.z - a0.z, b0.w - a0.w),
                        pk_bf16(b1.x - a1.x, b1.y - a1.y), pk_bf16(b1.z - a1.z, b1.w - a1.w)};
      *reinterpret_cast<u32x4*>(dst + 64 * h + 16 * q) = ci;
      *reinterpret_cast<u32x4*>(dst + 128 + 64 * h + 16 * q) = di;
    }
  }
}

// CK = K / 16: 1 (raw coordinates, C <= 4: E = [x_i, 0.. | x_j - x_i, 0..], 8 + 8 channels) or 8 (C = 64).  FB = F / 32.
template <int PASS, int CK, int FB>
__global__ __launch_bounds__(256) void edge_mlp_bf16_kernel(EdgeP p) {
  constexpr int K = 16 * CK;
  constexpr int S = 2 * K + 16;          // bytes per LDS row of E
  constexpr int FP = 32 * FB + 1;        // floats per LDS row of y (PASS 1)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Es = smem;
  float* Yt = reinterpret_cast<float*>(smem + RT * S);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, lh = lane >> 5;
  const int C = p.C, k = p.k, F = 32 * FB;
  const int64_t R = (int64_t)p.B * p.N;
  const int64_t Me = R * k;
  const int P = RT / k;                  // whole points per PASS-1 tile
  const int64_t ntiles = (PASS == 1) ? (R + P - 1) / P : (Me + RT - 1) / RT;

  // ---- the layer's weights as bf16 MFMA B fragments: wf[s][j] = W0[16 s + 8 lh .. + 7][32 j + l31] ----
  bf16x8 wf[CK][FB];
#pragma unroll
  for (int s = 0; s < CK; ++s)
#pragma unroll
    for (int j = 0; j < FB; ++j) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int kk = 16 * s + 8 * lh + q;
        int row;                            // row of W0 (2C x F) this channel of E multiplies, or -1 (pad channel)
        if (CK == 1) row = (kk < 8) ? (kk < C ? kk : -1) : (kk - 8 < C ? C + kk - 8 : -1);
        else row = kk;
        v[q] = row >= 0 ? p.W0[(int64_t)row * F + 32 * j + l31] : 0.f;
      }
      const u32x4 pk = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
      wf[s][j] = __builtin_bit_cast(bf16x8, pk);
    }

  // BatchNorm sums of this lane's columns: fp32 within a tile (16 values), DOUBLE across the tiles of the persistent workgroup
  // (5.2 M edge rows at configs[2] are ~640 tiles per lane, ~2500 with the capped grid of the deterministic mode: an fp32 running
  // sum there loses digits that var = E[y^2] - mean^2 then amplifies)
  double cs[FB], cq[FB];
#pragma unroll
  for (int j = 0; j < FB; ++j) cs[j] = cq[j] = 0.0;

#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    gather_tile<CK, PASS == 1>(p, Es, tile, P, R, Me, t, w, l31, lh);   // wave w stages rows 32 w .. 32 w + 31 of the tile
    __syncthreads();
    // ---- y tile of this wave: 32 rows x F ----
    f32x16 acc[FB];
#pragma unroll
    for (int j = 0; j < FB; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
    for (int s = 0; s < CK; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Es + (32 * w + l31) * S + 32 * s + 16 * lh);
#pragma unroll
      for (int j = 0; j < FB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[s][j], acc[j], 0, 0, 0);
    }
    // C/D layout: acc[j][q] = y[row 32 w + (q & 3) + 8 (q >> 2) + 4 lh][col 32 j + l31]
    if (PASS == 0) {
      // rows past the end of the edge list are zero rows of E: y = 0 exactly, they add nothing
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        float ts = 0.f, tq = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          ts += acc[j][q];
          tq += acc[j][q] * acc[j][q];
        }
        cs[j] += (double)ts;
        cq[j] += (double)tq;
      }
      __syncthreads();                                                  // (E rows are rewritten by the next tile)
    } else if (PASS == 2) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t e = tile * RT + 32 * w + (q & 3) + 8 * (q >> 2) + 4 * lh;
        if (e < Me) {
#pragma unroll
          for (int j = 0; j < FB; ++j) p.Y[e * F + 32 * j + l31] = acc[j][q];
        }
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < FB; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) Yt[(32 * w + (q & 3) + 8 * (q >> 2) + 4 * lh) * FP + 32 * j + l31] = acc[j][q];
      __syncthreads();
      const float invk = 1.0f / (float)k;
      for (int it = t; it < P * F; it += 256) {
        const int pi = it / F, c = it - pi * F;
        const int64_t gp = tile * P + pi;
        if (gp < R) {
          const float mu = p.mean[c], rs = p.rstd[c], be = p.beta[c];
          float mx = -INFINITY, sm = 0.f, cn = 0.f, np = 0.f;
          for (int m = 0; m < k; ++m) {
            const float z = bn_z1(Yt[(pi * k + m) * FP + c], mu, rs, be);
            const bool gt = z > mx;
            cn = gt ? 1.f : ((z == mx) ? cn + 1.f : cn);                // ties share the max gradient
            mx = gt ? z : mx;
            sm += z;
            np += (z > 0.f) ? 1.f : 0.f;
          }
          p.mx[gp * p.ldmx + c] = mx;
          p.mn[gp * p.ldmn + c] = sm * invk;
          if (p.cnt) p.cnt[gp * F + c] = p.pack ? cn + (float)CNT_POS * np : cn;
        }
      }
      __syncthreads();
    }
  }

  if (PASS == 0) {
    // column sums: lanes l and l + 32 hold the same columns (other rows); then the four waves through LDS, one writer per column
    double* red = reinterpret_cast<double*>(smem);                      // [4 waves][2][F]
#pragma unroll
    for (int j = 0; j < FB; ++j) {
      cs[j] += __shfl_xor(cs[j], 32);
      cq[j] += __shfl_xor(cq[j], 32);
    }
    __syncthreads();
    if (lh == 0) {
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        red[(w * 2 + 0) * F + 32 * j + l31] = cs[j];
        red[(w * 2 + 1) * F + 32 * j + l31] = cq[j];
      }
    }
    __syncthreads();
    const int slot = blockIdx.x % p.nslots;
    for (int i = t; i < 2 * F; i += 256) {
      const double v = red[i] + red[2 * F + i] + red[4 * F + i] + red[6 * F + i];
      atomicAdd(p.stats + (int64_t)slot * 2 * F + i, v);
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Backward of the fused layer (see the file header).
struct EdgeBwdP {
  EdgeP e;                               // x, idx, W0, shape, mean / rstd / beta, mx (forward max), cnt (packed ties / positives)
  const float* dmx; int64_t lddmx; const float* dmn; int64_t lddmn;
  const double* red;                     // [2][F]: sum dz, sum dz xhat (slot 0 after bn_bwd_finalize)
  uint16_t* dYb;                         // (B N k, F) bf16, or null
  float* dysum; int64_t lddysum;         // (B N, F), or null
  float* partial;                        // [gridDim.x][2C][F]
};

__device__ __forceinline__ bf16x8 tr_read2(const char* a0, const char* a1) {
  using s16x4 = __attribute__((ext_vector_type(4))) short;
  using s16x8 = __attribute__((ext_vector_type(8))) short;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// 512 threads: waves 0-3 compute (wave w owns rows 32 w .. 32 w + 31 of the tile), waves 4-7 load: while the compute waves work on
// tile T out of LDS buffer T & 1, the loaders gather tile T + 1 (neighbour index -> two 256-byte rows per edge, an L2 round trip
// each) and the per-point vectors, round them and fill the other buffer.  Without the split the kernel was a chain of exposed round
// trips (1.63 ms per layer at configs[2] against 0.50 ms for the forward's BatchNorm pass over the same tiles).
constexpr int VMAX = 8;                    // per-point vector items per loader thread: P F <= 16 x 128 = 8 x 256
template <int CK, int FB>
__global__ __launch_bounds__(512) void edge_mlp_bf16_bwd_kernel(EdgeBwdP bp) {
  const EdgeP& p = bp.e;
  constexpr int K = 16 * CK;
  constexpr int S = 2 * K + 16;          // bytes per LDS row of E
  constexpr int F = 32 * FB;
  constexpr int SD = 2 * F + 16;         // bytes per LDS row of dY (bf16)
  constexpr int MT = (K + 31) / 32;      // 32-channel row tiles of dW0: wave w < MT owns tile w
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l31 = lane & 31, lh = lane >> 5;
  const int C = p.C, k = p.k;
  const int64_t R = (int64_t)p.B * p.N;
  const int64_t Me = R * k;
  const int P = RT / k;
  const int64_t ntiles = (R + P - 1) / P;
  const float invk = 1.0f / (float)k;
  // LDS: [E buffer 0][E buffer 1][dY tile][vectors 0: pm | pg1 | pg2][vectors 1]
  char* Ds = smem + 2 * RT * S;
  auto Eb = [&](int b) { return smem + b * (RT * S); };
  auto vb = [&](int b) { return reinterpret_cast<float4*>(Ds + RT * SD) + b * (P * F); };    // (max, dmax / ties, dmean / k, -) per (point, column)

  if (w >= 4) {
    // ------------------------------------------------------------------------------------------------ loaders
    const int u = t - 256;
    // one row (CK = 1: threads 0..127) or half a row (32 channels of x_i and of x_j) per thread, in flight between its global loads
    // and its LDS image; same arithmetic as gather_tile<CK, true>
    constexpr int NQ = CK == 1 ? 1 : 8;
    const int gr = CK == 1 ? u : (u >> 1), gh = CK == 1 ? 0 : (u & 1);
    const bool gactive = CK == 1 ? (u < RT) : true;
    const int gpi = gr / k, gm = gr - gpi * k;
    float4 xi[NQ], xj[NQ];
    auto issue_row = [&](int64_t tile) {
      const int64_t gp = tile * P + gpi;
      const bool valid = gactive && gpi < P && gp < R;
#pragma unroll
      for (int q = 0; q < NQ; ++q) xi[q] = xj[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        const int64_t nb = (gp / p.N) * p.N + p.idx[gp * k + gm];
        if (CK == 1) {
          float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c < C) {
              a[c] = p.x[gp * p.ldx + c];
              b[c] = p.x[nb * p.ldx + c];
            }
          xi[0] = make_float4(a[0], a[1], a[2], a[3]);
          xj[0] = make_float4(b[0], b[1], b[2], b[3]);
        } else {
          const float4* pi4 = reinterpret_cast<const float4*>(p.x + gp * p.ldx + 32 * gh);
          const float4* pj4 = reinterpret_cast<const float4*>(p.x + nb * p.ldx + 32 * gh);
#pragma unroll
          for (int q = 0; q < NQ; ++q) { xi[q] = pi4[q]; xj[q] = pj4[q]; }
        }
      }
    };
    auto commit_row = [&](char* Es) {
      if (!gactive) return;
      char* dst = Es + gr * S;
      if (CK == 1) {
        const float4 a = xi[0], b = xj[0];                               // (channels >= C are zeros on both sides)
        const u32x4 ci = {pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), 0u, 0u};
        const u32x4 di = {pk_bf16(b.x - a.x, b.y - a.y), pk_bf16(b.z - a.z, b.w - a.w), 0u, 0u};
        *reinterpret_cast<u32x4*>(dst) = ci;
        *reinterpret_cast<u32x4*>(dst + 16) = di;
      } else {
#pragma unroll
        for (int q = 0; q < NQ / 2; ++q) {
          const float4 a0 = xi[2 * q], a1 = xi[2 * q + 1], b0 = xj[2 * q], b1 = xj[2 * q + 1];
          const u32x4 ci = {pk_bf16(a0.x, a0.y), pk_bf16(a0.z, a0.w), pk_bf16(a1.x, a1.y), pk_bf16(a1.z, a1.w)};
          const u32x4 di = {pk_bf16(b0.x - a0.x, b0.y - a0.y), pk_bf16(b0.z - a0.z, b0.w - a0.w),
                            pk_bf16(b1.x - a1.x, b1.y - a1.y), pk_bf16(b1.z - a1.z, b1.w - a1.w)};
          *reinterpret_cast<u32x4*>(dst + 64 * gh + 16 * q) = ci;
          *reinterpret_cast<u32x4*>(dst + 128 + 64 * gh + 16 * q) = di;
        }
      }
    };
    float vm[VMAX], vc[VMAX], vdx[VMAX], vdn[VMAX];
    auto issue_vec = [&](int64_t tile) {
#pragma unroll
      for (int i = 0; i < VMAX; ++i) {
        const int it = u + 256 * i;
        vm[i] = 0.f; vc[i] = 1.f; vdx[i] = 0.f; vdn[i] = 0.f;
        if (it < P * F) {
          const int pi = it / F, c = it - pi * F;
          const int64_t gp = tile * P + pi;
          if (gp < R) {
            vm[i] = p.mx[gp * p.ldmx + c];
            vc[i] = p.cnt[gp * F + c];
            vdx[i] = bp.dmx[gp * bp.lddmx + c];
            vdn[i] = bp.dmn[gp * bp.lddmn + c];
          }
        }
      }
    };
    auto commit_vec = [&](int64_t tile, float4* v) {
#pragma unroll
      for (int i = 0; i < VMAX; ++i) {
        const int it = u + 256 * i;
        if (it < P * F) {
          const int pi = it / F;
          const bool ok = tile * P + pi < R;
          float cn = vc[i];
          cn -= (float)CNT_POS * floorf(cn * (1.0f / CNT_POS));               // #ties of the max
          v[it] = ok ? make_float4(vm[i], vdx[i] / cn, vdn[i] * invk, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) {
      issue_row(tile);
      issue_vec(tile);
      commit_row(Eb(0));
      commit_vec(tile, vb(0));
    }
    __syncthreads();                                                      // buffer 0 is ready
    int it = 0;
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x, ++it) {
      const int64_t next = tile + gridDim.x;
      const bool more = next < ntiles;
      if (more) {
        issue_row(next);
        issue_vec(next);
      }
      __syncthreads();                                                    // (compute: dY tile written)
      if (more) {
        commit_row(Eb((it + 1) & 1));
        commit_vec(next, vb((it + 1) & 1));
      }
      // ---- dYsum of the tile's points (m ascending, as bn_bwd_apply_kernel adds them) ----
      if (bp.dysum) {
        for (int i2 = u; i2 < P * F; i2 += 256) {
          const int pi = i2 / F, c = i2 - pi * F;
          const int64_t gp = tile * P + pi;
          if (gp < R) {
            float a = 0.f;
            for (int m = 0; m < k; ++m)
              a += __uint_as_float((unsigned)*reinterpret_cast<const unsigned short*>(Ds + (pi * k + m) * SD + 2 * c) << 16);
            bp.dysum[gp * bp.lddysum + c] = a;
          }
        }
      }
      // ---- dY rows of the tile: P k consecutive edges, 16 bytes per store ----
      if (bp.dYb) {
        const int64_t e0 = tile * P * k;
        const int64_t left = Me - e0;
        const int nr = left < (int64_t)P * k ? (int)left : P * k;
        constexpr int CH = F / 8;
        for (int ch = u; ch < nr * CH; ch += 256) {
          const int row = ch / CH, cc = ch - row * CH;
          *reinterpret_cast<u32x4*>(bp.dYb + (e0 + row) * F + 8 * cc) = *reinterpret_cast<const u32x4*>(Ds + row * SD + 16 * cc);
        }
      }
      __syncthreads();                                                    // (compute: done with buffer it & 1)
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- compute
  bf16x8 wf[CK][FB];
#pragma unroll
  for (int s = 0; s < CK; ++s)
#pragma unroll
    for (int j = 0; j < FB; ++j) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int kk = 16 * s + 8 * lh + q;
        int row;
        if (CK == 1) row = (kk < 8) ? (kk < C ? kk : -1) : (kk - 8 < C ? C + kk - 8 : -1);
        else row = kk;
        v[q] = row >= 0 ? p.W0[(int64_t)row * F + 32 * j + l31] : 0.f;
      }
      const u32x4 pk = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
      wf[s][j] = __builtin_bit_cast(bf16x8, pk);
    }
  // per-column constants of this lane's FB columns (bn.hip: bn_bwd_apply_kernel)
  float mu[FB], rs[FB], be[FB], c1[FB], c2[FB];
  const double inv_cnt = 1.0 / ((double)R * (double)k);
#pragma unroll
  for (int j = 0; j < FB; ++j) {
    const int c = 32 * j + l31;
    mu[j] = p.mean[c]; rs[j] = p.rstd[c]; be[j] = p.beta[c];
    c1[j] = (float)(bp.red[c] * inv_cnt);
    c2[j] = (float)(bp.red[F + c] * inv_cnt);
  }
  int prow[16];                            // point (within the tile) of each of this lane's 16 accumulator rows
#pragma unroll
  for (int q = 0; q < 16; ++q) prow[q] = (32 * w + (q & 3) + 8 * (q >> 2) + 4 * lh) / k;

  f32x16 dacc[FB];
#pragma unroll
  for (int j = 0; j < FB; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) dacc[j][q] = 0.f;
  // transposed-read addressing (ds_read_b64_tr_b16: within a 16-lane group lane 4 jj + qq supplies the 8-byte piece
  // (row jj, channels 4 qq .. 4 qq + 3) and lane c receives channel c of rows 0..3)
  const int g = lane >> 4, jj = (lane >> 2) & 3, qq = lane & 3;
  const int trow = 8 * (g >> 1) + jj;      // + 16 ks + 4 tt
  const int tch = 16 * (g & 1) + 4 * qq;   // + 32 (tile of channels / columns)

  __syncthreads();                                                        // buffer 0 is ready
  int it = 0;
#pragma unroll 1
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
    const char* Es = Eb(it & 1);
    const float4* pv = vb(it & 1);
    // ---- y tile of this wave (the forward's instruction sequence) ----
    f32x16 acc[FB];
#pragma unroll
    for (int j = 0; j < FB; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
    for (int s = 0; s < CK; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Es + (32 * w + l31) * S + 32 * s + 16 * lh);
#pragma unroll
      for (int j = 0; j < FB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[s][j], acc[j], 0, 0, 0);
    }
    // ---- dY of the wave's 32 rows -> LDS as bf16 (v_cvt_pk_bf16_f32: round to nearest even, the dense kernels' integer form) ----
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = 32 * w + (q & 3) + 8 * (q >> 2) + 4 * lh;
      const int pi = prow[q];
      const bool valid = pi < P && tile * P + pi < R;
      float o[FB];
#pragma unroll
      for (int j = 0; j < FB; ++j) {
        const int c = 32 * j + l31;
        o[j] = 0.f;
        if (valid) {
          const float y = acc[j][q];
          const float xh = (y - mu[j]) * rs[j];
          const float z = fmaxf(xh + be[j], 0.f);
          const float4 v4 = pv[pi * F + c];
          float dz = ((z == v4.x) ? v4.y : 0.f) + v4.z;
          if (!(z > 0.f)) dz = 0.f;
          o[j] = rs[j] * (dz - c1[j] - xh * c2[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < FB; j += 2) {
        const unsigned pk = pk_bf16(o[j], j + 1 < FB ? o[j + 1] : 0.f);
        *reinterpret_cast<unsigned short*>(Ds + row * SD + 2 * (32 * j + l31)) = (unsigned short)(pk & 0xffffu);
        if (j + 1 < FB) *reinterpret_cast<unsigned short*>(Ds + row * SD + 2 * (32 * (j + 1) + l31)) = (unsigned short)(pk >> 16);
      }
    }
    __syncthreads();
    // ---- dW0 += E^T dY: reduction over the tile's 128 rows, both operands by transposed LDS reads ----
    if (w < MT) {
#pragma unroll
      for (int ks = 0; ks < RT / 16; ++ks) {
        const char* ea = Es + (16 * ks + trow) * S + 2 * (32 * w + tch);
        const bf16x8 a = tr_read2(ea, ea + 4 * S);
#pragma unroll
        for (int j = 0; j < FB; ++j) {
          const char* da = Ds + (16 * ks + trow) * SD + 2 * (32 * j + tch);
          const bf16x8 b = tr_read2(da, da + 4 * SD);
          dacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, dacc[j], 0, 0, 0);
        }
      }
    }
    // (the per-point sums of dY and the dY rows themselves leave through the loader waves, which idle here)
    __syncthreads();
  }

  // ---- this workgroup's partial dW0: acc row m = E channel 32 w + (q & 3) + 8 (q >> 2) + 4 lh, column 32 j + l31 ----
  float* out = bp.partial + (int64_t)blockIdx.x * 2 * C * F;
  if (w < MT) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int kk = 32 * w + (q & 3) + 8 * (q >> 2) + 4 * lh;
      int row;
      if (CK == 1) row = (kk < 8) ? (kk < C ? kk : -1) : ((kk < 16 && kk - 8 < C) ? C + kk - 8 : -1);
      else row = kk;
      if (row >= 0) {
#pragma unroll
        for (int j = 0; j < FB; ++j) out[(int64_t)row * F + 32 * j + l31] = dacc[j][q];
      }
    }
  }
}

template <int PASS>
int launch_pass(const EdgeP& p, hipStream_t st, const char* what) {
  const int CK = p.C <= 4 ? 1 : 8;
  const int FB = p.F / 32;
  const int K = 16 * CK;
  const int64_t R = (int64_t)p.B * p.N;
  const int64_t ntiles = (PASS == 1) ? dg::cdiv(R, (int64_t)(RT / p.k)) : dg::cdiv(R * p.k, (int64_t)RT);
  size_t sh = (size_t)RT * (2 * K + 16);
  if (PASS == 1) sh += (size_t)RT * (p.F + 1) * sizeof(float);
  if (PASS == 0 && sh < (size_t)8 * p.F * sizeof(double)) sh = (size_t)8 * p.F * sizeof(double);
  int64_t g = ntiles < 1024 ? ntiles : 1024;
  if (PASS == 0) g = dg::cap_writers(g);                               // (reproducible configuration: one writer per slot)
  if (g < 1) g = 1;
#define DG_E(CKV, FBV)                                                                                                        \
  do {                                                                                                                        \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_mlp_bf16_kernel<PASS, CKV, FBV>),                           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                        \
    dg::launch((edge_mlp_bf16_kernel<PASS, CKV, FBV>), dim3((unsigned)g), dim3(256), sh, st, p);                      \
  } while (0)
  if (CK == 1) {
    if (FB == 1) DG_E(1, 1); else if (FB == 2) DG_E(1, 2); else DG_E(1, 4);
  } else {
    if (FB == 1) DG_E(8, 1); else if (FB == 2) DG_E(8, 2); else DG_E(8, 4);
  }
#undef DG_E
  return dg::check_launch(what);
}

bool shape_ok(int C, int k, int F) { return (C <= 4 || C == 64) && (F == 32 || F == 64 || F == 128) && k <= RT; }
// LDS of the backward: two E buffers, the dY tile, two buffers of (P, F) float4 vectors for the tile's P = floor(128 / k) points
size_t bwd_lds_bytes(int C, int k, int F) {
  const int K = 16 * (C <= 4 ? 1 : 8), P = RT / k;
  return (size_t)2 * RT * (2 * K + 16) + (size_t)RT * (2 * F + 16) + (size_t)2 * 4 * P * F * sizeof(float);
}
// k >= 8 bounds P at 16; the widest layers (C = 64, F = 128) need k >= 10 to fit 160 KB
bool bwd_shape_ok(int C, int k, int F) { return shape_ok(C, k, F) && k >= 8 && k < CNT_POS && bwd_lds_bytes(C, k, F) <= 160 * 1024; }

}  // namespace

// 1 when the fused kernels take this layer shape (C <= 4 or C == 64; F in {32, 64, 128}; k <= 128), else 0
extern "C" int dgcnn_edge_mlp_bf16_supported(int C, int k, int F) { return shape_ok(C, k, F) ? 1 : 0; }
// 1 when dgcnn_edge_mlp_bf16_bwd takes this layer shape (the forward's shapes with 8 <= k < 256)
extern "C" int dgcnn_edge_mlp_bf16_bwd_supported(int C, int k, int F) { return bwd_shape_ok(C, k, F) ? 1 : 0; }

#define DG_EDGE_COMMON(name)                                                                                                  \
  DG_REQUIRE(x && idx && W0 && B > 0 && N > 0 && C > 0 && k > 0 && F > 0, DGCNN_EINVAL, name ": bad args");                    \
  DG_REQUIRE(shape_ok(C, k, F), DGCNN_EUNSUP, name ": needs C <= 4 or C == 64, F in {32, 64, 128}, k <= 128 (C=%d k=%d F=%d)", C, k, F); \
  DG_REQUIRE((int64_t)B * N * k < (1ll << 31), DGCNN_EUNSUP, name ": B*N*k >= 2^31");                                          \
  DG_REQUIRE(C <= 4 || (ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0), DGCNN_EINVAL, name ": x must be float4-loadable"); \
  EdgeP p = {};                                                                                                                \
  p.x = x; p.ldx = ldx; p.idx = idx; p.W0 = W0; p.B = B; p.N = N; p.C = C; p.k = k; p.F = F

// y = E W0 written out (B N k, F): the literal bf16 edge MLP (dgcnn_edge_mlp_f32 | bf16)
extern "C" int dgcnn_edge_mlp_bf16(const float* x, int64_t ldx, const int32_t* idx, const float* W0, int B, int N, int C, int k,
                                   int F, float* Y, void* stream) {
  DG_EDGE_COMMON("dgcnn_edge_mlp_bf16");
  DG_REQUIRE(Y, DGCNN_EINVAL, "dgcnn_edge_mlp_bf16: null output");
  p.Y = Y;
  return launch_pass<2>(p, (hipStream_t)stream, "dgcnn_edge_mlp_bf16");
}

// BatchNorm statistics of y without writing it: stats[slot][0][f] += sum_e y, stats[slot][1][f] += sum_e y^2
extern "C" int dgcnn_edge_mlp_bf16_stats(const float* x, int64_t ldx, const int32_t* idx, const float* W0, int B, int N, int C,
                                         int k, int F, double* stats, void* stream) {
  DG_EDGE_COMMON("dgcnn_edge_mlp_bf16_stats");
  DG_REQUIRE(stats, DGCNN_EINVAL, "dgcnn_edge_mlp_bf16_stats: null output");
  p.stats = stats;
  p.nslots = dg::stat_slots();
  return launch_pass<0>(p, (hipStream_t)stream, "dgcnn_edge_mlp_bf16_stats");
}

// relu(BatchNorm(y)) reduced over the k edges of every point, y recomputed: max -> mx, mean -> mn, #ties of the max -> cnt
extern "C" int dgcnn_edge_mlp_bf16_bn_kreduce(const float* x, int64_t ldx, const int32_t* idx, const float* W0, int B, int N,
                                              int C, int k, int F, const float* mean, const float* rstd, const float* beta,
                                              float* mx, int64_t ldmx, float* mn, int64_t ldmn, float* cnt, int pack_cnt,
                                              void* stream) {
  DG_EDGE_COMMON("dgcnn_edge_mlp_bf16_bn_kreduce");
  DG_REQUIRE(mean && rstd && beta && mx && mn && ldmx >= F && ldmn >= F, DGCNN_EINVAL, "dgcnn_edge_mlp_bf16_bn_kreduce: bad args");
  DG_REQUIRE(!pack_cnt || k < CNT_POS, DGCNN_EUNSUP, "dgcnn_edge_mlp_bf16_bn_kreduce: packed counts need k < %d", CNT_POS);
  p.mean = mean; p.rstd = rstd; p.beta = beta; p.mx = mx; p.ldmx = ldmx; p.mn = mn; p.ldmn = ldmn; p.cnt = cnt; p.pack = pack_cnt ? 1 : 0;
  return launch_pass<1>(p, (hipStream_t)stream, "dgcnn_edge_mlp_bf16_bn_kreduce");
}

namespace dg {
void launch_reduce_partials(const float* part, int splits, int M, int N, float* C, int64_t ldc, float beta, hipStream_t st);
void launch_bn_bwd_finalize(double* red, int F, float* dbeta, float dbeta_beta, hipStream_t st);
}

// Backward of the fused layer in one pass over the edges (file header).  `red` = the slots written by
// dgcnn_edge_bn_bwd_reduce_points_f32 from the forward's per-point outputs (cnt packed: pack_cnt = 1 in the forward); they are
// reduced here (d(beta) = dbeta_beta * dbeta + sum dz).  dW0 (2C, F) is ACCUMULATED.  dYb (B N k, F) bf16 and dysum (B N, F) may be
// null (no input gradient wanted).  ws: at least dgcnn_edge_mlp_bf16_bwd_workspace_bytes(B, N, C, k, F) bytes.
extern "C" int64_t dgcnn_edge_mlp_bf16_bwd_workspace_bytes(int B, int N, int C, int k, int F) {
  if (B <= 0 || N <= 0 || C <= 0 || k <= 0 || F <= 0 || k > RT) return 0;
  const int64_t ntiles = dg::cdiv((int64_t)B * N, (int64_t)(RT / k));
  const int64_t g = ntiles < 512 ? ntiles : 512;
  return g * 2 * C * F * (int64_t)sizeof(float);
}

extern "C" int dgcnn_edge_mlp_bf16_bwd(const float* x, int64_t ldx, const int32_t* idx, const float* W0, int B, int N, int C, int k,
                                       int F, const float* mean, const float* rstd, const float* beta, const float* mx,
                                       int64_t ldmx, const float* cnt, const float* dmx, int64_t lddmx, const float* dmn,
                                       int64_t lddmn, double* red, void* dYb, float* dysum, int64_t lddysum, float* dW0,
                                       float* dbeta, float dbeta_beta, void* ws, size_t ws_bytes, void* stream) {
  DG_EDGE_COMMON("dgcnn_edge_mlp_bf16_bwd");
  DG_REQUIRE(bwd_shape_ok(C, k, F), DGCNN_EUNSUP, "dgcnn_edge_mlp_bf16_bwd: needs 8 <= k < %d and %zu bytes of LDS <= 160 KB (k=%d)", CNT_POS, bwd_lds_bytes(C, k, F), k);
  DG_REQUIRE(mean && rstd && beta && mx && cnt && dmx && dmn && red && dW0 && ws && ldmx >= F && lddmx >= F && lddmn >= F, DGCNN_EINVAL,
             "dgcnn_edge_mlp_bf16_bwd: bad args");
  DG_REQUIRE(!dysum || lddysum >= F, DGCNN_EINVAL, "dgcnn_edge_mlp_bf16_bwd: lddysum < F");
  DG_REQUIRE(!dYb || (reinterpret_cast<uintptr_t>(dYb) & 15) == 0, DGCNN_EINVAL, "dgcnn_edge_mlp_bf16_bwd: dYb must be 16-byte aligned");
  const int64_t need = dgcnn_edge_mlp_bf16_bwd_workspace_bytes(B, N, C, k, F);
  DG_REQUIRE((int64_t)ws_bytes >= need, DGCNN_ENOSPC, "dgcnn_edge_mlp_bf16_bwd: workspace too small (%zu < %lld)", ws_bytes, (long long)need);
  hipStream_t st = (hipStream_t)stream;
  dg::launch_bn_bwd_finalize(red, F, dbeta, dbeta_beta, st);
  EdgeBwdP bp = {};
  bp.e = p;
  bp.e.mean = mean; bp.e.rstd = rstd; bp.e.beta = beta; bp.e.mx = const_cast<float*>(mx); bp.e.ldmx = ldmx; bp.e.cnt = const_cast<float*>(cnt);
  bp.dmx = dmx; bp.lddmx = lddmx; bp.dmn = dmn; bp.lddmn = lddmn; bp.red = red;
  bp.dYb = reinterpret_cast<uint16_t*>(dYb); bp.dysum = dysum; bp.lddysum = lddysum; bp.partial = reinterpret_cast<float*>(ws);
  const int CK = C <= 4 ? 1 : 8, FB = F / 32, P = RT / k;
  const int64_t ntiles = dg::cdiv((int64_t)B * N, (int64_t)P);
  const int64_t g = ntiles < 512 ? ntiles : 512;
  const size_t sh = bwd_lds_bytes(C, k, F);
#define DG_B(CKV, FBV)                                                                                                        \
  do {                                                                                                                        \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&edge_mlp_bf16_bwd_kernel<CKV, FBV>),                             \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                        \
    dg::launch((edge_mlp_bf16_bwd_kernel<CKV, FBV>), dim3((unsigned)g), dim3(512), sh, st, bp);                       \
  } while (0)
  if (CK == 1) {
    if (FB == 1) DG_B(1, 1); else if (FB == 2) DG_B(1, 2); else DG_B(1, 4);
  } else {
    if (FB == 1) DG_B(8, 1); else if (FB == 2) DG_B(8, 2); else DG_B(8, 4);
  }
#undef DG_B
  int rc = dg::check_launch("dgcnn_edge_mlp_bf16_bwd");
  if (rc) return rc;
  dg::launch_reduce_partials(reinterpret_cast<const float*>(ws), (int)g, 2 * C, F, dW0, (int64_t)F, 1.0f, st);
  return dg::check_launch("dgcnn_edge_mlp_bf16_bwd(reduce)");
}
