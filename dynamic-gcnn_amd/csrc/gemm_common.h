// gemm_common.h -- what the GEMM kernel families share: the launch descriptor, the operand-source /
// epilogue kinds and the accumulator epilogue (32x32 MFMA C/D layout is the same for every input type).
#pragma once
#include "common.h"
#include <stdlib.h>


namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

enum { A_ROW = 0, A_COL = 1, A_EDGE = 2, A_EDGE_T = 3 };
enum { B_ROW = 0, B_COL = 1 };
enum { E_STORE = 0, E_SCATTER = 1 };

constexpr int BK = 16;
constexpr int NT = 256;

struct GemmP {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int M, N, K;
  float beta;
  const float* gbias; int64_t ldgbias; int rpg;
  double* stats;
  // edge sources / scatter
  const float* x; int64_t ldx; const int32_t* idx; int npts; int cch; int knn;
  float* dx; int64_t lddx;
  // split-K
  int splits; int kchunk; float* partial;
  int avec, bvec;
  int mtiles, ntiles, xcd_group, bm, cvec;
  int edge_nbr;   // A_EDGE / A_EDGE_T rows are the raw neighbour features x_j (K or M = C) instead of [x_i, x_j - x_i]
  int gbvec;      // per-group bias rows are float4-loadable
  int zmajor;     // split-K launches of the bf16-split kernels: 1-D grid, XCD x (= block id % 8) owns the k-chunks z = x (mod 8)
  // per-group column maximum of the output (model.py:76-77 max-pool over the points of a cloud, taken in the epilogue of the GEMM
  // that produces the tensor): keys[group][N] <- atomicMax(order-preserving bits of the value << 32 | ~row-in-group), i.e. the
  // largest value and, among ties, the FIRST row.  Needs rows_per_group % tile rows == 0 (checked by the host).
  unsigned long long* colmax; int colmax_rpg;
  int stat_slots; // slots of p.stats (dg::stat_slots() at launch)
};

// order-preserving map float -> uint32 (larger float <=> larger unsigned), and back
__device__ __forceinline__ unsigned f32_ordered(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_ordered(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// Block -> (mt, nt, z).  MI355X hands workgroup b to XCD b % 8 (each XCD has a private 4 MB L2).
//   zmajor    (split-K: few output tiles, long reduction) every k-chunk z is pinned to ONE XCD: all mtiles x ntiles tiles
//             of the chunk run there side by side (<= 32 CUs), march through k in step and share each A / B slab
//             through that L2 -- every operand byte leaves HBM once.  (With z on the grid's z axis the tiles of a chunk
//             landed on all 8 XCDs: the A panel was fetched once per column tile, the B panel once per row tile:
//             3.2x the algorithmic bytes on FC0's weight gradient, profiles/r01e_pmc_hbm_bytes.txt.)
//   xcd_group (many row panels, several column tiles) the column tiles of one A row panel share an XCD;
//   otherwise row-major over the tiles, z from the grid.
__device__ __forceinline__ bool block_tile(const GemmP& p, int& mt, int& nt, int& z) {
  const int id = blockIdx.x;
  if (p.zmajor) {
    const int tiles = p.mtiles * p.ntiles;
    const int j = id >> 3;
    z = (id & 7) + 8 * (j / tiles);
    const int tl = j % tiles;
    mt = tl / p.ntiles;
    nt = tl % p.ntiles;
    return z < p.splits;
  }
  z = blockIdx.z;
  if (p.xcd_group) {
    mt = ((id >> 3) / p.ntiles) * 8 + (id & 7);
    nt = (id >> 3) % p.ntiles;
  } else {
    mt = id / p.ntiles;
    nt = id % p.ntiles;
  }
  return mt < p.mtiles;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
#define LD4(ptr) ld4(ptr)

__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// ---- shared epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
// WR = wave rows of the block (each wave owns BM/WR rows as TM 32-row tiles), NTH = threads taking part.
// WCN = wave columns (each wave owns BN/WCN columns as TN 32-column tiles).
template <int EPI, int BM, int BN, bool VEC, int TM, int TN, int WR = 2, int NTH = NT, int WCN = 2>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x16 (&acc)[TM][TN], float* smem, int m0, int n0,
                                              int mt, int z, int t, int wr, int wc, int l31, int lh) {
  const int colw = n0 + wc * (BN / WCN) + l31;
  const int roww = m0 + wr * (BM / WR) + 4 * lh;
  if (EPI == E_SCATTER) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = roww + i * 32 + (r & 3) + 8 * (r >> 2);
        if (row < p.M) {
          const int nb = (row / (p.knn * p.npts)) * p.npts + p.idx[row];
          float* d = p.dx + (int64_t)nb * p.lddx;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = colw + j * 32;
            if (col < p.N) atomicAdd(d + col, acc[i][j][r]);
          }
        }
      }
    return;
  }

  // ---- E_STORE: accumulators -> LDS (64 block rows at a time) -> coalesced float4 row stores.
  // Keeps the epilogue at one global_store_dwordx4 per 4 outputs, lets the read-modify-write
  // (beta) and the per-cloud bias be float4 loads, and needs no per-row pointer registers.
  float* tile = smem;                       // [WR*32][BN]
  constexpr int QV = BN / 4;                // float4 per row
  constexpr int RSTEP = NTH / QV;           // rows covered per pass of the NTH threads
  const int c4 = (t % QV) * 4;
  const int rr0 = t / QV;
  const int gcol = n0 + c4;
  const bool col_ok = gcol < p.N;
  const bool has_beta = (p.beta != 0.f);
  const bool has_gb = (p.gbias != nullptr);
  const bool split = (p.splits > 1);
  float* outp = split ? (p.partial + (int64_t)z * p.M * p.N) : p.C;
  const int64_t ldo = split ? (int64_t)p.N : p.ldc;
  const bool vec_st = VEC && p.cvec && (gcol + 3 < p.N);
  const int rlast = imin(m0 + BM, p.M) - 1;
  const bool gb_uniform = has_gb && ((m0 / p.rpg) == (rlast / p.rpg));
  float gbu[4] = {0.f, 0.f, 0.f, 0.f};
  if (gb_uniform && col_ok) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (gcol + q < p.N) gbu[q] = p.gbias[(int64_t)(m0 / p.rpg) * p.ldgbias + gcol + q];
  }
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  float cmx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int cmr[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  const bool want_max = (p.colmax != nullptr) && !split;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tile[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wc * (BN / WCN) + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int q0 = 0; q0 < (WR * 32) / RSTEP; ++q0) {
      const int rl = rr0 + q0 * RSTEP;
      const int grow = m0 + (rl >> 5) * (BM / WR) + i * 32 + (rl & 31);
      if (grow < p.M && col_ok) {
        const float4 tv = *reinterpret_cast<const float4*>(&tile[rl * BN + c4]);
        float v[4] = {tv.x, tv.y, tv.z, tv.w};
        float* dst = outp + (int64_t)grow * ldo + gcol;
        if (!split) {
          if (has_gb) {
            if (gb_uniform) {
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] += gbu[q];
            } else {
              const float* gb = p.gbias + (int64_t)(grow / p.rpg) * p.ldgbias + gcol;
              if (p.gbvec && gcol + 3 < p.N) {
                const float4 g4 = *reinterpret_cast<const float4*>(gb);
                v[0] += g4.x; v[1] += g4.y; v[2] += g4.z; v[3] += g4.w;
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  if (gcol + q < p.N) v[q] += gb[q];
              }
            }
          }
          if (has_beta) {
            if (vec_st) {
              const float4 o = *reinterpret_cast<const float4*>(dst);
              v[0] += p.beta * o.x; v[1] += p.beta * o.y; v[2] += p.beta * o.z; v[3] += p.beta * o.w;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (gcol + q < p.N) v[q] += p.beta * dst[q];
            }
          }
        }
        if (vec_st) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int q = 0; q < 4; ++q) { cs[q] += v[q]; cq[q] += v[q] * v[q]; }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (gcol + q < p.N) { dst[q] = v[q]; cs[q] += v[q]; cq[q] += v[q] * v[q]; }
        }
        if (want_max) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (v[q] > cmx[q] || (v[q] == cmx[q] && grow < cmr[q])) { cmx[q] = v[q]; cmr[q] = grow; }
        }
      }
    }
  }
  if ((p.stats || want_max) && !split) {
    // Column reductions of the tile.  The NTH / QV threads that share a column quad park their partial results as float4
    // slots [thread row l][kind][quad] (conflict-free ds_write_b128), one thread per column then walks down the l's (round 2
    // used 8 LDS float atomics per thread, 16 threads deep on every address).
    __syncthreads();
    constexpr int L = NTH / QV;
    float4* park = reinterpret_cast<float4*>(smem);          // [L][4 kinds][QV]
    const int l = t / QV, qd = t % QV;
    park[(l * 4 + 0) * QV + qd] = make_float4(cs[0], cs[1], cs[2], cs[3]);
    park[(l * 4 + 1) * QV + qd] = make_float4(cq[0], cq[1], cq[2], cq[3]);
    if (want_max) {
      park[(l * 4 + 2) * QV + qd] = make_float4(cmx[0], cmx[1], cmx[2], cmx[3]);
      park[(l * 4 + 3) * QV + qd] = make_float4(__int_as_float(cmr[0]), __int_as_float(cmr[1]), __int_as_float(cmr[2]), __int_as_float(cmr[3]));
    }
    __syncthreads();
    const float* pk = smem;
    if (t < BN) {
      const int c = n0 + t;
      float s0 = 0.f, s1 = 0.f, mx = -INFINITY;
      int mr = 0x7fffffff;
#pragma unroll 4
      for (int ll = 0; ll < L; ++ll) {
        s0 += pk[((ll * 4 + 0) * QV) * 4 + t];
        s1 += pk[((ll * 4 + 1) * QV) * 4 + t];
        if (want_max) {
          const float v = pk[((ll * 4 + 2) * QV) * 4 + t];
          const int r = __float_as_int(pk[((ll * 4 + 3) * QV) * 4 + t]);
          if (v > mx || (v == mx && r < mr)) { mx = v; mr = r; }
        }
      }
      if (c < p.N) {
        if (p.stats) {
          const int slot = mt % p.stat_slots;
          atomicAdd(p.stats + ((int64_t)slot * 2 + 0) * p.N + c, (double)s0);
          atomicAdd(p.stats + ((int64_t)slot * 2 + 1) * p.N + c, (double)s1);
        }
        if (want_max && mr != 0x7fffffff) {
          const int grp = m0 / p.colmax_rpg;                  // the tile lies inside one group (host check)
          const unsigned long long key = ((unsigned long long)f32_ordered(mx) << 32) |
                                         (unsigned long long)(0xffffffffu - (unsigned)(mr - grp * p.colmax_rpg));
          atomicMax(p.colmax + (int64_t)grp * p.N + c, key);
        }
      }
    }
  }
}

}  // namespace
