// gemm_x3.hip -- fp32 GEMM on the bf16 matrix pipe by exact operand splitting (gfx950).
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (157 TFLOP/s peak, 1/16 of the bf16 MFMA rate).
// Every fp32 value is EXACTLY the sum of three bf16 values:
//     a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)      (8 + 8 + 8 significand bits)
// so a*b = sum_ij a_i*b_j, each partial product exact in fp32, accumulated in the fp32 MFMA accumulator:
//     NP = 9  all nine partial products: nothing is dropped, only the fp32 accumulation rounds
//             (as it does in the native kernel)                               -> 9/16 of the native MFMA time
//     NP = 6  drops a2*b3, a3*b2, a3*b3 (<= 2^-23 |a*b| together, the size of ONE fp32 rounding of the
//             product -- what an unfused multiply-add chain commits anyway)    -> 6/16 of the native MFMA time
//     NP = 1  only a1*b1: plain bf16 OPERANDS (round to nearest even), fp32 accumulate; NOT fp32-class (2^-9 relative per
//             operand).  Kept for measurements (dgcnn_gemm_set_arith(1)): as the "bf16 edge-MLP" of BASELINE configs[2] it was
//             measured neither accurate (the folded conv0 rounds x_i and x_j separately: logits 0.6 off) nor faster, and the
//             model no longer offers it                                -> 1/16 of the native MFMA time
// Inputs, outputs and accumulators stay fp32; tests/test_gpu_parity.py::test_gemm_split_accuracy measures
// the error of both against an fp64 product next to the native fp32-MFMA kernel (profiles/r01_gemm_arith.txt).
// Inf/NaN inputs produce NaN (Inf - Inf in the split); the path's activations are finite.
//
// Kernel: block tile 128 x BN (BN in {64,128}) x 32, 256 threads = 2x2 waves, each wave 64 x BN/2 as
// 32x32x16 bf16 MFMA tiles.  Waves 0-1 fetch the A slab, waves 2-3 the B slab (8 global_load_dwordx4
// each, two slabs ahead, in registers); under the MFMAs of the current slab the 32 values per thread fetched
// for the next one are split into packed bf16 registers, and between the two barriers of a k-step only the
// 12 ds_write_b128 into the three bf16 planes remain (LDS image: struct Img).  A lane's MFMA operand (8
// consecutive k of one row) is one ds_read_b128 per plane.  Epilogue shared with the native kernel
// (gemm_common.h).
#include "gemm_common.h"
#include <type_traits>


#ifndef DGCNN_GEMM_ARITH_DEFAULT
#define DGCNN_GEMM_ARITH_DEFAULT 6
#endif

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int XK = 32;            // k per slab
enum { KCONTIG = 0, KSTRIDED = 1 };

// LDS image of one bf16 plane of a TILE x 32 operand slab: [4 chunks of 8 k][row'][16 B], where the tile
// row r sits at row' = (r & 3) * (TILE/4 + 4) + (r >> 2).  With that permutation
//   * an MFMA operand read (ds_read_b128, lanes = 32 consecutive tile rows, serviced in the hardware's four
//     16-lane groups) touches 16 distinct 16-byte slots,
//   * the transposing store of a k-strided source (a lane owns tile rows 4q..4q+3; 8 consecutive lanes store
//     rows 4q+e) and the store of a k-contiguous source (8 lanes = 2 rows x 4 chunks; chunk stride = 80 mod
//     128 bytes) are conflict-free too (ds_write_b128: 8-lane groups, 32 banks).
template <int TILE>
struct Img {
  static constexpr int RS = TILE / 4 + 4;
  static constexpr unsigned CS = (TILE == 256 ? 269u : (TILE == 128 ? 141u : 77u)) * 16u;    // chunk stride, = 80 (mod 128)
  static constexpr unsigned PB = 4u * CS;                              // bytes per plane
  static __device__ __forceinline__ unsigned at(int r, int chunk) {
    return (unsigned)chunk * CS + (unsigned)(((r & 3) * RS + (r >> 2)) << 4);
  }
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// (x0, x1) -> packed bf16 pairs of the three split terms (x0 in the low half)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16);
  const float r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16);
  const float s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(s0, s1);
}

struct Packed { uint4 h, m, l; };   // one 8-k chunk of one row, three planes

__device__ __forceinline__ Packed split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
  Packed o;
  split_pair(v0, v1, o.h.x, o.m.x, o.l.x);
  split_pair(v2, v3, o.h.y, o.m.y, o.l.y);
  split_pair(v4, v5, o.h.z, o.m.z, o.l.z);
  split_pair(v6, v7, o.h.w, o.m.w, o.l.w);
  return o;
}

__device__ __forceinline__ float4 keep(bool c, float4 v) { return c ? v : make_float4(0.f, 0.f, 0.f, 0.f); }

// One operand slab (TILE rows x 32 k) staged by 128 threads (u = 0..127), 4 chunks (of 8 k, one row) each:
//   KCONTIG  (k contiguous in memory: A[m][k], B stored [n][k]): chunk c = u&3 of rows (u>>2) + 32 j
//   KSTRIDED (row index contiguous:   A stored [k][m], B[k][n]):  chunk c of rows 4q..4q+3; 8 loads over k
// MASK = false: the block is interior and every slab is full -- no clamps, no predicates.
template <int KIND, int TILE>
struct Stage {
  static constexpr int NC = (KIND == KCONTIG) ? TILE / 32 : 4;    // chunks per thread
  const float* p[4];
  int64_t ld;
  unsigned rok;
  int c, r0;
  bool on;

  __device__ __forceinline__ void init(const float* base, int64_t ld_, int row0, int nrows, int u, int kbeg) {
    ld = ld_;
    rok = 0;
    if (KIND == KCONTIG) {
      on = true;
      c = u & 3;
      r0 = u >> 2;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const int row = row0 + r0 + 32 * j;
        if (row < nrows) rok |= 1u << j;
        p[j] = base + (int64_t)imin(row, nrows - 1) * ld + (kbeg + 8 * c);
      }
    } else {
      constexpr int QN = TILE / 4;
      on = u < QN * 4;
      c = (u / QN) & 3;
      r0 = 4 * (u % QN);
      const int row = row0 + r0;
      if (row < nrows) rok = 1u;
      p[0] = base + imin(row, nrows - 4) + (int64_t)(kbeg + 8 * c) * ld;
    }
  }

  // raw slab starting at k = kbeg + koff (koff a multiple of 32)
  template <bool MASK>
  __device__ __forceinline__ void load(float4 (&L)[8], unsigned& pm, int koff, int klen) const {
    if (KIND == KCONTIG) {
#pragma unroll
      for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int kk = koff + 8 * c + 4 * h;           // relative to kbeg
          if (MASK) {
            if (((rok >> j) & 1u) && kk < klen) pm |= 1u << (2 * j + h);
            L[2 * j + h] = ld4(p[j] + (imin(kk, klen - 4) - 8 * c));
          } else {
            L[2 * j + h] = ld4(p[j] + (koff + 4 * h));
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kk = koff + 8 * c + i;
        if (MASK) {
          if (rok && kk < klen) pm |= 1u << i;
          L[i] = ld4(p[0] + (int64_t)(imin(kk, klen - 1) - 8 * c) * ld);
        } else {
          L[i] = ld4(p[0] + (int64_t)(koff + i) * ld);
        }
      }
    }
  }

  template <bool MASK>
  __device__ __forceinline__ void split(const float4 (&L)[8], unsigned pm, Packed (&P)[4]) const {
    if (KIND == KCONTIG) {
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const float4 a = MASK ? keep((pm >> (2 * j)) & 1u, L[2 * j]) : L[2 * j];
        const float4 b = MASK ? keep((pm >> (2 * j + 1)) & 1u, L[2 * j + 1]) : L[2 * j + 1];
        P[j] = split8(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
      }
    } else {
      float4 Z[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) Z[i] = MASK ? keep((pm >> i) & 1u, L[i]) : L[i];
      P[0] = split8(Z[0].x, Z[1].x, Z[2].x, Z[3].x, Z[4].x, Z[5].x, Z[6].x, Z[7].x);
      P[1] = split8(Z[0].y, Z[1].y, Z[2].y, Z[3].y, Z[4].y, Z[5].y, Z[6].y, Z[7].y);
      P[2] = split8(Z[0].z, Z[1].z, Z[2].z, Z[3].z, Z[4].z, Z[5].z, Z[6].z, Z[7].z);
      P[3] = split8(Z[0].w, Z[1].w, Z[2].w, Z[3].w, Z[4].w, Z[5].w, Z[6].w, Z[7].w);
    }
  }

  template <int NPL = 3>      // planes written: 3, or only the leading bf16 term (single-product mode)
  __device__ __forceinline__ void write(const Packed (&P)[4], char* plane0) const {
    using I = Img<TILE>;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const unsigned off = (KIND == KCONTIG) ? I::at(r0 + 32 * j, c) : I::at(r0 + j, c);
      *reinterpret_cast<uint4*>(plane0 + off) = P[j].h;
      if (NPL == 3) {
        *reinterpret_cast<uint4*>(plane0 + I::PB + off) = P[j].m;
        *reinterpret_cast<uint4*>(plane0 + 2 * I::PB + off) = P[j].l;
      }
    }
  }
};

template <int AKIND, int BKIND, int BN, int NP, int BM = 128>
__global__ __launch_bounds__(NT, (BM == 64 ? 3 : 2)) void gemm_x3_kernel(GemmP p) {
  constexpr int TM = BM / 64;
  constexpr int TN = BN / 64;
  using IA = Img<BM>;
  using IB = Img<BN>;
  constexpr unsigned LDS_BYTES = 3 * IA::PB + 3 * IB::PB;
  constexpr unsigned EPI_BYTES = 64 * BN * 4;
  __shared__ __attribute__((aligned(16))) char smem_raw[LDS_BYTES > EPI_BYTES ? LDS_BYTES : EPI_BYTES];
  char* As = smem_raw;
  char* Bs = smem_raw + 3 * IA::PB;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = t >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  int mt, nt, z;
  if (!block_tile(p, mt, nt, z)) return;
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  const int kbeg = z * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? (kbeg + p.kchunk) : p.K;
  const int klen = kend - kbeg;
  const int nk = (klen + XK - 1) / XK;

  // waves 0-1 stage A, waves 2-3 stage B (wave-uniform role)
  const bool roleA = t < 128;
  const int u = t & 127;
  Stage<AKIND, BM> sa;
  Stage<BKIND, BN> sb;
  sa.init(p.A, p.lda, m0, p.M, u, kbeg);
  sb.init(p.B, p.ldb, n0, p.N, u, kbeg);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // operand addresses of this lane: tile row (wave base + 32 i + l31), chunk = 2*kstep + lh
  unsigned a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = IA::at(wr * (BM / 2) + i * 32 + l31, lh);
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = IB::at(wc * (BN / 2) + j * 32 + l31, lh);

  auto mfma_slab = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a[TM][3], b[TN][3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i][q] = *reinterpret_cast<const bf16x8*>(As + q * IA::PB + a_off[i] + 2 * s * IA::CS);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j][q] = *reinterpret_cast<const bf16x8*>(Bs + q * IB::PB + b_off[j] + 2 * s * IB::CS);
      }
      // partial products, largest first: (1,1) (1,2) (2,1) (2,2) (1,3) (3,1) [(2,3) (3,2) (3,3)]
      constexpr int PA[9] = {0, 0, 1, 1, 0, 2, 1, 2, 2};
      constexpr int PB[9] = {0, 1, 0, 1, 2, 0, 2, 1, 2};
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
    }
  };

  // Software pipeline, two slabs deep: while the MFMAs of slab kt read LDS, the raw registers L (slab kt+1,
  // fetched during iteration kt-1) are split into the packed registers P, then L is refilled with slab kt+2;
  // between the two barriers only the 12 ds_write_b128 of P remain.
  auto run = [&](auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    float4 L[8];
    Packed P[4];
    unsigned pm = 0;
    auto fetch = [&](int slab) {
      pm = 0;
      if (roleA) sa.template load<MASK>(L, pm, slab * XK, klen);
      else if (sb.on) sb.template load<MASK>(L, pm, slab * XK, klen);
    };
    auto do_split = [&]() {
      if (roleA) sa.template split<MASK>(L, pm, P);
      else if (sb.on) sb.template split<MASK>(L, pm, P);
    };
    auto do_write = [&]() {
      if (roleA) sa.template write<(NP == 1 ? 1 : 3)>(P, As);
      else if (sb.on) sb.template write<(NP == 1 ? 1 : 3)>(P, Bs);
    };
    fetch(0);
    do_split();
    do_write();
    fetch(imin(1, nk - 1));
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      do_split();                               // slab kt+1 (garbage-but-valid after the last one; not written)
      fetch(imin(kt + 2, nk - 1));
      mfma_slab();
      __syncthreads();
      if (kt + 1 < nk) do_write();
      __syncthreads();
    }
  };
  const bool edge = (m0 + BM > p.M) || (n0 + BN > p.N) || (klen % XK != 0);
  if (edge) run(std::true_type{});
  else run(std::false_type{});

  gemm_epilogue<E_STORE, BM, BN, true, TM, TN>(p, acc, reinterpret_cast<float*>(smem_raw), m0, n0, mt, z, t, wr, wc, l31, lh);
}

// ------------------------------------------------------------------------------------------------------
// 256 x 128 tile, 768 threads: waves 0-7 consumers (4 x 2, 64 x 64 each), waves 8-11 producers.  Three waves
// per SIMD (two consumers + one producer; <= 168 VGPRs).  The staging work per MFMA is 3/4 of the 128 x 128
// tile's (a 32-k slab of A and B is 384 rows for 256 x 128 outputs instead of 256 rows for 128 x 128), and it
// is the staging work next to the MFMAs that costs (profiles/r01_gemm_x3_ablation.txt): consumers alone reach the
// power-limited MFMA rate, and a wave-specialised 128 x 128 variant was no faster than gemm_x3_kernel (removed).
//   every producer thread stages 4 chunks of A (256 rows) and 2 chunks of B (128 rows) per slab,
//   ring of 2 raw slabs in registers, split + write chunk by chunk (12 live registers).
template <int KIND, int TILE>          // staged by 256 threads
struct Stage2 {
  static constexpr int NC = TILE / 64;                                  // chunks per thread: 4 (A) or 2 (B)
  static constexpr int VW = NC;                                         // KSTRIDED: rows per thread
  static constexpr int NL = (KIND == KCONTIG) ? 2 * NC : 8;             // loads per slab
  using Vt = typename std::conditional<(KIND == KCONTIG || VW == 4), float4, float2>::type;
  const float* p[NC];
  int64_t ld;
  unsigned rok;
  int c, r0;

  __device__ __forceinline__ void init(const float* base, int64_t ld_, int row0, int nrows, int u, int kbeg) {
    ld = ld_;
    rok = 0;
    if constexpr (KIND == KCONTIG) {
      c = u & 3;
      r0 = u >> 2;                                                      // 0..63
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const int row = row0 + r0 + 64 * j;
        if (row < nrows) rok |= 1u << j;
        p[j] = base + (int64_t)imin(row, nrows - 1) * ld + (kbeg + 8 * c);
      }
    } else {
      constexpr int QN = TILE / VW;                                     // 64 row groups
      c = (u / QN) & 3;
      r0 = VW * (u % QN);
      const int row = row0 + r0;
      if (row < nrows) rok = 1u;
      p[0] = base + imin(row, nrows - VW) + (int64_t)(kbeg + 8 * c) * ld;
    }
  }

  template <bool MASK>
  __device__ __forceinline__ void load(Vt (&L)[NL], unsigned& pm, int koff, int klen) const {
    pm = 0;
    if constexpr (KIND == KCONTIG) {
#pragma unroll
      for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int kk = koff + 8 * c + 4 * h;
          if (MASK) {
            if (((rok >> j) & 1u) && kk < klen) pm |= 1u << (2 * j + h);
            L[2 * j + h] = ld4(p[j] + (imin(kk, klen - 4) - 8 * c));
          } else {
            L[2 * j + h] = ld4(p[j] + (koff + 4 * h));
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kk = koff + 8 * c + i;
        const float* src = MASK ? (p[0] + (int64_t)(imin(kk, klen - 1) - 8 * c) * ld) : (p[0] + (int64_t)(koff + i) * ld);
        if (MASK && rok && kk < klen) pm |= 1u << i;
        L[i] = *reinterpret_cast<const Vt*>(src);
      }
    }
  }

  static __device__ __forceinline__ float comp(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }
  static __device__ __forceinline__ float comp(const float2& v, int e) { return e == 0 ? v.x : v.y; }

  // split + write, one chunk at a time
  template <bool MASK, int NPL = 3>
  __device__ __forceinline__ void stage(const Vt (&L)[NL], unsigned pm, char* plane0) const {
    using I = Img<TILE>;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      float v[8];
      unsigned off;
      if constexpr (KIND == KCONTIG) {
        const bool k0 = !MASK || ((pm >> (2 * j)) & 1u), k1 = !MASK || ((pm >> (2 * j + 1)) & 1u);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = k0 ? comp(L[2 * j], e) : 0.f;
          v[4 + e] = k1 ? comp(L[2 * j + 1], e) : 0.f;
        }
        off = I::at(r0 + 64 * j, c);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (!MASK || ((pm >> i) & 1u)) ? comp(L[i], j) : 0.f;
        off = I::at(r0 + j, c);
      }
      const Packed P = split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      *reinterpret_cast<uint4*>(plane0 + off) = P.h;
      if (NPL == 3) {
        *reinterpret_cast<uint4*>(plane0 + I::PB + off) = P.m;
        *reinterpret_cast<uint4*>(plane0 + 2 * I::PB + off) = P.l;
      }
    }
  }
};

template <int AKIND, int BKIND, int NP>
__global__ __launch_bounds__(768) void gemm_x3w2_kernel(GemmP p) {
  constexpr int BM = 256, BN = 128;
  constexpr int TM = 2, TN = 2;
  using IA = Img<BM>;
  using IB = Img<BN>;
  constexpr unsigned BUF = 3 * IA::PB + 3 * IB::PB;
  constexpr unsigned EPI_BYTES = 128 * BN * 4;
  static_assert(2 * BUF >= EPI_BYTES && 2 * BUF <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem_raw[2 * BUF];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = t >> 6;

  int mt, nt, z;
  if (!block_tile(p, mt, nt, z)) return;
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  const int kbeg = z * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? (kbeg + p.kchunk) : p.K;
  const int klen = kend - kbeg;
  const int nk = (klen + XK - 1) / XK;

  if (wv >= 8) {
    // ------------------------------------------------------------------ producers (256 threads)
    const int u = t - 512;
    using SA = Stage2<AKIND, BM>;
    using SB = Stage2<BKIND, BN>;
    SA sa;
    SB sb;
    sa.init(p.A, p.lda, m0, p.M, u, kbeg);
    sb.init(p.B, p.ldb, n0, p.N, u, kbeg);
    auto run = [&](auto mask_tag) {
      constexpr bool MASK = decltype(mask_tag)::value;
      typename SA::Vt LA[2][SA::NL];
      typename SB::Vt LB[2][SB::NL];
      unsigned pa[2], pb[2];
      auto fetch = [&](int r, int slab) {
        sa.template load<MASK>(LA[r], pa[r], slab * XK, klen);
        sb.template load<MASK>(LB[r], pb[r], slab * XK, klen);
      };
      auto stage = [&](int r, int buf) {
        char* base = smem_raw + buf * BUF;
        sa.template stage<MASK, (NP == 1 ? 1 : 3)>(LA[r], pa[r], base);
        sb.template stage<MASK, (NP == 1 ? 1 : 3)>(LB[r], pb[r], base + 3 * IA::PB);
      };
      fetch(0, 0);
      fetch(1, imin(1, nk - 1));
      stage(0, 0);
      fetch(0, imin(2, nk - 1));
      __syncthreads();
      // iteration kt stages slab kt+1 (ring slot and LDS buffer (kt+1)&1), then refills the slot with slab kt+3
#pragma unroll 1
      for (int kt0 = 0; kt0 < nk; kt0 += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int kt = kt0 + d;
          if (kt < nk) {
            if (kt + 1 < nk) {
              stage((d + 1) & 1, (d + 1) & 1);
              fetch((d + 1) & 1, imin(kt + 3, nk - 1));
            }
            __syncthreads();
          }
        }
      }
    };
    const bool edge = (m0 + BM > p.M) || (n0 + BN > p.N) || (klen % XK != 0);
    if (edge) run(std::true_type{});
    else run(std::false_type{});
    return;
  }

  // -------------------------------------------------------------------- consumers (512 threads, 4 x 2 waves)
  const int wr = wv >> 1, wc = wv & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  unsigned a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = IA::at(wr * 64 + i * 32 + l31, lh);
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = 3 * IA::PB + IB::at(wc * 64 + j * 32 + l31, lh);

  __syncthreads();                                 // buffer 0 is ready
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    const char* base = smem_raw + (kt & 1) * BUF;
    bf16x8 a[2][TM][3], b[2][TN][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[s][i][q] = *reinterpret_cast<const bf16x8*>(base + q * IA::PB + a_off[i] + 2 * s * IA::CS);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[s][j][q] = *reinterpret_cast<const bf16x8*>(base + q * IB::PB + b_off[j] + 2 * s * IB::CS);
      }
    constexpr int PA[9] = {0, 0, 1, 1, 0, 2, 1, 2, 2};
    constexpr int PB[9] = {0, 1, 0, 1, 2, 0, 2, 1, 2};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i][PA[q]], b[s][j][PB[q]], acc[i][j], 0, 0, 0);
    __syncthreads();
  }

  gemm_epilogue<E_STORE, BM, BN, true, TM, TN, 4, 512>(p, acc, reinterpret_cast<float*>(smem_raw), m0, n0, mt, z, t, wr, wc, l31, lh);
}

// ------------------------------------------------------------------------------------------------------
// 256 x 256 tile, 512 threads, UNspecialised: waves 4 x 2, 64 x 128 outputs each (128 accumulator registers), every wave fetches,
// splits and multiplies.  Per flop it stages 2/3 of the 256 x 128 kernel's operand bytes (a 16-k slab of A and B is 512 rows for
// 256 x 256 outputs; there 384 rows for 256 x 128) and reads 3/4 of its LDS bytes.  k-slab 16 (one MFMA k-step), two LDS buffers,
// one barrier per slab; raw operands one slab ahead in registers.  (The wave-specialised form does not fit: 8 consumers + 4
// producers are 3 waves per SIMD, and 128 accumulators + operands exceed the 170 registers that leaves each of them.)
// LDS image of one bf16 plane of a 256-row x 16-k slab: [2 chunks of 8 k][row ^ ((row >> 3) & 1)][16 B], chunk stride 4160 B
// (= 64 mod 128).  Checked against the instruction-level bank model (MI355X_MICROARCH.md "LDS"): the MFMA operand reads
// (ds_read_b128, 32 consecutive rows of one chunk) and both staging stores (k-contiguous source: 8 lanes = 4 rows x 2 chunks;
// k-strided source: 8 lanes = every second row of one chunk) are conflict free.
struct ImgQ {
  static constexpr unsigned CS = 4160u;
  static constexpr unsigned PB = 2u * CS;
  static __device__ __forceinline__ unsigned at(int r, int chunk) { return (unsigned)chunk * CS + (unsigned)((r ^ ((r >> 3) & 1)) << 4); }
};

// one 256-row operand slab (256 x 16 k).  KCONTIG: every thread of the block one chunk (row u >> 1, chunk u & 1: two float4);
// KSTRIDED: 256 threads (FIRST: u < 256, else u >= 256) two chunks each (rows 2 q, 2 q + 1 of chunk c: eight float2 down k).
template <int KIND, bool FIRST, int ROWS = 256>
struct StageQ {
  static constexpr int NR = (KIND == KCONTIG) ? 8 : 16;       // raw registers
  const float* p0;
  int64_t ld;
  int c, r0;
  bool on, rok0, rok1;

  __device__ __forceinline__ void init(const float* base, int64_t ld_, int row0, int nrows, int t, int kbeg) {
    ld = ld_;
    if constexpr (KIND == KCONTIG) {
      c = t & 1;
      r0 = t >> 1;
      on = r0 < ROWS;
      const int row = row0 + r0;
      rok0 = row < nrows;
      rok1 = false;
      p0 = base + (int64_t)imin(row, nrows - 1) * ld + (kbeg + 8 * c);
    } else {
      const int u = FIRST ? t : t - 256;
      c = (u >> 7) & 1;
      r0 = 2 * (u & 127);
      on = (FIRST ? (t < 256) : (t >= 256)) && r0 < ROWS;
      const int row = row0 + r0;
      rok0 = row < nrows;
      rok1 = row + 1 < nrows;
      p0 = base + imin(row, nrows - 2) + (int64_t)(kbeg + 8 * c) * ld;
    }
  }

  template <bool MASK>
  __device__ __forceinline__ void load(float (&L)[NR], unsigned& pm, int koff, int klen) const {
    pm = 0;
    if (!on) return;
    if constexpr (KIND == KCONTIG) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kk = koff + 8 * c + 4 * h;
        float4 v;
        if (MASK) {
          if (rok0 && kk < klen) pm |= 1u << h;
          v = ld4(p0 + (imin(kk, klen - 4) - 8 * c));
        } else {
          v = ld4(p0 + (koff + 4 * h));
        }
        L[4 * h + 0] = v.x; L[4 * h + 1] = v.y; L[4 * h + 2] = v.z; L[4 * h + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kk = koff + 8 * c + i;
        const float* src = MASK ? (p0 + (int64_t)(imin(kk, klen - 1) - 8 * c) * ld) : (p0 + (int64_t)(koff + i) * ld);
        if (MASK && kk < klen) pm |= 1u << i;
        const float2 v = *reinterpret_cast<const float2*>(src);
        L[2 * i] = v.x; L[2 * i + 1] = v.y;
      }
    }
  }

  template <bool MASK, int NPL>
  __device__ __forceinline__ void stage(const float (&L)[NR], unsigned pm, char* plane0) const {
    if (!on) return;
    if constexpr (KIND == KCONTIG) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (!MASK || ((pm >> (e >> 2)) & 1u)) ? L[e] : 0.f;
      const Packed P = split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      const unsigned off = ImgQ::at(r0, c);
      *reinterpret_cast<uint4*>(plane0 + off) = P.h;
      if (NPL == 3) {
        *reinterpret_cast<uint4*>(plane0 + ImgQ::PB + off) = P.m;
        *reinterpret_cast<uint4*>(plane0 + 2 * ImgQ::PB + off) = P.l;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float v[8];
        const bool rok = j ? rok1 : rok0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (!MASK || (rok && ((pm >> i) & 1u))) ? L[2 * i + j] : 0.f;
        const Packed P = split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const unsigned off = ImgQ::at(r0 + j, c);
        *reinterpret_cast<uint4*>(plane0 + off) = P.h;
        if (NPL == 3) {
          *reinterpret_cast<uint4*>(plane0 + ImgQ::PB + off) = P.m;
          *reinterpret_cast<uint4*>(plane0 + 2 * ImgQ::PB + off) = P.l;
        }
      }
    }
  }
};

// BM = 256: waves 4 x 2, 64 x 128 outputs each;  BM = 192: waves 2 x 4, 96 x 64 outputs each (M = 49152 = 256 x 192: one row panel
// per CU and round, no partial last round for any N).
template <int AKIND, int BKIND, int NP, int BM>
__global__ __launch_bounds__(512, 2) void gemm_x3q_kernel(GemmP p) {
  constexpr int BN = 256, QK = 16;
  constexpr int WR = (BM == 256) ? 4 : 2, WCN = 8 / WR;
  constexpr int TM = BM / WR / 32, TN = BN / WCN / 32;
  constexpr unsigned BUF = 6 * ImgQ::PB;                     // A planes 0-2, B planes 0-2
  constexpr unsigned EPI_BYTES = WR * 32 * BN * 4;
  constexpr unsigned SMEM = (2 * BUF > EPI_BYTES) ? 2 * BUF : EPI_BYTES;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem_raw[SMEM];
  constexpr int NPL = (NP == 1 ? 1 : 3);

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = t >> 6;
  const int wr = wv / WCN, wc = wv % WCN;
  const int l31 = lane & 31, lh = lane >> 5;

  int mt, nt, z;
  if (!block_tile(p, mt, nt, z)) return;
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  const int kbeg = z * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? (kbeg + p.kchunk) : p.K;
  const int klen = kend - kbeg;
  const int nk = (klen + QK - 1) / QK;

  using SA = StageQ<AKIND, true, BM>;
  using SB = StageQ<BKIND, false>;
  SA sa;
  SB sb;
  sa.init(p.A, p.lda, m0, p.M, t, kbeg);
  sb.init(p.B, p.ldb, n0, p.N, t, kbeg);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  unsigned a_off[TM], b_off[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = ImgQ::at(wr * (BM / WR) + i * 32 + l31, lh);
#pragma unroll
  for (int j = 0; j < TN; ++j) b_off[j] = 3 * ImgQ::PB + ImgQ::at(wc * (BN / WCN) + j * 32 + l31, lh);

  auto run = [&](auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    float LA[SA::NR], LB[SB::NR];
    unsigned pa = 0, pb = 0;
    auto fetch = [&](int slab) {
      sa.template load<MASK>(LA, pa, slab * QK, klen);
      sb.template load<MASK>(LB, pb, slab * QK, klen);
    };
    auto stage = [&](int buf) {
      char* base = smem_raw + buf * BUF;
      sa.template stage<MASK, NPL>(LA, pa, base);
      sb.template stage<MASK, NPL>(LB, pb, base + 3 * ImgQ::PB);
    };
    fetch(0);
    stage(0);
    if (nk > 1) fetch(1);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      const char* base = smem_raw + (kt & 1) * BUF;
      if (kt + 1 < nk) stage((kt + 1) & 1);                  // slab kt + 1 into the other buffer (its readers passed the last barrier)
      if (kt + 2 < nk) fetch(kt + 2);
      bf16x8 a[TM][3];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i][q] = *reinterpret_cast<const bf16x8*>(base + q * ImgQ::PB + a_off[i]);
      constexpr int PA[9] = {0, 0, 1, 1, 0, 2, 1, 2, 2};
      constexpr int PB_[9] = {0, 1, 0, 1, 2, 0, 2, 1, 2};
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bf16x8 b[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) b[q] = *reinterpret_cast<const bf16x8*>(base + q * ImgQ::PB + b_off[j]);
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[PB_[q]], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  };
  const bool edge = (m0 + BM > p.M) || (n0 + BN > p.N) || (klen % QK != 0);
  if (edge) run(std::true_type{});
  else run(std::false_type{});

  gemm_epilogue<E_STORE, BM, BN, true, TM, TN, WR, 512, WCN>(p, acc, reinterpret_cast<float*>(smem_raw), m0, n0, mt, z, t, wr, wc, l31, lh);
}

int g_arith = -1;      // -1: not resolved yet; 0 native fp32 MFMA; 6 / 9 partial products on the bf16 pipe

template <int AKIND, int BKIND>
void launch_kind(GemmP& p, hipStream_t st, int bn, int np, dim3 grid) {
  if (bn == 256) {              // the 256-column kernels run the default arithmetic only (other arithmetics: measurements, 256 x 128)
    if (p.bm == 192) dg::launch((gemm_x3q_kernel<AKIND, BKIND, 6, 192>), grid, dim3(512), 0, st, p);
    else dg::launch((gemm_x3q_kernel<AKIND, BKIND, 6, 256>), grid, dim3(512), 0, st, p);
    return;
  }
  if (p.bm == 256) {
    if (np == 9) dg::launch((gemm_x3w2_kernel<AKIND, BKIND, 9>), grid, dim3(768), 0, st, p);
    else if (np == 1) dg::launch((gemm_x3w2_kernel<AKIND, BKIND, 1>), grid, dim3(768), 0, st, p);
    else dg::launch((gemm_x3w2_kernel<AKIND, BKIND, 6>), grid, dim3(768), 0, st, p);
    return;
  }
  if (p.bm == 64) {          // short reductions over many rows (conv1, the point-level [U|V] product and their data gradients)
    if (bn == 64) dg::launch((gemm_x3_kernel<AKIND, BKIND, 64, 6, 64>), grid, dim3(NT), 0, st, p);
    else dg::launch((gemm_x3_kernel<AKIND, BKIND, 128, 6, 64>), grid, dim3(NT), 0, st, p);
    return;
  }
  if (bn == 64) {
    if (np == 9) dg::launch((gemm_x3_kernel<AKIND, BKIND, 64, 9>), grid, dim3(NT), 0, st, p);
    else if (np == 1) dg::launch((gemm_x3_kernel<AKIND, BKIND, 64, 1>), grid, dim3(NT), 0, st, p);
    else dg::launch((gemm_x3_kernel<AKIND, BKIND, 64, 6>), grid, dim3(NT), 0, st, p);
  } else {
    if (np == 9) dg::launch((gemm_x3_kernel<AKIND, BKIND, 128, 9>), grid, dim3(NT), 0, st, p);
    else if (np == 1) dg::launch((gemm_x3_kernel<AKIND, BKIND, 128, 1>), grid, dim3(NT), 0, st, p);
    else dg::launch((gemm_x3_kernel<AKIND, BKIND, 128, 6>), grid, dim3(NT), 0, st, p);
  }
}

}  // namespace

namespace dg {

int gemm_arith() {
  if (g_arith < 0) {
    const char* e = getenv("DGCNN_GEMM_ARITH");      // f32 | bf16x6 | bf16x9
    int v = DGCNN_GEMM_ARITH_DEFAULT;
    if (e) v = (e[0] == 'f' || e[0] == '0') ? 0 : ((e[0] == '9' || (e[0] == 'b' && e[5] == '9')) ? 9 : ((e[0] == '1' || (e[0] == 'b' && e[5] == '1')) ? 1 : 6));
    g_arith = v;
  }
  return g_arith;
}

void set_gemm_arith(int v) { g_arith = v; }

// asrc in {A_ROW, A_COL}, bsrc in {B_ROW, B_COL}; operands float4-loadable (checked by the caller).
// p.mtiles / ntiles / xcd_group / splits / kchunk (multiple of 32 when splits > 1) are set.
int g_x3_tile_override = 0;      // dgcnn_gemm_x3_tile_override (tools): 0 = the rule below, 128 / 256 = that row tile where legal,
                                 // 512 = the 256 x 256 kernel where legal

// Tile choice among the three big kernels (all give bit-identical outputs: the per-element k order is the same):
//   256 x 128 wave-specialised (gemm_x3w2_kernel), 256 x 256 and 192 x 256 unspecialised (gemm_x3q_kernel).
// Measured on the head's nine products (profiles/r05/gemm_quad.txt): the wave-specialised main loop is ~10 % better on long
// reductions (K = 1728: 424 vs 465 us), the 256-column kernels win where tile quantisation or padding hurt the 256 x 128 tiling --
// N = 256 (384 tiles = 1.5 rounds of the 256 CUs -> 256 tiles of 192 rows: 85 -> 72 us), N = 192 (one padded column tile instead of
// two: 144 -> 128 us), M = 192 (weight gradient of MergedEdgeConv: no row padding: 134 -> 116 us) -- and on very short reductions
// (K = 192: 150 -> 134 us).  cost = padding waste x (1 + half the idle share of the last round) / main-loop efficiency; a partial
// round costs only about half its idle share because the step runs at the package power cap (idle CUs leave power to the busy ones).
static double x3_cost(int M, int N, int K, int bm, int bn, bool split) {
  const double mt = (double)cdiv(M, bm), nt = (double)cdiv(N, bn);
  const double waste = mt * bm * nt * bn / ((double)M * (double)N);
  double rf = 1.0;
  if (!split) {
    const double t = mt * nt / 256.0;
    rf = 1.0 + 0.5 * (cdiv((int64_t)(mt * nt), 256) / t - 1.0);
  }
  const double eff = (bn != 256) ? 1.0 : (K <= 192 ? 1.1 : (K <= 1024 ? 1.0 : 0.9));
  return waste * rf / eff;
}

// (bm, bn) of a large product: bn = 256 selects gemm_x3q_kernel<bm>, bn = 0 leaves the column tile to gemm.hip:tile_n
static void x3_pick(int M, int N, int K, int& bm, int& bn) {
  bm = 256;
  bn = 0;
  if (dg::gemm_arith() != 6 || M <= 128 || N <= 128) return;
  if (g_x3_tile_override == 512 || g_x3_tile_override == 448) { bm = g_x3_tile_override == 448 ? 192 : 256; bn = 256; return; }
  if (g_x3_tile_override != 0) return;
  const bool split = cdiv(M, 256) * cdiv(N, 128) < 128;          // few output tiles: the caller splits the reduction
  double best = x3_cost(M, N, K, 256, 128, split) * 0.98;      // (ties go to the wave-specialised kernel)
  const int cand[2] = {256, 192};
  for (int c = 0; c < 2; ++c) {
    const double v = x3_cost(M, N, K, cand[c], 256, split);
    if (v < best) { best = v; bm = cand[c]; bn = 256; }
  }
}

int x3_tile_n(int M, int N, int K) {
  if (x3_tile_m(M, N, K) < 192) return 0;
  int bm, bn;
  x3_pick(M, N, K, bm, bn);
  return bn;
}

static int x3_tile_m_base(int M, int N, int K) {
  if (g_x3_tile_override == 128) return 128;
  if ((g_x3_tile_override == 256 || g_x3_tile_override == 512 || g_x3_tile_override == 448) && M > 128 && N > 64) return 256;
  // short reduction, many rows, one or two column tiles: the kernel is bound by the latency of its few slabs -- 64-row tiles put
  // twice the workgroups (and loads in flight) on a CU
  if (dg::gemm_arith() == 6 && K <= 256 && N <= 256 && M >= 8192 && cdiv(M, 128) * cdiv(N, 128) <= 1024) return 64;
  // small problems (configs[0]: 1024 rows, K ~ 1000): a 256 x 128 tiling, even split over K, leaves most CUs without work
  if (cdiv(M, 256) * cdiv(N, 128) * cdiv(K, 256) < 256) return 128;
  return (M > 128 && N > 64) ? 256 : 128;
}

int x3_tile_m(int M, int N, int K) {
  const int base = x3_tile_m_base(M, N, K);
  if (base != 256) return base;
  int bm, bn;
  x3_pick(M, N, K, bm, bn);
  return bn == 256 ? bm : 256;
}

void launch_gemm_x3(int asrc, int bsrc, void* pv, hipStream_t st, int bn, int np) {
  GemmP& p = *reinterpret_cast<GemmP*>(pv);
  const unsigned gx = p.xcd_group ? (unsigned)(cdiv(p.mtiles, 8) * 8 * p.ntiles) : (unsigned)(p.mtiles * p.ntiles);
  dim3 grid(gx, 1, (unsigned)p.splits);
  if (p.zmajor) grid = dim3((unsigned)(cdiv(p.splits, 8) * 8 * p.mtiles * p.ntiles), 1, 1);
  if (asrc == A_ROW && bsrc == B_ROW) launch_kind<KCONTIG, KSTRIDED>(p, st, bn, np, grid);
  else if (asrc == A_ROW && bsrc == B_COL) launch_kind<KCONTIG, KCONTIG>(p, st, bn, np, grid);
  else launch_kind<KSTRIDED, KSTRIDED>(p, st, bn, np, grid);
}

}  // namespace dg

extern "C" int dgcnn_gemm_set_arith(int mode) {
  DG_REQUIRE(mode == 0 || mode == 1 || mode == 6 || mode == 9, DGCNN_EINVAL, "dgcnn_gemm_set_arith: mode must be 0, 1, 6 or 9 (got %d)", mode);
  dg::set_gemm_arith(mode);
  return DGCNN_OK;
}

extern "C" int dgcnn_gemm_get_arith(void) { return dg::gemm_arith(); }

extern "C" int dgcnn_gemm_x3_tile_rows(int M, int N, int K) { return dg::x3_tile_m(M, N, K); }
extern "C" int dgcnn_gemm_x3_tile_cols(int M, int N, int K) { return dg::x3_tile_n(M, N, K); }

extern "C" int dgcnn_gemm_x3_tile_override(int bm) {      // tools: force the 128- or 256-row tile (0 = automatic); returns the previous value
  const int prev = dg::g_x3_tile_override;
  dg::g_x3_tile_override = (bm == 128 || bm == 256 || bm == 512 || bm == 448) ? bm : 0;      // 512: 256 x 256, 448: 192 x 256
  return prev;
}
