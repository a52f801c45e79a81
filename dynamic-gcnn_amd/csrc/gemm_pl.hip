// gemm_pl.hip -- fp32 GEMM on the 16-bit matrix pipe from PRE-SPLIT operand planes (gfx950).
//
// gemm_x3.hip splits every fp32 operand element into three bf16 terms inside the GEMM: every column tile of an
// output row panel repeats the split of the same A rows in VALU (4x for FC0, 8x for MergedEdgeConv), and that
// staging work next to the MFMAs is what the kernel pays for under the package power cap
// (profiles/r01_gemm_x3_ablation.txt).  Here the operands arrive already split -- "planes" written ONCE by the
// kernel that produced the tensor (or by dgcnn_split_planes_f32) -- and the GEMM does no VALU work on them at
// all: loader waves DMA the planes straight into LDS (global_load_lds_dwordx4), consumer waves read MFMA
// operands and issue MFMAs.
//
// Plane layout ("chunk-major"), one set per tensor X (rows x cols), NPL planes of 16-bit elements:
//     element (row, c) of plane p:  base + p * plane_stride + ((c / 8) * rows_alloc + row) * 16 + (c % 8) * 2   [bytes]
//   i.e. [plane][cols/8][rows_alloc][8]: the 8 channels of one row form one 16-byte slot, consecutive rows of the
//   same channel octet are consecutive slots.  rows_alloc = rows rounded up to 64; the pad rows hold zeros.
//   ONE layout serves both uses of a tensor:
//     PL_KC  the reduction runs over the channels (forward X W, dgrad dY W^T): an MFMA operand of lane (row, k-octet)
//            is exactly one slot; a DMA instruction moves 64 consecutive rows of one octet (1 KiB contiguous);
//     PL_TR  the reduction runs over the rows (wgrad X^T dY): the LDS tile keeps the [octet][row][8] slots and the
//            consumer reads it with ds_read_b64_tr_b16 (hardware 4x4 transpose), so no second copy of the tensor.
//   Weights are tiny: dgcnn_split_planes_f32 writes them in whichever orientation the product needs.
//
// Formats:  (round 3 also had DGCNN_PLANES_BF16X3, a = a1 + a2 + a3 exactly, 6 partial products: 1.5x the bytes of the fp32 operands
//           it replaced and slower than the in-kernel split at kernel and step level -- profiles/r03/gemm_planes.txt -- removed)
//                                same arithmetic as gemm_x3.hip NP = 6.
//           DGCNN_PLANES_F16X2   a * 2^e = h1 + h2 (2 x fp16, |a - (h1+h2) 2^-e| <= 2^-22 |a| while h2 is a normal
//                                fp16 number); 3 partial products (h1h1, h1h2, h2h1) -- see split_f16x2 below.
//
// Kernel: 256 x 128 output tile, 768 threads: waves 0-7 consumers (4 x 2, 64 x 64 each as 2 x 2 32x32x16 MFMA
// tiles), waves 8-11 loaders.  NS LDS stages of KS-k slabs: the DMA round trip of a slab is ~2 us under load, as long
// as a 32-k slab's MFMAs -- with two stages (one slab in flight) the kernel was latency bound (profiles/r03/gemm_planes.txt:
// no-MFMA ablation 0.34 ms of 0.58); NS - 1 slabs are in flight now.
// The consumers PING-PONG: waves w and w + 4 share a SIMD (and its matrix pipe); in every phase one of the two issues
// MFMAs on operands it already holds in registers while the other reads its next operands from LDS, and a barrier
// swaps the roles -- the matrix pipe never waits for an LDS read.  (With all eight waves in the same phase -- read,
// then MFMA, then barrier -- the pipe idled ~0.8 us of every 32-k slab: 174 TFLOP/s instead of the ~235 the pipe sustains
// on random operands, profiles/r03/gemm_planes.txt.)  PH = phases per slab and group: 2 (16 k of operands in
// registers: 48 VGPRs for three planes) or 1 (32 k).
#include "gemm_common.h"
#include "planes_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;

enum { PL_KC = 0, PL_TR = 1 };

template <int FMT> struct Fmt;
template <> struct Fmt<DGCNN_PLANES_F16X2> {
  static constexpr int NPL = 2, NP = 3;
  static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

struct PlP {
  GemmP g;                       // shapes, C / partial, epilogue options, tile mapping (A / B pointers unused)
  const char* Ap; int64_t a_ps; int64_t a_rows;      // plane 0 of the operand (at its first octet), plane stride [B], rows_alloc
  const char* Bp; int64_t b_ps; int64_t b_rows;
  const float* a_scale; const float* b_scale;        // device scalars (powers of two) the planes were multiplied by, or null
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KS = reduction rows / channels per slab (16 or 32), NS = LDS stages (slabs landed or in flight)
template <int FORM, int FMT, int KS, int NS>
__global__ __launch_bounds__(768) void gemm_pl_kernel(PlP q) {
  using F = Fmt<FMT>;
  constexpr int BM = 256, BN = 128, TM = 2, TN = 2, NPL = F::NPL;
  constexpr int OS = KS / 8;                                   // octets per slab
  constexpr int PH = KS / 16;                                  // 16-k MFMA steps (= ping-pong phase pairs) per slab
  constexpr int NB = 2 * PH;                                   // barriers per slab
  constexpr unsigned PA = BM * KS * 2, PB = BN * KS * 2;       // bytes per plane per slab
  constexpr unsigned STAGE = NPL * (PA + PB);
  constexpr unsigned EPI_BYTES = 128 * BN * 4;
  constexpr unsigned LDS_BYTES = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const GemmP& p = q.g;

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wv = t >> 6;

  int mt, nt, z;
  if (!block_tile(p, mt, nt, z)) return;
  const int m0 = mt * BM;
  const int n0 = nt * BN;
  const int kbeg = z * p.kchunk;
  const int kend = (kbeg + p.kchunk < p.K) ? (kbeg + p.kchunk) : p.K;
  const int nk = (kend - kbeg + KS - 1) / KS;                  // PL_TR: the last slab may run into the zero pad rows

  if (wv >= 8) {
    // ------------------------------------------------------------------ loaders: 4 waves x PER DMA instructions per slab
    constexpr int NIA = NPL * 4 * OS, NIB = NPL * 2 * OS, NI = NIA + NIB, PER = NI / 4;
    static_assert(NI % 4 == 0 && PER * (NS - 1) < 64, "DMA instructions per wave");
    const int lw = __builtin_amdgcn_readfirstlane(wv - 8);
    const char* src[PER];
    unsigned dst[PER];
    // (every instruction is always issued -- out-of-range rows / octets read a clamped, valid address and feed output rows
    //  the epilogue drops -- so that every slab counts exactly PER on this wave's vmcnt)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = lw + 4 * i;                                // wave-uniform
      const bool isA = e < NIA;
      const int ee = isA ? e : e - NIA;
      const int per_plane = isA ? 4 * OS : 2 * OS;
      const int pl = ee / per_plane, r = ee % per_plane;
      const char* base = (isA ? q.Ap + pl * q.a_ps : q.Bp + pl * q.b_ps);
      const int64_t rows = isA ? q.a_rows : q.b_rows;
      const int row0 = isA ? m0 : n0;
      const unsigned pbase = isA ? pl * PA : NPL * PA + pl * PB;
      if (FORM == PL_KC) {
        // 64 consecutive tile rows of one k-octet: image [octet c][tile row] slots
        const int nrb = isA ? 4 : 2;
        const int c = r / nrb, rb = r % nrb;
        const int64_t row = row0 + rb * 64;
        const int64_t rowc = row < rows ? row : 0;
        src[i] = base + (((int64_t)(kbeg >> 3) + c) * rows + rowc + lane) * 16;
        dst[i] = pbase + (unsigned)((c * (isA ? BM : BN) + rb * 64) * 16);
      } else {
        // 64 / KS channel octets x KS reduction rows; slot (octet cidx, row r) at cidx * KS + (r ^ 4 (cidx & 3)): the XOR
        // spreads the four octets a transpose-read touches over the four 64-byte quarters of a bank row
        constexpr int OPI = 64 / KS;                           // octets per instruction
        const int cidx = OPI * r + lane / KS;                  // tile-local octet
        const int rr = (lane & (KS - 1)) ^ (4 * (cidx & 3));
        const int64_t oct = (row0 >> 3) + cidx;
        const int64_t noct = ((isA ? p.M : p.N) + 7) >> 3;
        const int64_t octc = oct < noct ? oct : 0;
        src[i] = base + (octc * rows + kbeg + rr) * 16;
        dst[i] = pbase + (unsigned)(r * 64 * 16);
      }
    }
    const int64_t stepA = (FORM == PL_KC) ? q.a_rows * 16 * OS : KS * 16;       // bytes per slab
    const int64_t stepB = (FORM == PL_KC) ? q.b_rows * 16 * OS : KS * 16;
    auto issue = [&](int slab) {
      const int buf = slab % NS;
      const int sl = slab;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const bool isA = (lw + 4 * i) < NIA;
        const char* s = src[i] + (int64_t)sl * (isA ? stepA : stepB);
        char* d = smem + buf * STAGE + dst[i];
        __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)d, 16, 0, 0);
      }
    };
    // wait until at most `later` of the most recently issued slabs are still in flight (DMA completes in issue order)
    auto wait_all_but = [&](int later) {
      if (NS >= 4 && later >= 3) wait_vmcnt<(NS >= 4 ? 3 : 0) * PER>();
      else if (NS >= 3 && later == 2) wait_vmcnt<(NS >= 3 ? 2 : 0) * PER>();
      else if (later == 1) wait_vmcnt<PER>();
      else wait_vmcnt<0>();
    };
    static_assert(NS <= 5, "wait_all_but covers up to 3 slabs behind the awaited one");
    // slab u lives in stage u % NS; it is read during the NB phases of its period (group 0) and one phase later (group 1).
    // At the start of period u the stage of slab u - 1 is free: slab u - 1 + NS goes there.  Slab u + 1 must have landed
    // by the last barrier of period u.
    int issued = 0;                                  // slabs issued so far
    for (; issued < NS - 1 && issued < nk; ++issued) issue(issued);
    wait_all_but(issued - 1);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      if (issued < nk) { issue(issued); ++issued; }
#pragma unroll
      for (int bb = 0; bb < NB - 1; ++bb) __syncthreads();
      wait_all_but(issued - (kt + 2) > 0 ? issued - (kt + 2) : 0);        // slabs after kt + 1 may still be in flight
      __syncthreads();
    }
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

  // per-lane LDS offsets of the operand reads (16-k step s of the slab: + s * SSTEP)
  unsigned a_off[TM][2], b_off[TN][2];
  constexpr unsigned SSTEP_A = (FORM == PL_KC) ? 2 * BM * 16 : 256, SSTEP_B = (FORM == PL_KC) ? 2 * BN * 16 : 256;
  if (FORM == PL_KC) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a_off[i][0] = a_off[i][1] = (unsigned)((lh * BM + wr * 64 + i * 32 + l31) * 16);
#pragma unroll
    for (int j = 0; j < TN; ++j) b_off[j][0] = b_off[j][1] = NPL * PA + (unsigned)((lh * BN + wc * 64 + j * 32 + l31) * 16);
  } else {
    // ds_read_b64_tr_b16: within a 16-lane group lane 4 jj + qq supplies the 8-byte piece (row jj, channels 4 qq .. 4 qq + 3)
    // and lane c receives channel c of rows 0..3.  Groups: g & 1 -> channels 16..31 of the 32-row MFMA tile, g >> 1 -> k half.
    const int g = lane >> 4, jj = (lane >> 2) & 3, qq = lane & 3;
    const int cb3 = 2 * (g & 1) + (qq >> 1);                   // (tile-local octet) & 3: the tile bases are multiples of 4 octets
    const int kh = g >> 1;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int x = ((2 * kh + tt) ^ cb3);                     // swizzled (k row >> 2) & 3
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a_off[i][tt] = (unsigned)(((wr * 8 + i * 4 + cb3) * KS + x * 4 + jj) * 16 + (qq & 1) * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b_off[j][tt] = NPL * PA + (unsigned)(((wc * 8 + j * 4 + cb3) * KS + x * 4 + jj) * 16 + (qq & 1) * 8);
    }
  }

  const int grp = wv >> 2;                         // waves w and w + 4 sit on the same SIMD: group 1 runs one phase behind
  s16x8 a[TM][NPL], b[TN][NPL];
  auto load_ops = [&](const char* base, int s) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      if (FORM == PL_KC) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i][pl] = *reinterpret_cast<const s16x8*>(base + pl * PA + a_off[i][0] + s * SSTEP_A);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j][pl] = *reinterpret_cast<const s16x8*>(base + pl * PB + b_off[j][0] + s * SSTEP_B);
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + pl * PA + a_off[i][0] + s * SSTEP_A));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + pl * PA + a_off[i][1] + s * SSTEP_A));
          a[i][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + pl * PB + b_off[j][0] + s * SSTEP_B));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + pl * PB + b_off[j][1] + s * SSTEP_B));
          b[j][pl] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
    }
  };
  auto compute = [&]() {
    // partial products, largest first: (1,1) (1,2) (2,1) (2,2) (1,3) (3,1)
    constexpr int PRA[6] = {0, 0, 1, 1, 0, 2};
    constexpr int PRB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
    for (int pr = 0; pr < F::NP; ++pr)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = F::mfma(a[i][PRA[pr]], b[j][PRB[pr]], acc[i][j]);
  };

  // phase boundary: nothing may be scheduled across it (the compiler otherwise pulls the MFMAs above the barrier, next to
  // the LDS reads that feed them, and the two waves of a SIMD end up reading and multiplying in the same phase again)
  auto phase_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };
  __syncthreads();                                 // slab 0 landed
  if (grp) phase_barrier();                        // group 1 idles through phase 0
  int stage = 0;
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    const char* base = smem + stage * STAGE;
    stage = (stage + 1 == NS) ? 0 : stage + 1;
#pragma unroll
    for (int h = 0; h < PH; ++h) {
      load_ops(base, h);
      phase_barrier();
      compute();
      if (!(grp && kt == nk - 1 && h == PH - 1)) phase_barrier();      // (group 1's last compute is the kernel's last phase)
    }
  }

  if (q.a_scale || q.b_scale) {                    // planes hold x * scale (powers of two): exact rescaling of the result
    const float os = 1.f / ((q.a_scale ? *q.a_scale : 1.f) * (q.b_scale ? *q.b_scale : 1.f));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= os;
  }
  gemm_epilogue<E_STORE, BM, BN, true, TM, TN, 4, 512>(p, acc, reinterpret_cast<float*>(smem), m0, n0, mt, z, t, wr, wc, l31, lh);
}

// ------------------------------------------------------------------------------------------------------
// fp32 -> planes
// ------------------------------------------------------------------------------------------------------
// Activations: src (rows x cols) row-major, ld % 4 == 0, cols % 8 == 0.  Block = 64 rows x 16 octets; the split slots go
// through LDS so that every wave store is 64 consecutive rows of one octet (1 KiB contiguous).  Pad rows [rows, rows_alloc)
// are written as zeros.
template <int FMT>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int cols,
                                                         char* __restrict__ dst, int64_t plane_stride, int64_t rows_alloc,
                                                         const float* __restrict__ scale_dev) {
  // thread = one row x 4 adjacent channel octets (a full 128-byte line of the source row), lane = row: every wave store is
  // 64 consecutive slots of one octet (1 KiB per plane)
  const int lane = threadIdx.x & 63;
  const int oc0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4;
  const int noct = cols >> 3;
  if (oc0 >= noct) return;
  const float scale = (FMT == DGCNN_PLANES_F16X2 && scale_dev) ? *scale_dev : 1.f;
  const int64_t step = (int64_t)gridDim.x * 64;
  for (int64_t r = (int64_t)blockIdx.x * 64 + lane; r < rows_alloc; r += step) {
    float4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows && oc0 + u < noct) {
        a[u] = *reinterpret_cast<const float4*>(src + r * ld + (int64_t)(oc0 + u) * 8);
        b[u] = *reinterpret_cast<const float4*>(src + r * ld + (int64_t)(oc0 + u) * 8 + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (oc0 + u >= noct) break;
      const float v[8] = {a[u].x, a[u].y, a[u].z, a[u].w, b[u].x, b[u].y, b[u].z, b[u].w};
      uint4 o[3];
      split8<FMT>(v, scale, o);
#pragma unroll
      for (int pl = 0; pl < PlaneFmt<FMT>::NPL; ++pl)
        *reinterpret_cast<uint4*>(dst + pl * plane_stride + ((int64_t)(oc0 + u) * rows_alloc + r) * 16) = o[pl];
    }
  }
}

// Generic (small tensors: weights).  Plane element (row, c) = src[row * rs + c * cs]; rows / cols of the PLANE SET.
template <int FMT>
__global__ __launch_bounds__(256) void split_strided_kernel(const float* __restrict__ src, int64_t rs, int64_t cs, int64_t rows,
                                                            int cols, char* __restrict__ dst, int64_t plane_stride,
                                                            int64_t rows_alloc, const float* __restrict__ scale_dev) {
  constexpr int NPL = PlaneFmt<FMT>::NPL;
  const int noct = (cols + 7) >> 3;
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;        // (octet, row) with the row fastest
  if (item >= (int64_t)noct * rows_alloc) return;
  const int64_t row = item % rows_alloc;
  const int oc = (int)(item / rows_alloc);
  const float scale = (FMT == DGCNN_PLANES_F16X2 && scale_dev) ? *scale_dev : 1.f;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = oc * 8 + e;
    v[e] = (row < rows && c < cols) ? src[row * rs + (int64_t)c * cs] : 0.f;
  }
  uint4 o[3];
  split8<FMT>(v, scale, o);
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint4*>(dst + pl * plane_stride + item * 16) = o[pl];
}

// max |x| over a 2-D view -> the power of two that brings it to [2^14, 2^15) (fp16 planes keep 22 significant bits of every
// element down to 2^-17 of the tensor's largest and an absolute 2^-40 of it below that; 65504 is never exceeded)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int cols,
                                                     unsigned* __restrict__ out_bits) {
  const int64_t n4 = (int64_t)rows * (cols >> 2);
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / (cols >> 2);
    const int c = (int)(i % (cols >> 2)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld + c);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out_bits, __float_as_uint(m));     // non-negative floats order like their bit patterns
}

__global__ void scale_from_max_kernel(const unsigned* __restrict__ max_bits, float bound_mul, float* __restrict__ scale) {
  *scale = pow2_scale_for(__uint_as_float(*max_bits) * bound_mul);
}

inline bool aligned16p(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

namespace dg {
void launch_reduce_partials(const float* part, int splits, int M, int N, float* C, int64_t ldc, float beta, hipStream_t st);
}

#define ST ((hipStream_t)stream)

extern "C" int dgcnn_split_planes_f32(const float* src, int64_t row_stride, int64_t col_stride, int64_t rows, int cols, int fmt,
                                      const float* scale_dev, void* dst, int64_t plane_stride, int64_t rows_alloc, void* stream) {
  DG_REQUIRE(src && dst && rows > 0 && cols > 0, DGCNN_EINVAL, "dgcnn_split_planes_f32: bad args");
  DG_REQUIRE(fmt == DGCNN_PLANES_F16X2, DGCNN_EINVAL, "dgcnn_split_planes_f32: unknown format %d (the 3 x bf16 format was removed)", fmt);
  DG_REQUIRE(rows_alloc >= rows && rows_alloc % 64 == 0 && plane_stride % 16 == 0 && aligned16p(dst), DGCNN_EINVAL,
             "dgcnn_split_planes_f32: rows_alloc must be a multiple of 64 >= rows, planes 16-byte aligned");
  const int noct = (cols + 7) / 8;
  DG_REQUIRE(plane_stride >= (int64_t)noct * rows_alloc * 16, DGCNN_EINVAL, "dgcnn_split_planes_f32: plane stride too small");
  const bool fast = col_stride == 1 && cols % 8 == 0 && row_stride % 4 == 0 && aligned16p(src);
  if (fast) {
    const int64_t gy = dg::cdiv(noct, 16);
    int64_t gx = rows_alloc / 64;
    if (gx * gy > 8192) gx = dg::cdiv(8192, gy);
    dim3 grid((unsigned)gx, (unsigned)gy);
    dg::launch((split_rows_kernel<DGCNN_PLANES_F16X2>), grid, dim3(256), 0, ST, src, row_stride, rows, cols, (char*)dst, plane_stride, rows_alloc, scale_dev);
  } else {
    const unsigned g = (unsigned)dg::cdiv((int64_t)noct * rows_alloc, 256);
    dg::launch((split_strided_kernel<DGCNN_PLANES_F16X2>), dim3(g), dim3(256), 0, ST, src, row_stride, col_stride, rows, cols, (char*)dst, plane_stride, rows_alloc, scale_dev);
  }
  return dg::check_launch("dgcnn_split_planes_f32");
}

// scale <- power of two for a DGCNN_PLANES_F16X2 split of the view (max |x| * bound_mul lands in [2^14, 2^15)); ws = 4 bytes
extern "C" int dgcnn_planes_scale_f32(const float* src, int64_t ld, int64_t rows, int cols, float bound_mul, float* scale_dev,
                                      void* ws, void* stream) {
  DG_REQUIRE(src && scale_dev && ws && rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && aligned16p(src), DGCNN_EINVAL,
             "dgcnn_planes_scale_f32: bad args (float4-loadable view needed)");
  (void)dg::memset_async(ws, 0, 4, ST);
  const int64_t n4 = rows * (cols / 4);
  const unsigned g = (unsigned)(dg::cdiv(n4, 256 * 8) < 2048 ? dg::cdiv(n4, 256 * 8) : 2048);
  dg::launch(absmax_kernel, dim3(g ? g : 1), dim3(256), 0, ST, src, ld, rows, cols, (unsigned*)ws);
  dg::launch(scale_from_max_kernel, dim3(1), dim3(1), 0, ST, (const unsigned*)ws, bound_mul, scale_dev);
  return dg::check_launch("dgcnn_planes_scale_f32");
}

// C (M x N, fp32) (+)= A B^T-style product of two plane sets.
//   form = DGCNN_PL_KC: A rows = M, B rows = N, both reduce over their K channels (K % 32 == 0; plane pointers at the first octet)
//   form = DGCNN_PL_TR: A channels = M, B channels = N (both % 16 == 0), both reduce over their K rows (pad rows are zero)
extern "C" int dgcnn_gemm_planes_f32(int form, int fmt, int M, int N, int K,
                                     const void* A, int64_t a_plane_stride, int64_t a_rows_alloc,
                                     const void* B, int64_t b_plane_stride, int64_t b_rows_alloc,
                                     const float* a_scale_dev, const float* b_scale_dev,
                                     float* C, int64_t ldc, float beta,
                                     const float* gbias, int64_t ldgbias, int rows_per_group,
                                     double* stats, void* colmax_keys, int colmax_rows_per_group,
                                     void* ws, size_t ws_bytes, void* stream) {
  DG_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, DGCNN_EINVAL, "dgcnn_gemm_planes_f32: bad args");
  DG_REQUIRE(form == DGCNN_PL_KC || form == DGCNN_PL_TR, DGCNN_EINVAL, "dgcnn_gemm_planes_f32: unknown form %d", form);
  DG_REQUIRE(fmt == DGCNN_PLANES_F16X2, DGCNN_EINVAL, "dgcnn_gemm_planes_f32: unknown format %d (the 3 x bf16 format was removed)", fmt);
  DG_REQUIRE(aligned16p(A) && aligned16p(B) && a_plane_stride % 16 == 0 && b_plane_stride % 16 == 0 && a_rows_alloc % 64 == 0 &&
                 b_rows_alloc % 64 == 0, DGCNN_EINVAL, "dgcnn_gemm_planes_f32: planes must be 16-byte aligned, rows_alloc a multiple of 64");
  DG_REQUIRE(!gbias || rows_per_group > 0, DGCNN_EINVAL, "dgcnn_gemm_planes_f32: rows_per_group");
  PlP q = {};
  GemmP& p = q.g;
  p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.beta = beta;
  p.gbias = gbias; p.ldgbias = ldgbias; p.rpg = rows_per_group > 0 ? rows_per_group : 1;
  p.gbvec = gbias && (ldgbias % 4 == 0) && aligned16p(gbias);
  p.stats = stats;
  DG_REQUIRE(!colmax_keys || (form == DGCNN_PL_KC && colmax_rows_per_group > 0 && colmax_rows_per_group % 256 == 0), DGCNN_EUNSUP,
             "dgcnn_gemm_planes_f32: the column-maximum epilogue needs the KC form and rows_per_group %% 256 == 0");
  p.colmax = reinterpret_cast<unsigned long long*>(colmax_keys); p.colmax_rpg = colmax_rows_per_group;
  p.splits = 1; p.kchunk = K; p.bm = 256; p.stat_slots = dg::stat_slots();
  q.Ap = (const char*)A; q.a_ps = a_plane_stride; q.a_rows = a_rows_alloc;
  q.Bp = (const char*)B; q.b_ps = b_plane_stride; q.b_rows = b_rows_alloc;
  q.a_scale = a_scale_dev; q.b_scale = b_scale_dev;
  p.mtiles = (int)dg::cdiv(M, 256);
  p.ntiles = (int)dg::cdiv(N, 128);
  if (form == DGCNN_PL_KC) {
    DG_REQUIRE(K % 32 == 0, DGCNN_EUNSUP, "dgcnn_gemm_planes_f32(KC): K %% 32 != 0 (%d)", K);
    DG_REQUIRE(M <= a_rows_alloc && N <= b_rows_alloc, DGCNN_EINVAL, "dgcnn_gemm_planes_f32(KC): rows_alloc smaller than the operand");
  } else {
    DG_REQUIRE(M % 32 == 0 && N % 32 == 0, DGCNN_EUNSUP, "dgcnn_gemm_planes_f32(TR): M, N must be multiples of 32 (%d, %d)", M, N);
    DG_REQUIRE(a_rows_alloc == b_rows_alloc && K <= a_rows_alloc, DGCNN_EINVAL, "dgcnn_gemm_planes_f32(TR): both operands must share rows_alloc >= K");
    DG_REQUIRE(!gbias && !stats, DGCNN_EUNSUP, "dgcnn_gemm_planes_f32(TR): bias / statistics unsupported");
    // split the reduction over the rows: every k-chunk pinned to one XCD (gemm_common.h:block_tile), whole rounds of 32 workgroups
    const int64_t tiles = (int64_t)p.mtiles * p.ntiles;
    const int64_t maxs = K / 256 > 0 ? K / 256 : 1;
    if (maxs >= 8 && ws) {
      double best_eff = 0.0;
      int64_t best = 0;
      for (int64_t R = 1; R <= 3; ++R) {
        int64_t per = (32 * R) / tiles;
        if (per < 1) per = 1;
        if (8 * per > maxs) per = maxs / 8;
        const double eff = (double)(tiles * per) / (double)(32 * dg::cdiv(tiles * per, 32));
        if (eff > best_eff + 0.03) { best_eff = eff; best = 8 * per; }
      }
      if (best >= 8) {
        int64_t chunk = dg::cdiv(dg::cdiv(K, best), 32) * 32;
        const int64_t s = dg::cdiv(K, chunk);
        if (s > 1 && ws_bytes >= (size_t)s * M * N * sizeof(float)) {
          p.splits = (int)s; p.kchunk = (int)chunk; p.zmajor = 1; p.partial = reinterpret_cast<float*>(ws);
        }
      }
    }
  }
  if (p.splits > 1) p.cvec = (N % 4 == 0) && aligned16p(p.partial);
  else p.cvec = (ldc % 4 == 0) && aligned16p(C);
  p.xcd_group = (p.ntiles > 1 && p.mtiles >= 16 && p.splits == 1) ? 1 : 0;
  const unsigned gx = p.xcd_group ? (unsigned)(dg::cdiv(p.mtiles, 8) * 8 * p.ntiles) : (unsigned)(p.mtiles * p.ntiles);
  dim3 grid(gx, 1, 1);
  if (p.zmajor) grid = dim3((unsigned)(dg::cdiv(p.splits, 8) * 8 * p.mtiles * p.ntiles), 1, 1);
  // 32-k slabs x 3 stages (48 KB each): up to 96 KB of DMA in flight per CU
#define DG_PL(FORM, FMT) dg::launch((gemm_pl_kernel<FORM, FMT, 32, 3>), grid, dim3(768), 0, ST, q)
  if (form == DGCNN_PL_KC) DG_PL(PL_KC, DGCNN_PLANES_F16X2);
  else DG_PL(PL_TR, DGCNN_PLANES_F16X2);
#undef DG_PL
  int rc = dg::check_launch("dgcnn_gemm_planes_f32");
  if (rc) return rc;
  if (p.splits > 1) {
    dg::launch_reduce_partials(p.partial, p.splits, M, N, C, ldc, beta, ST);
    rc = dg::check_launch("dgcnn_gemm_planes_f32(reduce)");
  }
  return rc;
}
