// knn.hip -- K1: fused pairwise squared distance + per-row top-k select (dgcnn/ops.py:8-19).
//
// MUST be compiled with -ffp-contract=off: the arithmetic order is normative
// (oracle/knn_oracle.c) and the indices are bit-exact against it:
//     s_i  = sequential sum of fl(x*x)          (no FMA)
//     p_ij = fmaf chain over c ascending from +0
//     D_ij = fl( fl(s_i + s_j) - 2 p_ij )
// Selection: k smallest by (D_ij, j) lexicographic, ascending, self included.
//
// Mapping (wave64): a workgroup owns 64 query rows (one per lane, x_i in VGPRs) and 4 waves that
// split every 128-candidate LDS tile four ways, so B*N/64*4 waves are in flight (3072 at the
// headline shape).  All lanes of a wave read the same candidate row from LDS (broadcast, no bank
// conflicts); each lane keeps a private sorted (d, j) list of KC entries in registers and inserts
// with a branch-free v_med3/v_cndmask network, skipped wave-uniformly when no lane improves.
// The four per-wave lists are merged through LDS at the end.  The (B,N,N) matrix never exists.
#include "knn_common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int ROWS = 64;
constexpr int WAVES = 4;
constexpr int TJ = 128;
constexpr int PERW = TJ / WAVES;

// s_i = sequential sum of fl(x^2) over c ascending (the oracle's order: no FMA, no reassociation), one thread per row.
// 64 rows per block (B*N/64 blocks: 768 at the headline shape -- the round-1 kernel ran 192 blocks of 256 rows, under one
// block per CU, 24 us at C = 64): the block's rows are staged through LDS with coalesced loads (a thread walking its own
// 256-byte row in global memory touches 64 cache lines per wave-load), then 64 threads walk one row each, stride C + 1.
constexpr int SQ_ROWS = 64;
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int C,
                                                     float* __restrict__ sq) {
  extern __shared__ float sq_tile[];          // [64][C + 1]
  const int64_t r0 = (int64_t)blockIdx.x * SQ_ROWS;
  const int t = threadIdx.x;
  const int S = C + 1;
  for (int e = t; e < SQ_ROWS * C; e += 256) {
    const int rr = e / C, cc = e - rr * C;
    if (r0 + rr < rows) sq_tile[rr * S + cc] = x[(r0 + rr) * ldx + cc];
  }
  __syncthreads();
  if (t < SQ_ROWS && r0 + t < rows) {
    float s = 0.0f;
    for (int c = 0; c < C; ++c) {
      const float v = sq_tile[t * S + c];
      const float q = v * v;
      s = s + q;
    }
    sq[r0 + t] = s;
  }
}


// ------------------------------------------------------------------------------------------------
// A bound for the raw-coordinate layer (C <= 4, no previous graph to seed from; round 6).  One thread per query row counts the
// cloud's candidates (every STRIDE-th: k of a SAMPLE within t still puts the row's k-th distance below t) into half-octave bins of the
// NORMATIVE distance -- bin = (float bits of max(d, 0)) >> 22, i.e. exponent and leading mantissa bit, taken relative to the cloud's
// largest possible distance so that 48 bins span 24 octaves of d whatever the cloud's scale -- and reports the upper edge of the first
// bin at which the count reaches k.  The scan then drops every candidate at or above that edge before it reaches the sorted lists:
// ~1.3 k candidates pass per row instead of ~k (1 + ln(N / k)) per LIST (four lists per row), and the sorted inserts were three
// quarters of the scan's instructions (r05_pmc_sq.txt).  Rigorous for any input: a candidate dropped has k candidates strictly below it;
// a coarse bin (many equal distances) only makes the bound looser.  Per candidate: one LDS broadcast read, the 6-operation distance,
// clamp, shift, one ds_add_u32 into the thread's own column of the histogram ([bin][thread]: bank = thread, conflict free).
constexpr int HB_NB = 48;
template <int STRIDE>
__global__ __launch_bounds__(256) void knn_hist_bound_kernel(const float* __restrict__ x, const float* __restrict__ sq, int N, int C,
                                                             int64_t ldx, int k, float* __restrict__ tau0) {
  // 64 query rows per workgroup (lane = row); wave w counts quarter w of every 256-candidate tile into its own histogram
  // ([wave][bin][lane]: bank = lane), the four are added at the end -- 3072 waves at (24,2048) instead of 768: the per-candidate chain
  // (two LDS broadcast reads -> distance -> bin -> ds_add) is latency bound with one wave per SIMD (125 us; 4 waves per row: see
  // profiles/r06/knn_hist.txt)
  __shared__ unsigned hist[4 * HB_NB * 64];
  __shared__ float4 tile[256];                          // the sampled candidates' coordinates (zero padded to 4)
  __shared__ float ts[256];                             // ... and their s_j
  __shared__ float wmax[4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int b = blockIdx.y;
  const int row = blockIdx.x * 64 + lane;
  const float* xb = x + (int64_t)b * N * ldx;
  const float* sqb = sq + (int64_t)b * N;
  const int rowc = row < N ? row : N - 1;
  float xi[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) xi[c] = (c < C) ? xb[(int64_t)rowc * ldx + c] : 0.0f;
  const float si = sqb[rowc];
  unsigned* myh = hist + w * HB_NB * 64 + lane;
#pragma unroll
  for (int q = 0; q < HB_NB; ++q) myh[q * 64] = 0u;
  // the cloud's largest s_j bounds every distance: d <= 2 (s_i + s_j) <= 4 max s (4.5: room for the roundings of d)
  float m = 0.f;
  for (int j = tid; j < N; j += 256) m = fmaxf(m, sqb[j]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (lane == 0) wmax[w] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  const int raw_top = (int)(__float_as_uint(4.5f * m) >> 22);
  const int bin0 = raw_top - (HB_NB - 1);
  const int ns = (N + STRIDE - 1) / STRIDE;            // sampled candidates j = STRIDE u, u < ns
#pragma unroll 1
  for (int u0 = 0; u0 < ns; u0 += 256) {
    __syncthreads();
    if (u0 + tid < ns) {
      const int64_t j = (int64_t)(u0 + tid) * STRIDE;
      const float* src = xb + j * ldx;
      tile[tid] = make_float4(src[0], (C > 1) ? src[1] : 0.f, (C > 2) ? src[2] : 0.f, (C > 3) ? src[3] : 0.f);
      ts[tid] = sqb[j];
    }
    __syncthreads();
    const int lo = 64 * w;
    const int hi = (ns - u0 < lo + 64) ? ns - u0 : lo + 64;
    // the scan's arithmetic, operation for operation (knn_kernel<4, KC>): a candidate counted below an edge here IS below it there
    auto bin_of = [&](const float4& v, float sj) -> int {
      float p = 0.f;
      p = fmaf(xi[0], v.x, p);
      p = fmaf(xi[1], v.y, p);
      p = fmaf(xi[2], v.z, p);
      p = fmaf(xi[3], v.w, p);
      const float tt = si + sj;
      const float tp = 2.0f * p;
      const float d = fmaxf(tt - tp, 0.0f);
      const int bin = (int)(__float_as_uint(d) >> 22) - bin0;
      return bin < 0 ? 0 : (bin > HB_NB - 1 ? HB_NB - 1 : bin);
    };
    int t = lo;
    // groups of 8: all LDS reads of a group first, then the arithmetic, then the 8 adds (hipcc will not move a read across a
    // ds_add that may alias it: candidate by candidate the loop was one LDS round trip per candidate)
    for (; t + 8 <= hi; t += 8) {
      float4 v[8];
      float sj[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { v[q] = tile[t + q]; sj[q] = ts[t + q]; }
      int bn[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) bn[q] = bin_of(v[q], sj[q]);
#pragma unroll
      for (int q = 0; q < 8; ++q) atomicAdd(&myh[bn[q] * 64], 1u);
    }
    for (; t < hi; ++t) atomicAdd(&myh[bin_of(tile[t], ts[t]) * 64], 1u);
  }
  __syncthreads();
  if (w != 0) return;
  // first bin at which the count reaches k; its upper edge bounds the row's k-th distance (the top bin has none: +inf)
  unsigned cum = 0u;
  int bsel = HB_NB - 1;
#pragma unroll 1
  for (int q = 0; q < HB_NB; ++q) {
    cum += hist[q * 64 + lane] + hist[(HB_NB + q) * 64 + lane] + hist[(2 * HB_NB + q) * 64 + lane] + hist[(3 * HB_NB + q) * 64 + lane];
    if (cum >= (unsigned)k) { bsel = q; break; }
  }
  float tau = INFINITY;
  const int eraw = bin0 + bsel + 1;
  if (bsel < HB_NB - 1 && eraw > 0 && eraw < 0x1FE) tau = __uint_as_float((unsigned)eraw << 22);
  if (row < N) tau0[(int64_t)b * N + row] = tau;
}

// Shared selection bound.  The candidates of a query row are split over FOUR sorted lists (KC entries each, KC % 4 == 0).
// If every list holds at least KC/4 entries, the row has seen KC candidates with d <= tau := max_i list_i[KC/4 - 1], so a
// candidate with d > tau has KC candidates strictly before it in (d, j) order and can never enter the row's top KC:
// dropping it is exact.  Candidates with d == tau pass (the tie is decided by index in the final merge), hence the
// filter  d < min(own k-th, next_up(tau)).  tau is about the row's GLOBAL k-th distance (the KC/4-th best of a quarter
// of the candidates), where a list's own k-th is about the global 4k-th: ~2.5x fewer inserts.  The lists publish
// list_i[KC/4 - 1] in LDS after every drain; a stale (older = larger) value only makes the filter looser.
// Register budget: x_i (CP) + list (2*KC) + ~70 for the candidate stream -> waves/SIMD target.
template <int CP, int KC>
constexpr int knn_min_waves() {
  return (CP + 2 * KC + 70 <= 128) ? 4 : ((CP + 2 * KC + 70 <= 168) ? 3 : ((CP + 2 * KC + 70 <= 256) ? 2 : 1));
}

template <int CP, int KC>
__global__ __launch_bounds__(256, (knn_min_waves<CP, KC>())) void knn_kernel(const float* __restrict__ x,
                                                                            const float* __restrict__ sq, int N, int C,
                                                                            int64_t ldx, int k, int vec_ok,
                                                                            int32_t* __restrict__ idx, const float* __restrict__ tau0) {
  constexpr int TILE_F = TJ * CP;
  constexpr int DQ_F = PERW * 256;           // per-lane distance slots of the current 32 candidates
  constexpr int MERGE_F = ROWS * KC * 2;
  constexpr int SH = (TILE_F + DQ_F > MERGE_F ? TILE_F + DQ_F : MERGE_F);
  __shared__ __attribute__((aligned(16))) float smem[SH + TJ + WAVES * ROWS];
  float* xs = smem;
  float* dq = smem + TILE_F;
  float* sjs = smem + SH;
  volatile float* thrw = smem + SH + TJ;     // [4 lists = waves][64 rows]: list[KC/4 - 1] so far

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int b = blockIdx.y;
  const int row = blockIdx.x * ROWS + lane;
  const float* xb = x + (int64_t)b * N * ldx;
  const float* sqb = sq + (int64_t)b * N;

  const int rowc = row < N ? row : N - 1;
  float xi[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) xi[c] = (c < C) ? xb[(int64_t)rowc * ldx + c] : 0.0f;
  const float si = sqb[rowc];
  // a bound known in advance (knn_hist_bound_kernel: the row's k-th distance is < tau_row): candidates at or above it never enter
  const float tau_row = tau0 ? tau0[(int64_t)b * N + rowc] : INFINITY;

  float dl[KC];
  int jl[KC];
#pragma unroll
  for (int t = 0; t < KC; ++t) {
    dl[t] = INFINITY;
    jl[t] = 0x7fffffff;
  }

  thrw[tid] = INFINITY;
#pragma unroll 1
  for (int j0 = 0; j0 < N; j0 += TJ) {
    __syncthreads();
    // ---- stage TJ candidate rows (zero padded to CP channels) into LDS ----
    for (int e = tid; e < TJ * (CP / 4); e += 256) {
      const int r = e / (CP / 4);
      const int c4 = (e % (CP / 4)) * 4;
      const int j = j0 + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < N) {
        const float* src = xb + (int64_t)j * ldx + c4;
        if (vec_ok && c4 + 3 < C) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          if (c4 + 0 < C) v.x = src[0];
          if (c4 + 1 < C) v.y = src[1];
          if (c4 + 2 < C) v.z = src[2];
          if (c4 + 3 < C) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(&xs[r * CP + c4]) = v;
    }
    if (tid < TJ) {
      const int j = j0 + tid;
      sjs[tid] = (j < N) ? sqb[j] : INFINITY;
    }
    __syncthreads();
    if (j0 + w * PERW >= N) continue;        // wave-uniform: nothing for this wave in the tile

    // ---- phase A: distances of this wave's 32 candidates (2 independent fmaf chains at a time;
    // the channel loop is fully unrolled so x_i stays in registers, fenced every 16 channels so
    // the scheduler cannot hoist every LDS read).  Each lane parks d in its LDS slot and keeps a
    // 32-bit mask of the candidates that beat its current k-th distance. ----
    const float thr = fminf(dl[KC - 1], tau_row);
    unsigned mask = 0u;
#pragma unroll 1
    for (int g = 0; g < PERW; g += 2) {
      const int jl0 = w * PERW + g;
      const float* c0 = &xs[jl0 * CP];
      float p0 = 0.f, p1 = 0.f;
#pragma unroll
      for (int c = 0; c < CP; c += 4) {
        const float4 v0 = *reinterpret_cast<const float4*>(c0 + c);
        const float4 v1 = *reinterpret_cast<const float4*>(c0 + CP + c);
        p0 = fmaf(xi[c], v0.x, p0); p1 = fmaf(xi[c], v1.x, p1);
        p0 = fmaf(xi[c + 1], v0.y, p0); p1 = fmaf(xi[c + 1], v1.y, p1);
        p0 = fmaf(xi[c + 2], v0.z, p0); p1 = fmaf(xi[c + 2], v1.z, p1);
        p0 = fmaf(xi[c + 3], v0.w, p0); p1 = fmaf(xi[c + 3], v1.w, p1);
        if ((c & 15) == 12) __builtin_amdgcn_sched_barrier(0);
      }
      const float t0 = si + sjs[jl0], t1 = si + sjs[jl0 + 1];
      const float tp0 = 2.0f * p0, tp1 = 2.0f * p1;
      const float d0 = t0 - tp0, d1 = t1 - tp1;
      dq[g * 256 + tid] = d0;
      dq[(g + 1) * 256 + tid] = d1;
      mask |= (unsigned)sel_i(m_flt(d0, thr), (int)(1u << g), 0);
      mask |= (unsigned)sel_i(m_flt(d1, thr), (int)(2u << g), 0);
    }
    // ---- phase B: drain.  Every iteration each lane pops ITS lowest surviving candidate (ascending
    // j, which the tie rule needs) and all lanes run one insert: max-over-lanes(popcount) inserts
    // per 32 candidates instead of one per candidate with any taker. ----
    while (__any(mask != 0u)) {
      const lmask_t live = m_ine((int)mask, 0);
      const int g = sel_i(live, __builtin_ctz(mask | 0x80000000u), 0);
      const float d = sel_f(live, dq[g * 256 + tid], INFINITY);
      mask &= mask - 1u;
      list_insert<KC, false>(dl, jl, d, j0 + w * PERW + g);
    }
  }

  // ---- merge the 4 per-wave lists into wave 0 through LDS (lexicographic (d, j)) ----
  __syncthreads();
  float* md = smem;
  int* mj = reinterpret_cast<int*>(smem + ROWS * KC);
#pragma unroll 1
  for (int src = 1; src < WAVES; ++src) {
    if (src > 1) __syncthreads();
    if (w == src) {
#pragma unroll
      for (int t = 0; t < KC; ++t) {
        md[t * ROWS + lane] = dl[t];
        mj[t * ROWS + lane] = jl[t];
      }
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll 1
      for (int t = 0; t < KC; ++t) {
        const float d = md[t * ROWS + lane];
        const int j = mj[t * ROWS + lane];
        const lmask_t need = key_less<true>(d, j, dl[KC - 1], jl[KC - 1]);
        if (need == 0) break;  // wave-uniform; the source list is ascending: nothing later can enter either
        list_insert<KC, true>(dl, jl, d, j);
      }
    }
  }
  if (w == 0 && row < N) {
    int32_t* out = idx + ((int64_t)b * N + row) * k;
#pragma unroll
    for (int t = 0; t < KC; ++t)
      if (t < k) out[t] = jl[t];
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA variant for feature-space graphs (C > 4): the inner products p_ij come from
// v_mfma_f32_32x32x2_f32, which on gfx950 is bit-for-bit an fmaf chain in k order (one rounding per
// product, no wider accumulation; MI355X_MICROARCH.md "Matrix cores"): feeding c = 2s, 2s+1 at step
// s reproduces the normative chain exactly -- the bit-exact tests against oracle/knn_oracle.c run on
// this kernel.  The matrix pipe does the 2*N^2*C flops; the VALU is left for the selection.
//   A operand = candidates (k-major LDS tile [c][cand], one conflict-free ds_read_b32 per step),
//   B operand = this wave's 32 query rows, resident in CP/2 VGPRs for the whole kernel.
//   D layout: lane l holds query row (l & 31) and candidates (r&3) + 8(r>>2) + 4(l>>5), r = 0..15:
//   two lanes per row, each with its own register-resident sorted list over its candidate subset.
// Block = 64 query rows x 2 candidate halves (4 waves); 64-candidate LDS tiles, double buffered, next
// tile prefetched into registers under the MFMAs; 4 lists per row merged through LDS.
template <int CP, int KC, bool VEC>
__global__ __launch_bounds__(256, (KC <= 20 ? 3 : 2)) void knn_mfma_kernel(const float* __restrict__ x, const float* __restrict__ sq,
                                                                           int N, int C, int64_t ldx, int k,
                                                                           int32_t* __restrict__ idx, const float* __restrict__ tau0) {
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  constexpr int TJM = 64;                    // candidates per LDS tile: 32 per candidate-half wave
  constexpr int ST = TJM + 2;                // k-major candidate tile [CP][ST]
  constexpr int TILE_F = CP * ST;
  constexpr int DQ_F = 16 * 256;
  constexpr int MERGE_F = ROWS * KC * 2;
  constexpr int WORK_F = 2 * TILE_F + DQ_F + 2 * TJM + 4 * ROWS;
  constexpr int SH = (WORK_F > MERGE_F ? WORK_F : MERGE_F);
  constexpr int NV = (TJM * (CP / 4)) / 256;  // float4 staged per thread per tile
  static_assert(NV >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) float smem[SH];
  float* dq = smem + 2 * TILE_F;
  float* sjs = smem + 2 * TILE_F + DQ_F;     // [2][TJM]
  volatile float* thrw = smem + 2 * TILE_F + DQ_F + 2 * TJM;   // [4 lists][64 rows]: list[KC/4 - 1] so far

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int qg = w & 1;                      // which 32 query rows of the block
  const int cs = w >> 1;                     // which 32 candidates of every 64-candidate tile
  const int b = blockIdx.y;
  const int row = blockIdx.x * ROWS + qg * 32 + l31;
  const float* xb = x + (int64_t)b * N * ldx;
  const float* sqb = sq + (int64_t)b * N;
  const int rowc = row < N ? row : N - 1;

  float bq[CP / 2];                          // B operand: x_q[row][2s + h]
#pragma unroll
  for (int s2 = 0; s2 < CP / 2; ++s2) {
    const int c = 2 * s2 + h;
    bq[s2] = (c < C) ? xb[(int64_t)rowc * ldx + c] : 0.0f;
  }
  const float si = sqb[rowc];
  // seed bound (dgcnn_knn_seeded_f32): tau0 = the largest distance to k DISTINCT candidates, so the row's k-th distance is <= tau0
  // and a candidate with d > tau0 can never enter its top k (d == tau0 passes: ties are decided by index in the merge)
  const float t0 = tau0 ? next_up(tau0[(int64_t)b * N + rowc]) : INFINITY;

  float dl[KC];
  int jl[KC];
#pragma unroll
  for (int t = 0; t < KC; ++t) {
    dl[t] = INFINITY;
    jl[t] = 0x7fffffff;
  }

  // ---- candidate tiles: global -> registers one tile ahead, transposed ([c][cand]) into the other
  // LDS buffer after the current tile's MFMAs; one barrier per tile ----
  // VEC (16-byte aligned rows, C % 4 == 0): one unconditional float4 load per piece from a clamped, always valid address;
  // rows past N carry |x_j|^2 = +inf (their distance is +inf: never selected), channel quads past C are zeroed.
  float4 pre[NV];
  float pre_s = INFINITY;
  auto fetch = [&](int j0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + 256 * i;
      const int r = e / (CP / 4);
      const int c4 = (e % (CP / 4)) * 4;
      const int j = j0 + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (VEC) {
        const int jc = j < N ? j : N - 1;
        const int cc = c4 < C ? c4 : C - 4;
        const float4 t4 = *reinterpret_cast<const float4*>(xb + (int64_t)jc * ldx + cc);
        const bool ok = c4 < C;
        v.x = ok ? t4.x : 0.f; v.y = ok ? t4.y : 0.f; v.z = ok ? t4.z : 0.f; v.w = ok ? t4.w : 0.f;
      } else if (j < N) {
        const float* src = xb + (int64_t)j * ldx + c4;
        if (c4 + 0 < C) v.x = src[0];
        if (c4 + 1 < C) v.y = src[1];
        if (c4 + 2 < C) v.z = src[2];
        if (c4 + 3 < C) v.w = src[3];
      }
      pre[i] = v;
    }
    if (tid < TJM) pre_s = (j0 + tid < N) ? sqb[j0 + tid] : INFINITY;
  };
  auto stash = [&](int buf) {
    float* d = smem + buf * TILE_F;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + 256 * i;
      const int r = e / (CP / 4);
      const int c4 = (e % (CP / 4)) * 4;
      d[(c4 + 0) * ST + r] = pre[i].x;
      d[(c4 + 1) * ST + r] = pre[i].y;
      d[(c4 + 2) * ST + r] = pre[i].z;
      d[(c4 + 3) * ST + r] = pre[i].w;
    }
    if (tid < TJM) sjs[buf * TJM + tid] = pre_s;
  };

  const int nt = (N + TJM - 1) / TJM;
  fetch(0);
  stash(0);
  thrw[tid] = INFINITY;
  const int lid = (w >> 1) * 2 + h;          // list id of this lane among the 4 lists of its row (= `me` below)
  const int rslot = (w & 1) * 32 + l31;      // row within the block
  __syncthreads();

#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const int j0 = t * TJM;
    thrw[lid * ROWS + rslot] = dl[KC / 4 - 1];       // publish this list's KC/4-th entry (as of the previous tile)
    if (t + 1 < nt) fetch(j0 + TJM);
    const int cbase = cs * 32;
    if (j0 + cbase < N) {                    // wave-uniform
      const float* xsT = smem + buf * TILE_F;
      // the other three lists' KC/4-th entries (shared selection bound): requested before the chain, needed after it
      const float o1 = thrw[(lid ^ 1) * ROWS + rslot], o2 = thrw[(lid ^ 2) * ROWS + rslot], o3 = thrw[(lid ^ 3) * ROWS + rslot];
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* ap = xsT + h * ST + cbase + l31;
      {
        // A operand ring, two MFMAs deep: the operand of step s + 2 is requested right behind MFMA s, whose 64 clocks in
        // the pipe (the chain is dependent) cover the LDS round trip.  (Round 1 staged 8 + 8 operands: 12 registers more.)
        constexpr int NS = CP / 2;
        float a0 = ap[0], a1 = ap[2 * ST];
#pragma unroll
        for (int s2 = 0; s2 < NS; s2 += 2) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bq[s2], acc, 0, 0, 0);
          if (s2 + 2 < NS) a0 = ap[2 * (s2 + 2) * ST];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bq[s2 + 1], acc, 0, 0, 0);
          if (s2 + 3 < NS) a1 = ap[2 * (s2 + 3) * ST];
        }
      }

      // ---- distances of this lane's 16 candidates; park them, flag the ones that can still enter the row's top KC ----
      const float thr = fminf(fminf(dl[KC - 1], t0), next_up(fmaxf(fmaxf(dl[KC / 4 - 1], o1), fmaxf(o2, o3))));
      const float* sj = sjs + buf * TJM + cbase + 4 * h;
      unsigned mask = 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 s4 = *reinterpret_cast<const float4*>(sj + 8 * q);     // candidates 8 q + 4 h + (0..3)
        const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e;
          const float tt = si + sv[e];
          // (tt - 2 p in ONE rounding: 2 p is exact, so the fused form equals the oracle's  tt - fl(2 p)  bit for bit)
          const float d = __builtin_fmaf(-2.0f, acc[r], tt);
          dq[r * 256 + tid] = d;
          mask |= sel_01(m_flt(d, thr)) << r;
        }
      }
      // drain: the parked distance of the NEXT surviving candidate is fetched before the insert of
      // the current one (LDS round trip hidden behind ~90 VALU ops)
      int g = __builtin_ctz(mask | 0x80000000u) & 15;
      float dcur = dq[g * 256 + tid];
      while (__any(mask != 0u)) {
        const lmask_t live = m_ine((int)mask, 0);
        const int gc = g;
        mask &= mask - 1u;
        g = __builtin_ctz(mask | 0x80000000u) & 15;
        const float dnext = dq[g * 256 + tid];
        const float d = sel_f(live, dcur, INFINITY);
        const int i = (gc & 3) + 8 * (gc >> 2) + 4 * h;
        list_insert<KC, false>(dl, jl, d, j0 + cbase + i);
        dcur = dnext;
      }

    }
    if (t + 1 < nt) stash(buf ^ 1);
    __syncthreads();
  }

  // ---- merge: 4 lists per query row (2 lane halves x 2 candidate halves) -> lanes 0..31 of waves 0,1 ----
  __syncthreads();
  float* md = smem;
  int* mj = reinterpret_cast<int*>(smem + ROWS * KC);
  const int me = cs * 2 + h;                 // list id of this lane; id 0 is the destination
  const int slot = qg * 32 + l31;            // row within the block
#pragma unroll 1
  for (int src = 1; src < 4; ++src) {
    if (src > 1) __syncthreads();
    if (me == src) {
#pragma unroll
      for (int t = 0; t < KC; ++t) {
        md[t * ROWS + slot] = dl[t];
        mj[t * ROWS + slot] = jl[t];
      }
    }
    __syncthreads();
    if (cs == 0) {                           // wave-uniform; lanes with h == 1 idle along (d = +inf)
#pragma unroll 1
      for (int t = 0; t < KC; ++t) {
        const float d = (h == 0) ? md[t * ROWS + slot] : INFINITY;
        const int j = (h == 0) ? mj[t * ROWS + slot] : 0x7fffffff;
        const lmask_t need = key_less<true>(d, j, dl[KC - 1], jl[KC - 1]);
        if (need == 0) break;
        list_insert<KC, true>(dl, jl, d, j);
      }
    }
  }
  if (me == 0 && row < N) {
    int32_t* out = idx + ((int64_t)b * N + row) * k;
#pragma unroll
    for (int t = 0; t < KC; ++t)
      if (t < k) out[t] = jl[t];
  }
}

// ------------------------------------------------------------------------------------------------
// Large-N kernel (C = 64 feature graphs, N >= 8192): distances on the BF16 matrix pipe, exact fp32 only for the survivors.
//
// At N = 65536 the fp32 MFMA chain is ~80 % of the kernel above (almost nothing survives the filter), and on gfx950
// v_mfma_f32_32x32x2_f32 runs at the vector rate and blocks the vector pipe.  Here every 32 x 32 tile of inner products is
// computed APPROXIMATELY from the two leading bf16 terms of each operand (x = x1 + x2 + x3 exactly; products x1y1 + x1y2 +
// x2y1 on v_mfma_f32_32x32x16_bf16: 12 instructions of 32 clocks per tile instead of 32 of 64), with a rigorous bound:
//     |p' - P| <= 3.25 * 2^-16 * sum|x_c y_c|   (dropped terms x2y2, x3 y, (x1+x2) y3 and the fp32 accumulation)
//     |p  - P| <= C * 2^-24 * sum|x_c y_c|      (the oracle's fmaf chain p; P = exact inner product)
//  => |d' - d| <= 2 |p' - p| + 2^-22 t  <  2^-14 t,   t = fl(s_i + s_j)   (sum|xy| <= (S_i + S_j)/2; d, d' share t)
// A candidate can enter the list only if d < thr, hence only if d' < thr + eps with eps = 2^-13 t (2x margin).  For those
// (and only those) the NORMATIVE distance is recomputed on the VALU -- fmaf chain over c ascending from +0, x_i and x_j from
// fp32 LDS copies of the block's query rows and of the current candidate tile (a first version read x_j from global memory:
// 60 ms at (8,65536,64,20) against 9.8 ms without any re-check -- the re-check is latency, not work) -- and goes through the
// exact filter and the same insert as everywhere else.  Indices are therefore bit-identical to the oracle's; the tests that pin knn_mfma_kernel
// pin this kernel too (N = 16384 / 65536 compares, and every small-N case with the kernel forced on).
// No parking area: a survivor is identified by its mask bit, its distance is recomputed anyway.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// (x0, x1) -> packed bf16 pairs of the two leading split terms
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& m) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16);
  const float r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
}

template <int KC, bool UNUSED = true>
struct Bf16fCfg {
  static constexpr int OCC = (KC <= 20) ? 3 : 2;
};

template <int KC>
__global__ __launch_bounds__(256, (Bf16fCfg<KC>::OCC)) void knn_bf16f_kernel(const float* __restrict__ x, const float* __restrict__ sq,
                                                                            int N, int C, int64_t ldx, int k,
                                                                            int32_t* __restrict__ idx, const float* __restrict__ tau0) {
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  constexpr int CP = 64;
  constexpr int TJM = 64;
  constexpr unsigned CS = TJM * 16 + 16;     // bytes per 8-channel chunk of one bf16 plane: [cand][8 bf16], padded (bank shift of 4)
  constexpr unsigned PB = 8 * CS;            // bytes per plane (8 chunks = 64 channels)
  constexpr unsigned TILE_B = 2 * PB;        // two planes (single buffer: the MFMA phase of a tile is short, other blocks of the CU fill the refill)
  constexpr int RS = CP + 4;                 // fp32 row stride of the two re-check tiles: 16-byte aligned rows, bank = 4 row + c
  constexpr unsigned WORK_B = TILE_B + 4 * (2 * ROWS * RS + TJM + 4 * ROWS);
  constexpr unsigned MERGE_B = 4u * ROWS * KC * 2;
  constexpr unsigned SH_B = WORK_B > MERGE_B ? WORK_B : MERGE_B;
  __shared__ __attribute__((aligned(16))) char smem_raw[SH_B];
  float* xq = reinterpret_cast<float*>(smem_raw + TILE_B);         // [64 query rows][RS]    fp32, resident
  float* xc = xq + ROWS * RS;                                      // [64 candidates][RS]    fp32 copy of the current tile
  float* sjs = xc + TJM * RS;                                      // [TJM]
  volatile float* thrw = sjs + TJM;                                // [4 lists][64 rows]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int qg = w & 1;
  const int cs = w >> 1;
  const int b = blockIdx.y;
  const int row0 = blockIdx.x * ROWS;
  const int row = row0 + qg * 32 + l31;
  const float* xb = x + (int64_t)b * N * ldx;
  const float* sqb = sq + (int64_t)b * N;
  const int rowc = row < N ? row : N - 1;
  const int cbase = cs * 32;
  const int lid = cs * 2 + h;
  const int rslot = qg * 32 + l31;
  const float si = sqb[rowc];
  const float t0 = tau0 ? next_up(tau0[(int64_t)b * N + rowc]) : INFINITY;      // seed bound, as in knn_mfma_kernel
  const float* xi_row = xb + (int64_t)rowc * ldx;

  // B operand: this lane's query row, channels 16 s + 8 h + (0..7), two bf16 planes
  bf16x8 q1[4], q2[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = 16 * s + 8 * h + 4 * e;
      float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < C) t4 = *reinterpret_cast<const float4*>(xi_row + c);
      v[4 * e] = t4.x; v[4 * e + 1] = t4.y; v[4 * e + 2] = t4.z; v[4 * e + 3] = t4.w;
    }
    unsigned hh[4], mm[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2_pair(v[2 * e], v[2 * e + 1], hh[e], mm[e]);
    const uint4 H4 = make_uint4(hh[0], hh[1], hh[2], hh[3]), M4 = make_uint4(mm[0], mm[1], mm[2], mm[3]);
    q1[s] = *reinterpret_cast<const bf16x8*>(&H4);
    q2[s] = *reinterpret_cast<const bf16x8*>(&M4);
  }
  // fp32 copy of the block's query rows for the exact re-check
  for (int e = tid; e < ROWS * (CP / 4); e += 256) {
    const int r = e / (CP / 4), c4 = (e % (CP / 4)) * 4;
    const int rr = (row0 + r < N) ? row0 + r : N - 1;
    float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C) t4 = *reinterpret_cast<const float4*>(xb + (int64_t)rr * ldx + c4);
    *reinterpret_cast<float4*>(xq + r * RS + c4) = t4;
  }

  float dl[KC];
  int jl[KC];
#pragma unroll
  for (int t = 0; t < KC; ++t) {
    dl[t] = INFINITY;
    jl[t] = 0x7fffffff;
  }

  // candidate tile: 64 rows x 64 channels -> registers (float4 pieces) -> two bf16 planes in LDS
  constexpr int NV = (TJM * (CP / 4)) / 256;   // 4
  float4 pre[NV];
  float pre_s = INFINITY;
  auto fetch = [&](int j0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + 256 * i;
      const int r = e / (CP / 4);
      const int c4 = (e % (CP / 4)) * 4;
      const int j = j0 + r;
      const int jc = j < N ? j : N - 1;
      const int cc = c4 < C ? c4 : C - 4;
      const float4 t4 = *reinterpret_cast<const float4*>(xb + (int64_t)jc * ldx + cc);
      const bool ok = c4 < C;
      pre[i] = make_float4(ok ? t4.x : 0.f, ok ? t4.y : 0.f, ok ? t4.z : 0.f, ok ? t4.w : 0.f);
    }
    if (tid < TJM) pre_s = (j0 + tid < N) ? sqb[j0 + tid] : INFINITY;
  };
  auto stash = [&]() {
    char* base = smem_raw;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + 256 * i;
      const int r = e / (CP / 4);
      const int c4 = (e % (CP / 4)) * 4;
      unsigned h0, m0, h1, m1;
      split2_pair(pre[i].x, pre[i].y, h0, m0);
      split2_pair(pre[i].z, pre[i].w, h1, m1);
      const unsigned off = (unsigned)(c4 >> 3) * CS + (unsigned)r * 16u + (unsigned)(c4 & 4) * 2u;
      *reinterpret_cast<uint2*>(base + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(base + PB + off) = make_uint2(m0, m1);
      *reinterpret_cast<float4*>(xc + r * RS + c4) = pre[i];
    }
    if (tid < TJM) sjs[tid] = pre_s;
  };

  const int nt = (N + TJM - 1) / TJM;
  fetch(0);
  thrw[tid] = INFINITY;
  const unsigned a_off = (unsigned)h * CS + (unsigned)(cbase + l31) * 16u;     // chunk 2 s + h of candidate cbase + l31
  const float kappa = 1.0f / 8192.0f;                                          // eps = 2^-13 (s_i + s_j)

#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    const int j0 = t * TJM;
    if (t > 0) __syncthreads();              // every wave is done with tile t-1 (planes, fp32 copy)
    stash();                                 // tile t: registers -> LDS
    thrw[lid * ROWS + rslot] = dl[KC / 4 - 1];
    if (t + 1 < nt) fetch(j0 + TJM);         // tile t+1: global -> registers, in flight during this step
    __syncthreads();
    if (j0 + cbase < N) {                    // wave-uniform
      const float o1 = thrw[(lid ^ 1) * ROWS + rslot], o2 = thrw[(lid ^ 2) * ROWS + rslot], o3 = thrw[(lid ^ 3) * ROWS + rslot];
      const char* base = smem_raw + a_off;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + 2 * s * CS);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(base + PB + 2 * s * CS);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, q1[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, q2[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, q1[s], acc, 0, 0, 0);
      }
      // ---- conservative filter on the approximate distances
      const float thr = fminf(fminf(dl[KC - 1], t0), next_up(fmaxf(fmaxf(dl[KC / 4 - 1], o1), fmaxf(o2, o3))));
      const float* sj = sjs + cbase + 4 * h;
      unsigned mask = 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 s4 = *reinterpret_cast<const float4*>(sj + 8 * q);
        const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e;
          const float tt = si + sv[e];
          const float tp = 2.0f * acc[r];
          const float da = tt - tp;
          const float lim = fmaf(tt, kappa, thr);            // thr + eps(i, j); +inf while the list is not full
          mask |= sel_01(m_flt(da, lim)) << r;
        }
      }
      // ---- survivors: the normative distance on the VALU, the exact filter, the insert
      while (__any(mask != 0u)) {
        const lmask_t live = m_ine((int)mask, 0);
        const int g = __builtin_ctz(mask | 0x80000000u) & 15;
        mask &= mask - 1u;
        const int i = (g & 3) + 8 * (g >> 2) + 4 * h;
        const int j = j0 + cbase + i;
        const float* xj = xc + (cbase + i) * RS;             // fp32 copy of the candidate row (rows past N: zeros, s_j = +inf)
        const float* xi = xq + rslot * RS;
        float p = 0.f;
        // the oracle's chain, c ascending from +0 (C % 4 == 0).  C = 64 (every feature-space graph of the model): fully
        // unrolled with the row quads of step q + 2 requested before the four fmas of step q -- the chain is latency bound
        // (16 dependent LDS round trips otherwise; 0.68 -> 0.31 ms at (24,2048,64,20) with the kernel forced on).
        if (C == 64) {
          float4 a[3], v[3];
          a[0] = *reinterpret_cast<const float4*>(xi); v[0] = *reinterpret_cast<const float4*>(xj);
          a[1] = *reinterpret_cast<const float4*>(xi + 4); v[1] = *reinterpret_cast<const float4*>(xj + 4);
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            if (q + 2 < 16) {
              a[(q + 2) % 3] = *reinterpret_cast<const float4*>(xi + 4 * (q + 2));
              v[(q + 2) % 3] = *reinterpret_cast<const float4*>(xj + 4 * (q + 2));
            }
            const float4 aa = a[q % 3], vv = v[q % 3];
            p = fmaf(aa.x, vv.x, p); p = fmaf(aa.y, vv.y, p); p = fmaf(aa.z, vv.z, p); p = fmaf(aa.w, vv.w, p);
          }
        } else {
#pragma unroll 2
          for (int c0 = 0; c0 < C; c0 += 4) {
            const float4 aa = *reinterpret_cast<const float4*>(xi + c0);
            const float4 vv = *reinterpret_cast<const float4*>(xj + c0);
            p = fmaf(aa.x, vv.x, p); p = fmaf(aa.y, vv.y, p); p = fmaf(aa.z, vv.z, p); p = fmaf(aa.w, vv.w, p);
          }
        }
        const float tt = si + sjs[cbase + i];
        const float tp = 2.0f * p;
        const float d = tt - tp;
        const lmask_t pass = live & m_flt(d, thr);           // exact filter (thr as of the start of the tile: conservative)
        list_insert<KC, false>(dl, jl, sel_f(pass, d, INFINITY), j);
      }
    }
  }

  // ---- merge: 4 lists per query row -> lanes 0..31 of waves 0,1 ----
  __syncthreads();
  float* md = reinterpret_cast<float*>(smem_raw);
  int* mj = reinterpret_cast<int*>(smem_raw) + ROWS * KC;
#pragma unroll 1
  for (int src = 1; src < 4; ++src) {
    if (src > 1) __syncthreads();
    if (lid == src) {
#pragma unroll
      for (int t = 0; t < KC; ++t) {
        md[t * ROWS + rslot] = dl[t];
        mj[t * ROWS + rslot] = jl[t];
      }
    }
    __syncthreads();
    if (cs == 0) {
#pragma unroll 1
      for (int t = 0; t < KC; ++t) {
        const float d = (h == 0) ? md[t * ROWS + rslot] : INFINITY;
        const int j = (h == 0) ? mj[t * ROWS + rslot] : 0x7fffffff;
        const lmask_t need = key_less<true>(d, j, dl[KC - 1], jl[KC - 1]);
        if (need == 0) break;
        list_insert<KC, true>(dl, jl, d, j);
      }
    }
  }
  if (lid == 0 && row < N) {
    int32_t* out = idx + ((int64_t)b * N + row) * k;
#pragma unroll
    for (int t = 0; t < KC; ++t)
      if (t < k) out[t] = jl[t];
  }
}

// ------------------------------------------------------------------------------------------------
// Seeded scan, append form (round 5).  With a rigorous bound tau_r >= the row's k-th distance known IN ADVANCE (the seed bound), the
// scan does not need sorted lists at all: a candidate is either farther than tau_r (dropped) or it is APPENDED, unsorted, to the row's
// candidate buffer in global memory -- a superset of the row's true top k, ~1.7 ... 3 k entries (profiles/r05/knn_seed.txt) -- and a
// second kernel picks the k smallest of every buffer in (d, j) order.  What that removes from the bf16-filter kernel above is its
// dominant cost at N = 2048 (116 of 168 us; the filter alone runs in 52): the per-lane sorted insert (90 VALU ops per round, rounds
// = the maximum over 64 lanes of a count whose mean is 0.4), and most of the exact re-check, because the survivors of a wave are
// first COMPACTED across lanes through a small LDS queue (any lane re-checks any (row, candidate) pair: both rows are in LDS): one
// re-check round per tile at ~40 % lane use instead of ~2.5 rounds at 15 %.
//   entry  = 64-bit key (orderable bits of d << 32 | j): unsigned order == (d, j) lexicographic order (d never NaN / -0)
//   buffer = cap entries per row, cap >= k + 64; invariant at the start of every tile: count <= cap - 64 (a tile adds at most 64).
//            A row that crosses the mark is compacted at the next tile boundary by one wave: its k smallest entries stay, its bound
//            becomes their largest distance (rigorous again: k candidates at distance <= it).  Rows without a usable seed (tau0 =
//            +inf) work the same way -- they just compact after their first tiles.
// Append order is arbitrary (LDS atomics), the result is not: the selection is by a total order on distinct keys, and WHEN a row
// compacts depends only on its counts at tile boundaries.  Distances are the normative ones (same chain as above): bit-exact indices.
__device__ __forceinline__ unsigned long long knn_key(float d, int j) {
  const unsigned u = __float_as_uint(d + 0.0f);
  const unsigned o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)o << 32) | (unsigned)j;
}
__device__ __forceinline__ float knn_key_dist(unsigned long long key) {
  const unsigned o = (unsigned)(key >> 32);
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

constexpr int KA_MAXE = 8;                   // entries per lane: cap <= 512
// Levels of the append-form scan's bound tightening: T 0.9^m, m = 0..6 (down to 0.53 T: the seed bound is ~1.3 ... 1.5 x the true
// k-th distance, profiles/r05/knn_seed.txt).  The SAME float product T * ka_level(m) is what candidates are counted under and what
// the bound becomes, so the rounding of the constants does not matter.
constexpr int KA_SLACK = 192;                // appends a row can receive between two tile boundaries: 2 waves x (32 new + < 64 waiting)
constexpr int KA_SLACK_LX = 64;              // ... when nothing waits across tiles: the tile's 64 candidates
constexpr float KA_C1 = 0.5f * (1.0f - 1.0f / 8192.0f);
constexpr int KA_LEVELS = 7;
__device__ __forceinline__ constexpr float ka_level(int m) {
  constexpr float L[KA_LEVELS] = {1.0f, 0.9f, 0.81f, 0.729f, 0.6561f, 0.59049f, 0.531441f};
  return L[m];
}

// value of lane (l ^ M): quad permutes and bank-masked row shifts on the DPP path for M < 16 (no LDS round trip, no address register),
// ds_swizzle for 16, ds_bpermute for 32
template <int M>
__device__ __forceinline__ unsigned xchg32(unsigned v) {
  if constexpr (M == 1) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  else if constexpr (M == 4) {
    const int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);       // row_shl:4 into banks 0, 2 (lanes with bit 2 clear)
    return (unsigned)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);         // row_shr:4 into banks 1, 3
  } else if constexpr (M == 8) {
    const int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x108, 0xF, 0x3, false);       // row_shl:8 into banks 0, 1
    return (unsigned)__builtin_amdgcn_update_dpp(t, (int)v, 0x118, 0xF, 0xC, false);         // row_shr:8 into banks 2, 3
  } else if constexpr (M == 16) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);   // bit mode: and 0x1f, or 0, xor 16
  else return __shfl_xor(v, M, 64);
}
template <int M>
__device__ __forceinline__ unsigned long long xchg64(unsigned long long v) {
  const unsigned lo = xchg32<M>((unsigned)v), hi = xchg32<M>((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
template <int J2>
__device__ __forceinline__ unsigned long long bitonic_step(unsigned long long key, int lane, bool up) {
  const unsigned long long pk = xchg64<J2>(key);
  const bool take_min = ((lane & J2) == 0) == up;
  return (take_min == (pk < key)) ? pk : key;
}
template <int J2>
__device__ __forceinline__ unsigned long long bitonic_steps_down(unsigned long long key, int lane, bool up) {
  key = bitonic_step<J2>(key, lane, up);
  if constexpr (J2 > 1) key = bitonic_steps_down<J2 / 2>(key, lane, up);
  return key;
}
// bitonic merge: a bitonic sequence of one key per lane -> ascending by lane
__device__ __forceinline__ unsigned long long wave_merge64(unsigned long long key, int lane) {
  return bitonic_steps_down<32>(key, lane, true);
}
// bitonic sort of one key per lane across the wave, ascending by lane
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long key, int lane) {
  key = bitonic_steps_down<1>(key, lane, (lane & 2) == 0);
  key = bitonic_steps_down<2>(key, lane, (lane & 4) == 0);
  key = bitonic_steps_down<4>(key, lane, (lane & 8) == 0);
  key = bitonic_steps_down<8>(key, lane, (lane & 16) == 0);
  key = bitonic_steps_down<16>(key, lane, (lane & 32) == 0);
  return wave_merge64(key, lane);
}
// One wave: the 64 smallest of the S (<= 64 E) keys of a row, ascending by lane.  key[q] = entry lane + 64 q (~0 beyond S): every
// register is sorted across the lanes, then folded into the running lowest 64: min(a[l], b[63 - l]) is a bitonic sequence that
// holds the 64 smallest of a and b (the half-cleaner), one merge sorts it.
template <int E>
__device__ __forceinline__ unsigned long long wave_lowest64(const unsigned long long (&key)[E], int S, int lane) {
  unsigned long long res = wave_sort64(key[0], lane);
#pragma unroll
  for (int q = 1; q < E; ++q) {
    if (64 * q < S) {                        // wave-uniform
      const unsigned long long sq = wave_sort64(key[q], lane);
      const unsigned lo = __shfl((unsigned)sq, 63 - lane, 64), hi = __shfl((unsigned)(sq >> 32), 63 - lane, 64);
      const unsigned long long rb = ((unsigned long long)hi << 32) | lo;
      res = wave_merge64(rb < res ? rb : res, lane);
    }
  }
  return res;
}

// LX: the candidates' fp32 rows are kept in LDS next to the tile and a wave's survivors are re-checked tile by tile (N < 8192: ~10
// pairs per wave and tile, LDS is the cheaper source); !LX: pairs wait in the queue across tiles until 64 are there and read the
// candidate row from global memory (L2) -- at N = 65536 a wave meets a survivor every other tile.
// NPR = products of the filter's inner product: 3 = a1 q1 + a1 q2 + a2 q1 (two bf16 terms per operand; |d' - d| <= 2^-14 t), 1 = a1 q1
// only (plain bf16 operands, unit roundoff 2^-8: |p' - p| <= sum |a q| (2^-7 + 2^-16) <= (t / 2) 2^-7 (1 + 2^-9), fp32 accumulation of
// exact products on top: |d' - d| <= 2^-7 t (1 + 2^-8); tested with 2^-6 t): a third of the MFMAs, a quarter of the staging
// arithmetic, one plane in LDS -- for a wider margin, i.e. more pairs re-checked exactly.  The result is the same bit for bit (the
// re-check decides).
template <bool LX, int NPR = 3>
// N >= 8192 with the one-product filter: 4 workgroups per CU (128 VGPRs; the re-check's candidate ring 8 -> 4 deep keeps it out of
// scratch): (8,16384,64,40) 2048 workgroups = 2 rounds of 1024 instead of 2.67 of 768, 1.71 -> 1.54 ms (profiles/r06/knn_occ.txt)
__global__ __launch_bounds__(256, ((LX || NPR != 1) ? 3 : 4)) void knn_bf16a_kernel(const float* __restrict__ x, const float* __restrict__ sq, int N, int C,
                                                           int64_t ldx, int k, const float* __restrict__ tau0, int cap,
                                                           unsigned long long* __restrict__ ent, int* __restrict__ cnt,
                                                           int ka_tight_mask) {
  using f32x16 = __attribute__((ext_vector_type(16))) float;
  constexpr int CP = 64;
  constexpr int TJM = 64;
  constexpr unsigned CS = TJM * 16 + 16;
  constexpr unsigned PB = 8 * CS;
  constexpr unsigned TILE_B = 2 * PB;
  constexpr int RS = CP + 4;                 // (LX) fp32 row stride of the candidate copy: bank = 4 row + c
  constexpr int QN = 128;                    // queue entries per wave
  using qent_t = typename std::conditional<LX, unsigned short, unsigned>::type;
  // [planes | xq (64 query rows x 64 fp32, 16-byte granules XOR-swizzled by the row) | LX: xc (the tile's rows, fp32, padded) |
  //  sjs | thr_s | cnt_s | hist | queue | flags]
  constexpr unsigned SH_B = TILE_B + 4 * (ROWS * CP + (LX ? TJM * RS : 0) + TJM + ROWS + ROWS + 4 * ROWS) + sizeof(qent_t) * 4 * QN + 16;
  __shared__ __attribute__((aligned(16))) char smem_raw[SH_B];
  float* xq = reinterpret_cast<float*>(smem_raw + TILE_B);
  float* xc = xq + ROWS * CP;
  float* sjs = xc + (LX ? TJM * RS : 0);
  volatile float* thr_s = sjs + TJM;                               // [64 rows]: a candidate stays iff d < thr_s
  int* cnt_s = reinterpret_cast<int*>(const_cast<float*>(thr_s) + ROWS);   // [64 rows]: entries in the row's buffer
  // [64 rows][4 words]: level histogram of the row's candidates (below).  word 0 = bin 0 << 16 | upper half of T's bits, words 1-3 =
  // bins 1..6, two 16-bit counters each
  unsigned* hist = reinterpret_cast<unsigned*>(cnt_s + ROWS);
  // per wave: survivors of the conservative filter waiting for their exact distance: (row of the wave, candidate) as
  // LX: row << 6 | candidate of the tile;  !LX: row << 27 | candidate of the cloud
  volatile qent_t* queue = reinterpret_cast<qent_t*>(hist + 4 * ROWS);
  volatile int* flags = reinterpret_cast<int*>(const_cast<qent_t*>(queue) + 4 * QN);   // [2]: a row crossed the mark in tile t (t & 1)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int qg = w & 1;
  const int cs = w >> 1;
  const int b = blockIdx.y;
  const int row0 = blockIdx.x * ROWS;
  const int row = row0 + qg * 32 + l31;
  const float* xb = x + (int64_t)b * N * ldx;
  const float* sqb = sq + (int64_t)b * N;
  const int rowc = row < N ? row : N - 1;
  const int cbase = cs * 32;
  const int rslot = qg * 32 + l31;
  const float si = sqb[rowc];
  const float* xi_row = xb + (int64_t)rowc * ldx;
  const int64_t grow0 = (int64_t)b * N + row0;                     // global row of the block's first query row
  const int trig = cap - (LX ? KA_SLACK_LX : KA_SLACK);
  volatile qent_t* myq = queue + w * QN;
  constexpr float C1 = (NPR == 3) ? KA_C1 : 0.5f * (1.0f - 1.0f / 64.0f);       // the filter's (1 - E') / 2
  const int tight_mask = ka_tight_mask;

  bf16x8 q1[4], q2[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = 16 * s + 8 * h + 4 * e;
      float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < C) t4 = *reinterpret_cast<const float4*>(xi_row + c);
      v[4 * e] = t4.x; v[4 * e + 1] = t4.y; v[4 * e + 2] = t4.z; v[4 * e + 3] = t4.w;
    }
    unsigned hh[4], mm[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2_pair(v[2 * e], v[2 * e + 1], hh[e], mm[e]);
    const uint4 H4 = make_uint4(hh[0], hh[1], hh[2], hh[3]), M4 = make_uint4(mm[0], mm[1], mm[2], mm[3]);
    q1[s] = *reinterpret_cast<const bf16x8*>(&H4);
    q2[s] = *reinterpret_cast<const bf16x8*>(&M4);
  }
  for (int e = tid; e < ROWS * (CP / 4); e += 256) {
    const int r = e / (CP / 4), c4 = (e % (CP / 4)) * 4;
    const int rr = (row0 + r < N) ? row0 + r : N - 1;
    float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C) t4 = *reinterpret_cast<const float4*>(xb + (int64_t)rr * ldx + c4);
    *reinterpret_cast<float4*>(xq + r * CP + ((((c4 >> 2) ^ r) & 15) << 2)) = t4;
  }
  if (tid < ROWS) {
    const int rr = (row0 + tid < N) ? row0 + tid : N - 1;
    const float t0 = tau0[(int64_t)b * N + rr];
    thr_s[tid] = next_up(t0);
    cnt_s[tid] = 0;
    // levels T c_m, T = t0 truncated to its upper 16 bits (any positive T <= t0 serves; 16 bits fit beside a counter).  Rows
    // without a finite positive bound, and clouds whose candidates could overflow a 16-bit counter, do not count (T = 0).
    const bool lv = (t0 > 1e-30f) && (t0 < INFINITY) && (N <= 65536);
    hist[4 * tid] = lv ? (__float_as_uint(t0) >> 16) : 0u;
    hist[4 * tid + 1] = 0u; hist[4 * tid + 2] = 0u; hist[4 * tid + 3] = 0u;
  }
  if (tid < 2) flags[tid] = 0;

  const int fr = tid >> 4, fc4 = (tid & 15) << 2;
  constexpr int NV = (TJM * (CP / 4)) / 256;   // 4
  float4 pre[NV];
  float pre_s = INFINITY;
  auto fetch = [&](int j0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + 256 * i;
      const int r = e / (CP / 4);
      const int c4 = (e % (CP / 4)) * 4;
      const int j = j0 + r;
      const int jc = j < N ? j : N - 1;
      const int cc = c4 < C ? c4 : C - 4;
      const float4 t4 = *reinterpret_cast<const float4*>(xb + (int64_t)jc * ldx + cc);
      const bool ok = c4 < C;
      pre[i] = make_float4(ok ? t4.x : 0.f, ok ? t4.y : 0.f, ok ? t4.z : 0.f, ok ? t4.w : 0.f);
    }
    if (tid < TJM) pre_s = (j0 + tid < N) ? sqb[j0 + tid] : INFINITY;
  };
  auto stash = [&]() {
    char* base = smem_raw;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int r = fr + 16 * i;
      const int c4 = fc4;
      unsigned h0, m0, h1, m1;
      const unsigned off = (unsigned)(c4 >> 3) * CS + (unsigned)r * 16u + (unsigned)(c4 & 4) * 2u;
      if (NPR == 3) {
        split2_pair(pre[i].x, pre[i].y, h0, m0);
        split2_pair(pre[i].z, pre[i].w, h1, m1);
        *reinterpret_cast<uint2*>(base + PB + off) = make_uint2(m0, m1);
      } else {
        h0 = cvt_pk_bf16(pre[i].x, pre[i].y);
        h1 = cvt_pk_bf16(pre[i].z, pre[i].w);
      }
      *reinterpret_cast<uint2*>(base + off) = make_uint2(h0, h1);
      if (LX) *reinterpret_cast<float4*>(xc + r * RS + c4) = pre[i];
    }
    if (tid < TJM) sjs[tid] = pre_s * C1;             // the filter's per-candidate term (below); rows past N: +inf
  };

  // exact distance of the queued (row, candidate) pairs of this wave, one per lane, and the append.  x_i from the block's fp32 LDS
  // copy, x_j from global memory (L2: the cloud's rows) -- pairs wait in the queue until 64 of them are there, whatever tile they
  // came from (at N = 2048 a wave meets ~10 survivors per tile, at N = 65536 one every other tile: re-checking them tile by tile
  // runs the 64-step chain for a handful of lanes).  The row's bound is applied when the pair is appended, as it stands then.
  auto process = [&](int nvalid, int j0, int par) {
    const bool on = lane < nvalid;
    const unsigned e = on ? (unsigned)myq[lane] : 0u;
    const int rl = LX ? (int)(e >> 6) : (int)(e >> 27);
    const int j = LX ? j0 + (int)(e & 63u) : (int)(e & 0x7ffffffu);
    const int rs = qg * 32 + rl;
    const float* xi = xq + rs * CP;
    const int rx = (rs & 15) << 2;                                   // granule q of the row sits at float offset (4 q) ^ rx
    const float* xj = LX ? xc + (j - j0) * RS : xb + (int64_t)j * ldx;
    const float sjv = sqb[j];
    float p = 0.f;
    if (C == 64) {
      constexpr int VR = LX ? 3 : (NPR == 1 ? 4 : 8);                                 // ring of candidate quads: LDS two ahead, global eight ahead
      float4 v[VR], a[3];
#pragma unroll
      for (int q = 0; q < VR - (LX ? 1 : 0); ++q) v[q] = *reinterpret_cast<const float4*>(xj + 4 * q);
      a[0] = *reinterpret_cast<const float4*>(xi + (0 ^ rx));
      a[1] = *reinterpret_cast<const float4*>(xi + (4 ^ rx));
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (q + 2 < 16) a[(q + 2) % 3] = *reinterpret_cast<const float4*>(xi + ((4 * (q + 2)) ^ rx));
        if (LX && q + 2 < 16) v[(q + 2) % 3] = *reinterpret_cast<const float4*>(xj + 4 * (q + 2));
        const float4 aa = a[q % 3], vv = v[q % VR];
        if (!LX && q + VR < 16) v[q % VR] = *reinterpret_cast<const float4*>(xj + 4 * (q + VR));
        p = fmaf(aa.x, vv.x, p); p = fmaf(aa.y, vv.y, p); p = fmaf(aa.z, vv.z, p); p = fmaf(aa.w, vv.w, p);
      }
    } else {
#pragma unroll 2
      for (int c0 = 0; c0 < C; c0 += 4) {
        const float4 aa = *reinterpret_cast<const float4*>(xi + (c0 ^ rx));
        const float4 vv = *reinterpret_cast<const float4*>(xj + c0);
        p = fmaf(aa.x, vv.x, p); p = fmaf(aa.y, vv.y, p); p = fmaf(aa.z, vv.z, p); p = fmaf(aa.w, vv.w, p);
      }
    }
    const float sir = __shfl(si, rl, 64);                            // lane rl (h = 0) holds s_i of row rl of this wave
    const float tt = sir + sjv;
    const float tp = 2.0f * p;
    const float d = tt - tp;
    if (on && d < thr_s[rs]) {
      const int pos = atomicAdd(&cnt_s[rs], 1);                      // < cap by the invariant
      ent[(grow0 + rs) * cap + pos] = knn_key(d, j);
      if (pos >= trig) flags[par] = 1;
      // level histogram: the finest level T c_m the candidate lies under (the row itself is left out: one count less can
      // never overflow a 16-bit counter at N <= 65536)
      const float T = __uint_as_float(hist[4 * rs] << 16);
      int m = 0;
#pragma unroll
      for (int u = 0; u < KA_LEVELS; ++u) m += (d < T * ka_level(u)) ? 1 : 0;
      if (m > 0 && j != row0 + rs) atomicAdd(&hist[4 * rs + (m >> 1)], 1u << ((m & 1) << 4));      // bin m - 1 = 16-bit field m
    }
  };

  const int nt = (N + TJM - 1) / TJM;
  fetch(0);
  const unsigned a_off = (unsigned)h * CS + (unsigned)(cbase + l31) * 16u;
  int qn = 0;                                // pairs waiting in this wave's queue (wave-uniform)

#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    const int j0 = t * TJM;
    const int par = t & 1;
    __syncthreads();                         // every wave is done with tile t-1 (planes, fp32 copy, appends; first trip: the set-up)
    stash();
    if (t + 1 < nt) fetch(j0 + TJM);
    if (tid == 0) flags[par] = 0;            // (last read between the barriers of tile t-1, next set behind the barrier below)
    if (t > 0 && ((t & tight_mask) == 0 || t < 8) && lane < 16) {
      // (every tile for the first 8 tiles, where the bounds move most, then every (tight_mask + 1)-th: the ~80 instructions below
      //  were a third of a tile's vector work)
      // Bound tightening from the histogram: k candidates seen under T c_m  =>  the row's k-th distance is < T c_m, and a candidate
      // at d >= T c_m has k candidates strictly before it.  (Rows r = w + 4 lane': the wave that would also compact the row.)
      const int r = w + 4 * lane;
      const uint4 hw = *reinterpret_cast<const uint4*>(hist + 4 * r);
      const float T = __uint_as_float(hw.x << 16);
      const unsigned bin[KA_LEVELS] = {hw.x >> 16, hw.y & 0xffffu, hw.y >> 16, hw.z & 0xffffu, hw.z >> 16, hw.w & 0xffffu, hw.w >> 16};
      unsigned c = 0u;
      float tn = INFINITY;
#pragma unroll
      for (int u = KA_LEVELS - 1; u >= 0; --u) {
        c += bin[u];
        if (c >= (unsigned)k && tn == INFINITY) tn = T * ka_level(u);
      }
      if (T > 0.f && tn < thr_s[r]) thr_s[r] = tn;
    }
    if (t > 0 && flags[par ^ 1]) {           // block-uniform: some row crossed the mark in tile t-1 -> compact it to its k smallest
#pragma unroll 1
      for (int r = w; r < ROWS; r += 4) {
        const int S = cnt_s[r];
        if (S > trig) {                      // wave-uniform
          unsigned long long* buf = ent + (grow0 + r) * cap;
          unsigned long long key[KA_MAXE];
#pragma unroll
          for (int q = 0; q < KA_MAXE; ++q) {
            const int e = lane + 64 * q;
            key[q] = (e < S) ? __hip_atomic_load(buf + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
          }
          const unsigned long long best = wave_lowest64<KA_MAXE>(key, S, lane);       // lane l: the row's (l + 1)-th smallest key
          if (lane < k) buf[lane] = best;
          if (lane == k - 1) {
            thr_s[r] = next_up(knn_key_dist(best));
            cnt_s[r] = k;
          }
        }
      }
    }
    __syncthreads();
    if (j0 + cbase < N) {                    // wave-uniform
      const char* base = smem_raw + a_off;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(base + 2 * s * CS);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, q1[s], acc, 0, 0, 0);
        if (NPR == 3) {
          const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(base + PB + 2 * s * CS);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, q2[s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, q1[s], acc, 0, 0, 0);
        }
      }
      // Conservative filter on the approximate inner products a: the exact rule "d < thr" can only hold if  t - 2 a < thr + 2^-14 t
      // (t = s_i + s_j; bound on |d' - d| above), i.e.  a > (1 - 2^-14) t / 2 - thr / 2.  Tested with 2^-13 for 2^-14 and the right
      // side as  c1 s_j + (c1 s_i - thr / 2),  c1 = (1 - 2^-13) / 2: the roundings of that sum (< 2^-20 t) sit far inside the
      // 2^-15 t of slack; one add and one compare per candidate.  thr = +inf: everything passes; rows past N: nothing does.
      const float thr = (row < N) ? thr_s[rslot] : -INFINITY;
      const float gi = si * C1 - 0.5f * thr;
      const float* sj = sjs + cbase + 4 * h;
      unsigned mask = 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 s4 = *reinterpret_cast<const float4*>(sj + 8 * q);
        const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q + e;
          mask |= sel_01(m_flt(sv[e] + gi, acc[r])) << r;
        }
      }
      // survivors -> this wave's queue, compacted across lanes; a full batch of 64 is re-checked at once
      while (true) {
        const bool has = mask != 0u;
        const unsigned long long bal = __ballot(has);
        if (bal == 0ull) break;
        const int g = __builtin_ctz(mask | 0x80000000u) & 15;
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (has) {
          const unsigned ci = (unsigned)(cbase + (g & 3) + 8 * (g >> 2) + 4 * h);
          myq[qn + before] = LX ? (qent_t)(((unsigned)l31 << 6) | ci) : (qent_t)(((unsigned)l31 << 27) | ((unsigned)j0 + ci));
          mask &= mask - 1u;
        }
        qn += __popcll(bal);
        if (qn >= 64) {
          process(64, j0, par);
          qn -= 64;
          const qent_t mv = myq[64 + lane];
          if (lane < qn) myq[lane] = mv;
        }
      }
      if (LX && qn > 0) {                    // the tile's rows leave LDS at the next barrier
        process(qn, j0, par);
        qn = 0;
      }
    }
  }
  if (qn > 0) process(qn, 0, (nt - 1) & 1);
  __syncthreads();
  if (tid < ROWS && row0 + tid < N) cnt[grow0 + tid] = cnt_s[tid];
}

// the k smallest keys of every row's buffer, ascending: one wave per row
__global__ __launch_bounds__(256) void knn_select_kernel(const unsigned long long* __restrict__ ent, const int* __restrict__ cnt,
                                                         int64_t rows, int cap, int k, int32_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int S = cnt[row];
  const unsigned long long* buf = ent + row * cap;
  int32_t* out = idx + row * k;
  if (S <= 64) {
    unsigned long long key = (lane < S) ? buf[lane] : ~0ull;
    key = wave_sort64(key, lane);
    if (lane < k) out[lane] = (int32_t)(unsigned)key;
    return;
  }
  unsigned long long key[KA_MAXE];
#pragma unroll
  for (int q = 0; q < KA_MAXE; ++q) {
    const int e = lane + 64 * q;
    key[q] = (e < S) ? buf[e] : ~0ull;
  }
  const unsigned long long best = wave_lowest64<KA_MAXE>(key, S, lane);
  if (lane < k) out[lane] = (int32_t)(unsigned)best;
}

// Dispatch threshold: with the pipelined re-check the kernel also wins in isolation at (24,2048,64,20) (0.253 vs 0.272 ms), but
// inside the training step it LOSES there (5.19 vs 5.05 ms/step with the side stream on, 5.06 vs 5.11 with it off): its 54 KB x 3
// workgroups leave no LDS for the transposed-adjacency build that runs on the side stream.  So: N >= 8192 only.
int g_knn_bf16f = -1;      // -1: not resolved; 0 never; 1 always (where applicable); 2 auto (N >= 8192)
int knn_bf16f_mode() {
  if (g_knn_bf16f < 0) {
    const char* e = getenv("DGCNN_KNN_BF16F");   // A/B switch: 0 | 1 | auto
    g_knn_bf16f = e ? ((e[0] == 'a') ? 2 : atoi(e)) : 2;
  }
  return g_knn_bf16f;
}

int g_knn_valu = 0;            // dgcnn_knn_force_valu (tests): VALU fmaf distances for every C
bool knn_force_valu() { return g_knn_valu == 1; }

// tau0[row] >= max over the row's first k seed candidates j of D(row, j), D in the normative arithmetic of the scan kernels.
// One wave per query row; LP = C / 4 lanes per pair (lane = LP * slot + channel quad, 64 / LP pairs per step), every pair's
// candidate row read as LP consecutive float4 (whole lines), all steps' loads in flight before the first use.  The inner product
// is summed in a different order than the scan's fmaf chain (4 products per lane, then a butterfly over the LP lanes), so the
// value differs from the scan's by rounding only: both are within 64 * 2^-24 * sum|x_c y_c| <= 2^-19 (s_i + s_j) of the exact
// product; adding 2^-16 (s_i + s_j) (> 4x the worst case, and far below anything that matters to the filter) makes
//     tau0 = max_m [ (s_i + s_j) - 2 p~ + 2^-16 (s_i + s_j) ]
// an upper bound of the normative distances -- all the proof needs (k distinct candidates at distance <= tau0).
// Seeds that are not k DISTINCT in-range indices give +inf (no bound).
template <int LP, int MAXSTEPS>
__global__ __launch_bounds__(256) void knn_seed_bound_kernel(const float* __restrict__ x, const float* __restrict__ sq, int N,
                                                             int64_t ldx, const int32_t* __restrict__ seed, int64_t ldseed, int k,
                                                             int64_t rows, float* __restrict__ tau0) {
  constexpr int PPS = 64 / LP;                            // pairs per step
  constexpr int RW = 2;                                   // query rows per wave: their loads are all in flight together
  const int lane = threadIdx.x & 63;
  // XCD x (= block id % 8; each XCD has its own 4 MB L2) owns the x-th eighth of the rows: the neighbour rows its waves gather
  // are then those of a few clouds (1.5 MB at the headline shape) instead of all of them (12.6 MB through every L2: 38 us)
  const int64_t per = ((rows + 7) / 8 + 4 * RW - 1) / (4 * RW) * (4 * RW);       // rows per XCD, a whole number of blocks
  const int64_t g0 = (int64_t)(blockIdx.x & 7) * per + ((int64_t)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6)) * RW;
  if (g0 >= rows || ((int64_t)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6)) * RW >= per) return;       // (wave-uniform)
  const int slot = lane / LP, cq = lane % LP;
  const int steps = (k + PPS - 1) / PPS;
  int jj[RW];
  int64_t gr[RW];
  const float* xc[RW];                                    // the row's cloud: x and s_i of its first point (wave-uniform)
  const float* sc[RW];
#pragma unroll
  for (int w = 0; w < RW; ++w) {
    gr[w] = (g0 + w < rows) ? g0 + w : rows - 1;
    jj[w] = (lane < k) ? seed[gr[w] * ldseed + lane] : -1 - lane;      // the row's seeds, one per lane
    const int64_t cloud = (rows < 0x7fffffffll) ? (int64_t)((unsigned)gr[w] / (unsigned)N) : gr[w] / N;
    xc[w] = x + cloud * N * ldx;
    sc[w] = sq + cloud * N;
  }
  const unsigned uld = (unsigned)ldx;                     // (N * ldx < 2^31: checked by the host)
  float4 a[RW], v[RW][MAXSTEPS];
  float si[RW], sj[RW][MAXSTEPS];
#pragma unroll
  for (int w = 0; w < RW; ++w) {
    a[w] = *reinterpret_cast<const float4*>(x + gr[w] * ldx + 4 * cq);
    si[w] = sq[gr[w]];
#pragma unroll
    for (int s = 0; s < MAXSTEPS; ++s) {
      if (s < steps) {                                    // (wave-uniform)
        const int m = s * PPS + slot;
        int jm = __shfl(jj[w], m < 64 ? m : 63, 64);
        jm = (m < k && (unsigned)jm < (unsigned)N) ? jm : 0;
        v[w][s] = *reinterpret_cast<const float4*>(xc[w] + ((unsigned)jm * uld + 4u * (unsigned)cq));
        sj[w][s] = sc[w][jm];
      }
    }
  }
  // validity of the seeds (k in-range, pairwise distinct indices), under the loads: every seed is broadcast from its lane
  // (v_readlane -> SGPR) and must match exactly one of the k lanes
  const unsigned long long kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
  bool anybad[RW];
#pragma unroll
  for (int w = 0; w < RW; ++w) {
    bool b = false;
    for (int t = 0; t < k; ++t) {
      const int st = __builtin_amdgcn_readlane(jj[w], t);
      const unsigned long long eq = __ballot(jj[w] == st) & kmask;
      b = b || (__popcll(eq) != 1) || ((unsigned)st >= (unsigned)N);
    }
    anybad[w] = b;
  }
#pragma unroll
  for (int w = 0; w < RW; ++w) {
    float best = -INFINITY;
#pragma unroll
    for (int s = 0; s < MAXSTEPS; ++s) {
      if (s < steps) {
        float p = a[w].x * v[w][s].x;
        p = fmaf(a[w].y, v[w][s].y, p); p = fmaf(a[w].z, v[w][s].z, p); p = fmaf(a[w].w, v[w][s].w, p);
#pragma unroll
        for (int o = 1; o < LP; o <<= 1) p += __shfl_xor(p, o, 64);
        const float tt = si[w] + sj[w][s];
        const float d = fmaf(-2.0f, p, tt) + tt * (1.0f / 65536.0f);
        best = (s * PPS + slot < k) ? fmaxf(best, d) : best;
      }
    }
#pragma unroll
    for (int o = LP; o < 64; o <<= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
    if (lane == 0 && g0 + w < rows) tau0[g0 + w] = anybad[w] ? INFINITY : best;
  }
}

// C in {16, 32, 64} (LP = 4, 8, 16 lanes per pair), rows 16-byte aligned; k <= 64.  Returns false when the shape is not taken.
bool launch_seed_bound(const float* x, const float* sq, int B, int N, int C, int64_t ldx, const int32_t* seed, int64_t ldseed,
                       int k, float* tau0, hipStream_t st) {
  const int64_t rows = (int64_t)B * N;
  const int64_t per = dg::cdiv(dg::cdiv(rows, 8), 8) * 8;  // rows per XCD (blocks of 4 waves x 2 rows)
  const dim3 grid((unsigned)(per / 8 * 8));                // block id = 8 * (block within the XCD's share) + XCD
#define DG_SB(LPV, MS) dg::launch((knn_seed_bound_kernel<LPV, MS>), grid, dim3(256), 0, st, x, sq, N, ldx, seed, ldseed, k, rows, tau0)
  if (C == 64) { if (k <= 20) DG_SB(16, 5); else if (k <= 40) DG_SB(16, 10); else DG_SB(16, 16); }
  else if (C == 32) { if (k <= 24) DG_SB(8, 3); else DG_SB(8, 8); }
  else if (C == 16) { if (k <= 32) DG_SB(4, 2); else DG_SB(4, 4); }
  else return false;
#undef DG_SB
  return true;
}

template <int CP, int KC>
void launch_knn(const float* x, const float* sq, int B, int N, int C, int64_t ldx, int k, int vec_ok,
                int32_t* idx, const float* tau0, hipStream_t st) {
  dim3 grid((unsigned)dg::cdiv(N, ROWS), (unsigned)B);
  if constexpr (CP == 64) {                      // large feature-space graphs: bf16 matrix pipe + exact re-check of the survivors
    const int m = knn_bf16f_mode();
    if (!knn_force_valu() && vec_ok && C % 4 == 0 && C > 16 && (m == 1 || (m == 2 && (N >= 8192 || tau0)))) {
      dg::launch((knn_bf16f_kernel<KC>), grid, dim3(256), 0, st, x, sq, N, C, ldx, k, idx, tau0);
      return;
    }
  }
  if constexpr (CP >= 16 && CP <= 64) {
    if (!knn_force_valu()) {
      if (vec_ok && C % 4 == 0) dg::launch((knn_mfma_kernel<CP, KC, true>), grid, dim3(256), 0, st, x, sq, N, C, ldx, k, idx, tau0);
      else dg::launch((knn_mfma_kernel<CP, KC, false>), grid, dim3(256), 0, st, x, sq, N, C, ldx, k, idx, tau0);
      return;
    }
  }
  dg::launch((knn_kernel<CP, KC>), grid, dim3(256), 0, st, x, sq, N, C, ldx, k, vec_ok, idx, (CP <= 4) ? tau0 : (const float*)nullptr);
}

template <int CP>
int dispatch_k(const float* x, const float* sq, int B, int N, int C, int64_t ldx, int k, int vec_ok,
               int32_t* idx, const float* tau0, hipStream_t st) {
  if (k <= 8) launch_knn<CP, 8>(x, sq, B, N, C, ldx, k, vec_ok, idx, tau0, st);
  else if (k <= 20) launch_knn<CP, 20>(x, sq, B, N, C, ldx, k, vec_ok, idx, tau0, st);
  else if (k <= 40) launch_knn<CP, 40>(x, sq, B, N, C, ldx, k, vec_ok, idx, tau0, st);
  else launch_knn<CP, 64>(x, sq, B, N, C, ldx, k, vec_ok, idx, tau0, st);
  return dg::check_launch("dgcnn_knn_f32");
}

}  // namespace

extern "C" int dgcnn_knn_force_valu(int on) {   // A/B switch (tests): 1 = VALU fmaf distances for every C, 0 = MFMA for C > 4
  const int prev = knn_force_valu() ? 1 : 0;
  g_knn_valu = on ? 1 : 0;
  return prev;
}

extern "C" int dgcnn_knn_bf16_filter(int mode) {   // A/B switch (tests): 0 never, 1 whenever applicable, 2 auto (N >= 8192); returns the previous mode
  const int prev = knn_bf16f_mode();
  g_knn_bf16f = mode;
  return prev;
}

namespace dg {
size_t knn_grid_workspace_bytes(int B, int N);
bool knn_grid_applicable(int C, int k);
int knn_grid_min_n();
int launch_knn_grid(const float* x, const float* sq, int B, int N, int C, int64_t ldx, int k, int32_t* idx, void* ws, hipStream_t st);
}  // namespace dg

// workspace = [s_i of every row (B*N floats, padded to 256 bytes) | seed bounds (same size) | scratch of the cell-grid search (C <= 4,
//              k <= 40) or of the append-form scan (16 < C <= 64: one count and knn_append_cap(k, N) 8-byte entries per row)]
static size_t knn_sq_bytes(int B, int N) { return (((size_t)B * (size_t)N * sizeof(float)) + 255) & ~(size_t)255; }
static bool knn_append_lx(int N) { return N < 8192; }     // which form of the append scan (knn_bf16a_kernel<LX>)
static int knn_append_cap(int k, int N) {    // entries per row buffer: >= k + slack of the form, <= 64 KA_MAXE
  if (knn_append_lx(N)) return k <= 20 ? 256 : (k <= 40 ? 384 : 512);
  return k <= 20 ? 320 : (k <= 40 ? 448 : 512);
}
static bool knn_append_shape(int C, int k) { return C > 16 && C <= 64 && C % 4 == 0 && k <= 64; }
static size_t knn_append_bytes(int B, int N, int k) {
  const size_t rows = (size_t)B * (size_t)N;
  return ((rows * sizeof(int) + 255) & ~(size_t)255) + rows * (size_t)knn_append_cap(k, N) * sizeof(unsigned long long);
}

extern "C" int64_t dgcnn_knn_workspace_bytes(int B, int N, int C, int k) {
  if (B <= 0 || N <= 0) return 0;
  size_t n = 2 * knn_sq_bytes(B, N);                 // s_i, and the seed bounds of dgcnn_knn_seeded_f32
  if (dg::knn_grid_applicable(C, k)) n += dg::knn_grid_workspace_bytes(B, N);
  else if (knn_append_shape(C, k)) n += knn_append_bytes(B, N, k);       // counts + candidate buffers of the append-form scan
  return (int64_t)n;
}

// Seeds are used (a) whenever the append-form scan takes the shape (C in {32, 64}: the bound is what makes it possible) and (b) by
// the list-keeping kernels from N ~ 4096 on: at (24,2048,64,20) the bound kernel (33 us: 252 MB of gathered rows) costs them what
// the scan saves; at (8,16384,64,40) the step goes 39.0 -> 35.8 ms, at (8,65536,64,20) 101 -> 97 (profiles/r05/knn_seed.txt).
static int g_knn_seed_min_n = -2;            // -2: not resolved; -1: auto (the rule above); >= 0: explicit
static int knn_seed_min_n() {
  if (g_knn_seed_min_n == -2) {
    const char* e = getenv("DGCNN_KNN_SEED_MIN_N");       // A/B switch
    g_knn_seed_min_n = e ? atoi(e) : -1;
  }
  return g_knn_seed_min_n;
}
extern "C" int dgcnn_knn_seed_min_n(int n) {            // tools / tests: smallest N for which seeds are used (-1: the rule); returns the previous value
  const int prev = knn_seed_min_n();
  if (n >= -1) g_knn_seed_min_n = n;
  return prev;
}

static int g_knn_append = -1;                // DGCNN_KNN_APPEND=0: seeded searches keep per-lane lists (A/B switch)
static bool knn_append_on() {
  if (g_knn_append < 0) {
    const char* e = getenv("DGCNN_KNN_APPEND");
    g_knn_append = e ? atoi(e) : 1;
  }
  return g_knn_append != 0;
}
extern "C" int dgcnn_knn_append(int on) {               // tools / tests; returns the previous setting
  const int prev = knn_append_on() ? 1 : 0;
  if (on >= 0) g_knn_append = on ? 1 : 0;
  return prev;
}

// products of the append-form scan's filter per form (LX: N < 8192): 1 since round 6 (profiles/r06/knn_npr.txt: 187 -> 176 us per call
// at (24,2048,64,20), 2.05 -> 1.82 ms at (8,16384,64,40), 22.3 -> 18.9 ms at (8,65536,64,20)); DGCNN_KNN_APPEND_NPR="<lx>,<big>"
// (A/B switch; tests run both)
static int g_knn_npr[2] = {-1, -1};
static int knn_append_products(bool lx) {
  if (g_knn_npr[0] < 0) {
    int a = 1, b = 1;
    const char* e = getenv("DGCNN_KNN_APPEND_NPR");
    if (e) sscanf(e, "%d,%d", &a, &b);
    g_knn_npr[0] = (a == 3) ? 3 : 1;
    g_knn_npr[1] = (b == 3) ? 3 : 1;
  }
  return g_knn_npr[lx ? 0 : 1];
}
extern "C" int dgcnn_knn_append_products(int n) {       // tools / tests: 1 or 3 for both forms (other values: query only); returns the previous setting of the N < 8192 form
  const int prev = knn_append_products(true);
  if (n == 1 || n == 3) g_knn_npr[0] = g_knn_npr[1] = n;
  return prev;
}

// raw-coordinate layer (C <= 4): sample stride of the histogram bound (knn_hist_bound_kernel); 0 = no bound.  DGCNN_KNN_HIST=<0|1|2|4>
static int g_knn_hist = -1;
static int knn_hist_stride(int N) {
  if (g_knn_hist < 0) {
    const char* e = getenv("DGCNN_KNN_HIST");
    g_knn_hist = e ? atoi(e) : 2;
    if (g_knn_hist != 0 && g_knn_hist != 1 && g_knn_hist != 2 && g_knn_hist != 4) g_knn_hist = 2;
  }
  return N >= 256 ? g_knn_hist : 0;
}
extern "C" int dgcnn_knn_hist(int stride) {              // tools / tests: 0 off, 1 / 2 / 4 = sample stride; other values only query; returns the previous setting
  const int prev = knn_hist_stride(1 << 20);
  if (stride == 0 || stride == 1 || stride == 2 || stride == 4) g_knn_hist = stride;
  return prev;
}

static int knn_impl(const char* what, const float* x, int B, int N, int C, int64_t ldx, int k, const int32_t* seed, int64_t ldseed,
                    int kseed, int32_t* idx, void* ws, size_t ws_bytes, void* stream) {
  DG_REQUIRE(x && idx && ws, DGCNN_EINVAL, "%s: null pointer", what);
  DG_REQUIRE(B > 0 && N > 0 && C > 0 && ldx >= C, DGCNN_EINVAL, "%s: bad shape B=%d N=%d C=%d", what, B, N, C);
  DG_REQUIRE(k > 0 && k <= N, DGCNN_EINVAL, "%s: k=%d must be in [1, N=%d] (tf.nn.top_k raises otherwise)", what, k, N);
  DG_REQUIRE(k <= 64, DGCNN_EUNSUP, "%s: k=%d > 64 unsupported", what, k);
  DG_REQUIRE(C <= 128, DGCNN_EUNSUP, "%s: C=%d > 128 unsupported", what, C);
  DG_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0 && ws_bytes >= knn_sq_bytes(B, N), DGCNN_EINVAL,
             "%s: workspace must be 16-byte aligned and hold dgcnn_knn_workspace_bytes(B, N, C, k) bytes (got %zu)", what, ws_bytes);
  hipStream_t st = (hipStream_t)stream;
  float* sq_ws = reinterpret_cast<float*>(ws);
  const int64_t rows = (int64_t)B * N;
  const int vec_ok = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  // the seed bound is used by the matrix-pipe scan kernels (4 < C <= 64) on float4-loadable rows; needs >= k seeds per row
  const size_t grid_off = 2 * knn_sq_bytes(B, N);
  const bool append = knn_append_on() && knn_append_shape(C, k) && vec_ok && !knn_force_valu() && knn_bf16f_mode() != 0 &&
                      ws_bytes >= grid_off + knn_append_bytes(B, N, k);
  const int min_n = knn_seed_min_n() >= 0 ? knn_seed_min_n() : (append ? 0 : 4096);
  const bool seeded = seed && kseed >= k && kseed <= 64 && C > 4 && C <= 64 && C % 4 == 0 && vec_ok && !knn_force_valu() &&
                      N >= min_n && (int64_t)N * ldx < ((int64_t)1 << 31) && ws_bytes >= 2 * knn_sq_bytes(B, N);
  // raw coordinates (C <= 4) when the caller provided the scratch: exact search over a uniform cell grid (knn_grid.hip)
  const bool grid_ws = dg::knn_grid_applicable(C, k) && !knn_force_valu() && ws_bytes >= grid_off + dg::knn_grid_workspace_bytes(B, N);
  dg::launch(sqnorm_kernel, dim3((unsigned)dg::cdiv(rows, SQ_ROWS)), dim3(256), sizeof(float) * SQ_ROWS * (C + 1), st, x,
                     ldx, rows, C, sq_ws);
  if (grid_ws && N >= dg::knn_grid_min_n())
    return dg::launch_knn_grid(x, sq_ws, B, N, C, ldx, k, idx, reinterpret_cast<char*>(ws) + grid_off, st);
  float* tau0 = nullptr;
  if (seeded) {
    tau0 = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + knn_sq_bytes(B, N));
    if (!launch_seed_bound(x, sq_ws, B, N, C, ldx, seed, ldseed, k, tau0, st)) tau0 = nullptr;      // (the first k seeds of every row)
  }
  if (tau0 && append) {                          // bound known in advance: append-form scan + one selection per row
    char* base = reinterpret_cast<char*>(ws) + grid_off;
    int* cnt = reinterpret_cast<int*>(base);
    unsigned long long* ent = reinterpret_cast<unsigned long long*>(base + (((size_t)rows * sizeof(int) + 255) & ~(size_t)255));
    const int cap = knn_append_cap(k, N);
    const dim3 grid((unsigned)dg::cdiv(N, ROWS), (unsigned)B);
    const int npr = knn_append_products(knn_append_lx(N));
    static int tight = -1;                     // DGCNN_KNN_TIGHTEN_EVERY (power of two; A/B switch): bound tightening every n-th tile
    if (tight < 0) { const char* e = getenv("DGCNN_KNN_TIGHTEN_EVERY"); tight = e ? atoi(e) : 0; if (tight < 0 || (tight & (tight - 1))) tight = 1; }
    // default: every 4th tile below N = 8192, every 8th above (profiles/r06/knn_tight.txt: 175 -> 168 us, 1.83 -> 1.70 ms, 19.2 -> 17.6 ms per call)
    const int tmask = (tight ? tight : (knn_append_lx(N) ? 4 : 8)) - 1;
#define DG_KA(LXV, NPRV) dg::launch((knn_bf16a_kernel<LXV, NPRV>), grid, dim3(256), 0, st, x, (const float*)sq_ws, N, C, ldx, k, (const float*)tau0, cap, ent, cnt, tmask)
    if (knn_append_lx(N)) { if (npr == 1) DG_KA(true, 1); else DG_KA(true, 3); }
    else { if (npr == 1) DG_KA(false, 1); else DG_KA(false, 3); }
#undef DG_KA
    dg::launch(knn_select_kernel, dim3((unsigned)dg::cdiv(rows, 4)), dim3(256), 0, st, (const unsigned long long*)ent, (const int*)cnt, rows,
               cap, k, idx);
    return dg::check_launch(what);
  }
  if (C <= 4) {
    // raw coordinates below the cell grid's range: a histogram bound first (the workspace's second s_i-sized region holds it)
    const int hs = knn_hist_stride(N);
    if (hs > 0 && !knn_force_valu() && k <= 64 && N >= 4 * k * hs && ws_bytes >= 2 * knn_sq_bytes(B, N)) {
      float* tb = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + knn_sq_bytes(B, N));
      const dim3 hg((unsigned)dg::cdiv(N, 64), (unsigned)B);
      if (hs == 1) dg::launch(knn_hist_bound_kernel<1>, hg, dim3(256), 0, st, x, (const float*)sq_ws, N, C, ldx, k, tb);
      else if (hs == 2) dg::launch(knn_hist_bound_kernel<2>, hg, dim3(256), 0, st, x, (const float*)sq_ws, N, C, ldx, k, tb);
      else dg::launch(knn_hist_bound_kernel<4>, hg, dim3(256), 0, st, x, (const float*)sq_ws, N, C, ldx, k, tb);
      tau0 = tb;
    }
    return dispatch_k<4>(x, sq_ws, B, N, C, ldx, k, vec_ok, idx, tau0, st);
  }
  if (C <= 16) return dispatch_k<16>(x, sq_ws, B, N, C, ldx, k, vec_ok, idx, tau0, st);
  if (C <= 64) return dispatch_k<64>(x, sq_ws, B, N, C, ldx, k, vec_ok, idx, tau0, st);
  return dispatch_k<128>(x, sq_ws, B, N, C, ldx, k, vec_ok, idx, tau0, st);
}

extern "C" int dgcnn_knn_f32(const float* x, int B, int N, int C, int64_t ldx, int k, int32_t* idx,
                             void* ws, size_t ws_bytes, void* stream) {
  return knn_impl("dgcnn_knn_f32", x, B, N, C, ldx, k, nullptr, 0, 0, idx, ws, ws_bytes, stream);
}

// The same search, seeded: `seed` (B, N, >= kseed; row stride ldseed) lists kseed >= k DISTINCT candidates of every row (any:
// the previous EdgeConv layer's graph in dgcnn/ops.py:95-96's stack).  Their largest distance bounds the row's k-th distance from
// above, so the scan can drop farther candidates from its first tile on instead of inserting them into lists that are still
// filling up (~k (1 + ln(N / k)) inserts per row become ~k + the candidates between the k-th distance and the bound).  The result
// is the unseeded one bit for bit; rows whose seeds are not distinct in-range indices simply get no bound.
extern "C" int dgcnn_knn_seeded_f32(const float* x, int B, int N, int C, int64_t ldx, int k, const int32_t* seed, int64_t ldseed,
                                    int kseed, int32_t* idx, void* ws, size_t ws_bytes, void* stream) {
  DG_REQUIRE(!seed || (ldseed >= kseed && kseed > 0), DGCNN_EINVAL, "dgcnn_knn_seeded_f32: bad seed shape");
  return knn_impl("dgcnn_knn_seeded_f32", x, B, N, C, ldx, k, seed, ldseed, kseed, idx, ws, ws_bytes, stream);
}
