// knn_grid.hip -- K1 for low-dimensional clouds (C <= 4: the raw-coordinate layer of dgcnn/ops.py:8-19, points (x, y, z[, v])):
// the SAME k smallest (D_ij, j) as the brute-force kernels (knn.hip) and oracle/knn_oracle.c, bit for bit, without looking at
// all N candidates of a row.
//
//   s_i  = sequential sum of fl(x*x)     p_ij = fmaf chain over c ascending from +0     D_ij = fl( fl(s_i + s_j) - 2 p_ij )
//
// (-ffp-contract=off; the arithmetic per PAIR is exactly the brute-force kernel's, so a pair's D has the same bits whichever
// kernel evaluates it.)  What changes is WHICH pairs are evaluated:
//
//   build  (one 1024-thread block per cloud) bounding box of the first min(C, 3) coordinates, a uniform G^3 grid over it
//          (G^3 <= 4096 bins: histogram, scan and cursors in LDS), counting sort of the points by cell -- z fastest, so the cells
//          of a z column are one contiguous run of the sorted array.  Sorted records: (x, y, z, s_j) (C <= 3) or (x, y, z, v) + s_j.
//   query  (one wave per 64 consecutive SORTED points = neighbouring cells) every lane owns one query and walks the shells of
//          cells around its own cell: ring r = all cells within r of the own cell in every axis.  Candidates that may still enter
//          the lane's register-resident sorted (d, j) list are parked in LDS and drained by branch-free lexicographic inserts
//          (knn_common.h), all lanes together.  After ring r a lane is DONE when its k-th distance so far is smaller than
//          anything an unvisited point can have:
//              every unvisited point differs from the query by at least `gap` cells in some axis, so its true squared distance is
//              >= (gap * h)^2;  the computed D_ij is within  (2C + 4) u (s_i + s_j)  of the true value (u = 2^-24: C products and
//              C + 1 additions in s, C fused steps in p, two additions in D), the computed cell coordinate within 2.1 u * 2 G of
//              the true one.  With slack for all of that the test is
//                   kth < (max(0, gap - 1e-5) * h * (1 - 1e-6))^2 * (1 - 1e-6)  -  1e-6 (s_i + max_j s_j)
//          -- conservative, never tight: a lane that cannot prove it keeps walking (up to the whole grid = brute force: duplicate
//          points, clouds far from the origin, non-finite coordinates cost time, never correctness; a NaN makes every comparison
//          of the test false).
// Work per query at uniform density: the grid is sized for ~0.8 k points per cell, so ring 1 (27 cells, ~22 k candidates) decides
// almost every query: 432 pair evaluations instead of 2048 at (24, 2048, 3, 20), 864 instead of 16384 / 65536 at k = 40.
#include "knn_common.h"
#include <stdlib.h>

namespace {

constexpr int GMAX = 16;                 // cells per axis (G^3 <= 4096 LDS bins)
constexpr int PARK = 16;                 // parked candidates per lane between two drains
constexpr int UNR = 4;                   // candidates a lane looks at per wave iteration (loads in flight)
constexpr int QW = 4;                    // waves (of 64 queries) per query block
constexpr int LDS_CLOUD_MAX = 4096;      // clouds up to this many points are copied into LDS by every query block

struct GridInfo {                        // one per cloud (64 bytes)
  float mn[3];                           // lower corner of the bounding box
  float ih[3];                           // cells per unit length; 0 = the axis has a single cell (flat / non-finite / unused axis)
  float h[3];                            // (1 / ih) * (1 - 1e-6): a lower bound of the cell size
  float smax;                            // max_j s_j
  int G;
  int pad[5];
};

__device__ __forceinline__ int cell_axis(float x, float mn, float ih, int G) {
  const float t = (x - mn) * ih;         // two roundings (contraction is off for the whole library)
  int c = (int)t;                        // NaN -> 0, out of range saturates
  c = c < 0 ? 0 : c;
  return c > G - 1 ? G - 1 : c;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- (D, j) as ONE 64-bit key: the lexicographic order of the selection is the unsigned order of
//      key = ordered(D) << 32 | j,  ordered() = the usual order-preserving map float -> uint32 (after D + 0 so that -0 == +0).
// One v_cmp_lt_u64 per list slot instead of three compares and two mask operations (knn_common.h's pair form), and the
// candidate filter can compare keys exactly: a tie in D with a larger index never costs an insert round.
__device__ __forceinline__ unsigned f32_key(float d) {
  const unsigned u = __float_as_uint(d + 0.0f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__device__ __forceinline__ lmask_t m_ult64(unsigned long long a, unsigned long long b) { return __builtin_amdgcn_uicmpl(a, b, 36); }
__device__ __forceinline__ unsigned long long sel_u64(lmask_t m, unsigned long long t, unsigned long long f) {
  const unsigned tl = (unsigned)t, th = (unsigned)(t >> 32), fl = (unsigned)f, fh = (unsigned)(f >> 32);
  unsigned rl, rh;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(rl) : "v"(fl), "v"(tl), "s"(m));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(rh) : "v"(fh), "v"(th), "s"(m));
  return ((unsigned long long)rh << 32) | rl;
}
constexpr unsigned long long KEY_NONE = 0xff8000007fffffffull;      // (+inf, largest index): an empty slot

// Branch-free sorted insert of `key` into an ascending list of 64-bit keys in registers; a lane whose key is not smaller than its
// last entry is left unchanged (KEY_NONE is a no-op).
template <int KC>
__device__ __forceinline__ void list_insert64(unsigned long long (&kl)[KC], unsigned long long key) {
  lmask_t ct = m_ult64(key, kl[KC - 1]);
#pragma unroll
  for (int t = KC - 1; t >= 1; --t) {
    const lmask_t cp = m_ult64(key, kl[t - 1]);
    kl[t] = sel_u64(ct, sel_u64(cp, kl[t - 1], key), kl[t]);
    ct = cp;
  }
  kl[0] = sel_u64(ct, key, kl[0]);
}

__global__ __launch_bounds__(1024) void knn_grid_build_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ sq,
                                                              int N, int C, int G, float4* __restrict__ ps, float* __restrict__ s4,
                                                              int32_t* __restrict__ order, int32_t* __restrict__ cell_start,
                                                              GridInfo* __restrict__ info) {
  __shared__ int bins[GMAX * GMAX * GMAX];          // histogram, then cursors
  __shared__ int part[1024];
  __shared__ float red[7][16];
  __shared__ GridInfo gi;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x;
  const float* xb = x + (int64_t)b * N * ldx;
  const float* sqb = sq + (int64_t)b * N;
  const int D = C < 3 ? C : 3;
  const int G3 = G * G * G;

  // ---- bounding box of the grid axes, max s_j (fminf / fmaxf drop NaNs: such points land in cell 0 of their axis) ----
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, sm = 0.f;
  for (int j = tid; j < N; j += 1024) {
#pragma unroll
    for (int d = 0; d < 3; ++d)
      if (d < D) {
        const float v = xb[(int64_t)j * ldx + d];
        mn[d] = fminf(mn[d], v);
        mx[d] = fmaxf(mx[d], v);
      }
    sm = fmaxf(sm, sqb[j]);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    mn[d] = wave_min(mn[d]);
    mx[d] = wave_max(mx[d]);
  }
  sm = wave_max(sm);
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      red[d][wv] = mn[d];
      red[3 + d][wv] = mx[d];
    }
    red[6][wv] = sm;
  }
  for (int i = tid; i < G3; i += 1024) bins[i] = 0;
  __syncthreads();
  if (tid == 0) {
    float a[3] = {INFINITY, INFINITY, INFINITY}, z[3] = {-INFINITY, -INFINITY, -INFINITY}, s = 0.f;
    for (int w = 0; w < 16; ++w) {
      for (int d = 0; d < 3; ++d) {
        a[d] = fminf(a[d], red[d][w]);
        z[d] = fmaxf(z[d], red[3 + d][w]);
      }
      s = fmaxf(s, red[6][w]);
    }
    for (int d = 0; d < 3; ++d) {
      const float range = z[d] - a[d];
      const bool ok = d < D && G > 1 && range > 0.f && range < INFINITY;
      const float ih = ok ? (float)G / range : 0.f;
      gi.mn[d] = ok ? a[d] : 0.f;
      gi.ih[d] = (ih > 0.f && ih < INFINITY) ? ih : 0.f;
      gi.h[d] = gi.ih[d] > 0.f ? (1.0f / gi.ih[d]) * (1.0f - 1e-6f) : 0.f;
    }
    gi.smax = s;
    gi.G = G;
    for (int i = 0; i < 5; ++i) gi.pad[i] = 0;
    info[b] = gi;
  }
  __syncthreads();
  const GridInfo g = gi;
  auto cell_of = [&](int j) {
    int c[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < 3; ++d)
      if (d < D) c[d] = cell_axis(xb[(int64_t)j * ldx + d], g.mn[d], g.ih[d], G);
    return (c[0] * G + c[1]) * G + c[2];
  };
  for (int j = tid; j < N; j += 1024) atomicAdd(&bins[cell_of(j)], 1);
  __syncthreads();
  // ---- exclusive scan of the G^3 bins: every thread owns up to 4 consecutive bins ----
  const int per = (G3 + 1023) / 1024;
  const int lo = tid * per, hi = (lo + per < G3) ? lo + per : G3;
  int s = 0;
  for (int i = lo; i < hi; ++i) s += bins[i];
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = (tid >= d) ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - s;
  int32_t* cs = cell_start + (int64_t)b * (GMAX * GMAX * GMAX + 1);
  for (int i = lo; i < hi; ++i) {
    const int n = bins[i];
    cs[i] = run;
    bins[i] = run;                       // cursor
    run += n;
  }
  if (tid == 0) cs[G3] = N;
  __syncthreads();
  // ---- scatter (the order inside a cell is whatever the cursors hand out: the selection does not depend on it) ----
  for (int j = tid; j < N; j += 1024) {
    const int pos = atomicAdd(&bins[cell_of(j)], 1);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < 4; ++d)
      if (d < C) v[d] = xb[(int64_t)j * ldx + d];
    const float sj = sqb[j];
    if (C <= 3) v[3] = sj;
    ps[(int64_t)b * N + pos] = make_float4(v[0], v[1], v[2], v[3]);
    s4[(int64_t)b * N + pos] = sj;
    order[(int64_t)b * N + pos] = j;
  }
}

// LDSC: the block first copies the cloud's sorted records, original indices and cell table into LDS (N <= LDS_CLOUD_MAX): a lane's
// walk is a chain of dependent loads -- cell table -> run bounds -> candidates -- and at (24, 2048) there is less than one wave
// per SIMD to hide a global round trip behind (262 us with global loads against 155 us for the all-pairs kernel; LDS: see
// profiles/r04/knn_grid.txt).  Large clouds have thousands of waves and read the records through L1 / L2.
template <int KC, bool C4, bool LDSC>
__global__ __launch_bounds__(64 * QW) void knn_grid_query_kernel(const float4* __restrict__ ps, const float* __restrict__ s4,
                                                                 const int32_t* __restrict__ order, const int32_t* __restrict__ cell_start,
                                                                 const GridInfo* __restrict__ info, int N, int k, int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  float* dq = reinterpret_cast<float*>(smem) + wv * (PARK * 64);
  int* pj = reinterpret_cast<int*>(smem) + QW * (PARK * 64) + wv * (PARK * 64);
  const int b = blockIdx.y;
  const int q = blockIdx.x * (64 * QW) + tid;
  const bool valid = q < N;
  const int qc = valid ? q : N - 1;
  const GridInfo g = info[b];
  const int G = g.G;
  const float4* pbg = ps + (int64_t)b * N;
  const float* sbg = s4 + (int64_t)b * N;
  const int32_t* obg = order + (int64_t)b * N;
  const int32_t* csg = cell_start + (int64_t)b * (GMAX * GMAX * GMAX + 1);
  // LDS copies (LDSC): [records float4 N][order int N][cells int G^3 + 1][s_j float N (C4)]
  float4* pl = reinterpret_cast<float4*>(smem + 2 * QW * PARK * 64 * 4);
  int* ol = reinterpret_cast<int*>(pl + (LDSC ? N : 0));
  int* cl = ol + (LDSC ? N : 0);
  float* sl = reinterpret_cast<float*>(cl + (LDSC ? G * G * G + 1 : 0));
  if (LDSC) {
    for (int i = tid; i < N; i += 64 * QW) {
      pl[i] = pbg[i];
      ol[i] = obg[i];
      if (C4) sl[i] = sbg[i];
    }
    for (int i = tid; i <= G * G * G; i += 64 * QW) cl[i] = csg[i];
    __syncthreads();
  }
  auto rec = [&](int i) -> float4 { return LDSC ? pl[i] : pbg[i]; };
  auto sqv = [&](int i) -> float { return LDSC ? sl[i] : sbg[i]; };
  auto org = [&](int i) -> int { return LDSC ? ol[i] : obg[i]; };
  auto cst = [&](int i) -> int { return LDSC ? cl[i] : csg[i]; };

  const float4 P = rec(qc);
  const float xi0 = P.x, xi1 = P.y, xi2 = P.z, xi3 = C4 ? P.w : 0.f;
  const float si = C4 ? sqv(qc) : P.w;
  const float tq[3] = {(xi0 - g.mn[0]) * g.ih[0], (xi1 - g.mn[1]) * g.ih[1], (xi2 - g.mn[2]) * g.ih[2]};
  const int c0 = cell_axis(xi0, g.mn[0], g.ih[0], G), c1 = cell_axis(xi1, g.mn[1], g.ih[1], G), c2 = cell_axis(xi2, g.mn[2], g.ih[2], G);
  const float dmargin = 1e-6f * (si + g.smax);

  unsigned long long kl[KC];
#pragma unroll
  for (int t = 0; t < KC; ++t) kl[t] = KEY_NONE;
  float thr = INFINITY;                    // D of the list's last entry: candidates with d <= thr are parked
  int nb = 0;
  auto drain = [&]() {
    int i = 0;
    while (__any(i < nb)) {
      const bool live = i < nb;
      const unsigned kh = f32_key(live ? dq[i * 64 + lane] : INFINITY);
      const unsigned kj = (unsigned)(live ? pj[i * 64 + lane] : 0x7fffffff);
      ++i;
      list_insert64<KC>(kl, ((unsigned long long)kh << 32) | kj);
    }
    nb = 0;
    thr = key_f32((unsigned)(kl[KC - 1] >> 32));
  };

  bool done = !valid;
#pragma unroll 1
  for (int r = 0; r < GMAX; ++r) {
    // columns (dx, dy) of the ring, clipped to the grid; a column on the ring's rim (|dx| == r or |dy| == r) is one run of up to
    // 2 r + 1 cells along z, an inner column contributes its two end cells.  Stepped incrementally (no divisions: this loop runs
    // once per run and lane, and at N = 2048 it used to cost as much as the candidates themselves).
    const int dx0 = -r < -c0 ? -c0 : -r, dx1 = r > G - 1 - c0 ? G - 1 - c0 : r;
    const int dy0 = -r < -c1 ? -c1 : -r, dy1 = r > G - 1 - c1 ? G - 1 - c1 : r;
    int dx = dx0, dy = dy0, sub = 0, p = 0, e = 0;
    bool more = true;
#pragma unroll 1
    while (true) {
      // ---- lanes whose run is used up step to their next non-empty run of this ring ----
      while (p >= e && more && !done) {
        const bool rim = (dx == -r) || (dx == r) || (dy == -r) || (dy == r);
        const int base = ((c0 + dx) * G + (c1 + dy)) * G;
        int zlo, zhi;
        if (rim) {
          zlo = c2 - r < 0 ? 0 : c2 - r;
          zhi = c2 + r > G - 1 ? G - 1 : c2 + r;
        } else {
          zlo = zhi = sub ? c2 + r : c2 - r;
        }
        // next column / end cell
        if (!rim && sub == 0) {
          sub = 1;
        } else {
          sub = 0;
          if (++dy > dy1) {
            dy = dy0;
            if (++dx > dx1) more = false;
          }
        }
        if (zlo < 0 || zhi >= G) continue;
        p = cst(base + zlo);
        e = cst(base + zhi + 1);
      }
      const bool has = (p < e) && !done;
      if (!__any(has)) break;
      if (has) {
        float4 v[UNR];
        float sj[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int pp = (p + u < e) ? p + u : e - 1;
          v[u] = rec(pp);
          sj[u] = C4 ? sqv(pp) : v[u].w;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          float pr = fmaf(xi0, v[u].x, 0.f);
          pr = fmaf(xi1, v[u].y, pr);
          pr = fmaf(xi2, v[u].z, pr);
          pr = fmaf(xi3, C4 ? v[u].w : 0.f, pr);
          const float t0 = si + sj[u];
          const float tp = 2.0f * pr;
          const float d = t0 - tp;
          if (p + u < e && d <= thr) {
            dq[nb * 64 + lane] = d;
            pj[nb * 64 + lane] = org(p + u);
            ++nb;
          }
        }
        p += UNR;
      }
      if (__any(nb > PARK - UNR)) drain();
    }
    drain();
    // ---- can an unvisited point still enter this lane's list? ----
    unsigned long long kk = kl[KC - 1];
#pragma unroll
    for (int t = 0; t < KC - 1; ++t)
      if (t == k - 1) kk = kl[t];
    const float kth = key_f32((unsigned)(kk >> 32));
    const int cc[3] = {c0, c1, c2};
    float lb = INFINITY;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (g.ih[d] > 0.f) {
        const int lo = cc[d] - r, hi = cc[d] + r;
        if (lo > 0) lb = fminf(lb, fmaxf(0.f, (tq[d] - (float)lo) - 1e-5f) * g.h[d]);
        if (hi < G - 1) lb = fminf(lb, fmaxf(0.f, ((float)(hi + 1) - tq[d]) - 1e-5f) * g.h[d]);
      }
    }
    if (lb == INFINITY) done = true;                                  // the ring covered the whole grid
    else if (kth < lb * lb * (1.0f - 1e-6f) - dmargin) done = true;
    if (__all(done)) break;
  }

  if (valid) {
    int32_t* out = idx + ((int64_t)b * N + org(q)) * k;
#pragma unroll
    for (int t = 0; t < KC; ++t)
      if (t < k) out[t] = (int32_t)(unsigned)kl[t];
  }
}

bool g_knn_grid_all = false;
int g_knn_grid = -1;
bool knn_grid_on() {
  if (g_knn_grid < 0) {
    const char* e = getenv("DGCNN_KNN_GRID");      // A/B switch: 0 = brute-force kernel for C <= 4 as well
    g_knn_grid = (e && e[0] == '0') ? 0 : 1;
  }
  return g_knn_grid == 1;
}

}  // namespace

namespace dg {

size_t knn_grid_workspace_bytes(int B, int N) {
  const size_t rows = (size_t)B * (size_t)N;
  return rows * (sizeof(float4) + sizeof(float) + sizeof(int32_t)) + (size_t)B * ((GMAX * GMAX * GMAX + 1) * sizeof(int32_t) + sizeof(GridInfo)) + 256;
}

// C <= 4, k <= 40.  sq = the s_j of the cloud (already computed); ws >= knn_grid_workspace_bytes(B, N), 16-byte aligned.
bool knn_grid_applicable(int C, int k) { return knn_grid_on() && C <= 4 && k <= 40; }

// Where it pays (profiles/r04/knn_grid.txt): the walk costs ~22 k candidates per row whatever N is, the all-pairs kernel N.
// Measured cross-over between N = 2048 (all pairs 155 us, grid 207 us at B = 24: one 64-query wave per SIMD, bound by its own
// dependent instruction chain) and N = 4096; at N = 16384 / 65536 the grid is 3.1x / 14x faster.  $DGCNN_KNN_GRID_MIN_N moves it
// (tests run the grid at every N).
int knn_grid_min_n() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("DGCNN_KNN_GRID_MIN_N"); v = e ? atoi(e) : 4096; }
  return g_knn_grid_all ? 0 : v;
}

int launch_knn_grid(const float* x, const float* sq, int B, int N, int C, int64_t ldx, int k, int32_t* idx, void* ws, hipStream_t st) {
  char* w = reinterpret_cast<char*>(ws);
  const size_t rows = (size_t)B * (size_t)N;
  float4* ps = reinterpret_cast<float4*>(w);
  w += rows * sizeof(float4);
  float* s4 = reinterpret_cast<float*>(w);
  w += rows * sizeof(float);
  int32_t* order = reinterpret_cast<int32_t*>(w);
  w += rows * sizeof(int32_t);
  w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(w) + 63) & ~(uintptr_t)63);
  GridInfo* info = reinterpret_cast<GridInfo*>(w);
  w += (size_t)B * sizeof(GridInfo);
  int32_t* cell_start = reinterpret_cast<int32_t*>(w);
  // ~0.1 k points per cell (cell edge ~ 0.75 of the radius of a ball holding k points): rings are thin shells, a lane stops within
  // a fraction of a cell of its k-th distance.  Measured against 0.8 k / 0.2 k points per cell (profiles/r04/knn_grid.txt):
  // (8, 16384, 3, 40) 648 / 544 / 386 us, (24, 2048, 3, 20) 191 / 160 / 151 us; finer than that the empty runs cost more than the
  // pairs they save.  G <= GMAX = 16: clouds of 65536 points stay at 16 points per cell.
  int G = (int)floorf(cbrtf((float)N / (0.1f * (float)k)));
  G = G < 1 ? 1 : (G > GMAX ? GMAX : G);
  dg::launch(knn_grid_build_kernel, dim3((unsigned)B), dim3(1024), 0, st, x, ldx, sq, N, C, G, ps, s4, order, cell_start, info);
  dim3 grid((unsigned)cdiv(N, 64 * QW), (unsigned)B);
  const size_t park = (size_t)2 * QW * PARK * 64 * 4;
  const size_t G3 = (size_t)G * G * G;
  const size_t cloud = (size_t)N * (16 + 4 + (C == 4 ? 4 : 0)) + (G3 + 1) * 4;
  const bool in_lds = N <= LDS_CLOUD_MAX && park + cloud + 16 <= 160 * 1024;
#define DG_GRID2(KC, C4)                                                                                                          \
  do {                                                                                                                            \
    if (in_lds) {                                                                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_grid_query_kernel<KC, C4, true>),                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                          \
      dg::launch((knn_grid_query_kernel<KC, C4, true>), grid, dim3(64 * QW), park + cloud + 16, st, ps, s4, order,        \
                         cell_start, info, N, k, idx);                                                                            \
    } else {                                                                                                                      \
      dg::launch((knn_grid_query_kernel<KC, C4, false>), grid, dim3(64 * QW), park, st, ps, s4, order, cell_start, info,  \
                         N, k, idx);                                                                                              \
    }                                                                                                                             \
  } while (0)
#define DG_GRID(KC)                    \
  do {                                 \
    if (C == 4) DG_GRID2(KC, true);    \
    else DG_GRID2(KC, false);          \
  } while (0)
  if (k <= 8) DG_GRID(8);
  else if (k <= 20) DG_GRID(20);
  else DG_GRID(40);
#undef DG_GRID
#undef DG_GRID2
  return check_launch("dgcnn_knn_f32 (grid)");
}

}  // namespace dg

// 0 = never, 1 = where it pays (N >= knn_grid_min_n), 2 = whenever applicable (tests)
extern "C" int dgcnn_knn_grid(int mode) {
  const int prev = knn_grid_on() ? (g_knn_grid_all ? 2 : 1) : 0;
  g_knn_grid = mode ? 1 : 0;
  g_knn_grid_all = mode == 2;
  return prev;
}
