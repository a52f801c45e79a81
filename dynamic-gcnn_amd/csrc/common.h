// common.h -- shared helpers of libdgcnn_hip.so (gfx950 only; no CUDA compatibility paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dgcnn_hip.h"

namespace dg {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DGCNN_ELAUNCH;
  }
  return DGCNN_OK;
}

#define DG_REQUIRE(cond, code, ...)          \
  do {                                       \
    if (!(cond)) {                           \
      dg::set_error(__VA_ARGS__);            \
      return (code);                         \
    }                                        \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Accumulation slots of every `stats` / `red` buffer (double[slots][2][F]): a producer workgroup adds its partial sums to slot
// (writer index % slots).  Default DGCNN_STAT_SLOTS; dgcnn_set_stat_slots(n) with n >= the number of writers gives every slot a
// single writer -- the sums then do not depend on the order in which workgroups finish (bit-reproducible runs), api.cc.
int stat_slots();
// a grid of `g` writers, capped so that every slot keeps a single writer when the reproducible configuration is on
static inline int64_t cap_writers(int64_t g) {
  const int64_t n = stat_slots();
  return (n > DGCNN_STAT_SLOTS && g > n) ? n : g;
}

// gemm_x3.hip: arithmetic of the plain GEMMs (0 native fp32 MFMA, 6 / 9 = partial products of the exact
// three-way bf16 split on the bf16 matrix pipe) and its launcher (pv = GemmP*, tiles already planned)
int gemm_arith();
void set_gemm_arith(int v);
int x3_tile_m(int M, int N, int K);
void launch_gemm_x3(int asrc, int bsrc, void* pv, hipStream_t st, int bn, int np);

}  // namespace dg

// ---- dropout (tf.nn.dropout(net, 0.7), model.py:91): counter-based mask, element i of the flattened tensor is kept iff
// mix32(seed * K + i) < keep * 2^32; the backward regenerates the mask from the same seed (misc.hip, bn.hip)
__device__ __forceinline__ uint32_t mix32(uint64_t z) {   // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

__device__ __forceinline__ bool dropout_keeps(uint64_t seed, uint64_t i, uint32_t thr) {
  return mix32(seed * 0xD1342543DE82EF95ull + i) < thr;
}
__device__ __forceinline__ uint32_t dropout_threshold(float keep) {
  return (keep >= 1.f) ? 0xffffffffu : (uint32_t)((double)keep * 4294967296.0);
}
