// common.h -- shared helpers of libdgcnn_hip.so (gfx950 only; no CUDA compatibility paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <tuple>
#include <utility>
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

// ---- launch plans (plan.cc) --------------------------------------------------------------------------------------------
// Every kernel launch, memset, cross-stream wait and collective of the library goes through the four functions below.  While a
// plan is being recorded (dgcnn_plan_begin .. dgcnn_plan_end) each of them ALSO appends itself -- kernel address, launch
// geometry, a private copy of the argument values, the stream -- to the plan; dgcnn_plan_replay then re-issues the whole list
// from one C loop.  The reference replays a static TF graph with sess.run (dgcnn/trainval.py:103-129); this is that, with plain
// stream semantics (same two-stream schedule as the eager step, RCCL calls included) and ~2 us of host time per launch.
bool plan_recording();
void plan_add_kernel(const void* fn, dim3 g, dim3 b, size_t sh, hipStream_t st, void** argv, const size_t* sizes, int n);
int memset_async(void* p, int value, size_t bytes, hipStream_t st);
int stream_wait(hipStream_t waiter, hipStream_t signaller);

template <class Tuple, size_t... I>
inline void launch_tuple(const void* fn, dim3 g, dim3 b, size_t sh, hipStream_t st, Tuple& args, std::index_sequence<I...>) {
  void* argv[] = {(void*)&std::get<I>(args)..., nullptr};
  (void)hipLaunchKernel(fn, g, b, argv, sh, st);
  if (plan_recording()) {
    const size_t sizes[] = {sizeof(std::get<I>(args))..., 0};
    plan_add_kernel(fn, g, b, sh, st, argv, sizes, (int)sizeof...(I));
  }
}

// launch(kernel, grid, block, dynamic LDS bytes, stream, kernel arguments...): arguments are converted to the kernel's
// parameter types exactly as a call would convert them
template <class... KA, class... A>
inline void launch(void (*k)(KA...), dim3 g, dim3 b, size_t sh, hipStream_t st, A&&... a) {
  static_assert(sizeof...(KA) == sizeof...(A), "dg::launch: argument count differs from the kernel's parameter list");
  std::tuple<KA...> args{static_cast<KA>(std::forward<A>(a))...};
  launch_tuple(reinterpret_cast<const void*>(k), g, b, sh, st, args, std::index_sequence_for<KA...>{});
}

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
int x3_tile_n(int M, int N, int K);      // 256: the 256 x 256 kernel takes this product; 0: column tile from gemm.hip:tile_n
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
