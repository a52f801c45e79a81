// knn_common.h -- lane-mask selects and the register-resident sorted (d, j) list shared by the k-NN kernels (knn.hip, knn_grid.hip).
#pragma once
#include "common.h"
#include <math.h>

namespace {

// ---- lane-mask helpers.  hipcc turns nested ?: on register arrays into exec-masked branches (20
// s_and_saveexec/s_cbranch per insert, measured 10x slower); v_cmp -> SGPR-pair mask -> v_cndmask
// is forced with the fcmp/icmp builtins and a one-instruction asm select.
typedef unsigned long long lmask_t;
__device__ __forceinline__ lmask_t m_flt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 4); }   // a <  b (ordered)
__device__ __forceinline__ lmask_t m_feq(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 1); }   // a == b
__device__ __forceinline__ lmask_t m_ilt(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 40); }      // a <  b (signed)
__device__ __forceinline__ lmask_t m_ine(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 33); }      // a != b
__device__ __forceinline__ float sel_f(lmask_t m, float t, float f) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
  return r;
}
__device__ __forceinline__ int sel_i(lmask_t m, int t, int f) {
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
  return r;
}

// smallest float greater than f (f finite or +inf; -0 counts as +0): d <= f  <=>  d < next_up(f)
__device__ __forceinline__ float next_up(float f) {
  const float g = f + 0.0f;
  const unsigned u = __float_as_uint(g);
  const unsigned v = (g >= 0.0f) ? u + 1u : u - 1u;
  return (g == INFINITY) ? g : __uint_as_float(v);
}

// lane mask -> 0 / 1 with inline constants (sel_i would park its two constants in VGPRs)
__device__ __forceinline__ unsigned sel_01(lmask_t m) {
  unsigned r;
  asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(m));
  return r;
}

template <bool LEX>
__device__ __forceinline__ lmask_t key_less(float d, int j, float dt, int jt) {
  return LEX ? (m_flt(d, dt) | (m_feq(d, dt) & m_ilt(j, jt))) : m_flt(d, dt);
}

// Branch-free sorted insert of (d, j) into an ascending list held in registers: per slot one
// v_cmp, one v_med3_f32 (new dl[t] = clamp(d, dl[t-1], dl[t])) and two v_cndmask for the index.
// A lane whose (d, j) is not smaller than its last entry is left unchanged (d = +inf is a no-op).
template <int KC, bool LEX>
__device__ __forceinline__ void list_insert(float (&dl)[KC], int (&jl)[KC], float d, int j) {
  lmask_t ct = key_less<LEX>(d, j, dl[KC - 1], jl[KC - 1]);
#pragma unroll
  for (int t = KC - 1; t >= 1; --t) {
    const lmask_t cp = key_less<LEX>(d, j, dl[t - 1], jl[t - 1]);
    dl[t] = __builtin_amdgcn_fmed3f(dl[t - 1], d, dl[t]);
    jl[t] = sel_i(ct, sel_i(cp, jl[t - 1], j), jl[t]);
    ct = cp;
  }
  dl[0] = sel_f(ct, d, dl[0]);
  jl[0] = sel_i(ct, j, jl[0]);
}

}  // namespace
