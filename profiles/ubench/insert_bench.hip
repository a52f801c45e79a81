// Throughput of the register-list insert of knn.hip (KC = 20) in isolation: cycles per insert round per wave for
//   V0  v_cmp -> SGPR mask -> v_cndmask   (the kernel's form)
//   V1  sign-of-difference mask in a VGPR -> v_bfi   (no SGPR hop)
// at 1, 2, 3 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off insert_bench.hip -o insert_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>

typedef unsigned long long lmask_t;
__device__ __forceinline__ lmask_t m_flt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 4); }
__device__ __forceinline__ float sel_f(lmask_t m, float t, float f) { float r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m)); return r; }
__device__ __forceinline__ int sel_i(lmask_t m, int t, int f) { int r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m)); return r; }

template <int KC>
__device__ __forceinline__ void insert_v0(float (&dl)[KC], int (&jl)[KC], float d, int j) {
  lmask_t ct = m_flt(d, dl[KC - 1]);
#pragma unroll
  for (int t = KC - 1; t >= 1; --t) {
    const lmask_t cp = m_flt(d, dl[t - 1]);
    dl[t] = __builtin_amdgcn_fmed3f(dl[t - 1], d, dl[t]);
    jl[t] = sel_i(ct, sel_i(cp, jl[t - 1], j), jl[t]);
    ct = cp;
  }
  dl[0] = sel_f(ct, d, dl[0]);
  jl[0] = sel_i(ct, j, jl[0]);
}

__device__ __forceinline__ int bfi(int m, int a, int b) { return (m & a) | (~m & b); }
template <int KC>
__device__ __forceinline__ void insert_v1(float (&dl)[KC], int (&jl)[KC], float d, int j) {
  int ct = __float_as_int(d - dl[KC - 1]) >> 31;      // all ones iff d < dl[KC-1]
#pragma unroll
  for (int t = KC - 1; t >= 1; --t) {
    const int cp = __float_as_int(d - dl[t - 1]) >> 31;
    dl[t] = __builtin_amdgcn_fmed3f(dl[t - 1], d, dl[t]);
    jl[t] = bfi(ct, bfi(cp, jl[t - 1], j), jl[t]);
    ct = cp;
  }
  dl[0] = __int_as_float(bfi(ct, __float_as_int(d), __float_as_int(dl[0])));
  jl[0] = bfi(ct, j, jl[0]);
}

template <int V>
__global__ __launch_bounds__(256) void bench(const float* __restrict__ in, int rounds, float* __restrict__ out, long long* cyc) {
  constexpr int KC = 20;
  float dl[KC]; int jl[KC];
#pragma unroll
  for (int t = 0; t < KC; ++t) { dl[t] = INFINITY; jl[t] = 0x7fffffff; }
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float d = in[tid];
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int r = 0; r < rounds; ++r) {
    d = d * 0.9993f + 0.00037f * (float)(r & 7);        // a changing candidate
    if (V == 0) insert_v0<KC>(dl, jl, d, r); else insert_v1<KC>(dl, jl, d, r);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f; int q = 0;
#pragma unroll
  for (int t = 0; t < KC; ++t) { s += dl[t]; q += jl[t]; }
  out[tid] = s + (float)q;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int rounds = 2000;
  for (int wps = 1; wps <= 3; ++wps) {                  // waves per SIMD = blocks per CU (a block = 4 waves = 1 per SIMD)
    const int blocks = 256 * wps;
    float *in, *out; long long* cyc;
    hipMalloc(&in, blocks * 256 * 4); hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    std::vector<float> h(blocks * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int v = 0; v < 2; ++v) {
      for (int rep = 0; rep < 2; ++rep) {
        if (v == 0) hipLaunchKernelGGL(bench<0>, dim3(blocks), dim3(256), 0, 0, in, rounds, out, cyc);
        else hipLaunchKernelGGL(bench<1>, dim3(blocks), dim3(256), 0, 0, in, rounds, out, cyc);
        hipDeviceSynchronize();
      }
      std::vector<long long> c(blocks);
      hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto x : c) s += (double)x;
      printf("V%d  %d wave(s)/SIMD: %.1f clk per insert round per wave  (%.1f clk of SIMD time per round)\n", v, wps,
             s / blocks / rounds, s / blocks / rounds / wps);
    }
    hipFree(in); hipFree(out); hipFree(cyc);
  }
  // same results?
  return 0;
}
