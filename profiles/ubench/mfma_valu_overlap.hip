// Do v_mfma_f32_32x32x2_f32 (fp32 in) and VALU work of the SAME wave overlap on gfx950?  One wave per SIMD issues
//   A: N dependent fp32 MFMAs,   B: N x 12 independent VALU ops (v_med3 / v_cndmask mix),   C: both interleaved (1 MFMA, 12 VALU).
// If C ~ max(A, B) they overlap; if C ~ A + B the fp32 MFMA occupies the vector pipe.  Same for v_mfma_f32_32x32x16_bf16.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int MODE, int KIND>   // MODE 0 = MFMA only, 1 = VALU only, 2 = interleaved; KIND 0 = f32 MFMA, 1 = bf16 MFMA
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int n) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  bf16x8 ab, bb;
  for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)(a + e); bb[e] = (__bf16)(b - e); }
  float v[12];
  for (int i = 0; i < 12; ++i) v[i] = a * (i + 1);
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE != 1) {
        __builtin_amdgcn_sched_barrier(0);
        if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE != 0) {
#pragma unroll
        for (int q = 0; q < 12; ++q) v[q] = __builtin_amdgcn_fmed3f(v[q], v[(q + 1) % 12], b);   // 12 VALU, two independent chains of dependence each
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int i = 0; i < 12; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int KIND>
double run(int blocks, int n, float* out, long long* cyc) {
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<MODE, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, n); hipDeviceSynchronize(); }
  std::vector<long long> c(blocks);
  hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto x : c) s += (double)x;
  return s / blocks / (n * 4.0);
}

int main() {
  float* out; long long* cyc;
  const int n = 4000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    printf("%d wave(s)/SIMD, clk per (1 MFMA | 12 VALU | 1 MFMA + 12 VALU):  f32 MFMA: %.1f | %.1f | %.1f     bf16 MFMA: %.1f | %.1f | %.1f\n", wps,
           run<0, 0>(blocks, n, out, cyc), run<1, 0>(blocks, n, out, cyc), run<2, 0>(blocks, n, out, cyc),
           run<0, 1>(blocks, n, out, cyc), run<1, 1>(blocks, n, out, cyc), run<2, 1>(blocks, n, out, cyc));
    hipFree(out); hipFree(cyc);
  }
  return 0;
}
