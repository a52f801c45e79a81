// Sustained v_mfma_f32_32x32x2_f32 rate with random (non-zero) register operands, no memory traffic.
// build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int DYN>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; a[i] = (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f + seed;
                                s = s * 1664525u + 1013904223u; b[i] = (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
  for (int it = 0; it < iters; ++it) {
    if (DYN) {   // fresh pseudo-random mantissas every iteration (data toggling as in a real GEMM)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        s = s * 1664525u + 1013904223u;
        a[u] = __uint_as_float((__float_as_uint(a[u]) & 0xff800000u) | (s >> 9));
        b[u] = __uint_as_float((__float_as_uint(b[u]) & 0xff800000u) | ((s * 2654435761u) >> 9));
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + 1) & 7], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 1) & 7], b[u], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + 3) & 7], b[(u + 5) & 7], acc[3], 0, 0, 0);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) r += acc[i][q];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int DYN> void run(float* d) {
  for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
    int blocks = 256 * waves_per_simd, iters = 20000;
    hipLaunchKernelGGL(k<DYN>, dim3(blocks), dim3(256), 0, 0, d, 100, 0.f); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<DYN>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.25f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 32 * (2.0 * 32 * 32 * 2);
    printf("%s operands, waves/SIMD %d: %.2f ms  %.1f TFLOP/s\n", DYN ? "changing" : "static", waves_per_simd, ms, flops / ms / 1e9);
  }
}
int main() {
  float* d; hipMalloc(&d, 4 * 256 * 4096);
  run<0>(d);
  run<1>(d);
  return 0;
}
