// planes_common.h -- the 16-bit plane formats shared by the plane GEMM (gemm_pl.hip) and the kernels that write planes
// (planes_bn.hip): how one fp32 value is split, and where an element lives (include/dgcnn_hip.h "plane set").
#pragma once
#include "common.h"

namespace {

template <int FMT> struct PlaneFmt;
template <> struct PlaneFmt<DGCNN_PLANES_F16X2> { static constexpr int NPL = 2; };

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// (x0, x1) -> packed pairs of the split terms (x0 in the low half)
// fp16 pair: h1 = rn16(x), h2 = rn16(x - h1) (x already scaled by the tensor's power of two; |x| <= 65504 or it saturates)
__device__ __forceinline__ void split_f16x2(float x0, float x1, unsigned (&o)[3]) {
  const float lim = 65504.f;
  x0 = fminf(fmaxf(x0, -lim), lim);
  x1 = fminf(fmaxf(x1, -lim), lim);
  const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
  const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
  o[0] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
  o[1] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
  o[2] = 0;
}

template <int FMT>
__device__ __forceinline__ void split8(const float (&v)[8], float scale, uint4 (&o)[3]) {
  unsigned w[4][3];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    split_f16x2(v[2 * e] * scale, v[2 * e + 1] * scale, w[e]);
  }
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) o[pl] = make_uint4(w[0][pl], w[1][pl], w[2][pl], w[3][pl]);
}


// Block-level transposing store of 64 rows x 16 octets of split slots (256 threads): thread t staged the slots of row
// t >> 2, octets (t & 3) + 4 u; every wave store then covers 64 consecutive rows of one octet (1 KiB contiguous).
constexpr int PL_CS = 65;                                    // slots per octet in LDS (+1: spreads the ds_write_b128 banks)

template <int FMT>
__device__ __forceinline__ void planes_stage(uint4 (*sh)[16 * PL_CS], int ol, int rl, const uint4 (&o)[3]) {
#pragma unroll
  for (int pl = 0; pl < PlaneFmt<FMT>::NPL; ++pl) sh[pl][ol * PL_CS + rl] = o[pl];
}

template <int FMT>
__device__ __forceinline__ void planes_flush(uint4 (*sh)[16 * PL_CS], char* dst, int64_t plane_stride, int64_t rows_alloc,
                                             int64_t r0, int o0, int noct) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int ol = wv * 4 + u;
    const int64_t row = r0 + lane;
    if (o0 + ol < noct && row < rows_alloc) {
#pragma unroll
      for (int pl = 0; pl < PlaneFmt<FMT>::NPL; ++pl)
        *reinterpret_cast<uint4*>(dst + pl * plane_stride + ((int64_t)(o0 + ol) * rows_alloc + row) * 16) = sh[pl][ol * PL_CS + lane];
    }
  }
}

// the power of two that brings `bound` into [2^14, 2^15) (fp16 planes: nothing reaches 65504)
__device__ __forceinline__ float pow2_scale_for(float bound) {
  if (!(bound > 0.f) || !(bound < INFINITY)) return 1.f;
  int e;
  frexpf(bound, &e);                       // bound = f * 2^e, f in [0.5, 1)
  int sh = 15 - e;
  sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
  return ldexpf(1.f, sh);
}

}  // namespace
