// The 16-bit storage format of the dense 2D kernels (conv2d.hip, bn2d.hip, misc2d.hip) is chosen per OBJECT FILE: bf16 by
// default, IEEE fp16 with -DMM_ACT_FP16.  The Makefile builds the three sources twice; the fp16 objects export the same entry
// points under the suffix _f16 (MM_SYM / MM_H).  Inside the kernels "bf2f" / "f2bf" mean "stored 16 bits <-> float" in either
// build, h_lo / h_hi unpack the two halves of a dword, and the MFMA macros select v_mfma_*_bf16 or v_mfma_*_f16 (same rate).
// The reference trains with ``precision: 16`` = fp16 autocast + GradScaler (config/run/train.yaml:11); fp16 stores 11
// significand bits against bf16's 8 and needs the loss scale of mm2d3d_amd/amp.py for its gradient maps.
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned short u16;

#ifdef MM_ACT_FP16
#define MM_SYM(name) name##_f16
#define MM_H(name) name##_f16
#define MM_H2(name, tail) name##_f16##tail
typedef _Float16 h16;
__device__ inline float bf2f(u16 v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ inline u16 f2bf(float f) { return __builtin_bit_cast(u16, (_Float16)f); }  // v_cvt_f16_f32: round to nearest even
__device__ inline float h_lo(unsigned w) { return bf2f((u16)(w & 0xFFFFu)); }
__device__ inline float h_hi(unsigned w) { return bf2f((u16)(w >> 16)); }
#define MM_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define MM_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#else
#define MM_SYM(name) name
#define MM_H(name) name##_bf16
#define MM_H2(name, tail) name##_bf16##tail
typedef __bf16 h16;
__device__ inline float bf2f(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ inline u16 f2bf(float f) {  // round to nearest even (inputs are finite)
  unsigned u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__device__ inline float h_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ inline float h_hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
#define MM_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define MM_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
