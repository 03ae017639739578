// Row-major [N, C] batch normalisation + (leaky) ReLU over the active rows of the whole batch
// (reference call sites scn_unet.py:42,44,51,66,73,116).
// HBM-bound: two passes over x in training (stats, then normalise+activate), fp64 accumulation of the
// per-channel sums so the variance does not lose digits at N ~ 2.5e5 rows.
#include "common.h"
#include "fused_bn.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int T = 256;
constexpr int MAX_PART = 1024;

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  typedef f32x4 type;
};
template <>
struct Vec<1> {
  typedef float type;
};

// bf16 rows (the 16-bit activation mode): 4 elements = one 8-byte access, arithmetic stays fp32
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ inline void ldv(const __bf16* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    bf16x4 t = *(const bf16x4*)p;
    v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w;
  } else {
    v[0] = (float)*p;
  }
}
template <int VEC>
__device__ inline void stv(__bf16* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *(bf16x4*)p = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  } else {
    *p = (__bf16)v[0];
  }
}

// IEEE fp16 rows (the fp16 kind of the 16-bit activation mode)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ inline void ldv(const _Float16* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    f16x4 t = *(const f16x4*)p;
    v[0] = (float)t.x; v[1] = (float)t.y; v[2] = (float)t.z; v[3] = (float)t.w;
  } else {
    v[0] = (float)*p;
  }
}
template <int VEC>
__device__ inline void stv(_Float16* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *(f16x4*)p = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
  } else {
    *p = (_Float16)v[0];
  }
}

template <int VEC>
__device__ inline void ldv(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    f32x4 t = *(const f32x4*)p;
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}
template <int VEC>
__device__ inline void stv(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    *p = v[0];
  }
}

// partial[block][0][C] = sum a, partial[block][1][C] = sum b  where (a, b) are produced by F per element
// MODE 0: a = x, b = x*x            (forward statistics)
// MODE 1: a = dy', b = dy' * xhat    (backward reductions), dy' = dy * act'(y)
template <int VEC, int MODE, typename E>
__global__ __launch_bounds__(T) void k_bn_reduce(const E* __restrict__ x, int ld_x, const E* __restrict__ dy,
                                                  int ld_dy, int64_t N, int C, const float* __restrict__ mean,
                                                  const float* __restrict__ invstd, const float* __restrict__ weight,
                                                  const float* __restrict__ bias, float leak,
                                                  double* __restrict__ partial, int64_t Ns, int nb0) {
  // rows [0, Ns) are statistics group 0 (blocks [0, nb0)), rows [Ns, N) group 1 (the other blocks): the two domains of a
  // jointly batched training step keep their own batch statistics.  Ns == N: one group.
  __shared__ double red[2 * T];
  const int grp = (int)blockIdx.x >= nb0;
  const int lb = grp ? blockIdx.x - nb0 : blockIdx.x, nbg = grp ? gridDim.x - nb0 : nb0;
  const int64_t gbase = grp ? Ns : 0, Ng = grp ? N - Ns : Ns;
  if (MODE == 1) mean += grp * C, invstd += grp * C;  // reused once per vector lane below
  const int CV = C / VEC;
  const int rs = T / CV;  // row slots per block
  const int tid = threadIdx.x;
  const int slot = tid / CV, cv = tid - slot * CV;
  float a[VEC], b[VEC];  // per-thread partials cover <= ~100 rows; block / grid combination is fp64
#pragma unroll
  for (int i = 0; i < VEC; i++) a[i] = b[i] = 0.f;
  float m[VEC], is[VEC], w[VEC], bs[VEC];
  if (MODE == 1 && slot < rs) {
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      m[i] = mean[cv * VEC + i];
      is[i] = invstd[cv * VEC + i];
      w[i] = weight ? weight[cv * VEC + i] : 1.f;
      bs[i] = bias ? bias[cv * VEC + i] : 0.f;
    }
  }
  const int64_t rows_per_block = (Ng + nbg - 1) / nbg;
  const int64_t r0 = gbase + (int64_t)lb * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < gbase + Ng ? r0 + rows_per_block : gbase + Ng;
  if (slot < rs) {
    // four rows (statistics) / two rows (backward reductions) per trip, all loads issued before any value is used (round 3: the one-row loop was one exposed round trip per
    // row and thread); rows accumulate in the same order as before: bit-identical
    auto row_acc = [&](const float (&xv)[VEC], const float (&dv)[VEC]) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          a[i] += xv[i];
          b[i] = fmaf(xv[i], xv[i], b[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          float xh = (xv[i] - m[i]) * is[i];
          float y = xh * w[i] + bs[i];
          float g = y > 0.f ? dv[i] : dv[i] * leak;
          a[i] += g;
          b[i] = fmaf(g, xh, b[i]);
        }
      }
    };
    constexpr int U = MODE == 1 ? 2 : 4;
    int64_t r = r0 + slot;
    for (; r + (U - 1) * (int64_t)rs < r1; r += U * (int64_t)rs) {
      float xv[U][VEC], dv[U][VEC];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int64_t ru = r + (int64_t)u * rs;
        ldv<VEC>(x + ru * ld_x + cv * VEC, xv[u]);
        if (MODE == 1) ldv<VEC>(dy + ru * ld_dy + cv * VEC, dv[u]);
      }
#pragma unroll
      for (int u = 0; u < U; u++) row_acc(xv[u], dv[u]);
    }
    for (; r < r1; r += rs) {
      float xv[VEC], dv[VEC];
      ldv<VEC>(x + r * ld_x + cv * VEC, xv);
      if (MODE == 1) ldv<VEC>(dy + r * ld_dy + cv * VEC, dv);
      row_acc(xv, dv);
    }
  }
  // reduce over row slots through LDS, one vector lane at a time
#pragma unroll
  for (int i = 0; i < VEC; i++) {
    __syncthreads();
    red[tid] = (double)a[i];
    red[T + tid] = (double)b[i];
    __syncthreads();
    if (tid < CV) {
      double sa = 0.0, sb = 0.0;
      for (int s = 0; s < rs; s++) {
        sa += red[s * CV + tid];
        sb += red[T + s * CV + tid];
      }
      partial[((int64_t)blockIdx.x * 2 + 0) * C + tid * VEC + i] = sa;
      partial[((int64_t)blockIdx.x * 2 + 1) * C + tid * VEC + i] = sb;
    }
  }
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;  // lane 0 holds the sum; fixed tree order -> bit-stable
}

// one wave per channel: lane l sums partials l, l+64, ... then a fixed shuffle tree.  With two statistics groups the
// running buffers are updated group 0 first, then group 1 - what two consecutive forward calls do.
__global__ __launch_bounds__(64) void k_bn_finalize_fwd(const double* __restrict__ partial, int nb0, int nb1, int64_t Ns, int64_t N,
                                                         int C, float eps, float momentum, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float* __restrict__ save_mean,
                                                         float* __restrict__ save_invstd) {
  const int c = blockIdx.x;
  const int G = nb1 > 0 ? 2 : 1;
  for (int g = 0; g < G; g++) {
    const int b0 = g ? nb0 : 0, b1 = g ? nb0 + nb1 : nb0;
    const int64_t Ng = g ? N - Ns : Ns;
    double s = 0.0, q = 0.0;
    for (int b = b0 + threadIdx.x; b < b1; b += 64) {
      s += partial[((int64_t)b * 2 + 0) * C + c];
      q += partial[((int64_t)b * 2 + 1) * C + c];
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (threadIdx.x == 0) {
      double mean = Ng > 0 ? s / (double)Ng : 0.0;
      double var = Ng > 0 ? q / (double)Ng - mean * mean : 0.0;
      if (var < 0.0) var = 0.0;
      save_mean[g * C + c] = (float)mean;
      save_invstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        double unbiased = Ng > 1 ? var * (double)Ng / (double)(Ng - 1) : var;
        running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * (float)mean;
        running_var[c] = momentum * running_var[c] + (1.f - momentum) * (float)unbiased;
      }
    }
  }
}

// sums[g][0][C] = sum dy', sums[g][1][C] = sum dy'*xhat per group; dweight / dbias are the totals over the groups
__global__ __launch_bounds__(64) void k_bn_finalize_bwd(const double* __restrict__ partial, int nb0, int nb1, int C,
                                                         float* __restrict__ sums, float* __restrict__ dweight,
                                                         float* __restrict__ dbias, int accumulate) {
  const int c = blockIdx.x;
  const int G = nb1 > 0 ? 2 : 1;
  float ts = 0.f, tq = 0.f;
  for (int g = 0; g < G; g++) {
    const int b0 = g ? nb0 : 0, b1 = g ? nb0 + nb1 : nb0;
    double s = 0.0, q = 0.0;
    for (int b = b0 + threadIdx.x; b < b1; b += 64) {
      s += partial[((int64_t)b * 2 + 0) * C + c];
      q += partial[((int64_t)b * 2 + 1) * C + c];
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (threadIdx.x == 0) {
      sums[(g * 2 + 0) * C + c] = (float)s;
      sums[(g * 2 + 1) * C + c] = (float)q;
      ts += (float)s, tq += (float)q;
    }
  }
  if (threadIdx.x != 0) return;
  if (dweight) dweight[c] = accumulate ? dweight[c] + tq : tq;
  if (dbias) dbias[c] = accumulate ? dbias[c] + ts : ts;
}

constexpr int APPLY_ROWS = 8;  // rows per thread in the apply kernels (parameters live in registers)

// y = act((x - mean) * invstd * w + b)
template <int VEC, typename E>
__global__ __launch_bounds__(T) void k_bn_apply(const E* __restrict__ x, int ld_x, int64_t N, int C,
                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                 int invstd_is_var, float eps, const float* __restrict__ weight,
                                                 const float* __restrict__ bias, float leak, E* __restrict__ y,
                                                 int ld_y, int64_t Ns, int ab0) {
  const int CV = C / VEC;
  const int rs = T / CV;
  const int slot = threadIdx.x / CV, cv = threadIdx.x - slot * CV;
  if (slot >= rs) return;
  const int grp = (int)blockIdx.x >= ab0;  // statistics group of this block's rows (see k_bn_reduce)
  const int64_t gbase = grp ? Ns : 0, gend = grp ? N : Ns;
  mean += grp * C, invstd += grp * C;
  float m[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int i = 0; i < VEC; i++) {
    int c = cv * VEC + i;
    float is = invstd_is_var ? 1.f / sqrtf(invstd[c] + eps) : invstd[c];
    m[i] = mean[c];
    sc[i] = is * (weight ? weight[c] : 1.f);
    sh[i] = bias ? bias[c] : 0.f;
  }
  const int64_t r0 = gbase + (int64_t)(grp ? blockIdx.x - ab0 : blockIdx.x) * rs * APPLY_ROWS + slot;
#pragma unroll 4
  for (int k = 0; k < APPLY_ROWS; k++) {
    const int64_t r = r0 + (int64_t)k * rs;
    if (r >= gend) break;
    float xv[VEC], yv[VEC];
    ldv<VEC>(x + r * ld_x + cv * VEC, xv);
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      float v = (xv[i] - m[i]) * sc[i] + sh[i];
      yv[i] = v > 0.f ? v : v * leak;
    }
    stv<VEC>(y + r * ld_y + cv * VEC, yv);
  }
}

// dx = w*invstd * (dy' - sum_dy/N - xhat * sum_dy_xhat/N)
template <int VEC, typename E>
__global__ __launch_bounds__(T) void k_bn_bwd_apply(const E* __restrict__ x, int ld_x, const E* __restrict__ dy,
                                                     int ld_dy, int64_t N, int C, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const float* __restrict__ weight,
                                                     const float* __restrict__ bias, float leak,
                                                     const float* __restrict__ sums /*[G][2][C]*/, E* __restrict__ dx,
                                                     int ld_dx, int64_t Ns, int ab0) {
  const int CV = C / VEC;
  const int rs = T / CV;
  const int slot = threadIdx.x / CV, cv = threadIdx.x - slot * CV;
  if (slot >= rs) return;
  const int grp = (int)blockIdx.x >= ab0;
  const int64_t gbase = grp ? Ns : 0, gend = grp ? N : Ns;
  mean += grp * C, invstd += grp * C;
  const float* sum_dy = sums + grp * 2 * C;
  const float* sum_dy_xhat = sum_dy + C;
  const float invN = 1.f / (float)(gend - gbase);
  float m[VEC], is[VEC], w[VEC], b[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; i++) {
    int c = cv * VEC + i;
    m[i] = mean[c];
    is[i] = invstd[c];
    w[i] = weight ? weight[c] : 1.f;
    b[i] = bias ? bias[c] : 0.f;
    s1[i] = sum_dy[c] * invN;
    s2[i] = sum_dy_xhat[c] * invN;
  }
  const int64_t r0 = gbase + (int64_t)(grp ? blockIdx.x - ab0 : blockIdx.x) * rs * APPLY_ROWS + slot;
#pragma unroll 4
  for (int k = 0; k < APPLY_ROWS; k++) {
    const int64_t r = r0 + (int64_t)k * rs;
    if (r >= gend) break;
    float xv[VEC], dv[VEC], ov[VEC];
    ldv<VEC>(x + r * ld_x + cv * VEC, xv);
    ldv<VEC>(dy + r * ld_dy + cv * VEC, dv);
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      float xh = (xv[i] - m[i]) * is[i];
      float yy = xh * w[i] + b[i];
      float g = yy > 0.f ? dv[i] : dv[i] * leak;
      ov[i] = w[i] * is[i] * (g - s1[i] - xh * s2[i]);
    }
    stv<VEC>(dx + r * ld_dx + cv * VEC, ov);
  }
}


// ---- single-launch training kernels over fp32 rows (skeleton and grid-barrier rules: fused_bn.h / bn2d.hip): a thread owns
// 4-channel (16-byte) pieces of its rows; x is read once in the forward pass, x and dy once in the backward pass.
struct Fused3P {
  const float *x, *dy;
  float *y, *dx;
  int ld_x, ld_y, ld_dy, ld_dx;
  int64_t N, Ns;
  int C, G0, G1, R;
  const float *weight, *bias;
  float *running_mean, *running_var;
  float eps, momentum, leak;
  float *save_mean, *save_invstd;
  float *sums, *dweight, *dbias;
  int accumulate;
  double* partial;
  unsigned* sync;
  unsigned* fault;
};

__device__ inline FusedBuf fused_buf32(const void* base, int64_t N, int ld, int C, int slot, int cv) {
  return fused_buf_bytes(base, N, ld, C, slot, cv, 4, 4);
}
__device__ inline void unpack4(const u32x4 t, float (&v)[4]) {
  v[0] = __uint_as_float(t.x), v[1] = __uint_as_float(t.y), v[2] = __uint_as_float(t.z), v[3] = __uint_as_float(t.w);
}
__device__ inline u32x4 pack4(const float (&v)[4]) {
  return (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
}

template <int RMAX>
__global__ __launch_bounds__(FT) void k_bn_fused_fwd(const Fused3P p) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* red = (float*)smem;
  double* red2 = (double*)(smem + FUSED_RED);
  double* outp = red2 + 1024;
  u32x4* rows = (u32x4*)(smem + FUSED_RED + 2 * 8192);
  constexpr int NREG = RMAX > FUSED_NL ? RMAX - FUSED_NL : 1;
  static_assert(RMAX % 4 == 0, "row groups of four");
  const int tid = threadIdx.x;
  const int C = p.C;
  const FusedGeom g = fused_geom(p.N, p.Ns, C >> 2, p.G0, p.G1);
  unsigned flag0 = 0;
  if (tid == 0) flag0 = xcd_load(&p.sync[FUSED_FLAG]);
  const int cvc = g.active ? g.cv : 0, slc = g.active ? g.slot : 0;
  const FusedBuf bx = fused_buf32(p.x, p.N, p.ld_x, C, slc, cvc), by = fused_buf32(p.y, p.N, p.ld_y, C, slc, cvc);
  u32x4 xr[NREG];
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) a[i] = b[i] = 0.f;
#pragma unroll
  for (int k = 0; k < RMAX; k++) {
    const bool ok = g.active && k < p.R && g.slot + k * g.rs < g.nrows;
    const u32x4 t = fused_ld(bx, g.r0 + (int64_t)k * g.rs, ok);  // zeros where !ok
    if (k < FUSED_NL) rows[k * FT + tid] = t;
    else xr[k < FUSED_NL ? 0 : k - FUSED_NL] = t;
    float xv[4];
    unpack4(t, xv);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      a[i] += xv[i];
      b[i] = fmaf(xv[i], xv[i], b[i]);
    }
    if ((k & 3) == 3) {
      MM_PIN8(a, b);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  fused_block_sums<4>(a, b, red, red2, outp, C, g.CV, g.rs, g.active);
  for (int pr = tid; pr < 2 * C; pr += FT) xcd_store(p.partial + (size_t)blockIdx.x * 2 * C + pr, outp[pr]);
  const int G = p.G0 + p.G1;
  fused_barrier(p.sync, p.fault, (unsigned)G, flag0);
  {
    const int ngrp = p.G1 > 0 ? 2 : 1;
    for (int c = blockIdx.x + (tid >> 6) * G; c < C; c += (FT / 64) * G) {
      for (int gi = 0; gi < ngrp; gi++) {
        double sm, sq;
        fused_wave_sums(p.partial, gi ? p.G0 : 0, gi ? G : p.G0, C, c, sm, sq);
        if ((tid & 63) == 0) {
          const int64_t Ng = gi ? p.N - p.Ns : p.Ns;
          const double mean = Ng > 0 ? sm / (double)Ng : 0.0;
          double var = Ng > 0 ? sq / (double)Ng - mean * mean : 0.0;
          if (var < 0.0) var = 0.0;
          xcd_store(p.save_mean + gi * C + c, (float)mean);
          xcd_store(p.save_invstd + gi * C + c, (float)(1.0 / sqrt(var + (double)p.eps)));
          if (p.running_mean) {  // scn semantics: momentum = the fraction of the old value kept (see k_bn_finalize_fwd)
            const double unbiased = Ng > 1 ? var * (double)Ng / (double)(Ng - 1) : var;
            p.running_mean[c] = p.momentum * p.running_mean[c] + (1.f - p.momentum) * (float)mean;
            p.running_var[c] = p.momentum * p.running_var[c] + (1.f - p.momentum) * (float)unbiased;
          }
        }
      }
    }
  }
  fused_barrier(p.sync, p.fault, (unsigned)G, flag0 + 1u);
  if (!g.active) return;
  const float *wp = p.weight ? p.weight : p.save_mean, *bp = p.bias ? p.bias : p.save_mean;
#pragma unroll
  for (int k = 0; k < NREG; k++) asm volatile("" : "+v"(xr[k]));
  float m[4], sc[4], sh[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = g.cv * 4 + i;
    const float is = xcd_load(p.save_invstd + g.grp * C + c);
    const float wv = wp[c], bv = bp[c];
    m[i] = xcd_load(p.save_mean + g.grp * C + c);
    sc[i] = is * (p.weight ? wv : 1.f);
    sh[i] = p.bias ? bv : 0.f;
  }
#pragma unroll
  for (int k0 = 0; k0 < RMAX; k0 += 4) {
    if (k0 < p.R) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = k0 + j;
        u32x4 t;
        if (k < FUSED_NL) t = rows[k * FT + tid];
        else t = xr[k < FUSED_NL ? 0 : k - FUSED_NL];
        float xv[4], yv[4];
        unpack4(t, xv);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float v = (xv[i] - m[i]) * sc[i] + sh[i];  // the arithmetic of k_bn_apply
          yv[i] = v > 0.f ? v : v * p.leak;
        }
        if (k < p.R && g.slot + k * g.rs < g.nrows) fused_st(by, g.r0 + (int64_t)k * g.rs, pack4(yv));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

template <int RMAX>
__global__ __launch_bounds__(FT) void k_bn_fused_bwd(const Fused3P p) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* red = (float*)smem;
  double* red2 = (double*)(smem + FUSED_RED);
  double* outp = red2 + 1024;
  u32x4* rows = (u32x4*)(smem + FUSED_RED + 2 * 8192);  // the masked gradient g of the first FUSED_NL rows
  static_assert(RMAX % 4 == 0, "row groups of four");
  const int tid = threadIdx.x;
  const int C = p.C;
  const FusedGeom g = fused_geom(p.N, p.Ns, C >> 2, p.G0, p.G1);
  unsigned flag0 = 0;
  if (tid == 0) flag0 = xcd_load(&p.sync[FUSED_FLAG]);
  const int cvc = g.active ? g.cv : 0, slc = g.active ? g.slot : 0;
  const FusedBuf bx = fused_buf32(p.x, p.N, p.ld_x, C, slc, cvc), bd = fused_buf32(p.dy, p.N, p.ld_dy, C, slc, cvc),
                 bdx = fused_buf32(p.dx, p.N, p.ld_dx, C, slc, cvc);
  const float *wp = p.weight ? p.weight : p.save_mean, *bp = p.bias ? p.bias : p.save_mean;
  u32x4 xr[RMAX];
  float m[4], is[4], w[4], bb[4], a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = cvc * 4 + i;
    const float wv = wp[c], bv = bp[c];
    m[i] = p.save_mean[g.grp * C + c];
    is[i] = p.save_invstd[g.grp * C + c];
    w[i] = p.weight ? wv : 1.f;
    bb[i] = p.bias ? bv : 0.f;
    a[i] = b[i] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < RMAX; k++) {
    const bool ok = g.active && k < p.R && g.slot + k * g.rs < g.nrows;
    const int64_t row = g.r0 + (int64_t)k * g.rs;
    const u32x4 tx = fused_ld(bx, row, ok);
    const u32x4 td = fused_ld(bd, row, ok);
    float xv[4], dv[4], gv[4];
    unpack4(tx, xv);
    unpack4(td, dv);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float xh = (xv[i] - m[i]) * is[i];  // the arithmetic of k_bn_reduce<*, 1> / k_bn_bwd_apply
      const float yy = xh * w[i] + bb[i];
      const float gg = ok ? (yy > 0.f ? dv[i] : dv[i] * p.leak) : 0.f;
      gv[i] = gg;
      a[i] += gg;
      b[i] = fmaf(gg, xh, b[i]);
    }
    xr[k] = tx;
    if (k < FUSED_NL) rows[k * FT + tid] = pack4(gv);
    if ((k & 1) == 1) {
      MM_PIN8(a, b);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  fused_block_sums<4>(a, b, red, red2, outp, C, g.CV, g.rs, g.active);
  for (int pr = tid; pr < 2 * C; pr += FT) xcd_store(p.partial + (size_t)blockIdx.x * 2 * C + pr, outp[pr]);
  const int G = p.G0 + p.G1;
  fused_barrier(p.sync, p.fault, (unsigned)G, flag0);
  {
    const int ngrp = p.G1 > 0 ? 2 : 1;
    for (int c = blockIdx.x + (tid >> 6) * G; c < C; c += (FT / 64) * G) {
      float ts = 0.f, tq = 0.f;
      for (int gi = 0; gi < ngrp; gi++) {
        double sg, sq;
        fused_wave_sums(p.partial, gi ? p.G0 : 0, gi ? G : p.G0, C, c, sg, sq);
        if ((tid & 63) == 0) {
          xcd_store(p.sums + (gi * 2 + 0) * C + c, (float)sg);
          xcd_store(p.sums + (gi * 2 + 1) * C + c, (float)sq);
        }
        ts += (float)sg, tq += (float)sq;
      }
      if ((tid & 63) == 0) {
        if (p.dweight) p.dweight[c] = p.accumulate ? p.dweight[c] + tq : tq;
        if (p.dbias) p.dbias[c] = p.accumulate ? p.dbias[c] + ts : ts;
      }
    }
  }
  fused_barrier(p.sync, p.fault, (unsigned)G, flag0 + 1u);
  if (!g.active) return;
#pragma unroll
  for (int k = 0; k < RMAX; k++) asm volatile("" : "+v"(xr[k]));
  float s1[4], s2[4];
  const float invN = 1.f / (float)g.Ng;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = g.cv * 4 + i;
    s1[i] = xcd_load(p.sums + (g.grp * 2 + 0) * C + c) * invN;
    s2[i] = xcd_load(p.sums + (g.grp * 2 + 1) * C + c) * invN;
  }
#pragma unroll
  for (int k0 = 0; k0 < RMAX; k0 += 4) {
    if (k0 < p.R) {
      u32x4 d1[4];
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (k0 + j >= FUSED_NL)  // rows beyond the LDS budget: dy again
          d1[j] = fused_ld(bd, g.r0 + (int64_t)(k0 + j) * g.rs, g.active && k0 + j < p.R && g.slot + (k0 + j) * g.rs < g.nrows);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = k0 + j;
        float xv[4], gv[4], ov[4];
        unpack4(xr[k], xv);
        if (k < FUSED_NL) {
          unpack4(rows[k * FT + tid], gv);
        } else {
          float dv[4];
          unpack4(d1[j], dv);
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float xh = (xv[i] - m[i]) * is[i];
            gv[i] = (xh * w[i] + bb[i]) > 0.f ? dv[i] : dv[i] * p.leak;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float xh = (xv[i] - m[i]) * is[i];
          ov[i] = w[i] * is[i] * (gv[i] - s1[i] - xh * s2[i]);
        }
        if (k < p.R && g.slot + k * g.rs < g.nrows) fused_st(bdx, g.r0 + (int64_t)k * g.rs, pack4(ov));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

inline unsigned apply_blocks(int64_t N, int C, int VEC) {
  int rs = T / (C / VEC);
  return (unsigned)mm_cdiv(N, (int64_t)rs * APPLY_ROWS);
}

inline int stat_blocks(int64_t N, int C, int VEC) {
  int rs = T / (C / VEC);
  int64_t nb = mm_cdiv(N, (int64_t)rs * 32);
  if (nb < 1) nb = 1;
  if (nb > MAX_PART) nb = MAX_PART;
  return (int)nb;
}
}  // namespace


static const void* const k_fused3_fns[] = {(const void*)k_bn_fused_fwd<8>,  (const void*)k_bn_fused_fwd<20>, (const void*)k_bn_fused_fwd<36>,
                                           (const void*)k_bn_fused_bwd<8>,  (const void*)k_bn_fused_bwd<20>, (const void*)k_bn_fused_bwd<36>};

// true: launched.  fp32 rows with 16-byte pieces only; maps too large for the chip fall through to the three-kernel path
static int bn_fused_try(MMHandle* H, bool backward, Fused3P& p, int64_t ldmax, hipStream_t s, bool* done) {
  *done = false;
  if (p.C % 4 != 0 || (p.ld_x % 4) || p.N * ldmax * 4 >= (1ll << 31)) return MM_OK;
  FusedPlan pl;
  int rc = fused_plan(H, MM_OPT_BN3D_FUSED, 0, p.N, p.Ns, p.C, 4, 36, backward, k_fused3_fns, 6, s, &pl);
  if (rc || !pl.ok) return rc;
  p.G0 = pl.G0, p.G1 = pl.G1, p.R = pl.R, p.sync = pl.sync, p.fault = pl.fault;
  const dim3 grid(pl.G0 + pl.G1), blk(FT);
  if (!backward) {
    if (pl.R <= 8) hipLaunchKernelGGL(k_bn_fused_fwd<8>, grid, blk, FUSED_LDS, s, p);
    else if (pl.R <= 20) hipLaunchKernelGGL(k_bn_fused_fwd<20>, grid, blk, FUSED_LDS, s, p);
    else hipLaunchKernelGGL(k_bn_fused_fwd<36>, grid, blk, FUSED_LDS, s, p);
  } else {
    if (pl.R <= 8) hipLaunchKernelGGL(k_bn_fused_bwd<8>, grid, blk, FUSED_LDS, s, p);
    else if (pl.R <= 20) hipLaunchKernelGGL(k_bn_fused_bwd<20>, grid, blk, FUSED_LDS, s, p);
    else hipLaunchKernelGGL(k_bn_fused_bwd<36>, grid, blk, FUSED_LDS, s, p);
  }
  MM_LAUNCH_CHECK();
  *done = true;
  return MM_OK;
}

static void split_blocks(int64_t N, int64_t& Ns, int C, int VEC, bool stats, int& b0, int& b1) {
  if (Ns <= 0 || Ns >= N) Ns = N;
  if (stats) {
    b0 = stat_blocks(Ns, C, VEC);
    b1 = Ns < N ? stat_blocks(N - Ns, C, VEC) : 0;
    if (b1 > 0) {  // both groups share the MAX_PART partial slots
      if (b0 > MAX_PART / 2) b0 = MAX_PART / 2;
      if (b1 > MAX_PART / 2) b1 = MAX_PART / 2;
    }
  } else {
    b0 = (int)apply_blocks(Ns, C, VEC);
    b1 = Ns < N ? (int)apply_blocks(N - Ns, C, VEC) : 0;
  }
}

// training forward: batch statistics over the N rows; running stats updated in place (scn "momentum" = keep fraction).
// Ns: rows [0,Ns) and [Ns,N) (the active sites of the source and of the target scenes of a jointly batched step) are
// normalised with their OWN statistics; Ns = N (or 0): ordinary single batch.  save_mean / save_invstd: [G][C].
template <typename E>
static int bn_fwd_train(MMHandle* H, const E* x, int ld_x, int64_t N, int64_t Ns, int C, const float* weight, const float* bias,
                    float* running_mean, float* running_var, float eps, float momentum, float leak, E* y, int ld_y,
                    float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(C > 0 && C <= 4 * T && ld_x >= C && ld_y >= C, "bn_fwd: bad shape C=%d", C);
  if (ws_bytes < (size_t)MAX_PART * 2 * C * sizeof(double)) {
    mm_set_error("bn_fwd: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  double* partial = (double*)ws;
  const bool v4 = (C % 4 == 0) && (ld_x % 4 == 0) && (ld_y % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % (4 * sizeof(E)) == 0);
  MM_CHECK_ARG(C / (v4 ? 4 : 1) <= T, "bn_fwd: %d channels need 16-byte aligned rows (C multiple of 4)", C);
  if constexpr (sizeof(E) == 4) {
    if (v4 && N > 0) {
      Fused3P p = {};
      p.x = (const float*)x, p.y = (float*)y, p.ld_x = ld_x, p.ld_y = ld_y, p.N = N, p.Ns = (Ns <= 0 || Ns >= N) ? N : Ns, p.C = C;
      p.weight = weight, p.bias = bias, p.running_mean = running_mean, p.running_var = running_var;
      p.eps = eps, p.momentum = momentum, p.leak = leak, p.save_mean = save_mean, p.save_invstd = save_invstd, p.partial = partial;
      bool done;
      int rc = bn_fused_try(H, false, p, std::max(ld_x, ld_y), s, &done);
      if (rc || done) return rc;
    }
  }
  int nb0, nb1, ab0, ab1;
  split_blocks(N, Ns, C, v4 ? 4 : 1, true, nb0, nb1);
  if (v4)
    hipLaunchKernelGGL((k_bn_reduce<4, 0, E>), dim3(nb0 + nb1), dim3(T), 0, s, x, ld_x, (const E*)nullptr, 0, N, C, nullptr, nullptr, nullptr,
                       nullptr, 0.f, partial, Ns, nb0);
  else
    hipLaunchKernelGGL((k_bn_reduce<1, 0, E>), dim3(nb0 + nb1), dim3(T), 0, s, x, ld_x, (const E*)nullptr, 0, N, C, nullptr, nullptr, nullptr,
                       nullptr, 0.f, partial, Ns, nb0);
  hipLaunchKernelGGL(k_bn_finalize_fwd, dim3(C), dim3(64), 0, s, partial, nb0, nb1, Ns, N, C, eps, momentum, running_mean, running_var,
                     save_mean, save_invstd);
  if (N > 0) {
    split_blocks(N, Ns, C, v4 ? 4 : 1, false, ab0, ab1);
    if (v4)
      hipLaunchKernelGGL((k_bn_apply<4, E>), dim3(ab0 + ab1), dim3(T), 0, s, x, ld_x, N, C, save_mean, save_invstd, 0, eps, weight, bias,
                         leak, y, ld_y, Ns, ab0);
    else
      hipLaunchKernelGGL((k_bn_apply<1, E>), dim3(ab0 + ab1), dim3(T), 0, s, x, ld_x, N, C, save_mean, save_invstd, 0, eps, weight, bias,
                         leak, y, ld_y, Ns, ab0);
  }
  MM_LAUNCH_CHECK();
  return MM_OK;
}

template <typename E>
static int bn_fwd_eval(const E* x, int ld_x, int64_t N, int C, const float* weight, const float* bias,
                   const float* running_mean, const float* running_var, float eps, float leak, E* y, int ld_y,
                   hipStream_t s) {
  MM_CHECK_ARG(C > 0 && ld_x >= C && ld_y >= C, "bn_eval: bad shape");
  if (N == 0) return MM_OK;
  const bool v4 = (C % 4 == 0) && (ld_x % 4 == 0) && (ld_y % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % (4 * sizeof(E)) == 0);
  const int ab = (int)apply_blocks(N, C, v4 ? 4 : 1);
  if (v4)
    hipLaunchKernelGGL((k_bn_apply<4, E>), dim3(ab), dim3(T), 0, s, x, ld_x, N, C, running_mean, running_var, 1, eps, weight, bias, leak, y,
                       ld_y, N, ab);
  else
    hipLaunchKernelGGL((k_bn_apply<1, E>), dim3(ab), dim3(T), 0, s, x, ld_x, N, C, running_mean, running_var, 1, eps, weight, bias, leak, y,
                       ld_y, N, ab);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// training backward; dweight/dbias may be null; accumulate != 0 adds into them; Ns and [G][C] statistics as in the forward
template <typename E>
static int bn_bwd(MMHandle* H, const E* x, int ld_x, const E* dy, int ld_dy, int64_t N, int64_t Ns, int C, const float* weight,
              const float* bias, const float* save_mean, const float* save_invstd, float leak, E* dx, int ld_dx,
              float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(C > 0 && C <= 4 * T && ld_x >= C && ld_dy >= C && ld_dx >= C, "bn_bwd: bad shape");
  size_t need = mm_align((size_t)MAX_PART * 2 * C * sizeof(double));
  if (ws_bytes < need + 4 * C * sizeof(float)) {
    mm_set_error("bn_bwd: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  double* partial = (double*)ws;
  float* sums = (float*)((char*)ws + need);
  const bool v4 = (C % 4 == 0) && (ld_x % 4 == 0) && (ld_dy % 4 == 0) && (ld_dx % 4 == 0) &&
                  (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) % (4 * sizeof(E)) == 0);
  MM_CHECK_ARG(C / (v4 ? 4 : 1) <= T, "bn_bwd: %d channels need 16-byte aligned rows (C multiple of 4)", C);
  if constexpr (sizeof(E) == 4) {
    if (v4 && N > 0) {
      Fused3P p = {};
      p.x = (const float*)x, p.dy = (const float*)dy, p.dx = (float*)dx, p.ld_x = ld_x, p.ld_dy = ld_dy, p.ld_dx = ld_dx;
      p.N = N, p.Ns = (Ns <= 0 || Ns >= N) ? N : Ns, p.C = C, p.weight = weight, p.bias = bias, p.leak = leak;
      p.save_mean = (float*)save_mean, p.save_invstd = (float*)save_invstd, p.sums = sums, p.dweight = dweight, p.dbias = dbias;
      p.accumulate = accumulate, p.partial = partial;
      bool done;
      int rc = bn_fused_try(H, true, p, std::max(std::max(ld_x, ld_dy), ld_dx), s, &done);
      if (rc || done) return rc;
    }
  }
  int nb0, nb1, ab0, ab1;
  split_blocks(N, Ns, C, v4 ? 4 : 1, true, nb0, nb1);
  if (v4)
    hipLaunchKernelGGL((k_bn_reduce<4, 1, E>), dim3(nb0 + nb1), dim3(T), 0, s, x, ld_x, dy, ld_dy, N, C, save_mean, save_invstd, weight,
                       bias, leak, partial, Ns, nb0);
  else
    hipLaunchKernelGGL((k_bn_reduce<1, 1, E>), dim3(nb0 + nb1), dim3(T), 0, s, x, ld_x, dy, ld_dy, N, C, save_mean, save_invstd, weight,
                       bias, leak, partial, Ns, nb0);
  hipLaunchKernelGGL(k_bn_finalize_bwd, dim3(C), dim3(64), 0, s, partial, nb0, nb1, C, sums, dweight, dbias, accumulate);
  if (N > 0) {
    split_blocks(N, Ns, C, v4 ? 4 : 1, false, ab0, ab1);
    if (v4)
      hipLaunchKernelGGL((k_bn_bwd_apply<4, E>), dim3(ab0 + ab1), dim3(T), 0, s, x, ld_x, dy, ld_dy, N, C, save_mean, save_invstd, weight,
                         bias, leak, sums, dx, ld_dx, Ns, ab0);
    else
      hipLaunchKernelGGL((k_bn_bwd_apply<1, E>), dim3(ab0 + ab1), dim3(T), 0, s, x, ld_x, dy, ld_dy, N, C, save_mean, save_invstd, weight,
                         bias, leak, sums, dx, ld_dx, Ns, ab0);
  }
  MM_LAUNCH_CHECK();
  return MM_OK;
}

extern "C" {

// Which path runs (single-launch grid-barrier kernels or reduce / finalize / apply) is the HANDLE's choice:
// mm_set_option(h, MM_OPT_BN3D_FUSED, mask), bit 0 = forward, bit 1 = backward.  Same residency rules as the 2D kernels.
size_t mm_bn_ws_bytes(int C) { return mm_align((size_t)MAX_PART * 2 * C * sizeof(double)) + mm_align(4 * C * sizeof(float)) + 256; }


int mm_bn_fwd_train(void* h, const float* x, int ld_x, int64_t N, int64_t Ns, int C, const float* weight, const float* bias,
                    float* running_mean, float* running_var, float eps, float momentum, float leak, float* y, int ld_y,
                    float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_HANDLE(h);
  return bn_fwd_train<float>(H, x, ld_x, N, Ns, C, weight, bias, running_mean, running_var, eps, momentum, leak, y, ld_y, save_mean,
                             save_invstd, ws, ws_bytes, s);
}
int mm_bn_fwd_eval(const float* x, int ld_x, int64_t N, int C, const float* weight, const float* bias,
                   const float* running_mean, const float* running_var, float eps, float leak, float* y, int ld_y,
                   hipStream_t s) {
  return bn_fwd_eval<float>(x, ld_x, N, C, weight, bias, running_mean, running_var, eps, leak, y, ld_y, s);
}
int mm_bn_bwd(void* h, const float* x, int ld_x, const float* dy, int ld_dy, int64_t N, int64_t Ns, int C, const float* weight,
              const float* bias, const float* save_mean, const float* save_invstd, float leak, float* dx, int ld_dx,
              float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_HANDLE(h);
  return bn_bwd<float>(H, x, ld_x, dy, ld_dy, N, Ns, C, weight, bias, save_mean, save_invstd, leak, dx, ld_dx, dweight, dbias,
                       accumulate, ws, ws_bytes, s);
}
// the same three entry points over bf16 rows (16-bit activation mode): x / y / dy / dx are bf16 [N, C], statistics and
// parameters stay fp32
int mm_bn_fwd_train_bf16(void* h, const void* x, int ld_x, int64_t N, int64_t Ns, int C, const float* weight, const float* bias,
                         float* running_mean, float* running_var, float eps, float momentum, float leak, void* y, int ld_y,
                         float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_HANDLE(h);
  return bn_fwd_train<__bf16>(H, (const __bf16*)x, ld_x, N, Ns, C, weight, bias, running_mean, running_var, eps, momentum, leak,
                              (__bf16*)y, ld_y, save_mean, save_invstd, ws, ws_bytes, s);
}
int mm_bn_fwd_eval_bf16(const void* x, int ld_x, int64_t N, int C, const float* weight, const float* bias,
                        const float* running_mean, const float* running_var, float eps, float leak, void* y, int ld_y,
                        hipStream_t s) {
  return bn_fwd_eval<__bf16>((const __bf16*)x, ld_x, N, C, weight, bias, running_mean, running_var, eps, leak, (__bf16*)y, ld_y, s);
}
int mm_bn_bwd_bf16(void* h, const void* x, int ld_x, const void* dy, int ld_dy, int64_t N, int64_t Ns, int C, const float* weight,
                   const float* bias, const float* save_mean, const float* save_invstd, float leak, void* dx, int ld_dx,
                   float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_HANDLE(h);
  return bn_bwd<__bf16>(H, (const __bf16*)x, ld_x, (const __bf16*)dy, ld_dy, N, Ns, C, weight, bias, save_mean, save_invstd, leak,
                        (__bf16*)dx, ld_dx, dweight, dbias, accumulate, ws, ws_bytes, s);
}

// ... and over IEEE fp16 rows
int mm_bn_fwd_train_f16(void* h, const void* x, int ld_x, int64_t N, int64_t Ns, int C, const float* weight, const float* bias,
                        float* running_mean, float* running_var, float eps, float momentum, float leak, void* y, int ld_y,
                        float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_HANDLE(h);
  return bn_fwd_train<_Float16>(H, (const _Float16*)x, ld_x, N, Ns, C, weight, bias, running_mean, running_var, eps, momentum, leak,
                                (_Float16*)y, ld_y, save_mean, save_invstd, ws, ws_bytes, s);
}
int mm_bn_fwd_eval_f16(const void* x, int ld_x, int64_t N, int C, const float* weight, const float* bias,
                       const float* running_mean, const float* running_var, float eps, float leak, void* y, int ld_y,
                       hipStream_t s) {
  return bn_fwd_eval<_Float16>((const _Float16*)x, ld_x, N, C, weight, bias, running_mean, running_var, eps, leak, (_Float16*)y, ld_y, s);
}
int mm_bn_bwd_f16(void* h, const void* x, int ld_x, const void* dy, int ld_dy, int64_t N, int64_t Ns, int C, const float* weight,
                  const float* bias, const float* save_mean, const float* save_invstd, float leak, void* dx, int ld_dx,
                  float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_HANDLE(h);
  return bn_bwd<_Float16>(H, (const _Float16*)x, ld_x, (const _Float16*)dy, ld_dy, N, Ns, C, weight, bias, save_mean, save_invstd, leak,
                          (_Float16*)dx, ld_dx, dweight, dbias, accumulate, ws, ws_bytes, s);
}

}  // extern "C"
