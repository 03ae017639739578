// Row-major [N, C] batch normalisation + (leaky) ReLU over the active rows of the whole batch
// (reference call sites scn_unet.py:42,44,51,66,73,116).
// HBM-bound: two passes over x in training (stats, then normalise+activate), fp64 accumulation of the
// per-channel sums so the variance does not lose digits at N ~ 2.5e5 rows.
#include "common.h"

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
    for (int64_t r = r0 + slot; r < r1; r += rs) {
      float xv[VEC];
      ldv<VEC>(x + r * ld_x + cv * VEC, xv);
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          a[i] += xv[i];
          b[i] = fmaf(xv[i], xv[i], b[i]);
        }
      } else {
        float dv[VEC];
        ldv<VEC>(dy + r * ld_dy + cv * VEC, dv);
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          float xh = (xv[i] - m[i]) * is[i];
          float y = xh * w[i] + bs[i];
          float g = y > 0.f ? dv[i] : dv[i] * leak;
          a[i] += g;
          b[i] = fmaf(g, xh, b[i]);
        }
      }
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
static int bn_fwd_train(const E* x, int ld_x, int64_t N, int64_t Ns, int C, const float* weight, const float* bias,
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
static int bn_bwd(const E* x, int ld_x, const E* dy, int ld_dy, int64_t N, int64_t Ns, int C, const float* weight,
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

size_t mm_bn_ws_bytes(int C) { return mm_align((size_t)MAX_PART * 2 * C * sizeof(double)) + mm_align(4 * C * sizeof(float)) + 256; }


int mm_bn_fwd_train(const float* x, int ld_x, int64_t N, int64_t Ns, int C, const float* weight, const float* bias,
                    float* running_mean, float* running_var, float eps, float momentum, float leak, float* y, int ld_y,
                    float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, hipStream_t s) {
  return bn_fwd_train<float>(x, ld_x, N, Ns, C, weight, bias, running_mean, running_var, eps, momentum, leak, y, ld_y, save_mean,
                             save_invstd, ws, ws_bytes, s);
}
int mm_bn_fwd_eval(const float* x, int ld_x, int64_t N, int C, const float* weight, const float* bias,
                   const float* running_mean, const float* running_var, float eps, float leak, float* y, int ld_y,
                   hipStream_t s) {
  return bn_fwd_eval<float>(x, ld_x, N, C, weight, bias, running_mean, running_var, eps, leak, y, ld_y, s);
}
int mm_bn_bwd(const float* x, int ld_x, const float* dy, int ld_dy, int64_t N, int64_t Ns, int C, const float* weight,
              const float* bias, const float* save_mean, const float* save_invstd, float leak, float* dx, int ld_dx,
              float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  return bn_bwd<float>(x, ld_x, dy, ld_dy, N, Ns, C, weight, bias, save_mean, save_invstd, leak, dx, ld_dx, dweight, dbias,
                       accumulate, ws, ws_bytes, s);
}
// the same three entry points over bf16 rows (16-bit activation mode): x / y / dy / dx are bf16 [N, C], statistics and
// parameters stay fp32
int mm_bn_fwd_train_bf16(const void* x, int ld_x, int64_t N, int64_t Ns, int C, const float* weight, const float* bias,
                         float* running_mean, float* running_var, float eps, float momentum, float leak, void* y, int ld_y,
                         float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, hipStream_t s) {
  return bn_fwd_train<__bf16>((const __bf16*)x, ld_x, N, Ns, C, weight, bias, running_mean, running_var, eps, momentum, leak,
                              (__bf16*)y, ld_y, save_mean, save_invstd, ws, ws_bytes, s);
}
int mm_bn_fwd_eval_bf16(const void* x, int ld_x, int64_t N, int C, const float* weight, const float* bias,
                        const float* running_mean, const float* running_var, float eps, float leak, void* y, int ld_y,
                        hipStream_t s) {
  return bn_fwd_eval<__bf16>((const __bf16*)x, ld_x, N, C, weight, bias, running_mean, running_var, eps, leak, (__bf16*)y, ld_y, s);
}
int mm_bn_bwd_bf16(const void* x, int ld_x, const void* dy, int ld_dy, int64_t N, int64_t Ns, int C, const float* weight,
                   const float* bias, const float* save_mean, const float* save_invstd, float leak, void* dx, int ld_dx,
                   float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  return bn_bwd<__bf16>((const __bf16*)x, ld_x, (const __bf16*)dy, ld_dy, N, Ns, C, weight, bias, save_mean, save_invstd, leak,
                        (__bf16*)dx, ld_dx, dweight, dbias, accumulate, ws, ws_bytes, s);
}

}  // extern "C"
