// Exact-fp32 2D convolutions (the `precision: 32` mode of the reference's test runs, config/run/test.yaml:8).
//
// The bf16 MFMA kernels of conv2d.hip are the training hot path; this file is the fp32 counterpart used when the 2D branch
// has to match an fp32 reference to 1e-3 (north_star): plain fp32 FMAs (v_fma_f32, no reduced-precision step anywhere) in
// an LDS-tiled implicit GEMM.  One pair of kernels expresses every convolution of the 2D net through index maps:
//   k_conv_f32   O[b,oy,ox,n] = bias[n] + sum_{ky,kx,c} A[b, ty/up, tx/up, c] * W[n*w_sn + c*w_sc + ky*w_sy + kx*w_sx]
//                with ty = oy*so + sgn*ky + off (tx alike), the term present iff ty >= 0, ty % up == 0, ty/up < Hi:
//                  Conv2d forward            so = stride, sgn = +1, off = -pad, up = 1       (W[n=co][c=ci])
//                  Conv2d data gradient      so = 1, sgn = -1, off = +pad, up = stride        (W[c=co][n=ci])
//                  ConvTranspose2d forward   so = 1, sgn = -1, off = 0, up = stride           (W[c=ci][n=co])
//                  ConvTranspose2d data grad so = stride, sgn = +1, off = 0, up = 1           (W[n=ci][c=co])
//   k_wgrad_f32  dW[n*w_sn + c*w_sc + ky*w_sy + kx*w_sx] (+)= sum_{b,oy,ox} G[b,oy,ox,n] * A[b, ty/up, tx/up, c]
//                (Conv2d: G = dout, A = x; ConvTranspose2d: G = x on the coarse grid, A = dout), split over pixel chunks
//                into partial slabs that are summed in a fixed order (bit-stable).
// Reference call sites: backbones.py:23-25,49-63 (ResNet34 stacks), 2d_net/model.py:44-56,64-82 (decoder), :59-60 (heads).
#include "common.h"

namespace {

struct CF {
  const float* A;
  const float* W;
  const float* bias;
  float* O;
  int B, Hi, Wi, Ca, ldA, Ho, Wo, Cn, ldO, KH, KW, so, sgn, off, up;
  int64_t w_sn, w_sc, w_sy, w_sx;
};

constexpr int TM = 64, TN = 64, TK = 16;

__device__ inline bool src_coord(int o, int k, const CF& p, int limit, int& s) {
  const int t = o * p.so + p.sgn * k + p.off;
  if (t < 0) return false;
  if (p.up > 1) {
    if (t % p.up) return false;
    s = t / p.up;
  } else {
    s = t;
  }
  return s < limit;
}

__global__ __launch_bounds__(256) void k_conv_f32(CF p) {
  __shared__ float As[TK][TM + 4], Bs[TK][TN + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t M = (int64_t)p.B * p.Ho * p.Wo;
  const int64_t m0 = (int64_t)blockIdx.x * TM;
  const int n0 = blockIdx.y * TN;
  const int K = p.KH * p.KW * p.Ca;
  // this thread stages A elements (m = a_m, k = kk0 + a_k) with a_k = tid & 15, a_m = (tid >> 4) + 16 i
  int pb[4], py[4], px[4];
  bool pv[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int64_t m = m0 + (tid >> 4) + 16 * i;
    pv[i] = m < M;
    const int64_t mm = pv[i] ? m : 0;
    px[i] = (int)(mm % p.Wo);
    py[i] = (int)((mm / p.Wo) % p.Ho);
    pb[i] = (int)(mm / ((int64_t)p.Wo * p.Ho));
  }
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += TK) {
    {
      const int k = k0 + (tid & 15);
      const bool kv = k < K;
      const int tap = kv ? k / p.Ca : 0, c = kv ? k - tap * p.Ca : 0;
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float v = 0.f;
        int sy, sx;
        if (kv && pv[i] && src_coord(py[i], ky, p, p.Hi, sy) && src_coord(px[i], kx, p, p.Wi, sx))
          v = p.A[(((int64_t)pb[i] * p.Hi + sy) * p.Wi + sx) * p.ldA + c];
        As[tid & 15][(tid >> 4) + 16 * i] = v;
      }
      // B elements (k = k0 + b_k, n = n0 + b_n) with b_n = tid & 63, b_k = (tid >> 6) + 4 i
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int kb = k0 + (tid >> 6) + 4 * i, n = n0 + (tid & 63);
        float v = 0.f;
        if (kb < K && n < p.Cn) {
          const int tapb = kb / p.Ca, cb = kb - tapb * p.Ca;
          const int kyb = tapb / p.KW, kxb = tapb - kyb * p.KW;
          v = p.W[n * p.w_sn + cb * p.w_sc + kyb * p.w_sy + kxb * p.w_sx];
        }
        Bs[(tid >> 6) + 4 * i][tid & 63] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; kk++) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int64_t m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n = n0 + tx * 4 + j;
      if (n < p.Cn) p.O[m * p.ldO + n] = acc[i][j] + (p.bias ? p.bias[n] : 0.f);
    }
  }
}

struct WF {
  const float* G;
  const float* A;
  float* partial;
  int B, Hg, Wg, Cg, ldG, Hi, Wi, Ca, ldA, KH, KW, so, sgn, off, up;
  int64_t chunk;  // pixels of the G grid per blockIdx.z
};

// partial[z][n][kflat], kflat = (ky*KW + kx)*Ca + c
__global__ __launch_bounds__(256) void k_wgrad_f32(WF p) {
  __shared__ float Gs[TK][TM + 4], As[TK][TN + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int n0 = blockIdx.x * TM, k0 = blockIdx.y * TN;
  const int Kf = p.KH * p.KW * p.Ca;
  const int64_t M = (int64_t)p.B * p.Hg * p.Wg;
  const int64_t mb = (int64_t)blockIdx.z * p.chunk, me = mb + p.chunk < M ? mb + p.chunk : M;
  // the A element this thread stages: kflat = k0 + (tid & 63) fixed, pixel = mm + (tid >> 6) + 4 i
  const int kf = k0 + (tid & 63);
  const bool kv = kf < Kf;
  const int tap = kv ? kf / p.Ca : 0, c = kv ? kf - tap * p.Ca : 0;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  CF q;
  q.so = p.so, q.sgn = p.sgn, q.off = p.off, q.up = p.up;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.f;
  for (int64_t mm = mb; mm < me; mm += TK) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int64_t m = mm + (tid >> 6) + 4 * i;
      float g = 0.f, a = 0.f;
      if (m < me) {
        const int n = n0 + (tid & 63);
        if (n < p.Cg) g = p.G[m * p.ldG + n];
        if (kv) {
          const int x = (int)(m % p.Wg), y = (int)((m / p.Wg) % p.Hg), b = (int)(m / ((int64_t)p.Wg * p.Hg));
          int sy, sx;
          if (src_coord(y, ky, q, p.Hi, sy) && src_coord(x, kx, q, p.Wi, sx)) a = p.A[(((int64_t)b * p.Hi + sy) * p.Wi + sx) * p.ldA + c];
        }
      }
      Gs[(tid >> 6) + 4 * i][tid & 63] = g;
      As[(tid >> 6) + 4 * i][tid & 63] = a;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; kk++) {
      float g[4], a[4];
#pragma unroll
      for (int i = 0; i < 4; i++) g[i] = Gs[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; j++) a[j] = As[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = fmaf(g[i], a[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* P = p.partial + (int64_t)blockIdx.z * p.Cg * Kf;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int n = n0 + ty * 4 + i;
    if (n >= p.Cg) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = k0 + tx * 4 + j;
      if (k < Kf) P[(int64_t)n * Kf + k] = acc[i][j];
    }
  }
}

__global__ __launch_bounds__(256) void k_wgrad_f32_reduce(const float* __restrict__ partial, int nz, int Cg, int Ca, int KH, int KW,
                                                           float* __restrict__ dW, int64_t w_sn, int64_t w_sc, int64_t w_sy,
                                                           int64_t w_sx, int accumulate) {
  const int Kf = KH * KW * Ca;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)Cg * Kf) return;
  double s = 0.0;
  for (int z = 0; z < nz; z++) s += (double)partial[(int64_t)z * Cg * Kf + e];
  const int n = (int)(e / Kf), kf = (int)(e - (int64_t)n * Kf);
  const int tap = kf / Ca, c = kf - tap * Ca, ky = tap / KW, kx = tap - ky * KW;
  float* d = dW + n * w_sn + c * w_sc + ky * w_sy + kx * w_sx;
  *d = accumulate ? *d + (float)s : (float)s;
}

// out[c] (+)= sum over rows of x[:, c]   (bias gradients), fp64 accumulation, fixed order
__global__ __launch_bounds__(256) void k_colsum_f32(const float* __restrict__ x, int ld, int64_t N, int C, float* __restrict__ out,
                                                     int accumulate) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int64_t r = threadIdx.x; r < N; r += 256) s += (double)x[r * ld + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)red[0] : (float)red[0];
}

}  // namespace

extern "C" {

// see the file header for the index maps; A [B,Hi,Wi,Ca] and O [B,Ho,Wo,Cn] are NHWC fp32 with pixel pitches ldA / ldO
int mm_conv2d_f32(const float* A, int B, int Hi, int Wi, int Ca, int ldA, float* O, int Ho, int Wo, int Cn, int ldO, int KH, int KW,
                  int so, int sgn, int off, int up, const float* W, int64_t w_sn, int64_t w_sc, int64_t w_sy, int64_t w_sx,
                  const float* bias, hipStream_t s) {
  MM_CHECK_ARG(A && O && W && B > 0 && Hi > 0 && Wi > 0 && Ca > 0 && Ho > 0 && Wo > 0 && Cn > 0 && KH > 0 && KW > 0 && up >= 1 &&
                   (sgn == 1 || sgn == -1) && ldA >= Ca && ldO >= Cn,
               "conv2d_f32: bad arguments");
  CF p{A, W, bias, O, B, Hi, Wi, Ca, ldA, Ho, Wo, Cn, ldO, KH, KW, so, sgn, off, up, w_sn, w_sc, w_sy, w_sx};
  const int64_t M = (int64_t)B * Ho * Wo;
  hipLaunchKernelGGL(k_conv_f32, dim3((unsigned)mm_cdiv(M, TM), (unsigned)mm_cdiv(Cn, TN)), dim3(256), 0, s, p);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

static int wgrad_chunks(int64_t M) {
  int64_t z = mm_cdiv(M, 4096);
  return (int)(z < 1 ? 1 : z > 512 ? 512 : z);
}
size_t mm_conv2d_f32_wgrad_ws_bytes(int64_t n_pixels, int Cg, int Ca, int KH, int KW) {
  return (size_t)wgrad_chunks(n_pixels) * Cg * Ca * KH * KW * sizeof(float) + 256;
}
// dW[n*w_sn + c*w_sc + ky*w_sy + kx*w_sx] (+)= sum over the G grid of G[b,y,x,n] * A[b, (y*so + sgn*ky + off)/up, .., c]
int mm_conv2d_f32_wgrad(const float* G, int B, int Hg, int Wg, int Cg, int ldG, const float* A, int Hi, int Wi, int Ca, int ldA,
                        int KH, int KW, int so, int sgn, int off, int up, float* dW, int64_t w_sn, int64_t w_sc, int64_t w_sy,
                        int64_t w_sx, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(G && A && dW && B > 0 && Cg > 0 && Ca > 0 && KH > 0 && KW > 0 && up >= 1, "conv2d_f32_wgrad: bad arguments");
  const int64_t M = (int64_t)B * Hg * Wg;
  const int nz = wgrad_chunks(M);
  const int Kf = KH * KW * Ca;
  if ((size_t)nz * Cg * Kf * sizeof(float) > ws_bytes) {
    mm_set_error("conv2d_f32_wgrad: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  WF p{G, A, (float*)ws, B, Hg, Wg, Cg, ldG, Hi, Wi, Ca, ldA, KH, KW, so, sgn, off, up, mm_cdiv(mm_cdiv(M, nz), TK) * TK};
  hipLaunchKernelGGL(k_wgrad_f32, dim3((unsigned)mm_cdiv(Cg, TM), (unsigned)mm_cdiv(Kf, TN), nz), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_wgrad_f32_reduce, dim3((unsigned)mm_cdiv((int64_t)Cg * Kf, 256)), dim3(256), 0, s, (const float*)ws, nz, Cg, Ca,
                     KH, KW, dW, w_sn, w_sc, w_sy, w_sx, accumulate);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_colsum_f32(const float* x, int ld, int64_t N, int C, float* out, int accumulate, hipStream_t s) {
  MM_CHECK_ARG(x && out && C > 0 && ld >= C, "colsum_f32: bad arguments");
  hipLaunchKernelGGL(k_colsum_f32, dim3(C), dim3(256), 0, s, x, ld, N, C, out, accumulate);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
