// Per-point / per-voxel row kernels of the 3D branch (SURVEY.md K1 mean, K6, K8, K9).
//   gate        feats *= sigmoid(feats.w + b)           (/root/reference/.../3d_net/model.py:46-48)
//   input mean  voxel row = mean of its points' rows    (scn.InputLayer mode 4, scn_unet.py:113)
//   output      point row = its voxel's row             (scn.OutputLayer, scn_unet.py:117)
//   linear      [N,Cin].[Cin,Cout] + b heads            (model.py:50,85)
// All reductions run in a fixed order (CSR lists ascending, block partials summed in order): bit-stable.
#include "common.h"

namespace {
constexpr int T = 256;
constexpr int MAX_PART = 512;

__device__ inline float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

__global__ __launch_bounds__(T) void k_gate_fwd(const float* __restrict__ x, int64_t N, int C,
                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                 float* __restrict__ y, float* __restrict__ mask) {
  int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
  if (p >= N) return;
  float z = b[0];
  for (int c = 0; c < C; c++) z = fmaf(x[p * C + c], w[c], z);
  float m = sigmoidf_(z);
  mask[p] = m;
  for (int c = 0; c < C; c++) y[p * C + c] = x[p * C + c] * m;
}

// dy -> (dx optional), partial sums for dw[C], db.  y = x*m, m = sigmoid(z), z = x.w+b
__global__ __launch_bounds__(T) void k_gate_bwd(const float* __restrict__ x, const float* __restrict__ mask,
                                                 const float* __restrict__ dy, int64_t N, int C,
                                                 const float* __restrict__ w, float* __restrict__ dx,
                                                 double* __restrict__ partial /*[grid][C+1]*/) {
  __shared__ double red[T];
  double acc[9];
  for (int c = 0; c <= C; c++) acc[c] = 0.0;
  for (int64_t p = (int64_t)blockIdx.x * T + threadIdx.x; p < N; p += (int64_t)gridDim.x * T) {
    float m = mask[p];
    float dm = 0.f;
    for (int c = 0; c < C; c++) dm = fmaf(dy[p * C + c], x[p * C + c], dm);
    float dz = dm * m * (1.f - m);
    for (int c = 0; c < C; c++) {
      acc[c] += (double)(dz * x[p * C + c]);
      if (dx) dx[p * C + c] = dy[p * C + c] * m + dz * w[c];
    }
    acc[C] += (double)dz;
  }
  for (int c = 0; c <= C; c++) {
    __syncthreads();
    red[threadIdx.x] = acc[c];
    __syncthreads();
    for (int s = T / 2; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * (C + 1) + c] = red[0];
  }
}

__global__ void k_sum_partials(const double* __restrict__ partial, int nblk, int ne, float* __restrict__ out0, int n0,
                               float* __restrict__ out1, int accumulate) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  double s = 0.0;
  for (int b = 0; b < nblk; b += 8) {  // eight independent loads per pass (a rolled loop is one memory round trip per block)
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = partial[(int64_t)(b + j < nblk ? b + j : nblk - 1) * ne + e];
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (b + j < nblk) s += v[j];
  }
  float* d = e < n0 ? out0 + e : out1 + (e - n0);
  *d = accumulate ? *d + (float)s : (float)s;
}

__global__ __launch_bounds__(T) void k_seg_mean(const float* __restrict__ feats, int ld_f, int C,
                                                 const int32_t* __restrict__ csr_off,
                                                 const int32_t* __restrict__ csr_items, int64_t n_vox, int mean,
                                                 float* __restrict__ out, int ld_o) {
  const unsigned gid = blockIdx.x * (unsigned)T + threadIdx.x;  // 32-bit index arithmetic (host: n_vox * C < 2^32)
  const unsigned vu = gid / (unsigned)C;
  const int c = (int)(gid - vu * (unsigned)C);
  const int64_t v = vu;
  if (v >= n_vox) return;
  int a = csr_off[v], b = csr_off[v + 1];
  float s = 0.f;
  for (int e = a; e < b; e++) s += feats[(int64_t)csr_items[e] * ld_f + c];
  out[v * ld_o + c] = mean ? s / (float)(b - a) : s;
}

// out[p] = vox[p2v[p]] * (scale_by_count ? 1/count : 1)
__global__ __launch_bounds__(T) void k_row_gather(const float* __restrict__ vox, int ld_v, int C,
                                                   const int32_t* __restrict__ p2v, const int32_t* __restrict__ csr_off,
                                                   int div_count, int64_t N, float* __restrict__ out, int ld_o) {
  const unsigned gid = blockIdx.x * (unsigned)T + threadIdx.x;  // 32-bit index arithmetic (host: N * C < 2^32)
  const unsigned pu = gid / (unsigned)C;
  const int c = (int)(gid - pu * (unsigned)C);
  const int64_t p = pu;
  if (p >= N) return;
  int v = p2v[p];
  float x = 0.f;
  if (v >= 0) {
    x = vox[(int64_t)v * ld_v + c];
    if (div_count) x /= (float)(csr_off[v + 1] - csr_off[v]);
  }
  out[p * ld_o + c] = x;
}

// y[n, co] = sum_ci x[n, ci] * w[co, ci] + b[co]      (torch nn.Linear layout: weight [Cout, Cin])
__global__ __launch_bounds__(T) void k_linear_fwd(const float* __restrict__ x, int ld_x, int64_t N, int Cin, int Cout,
                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                   float* __restrict__ y, int ld_y) {
  const unsigned gid = blockIdx.x * (unsigned)T + threadIdx.x;  // 32-bit index arithmetic (host: N * Cout < 2^32)
  const unsigned nu = gid / (unsigned)Cout;
  const int co = (int)(gid - nu * (unsigned)Cout);
  const int64_t n = nu;
  if (n >= N) return;
  float acc = b ? b[co] : 0.f;
  for (int ci = 0; ci < Cin; ci++) acc = fmaf(x[n * ld_x + ci], w[co * Cin + ci], acc);
  y[n * ld_y + co] = acc;
}

// dx[n, ci] (+)= sum_co dy[n, co] * w[co, ci]
__global__ __launch_bounds__(T) void k_linear_bwd_x(const float* __restrict__ dy, int ld_dy, int64_t N, int Cin, int Cout,
                                                     const float* __restrict__ w, float* __restrict__ dx, int ld_dx,
                                                     int accumulate) {
  const unsigned gid = blockIdx.x * (unsigned)T + threadIdx.x;  // 32-bit index arithmetic (host: N * Cin < 2^32)
  const unsigned nu = gid / (unsigned)Cin;
  const int ci = (int)(gid - nu * (unsigned)Cin);
  const int64_t n = nu;
  if (n >= N) return;
  float acc = 0.f;
  for (int co = 0; co < Cout; co++) acc = fmaf(dy[n * ld_dy + co], w[co * Cin + ci], acc);
  float* d = dx + n * ld_dx + ci;
  *d = accumulate ? *d + acc : acc;
}

// partial[block][co*Cin+ci] = sum_n dy[n,co]*x[n,ci] ; partial[block][Cout*Cin + co] = sum_n dy[n,co]
__global__ __launch_bounds__(T) void k_linear_bwd_w(const float* __restrict__ x, int ld_x, const float* __restrict__ dy,
                                                     int ld_dy, int64_t N, int Cin, int Cout,
                                                     double* __restrict__ partial) {
  extern __shared__ float sm[];  // [64][Cin] + [64][Cout]
  float* xs = sm;
  float* ds = sm + 64 * Cin;
  const int ne = Cout * Cin + Cout;
  const int64_t rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
  for (int e0 = 0; e0 < ne; e0 += T) {
    const int e = e0 + threadIdx.x;
    const bool is_b = e >= Cout * Cin;
    const int co = is_b ? e - Cout * Cin : e / Cin;
    const int ci = is_b ? 0 : e - co * Cin;
    double acc = 0.0;
    for (int64_t t0 = r0; t0 < r1; t0 += 64) {
      __syncthreads();
      for (int i = threadIdx.x; i < 64 * Cin; i += T) {
        int64_t r = t0 + i / Cin;
        xs[i] = r < r1 ? x[r * ld_x + (i % Cin)] : 0.f;
      }
      for (int i = threadIdx.x; i < 64 * Cout; i += T) {
        int64_t r = t0 + i / Cout;
        ds[i] = r < r1 ? dy[r * ld_dy + (i % Cout)] : 0.f;
      }
      __syncthreads();
      if (e < ne) {
        float a = 0.f;
        if (is_b)
          for (int rr = 0; rr < 64; rr++) a += ds[rr * Cout + co];
        else
          for (int rr = 0; rr < 64; rr++) a = fmaf(ds[rr * Cout + co], xs[rr * Cin + ci], a);
        acc += (double)a;
      }
    }
    if (e < ne) partial[(int64_t)blockIdx.x * ne + e] = acc;
  }
}

// ---- the per-point heads (3d_net/model.py:50,85: 16 features -> <= 16 classes over ~560k points): one thread per ROW.
// The generic kernels above give every (row, output) pair its own thread, which re-reads the 64-byte row Cout times through
// 16 scalar loads each (121 us for a 36 MB input); here a thread loads its row with four 16-byte loads and keeps the whole
// [Cout][16] weight matrix in registers via LDS broadcast reads.
typedef float f32x4p __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(T) void k_linear16_fwd(const float* __restrict__ x, int ld_x, int64_t N, int Cout, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y, int ld_y) {
  __shared__ float ws[16 * 16 + 16];
  for (int i = threadIdx.x; i < Cout * 16; i += T) ws[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += T) ws[256 + i] = b ? b[i] : 0.f;
  __syncthreads();
  const int64_t n = (int64_t)blockIdx.x * T + threadIdx.x;
  if (n >= N) return;
  float xv[16];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const f32x4p t = *(const f32x4p*)(x + n * ld_x + 4 * q);
    xv[4 * q] = t.x, xv[4 * q + 1] = t.y, xv[4 * q + 2] = t.z, xv[4 * q + 3] = t.w;
  }
  for (int co = 0; co < Cout; co++) {
    float acc = ws[256 + co];
#pragma unroll
    for (int ci = 0; ci < 16; ci++) acc = fmaf(xv[ci], ws[co * 16 + ci], acc);
    y[n * ld_y + co] = acc;
  }
}

__global__ __launch_bounds__(T) void k_linear16_bwd_x(const float* __restrict__ dy, int ld_dy, int64_t N, int Cout,
                                                       const float* __restrict__ w, float* __restrict__ dx, int ld_dx, int accumulate) {
  __shared__ float ws[16 * 16];
  for (int i = threadIdx.x; i < Cout * 16; i += T) ws[i] = w[i];
  __syncthreads();
  const int64_t n = (int64_t)blockIdx.x * T + threadIdx.x;
  if (n >= N) return;
  float acc[16];
#pragma unroll
  for (int ci = 0; ci < 16; ci++) acc[ci] = 0.f;
  for (int co = 0; co < Cout; co++) {
    const float g = dy[n * ld_dy + co];
#pragma unroll
    for (int ci = 0; ci < 16; ci++) acc[ci] = fmaf(g, ws[co * 16 + ci], acc[ci]);
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    f32x4p* d = (f32x4p*)(dx + n * ld_dx + 4 * q);
    f32x4p v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    if (accumulate) v += *d;
    *d = v;
  }
}

// partial[block][co*16+ci] = sum_n dy[n,co]*x[n,ci]; partial[block][Cout*16 + co] = sum_n dy[n,co].  Every thread walks rows
// with a block stride and accumulates SIX classes at a time (6 x 17 registers; x is read once per group of six classes, not
// once per class); the 64 lanes of a wave are added by a fixed butterfly in fp32, the four waves in fp64.
__global__ __launch_bounds__(T) void k_linear16_bwd_w(const float* __restrict__ x, int ld_x, const float* __restrict__ dy, int ld_dy,
                                                       int64_t N, int Cout, double* __restrict__ partial) {
  constexpr int CG = 6;
  __shared__ float red[T / 64][CG * 17];
  const int ne = Cout * 16 + Cout;
  const int64_t rows_per_block = (N + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c0 = 0; c0 < Cout; c0 += CG) {
    float acc[CG][17];
#pragma unroll
    for (int c = 0; c < CG; c++)
#pragma unroll
      for (int i = 0; i < 17; i++) acc[c][i] = 0.f;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += T) {
      float xv[16], g[CG];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4p t = *(const f32x4p*)(x + r * ld_x + 4 * q);
        xv[4 * q] = t.x, xv[4 * q + 1] = t.y, xv[4 * q + 2] = t.z, xv[4 * q + 3] = t.w;
      }
#pragma unroll
      for (int c = 0; c < CG; c++) g[c] = c0 + c < Cout ? dy[r * ld_dy + c0 + c] : 0.f;
#pragma unroll
      for (int c = 0; c < CG; c++) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[c][i] = fmaf(g[c], xv[i], acc[c][i]);
        acc[c][16] += g[c];
      }
    }
#pragma unroll
    for (int c = 0; c < CG; c++)
#pragma unroll
      for (int i = 0; i < 17; i++) {
        float v = acc[c][i];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0) red[wave][c * 17 + i] = v;
      }
    __syncthreads();
    if ((int)threadIdx.x < CG * 17) {
      const int c = threadIdx.x / 17, i = threadIdx.x - c * 17;
      if (c0 + c < Cout) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < T / 64; w++) sum += (double)red[w][threadIdx.x];
        partial[(int64_t)blockIdx.x * ne + (i < 16 ? (c0 + c) * 16 + i : Cout * 16 + c0 + c)] = sum;
      }
    }
    __syncthreads();
  }
}

static bool linear16_ok(const float* x, int ld_x, int Cin, int Cout) {
  return Cin == 16 && Cout <= 16 && ld_x % 4 == 0 && ((uintptr_t)x % 16) == 0;
}

}  // namespace

extern "C" {

size_t mm_point_ws_bytes(int Cin, int Cout) {
  return mm_align((size_t)MAX_PART * (size_t)(Cin * Cout + Cout + Cin + 2) * sizeof(double)) + 256;
}

int mm_gate_fwd(const float* x, int64_t N, int C, const float* w, const float* b, float* y, float* mask, hipStream_t s) {
  MM_CHECK_ARG(C > 0 && C <= 8, "gate: C must be in [1,8]");
  if (N == 0) return MM_OK;
  hipLaunchKernelGGL(k_gate_fwd, dim3((unsigned)mm_cdiv(N, T)), dim3(T), 0, s, x, N, C, w, b, y, mask);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_gate_bwd(const float* x, const float* mask, const float* dy, int64_t N, int C, const float* w, float* dx, float* dw,
                float* db, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(C > 0 && C <= 8, "gate: C must be in [1,8]");
  int nb = (int)mm_cdiv(N > 0 ? N : 1, (int64_t)T * 8);
  if (nb > MAX_PART) nb = MAX_PART;
  if (ws_bytes < (size_t)nb * (C + 1) * sizeof(double)) {
    mm_set_error("gate_bwd: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  double* partial = (double*)ws;
  hipLaunchKernelGGL(k_gate_bwd, dim3(nb), dim3(T), 0, s, x, mask, dy, N, C, w, dx, partial);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, s, partial, nb, C + 1, dw, C, db, accumulate);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// voxel rows from point rows through the CSR voxel->points lists (mean != 0: InputLayer mode 4; 0: sum)
int mm_segment_reduce(const float* feats, int ld_f, int C, const int32_t* csr_off, const int32_t* csr_items, int64_t n_vox,
                      int mean, float* out, int ld_o, hipStream_t s) {
  if (n_vox == 0) return MM_OK;
  MM_CHECK_ARG((int64_t)(n_vox * C) < (1ll << 32) - 4096, "k_seg_mean: too many elements for 32-bit thread indices");
  hipLaunchKernelGGL(k_seg_mean, dim3((unsigned)mm_cdiv(n_vox * C, T)), dim3(T), 0, s, feats, ld_f, C, csr_off, csr_items, n_vox,
                     mean, out, ld_o);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// point rows from voxel rows (OutputLayer fwd; InputLayer bwd with div_count != 0)
int mm_row_gather(const float* vox, int ld_v, int C, const int32_t* p2v, const int32_t* csr_off, int div_count, int64_t N,
                  float* out, int ld_o, hipStream_t s) {
  if (N == 0) return MM_OK;
  MM_CHECK_ARG((int64_t)(N * C) < (1ll << 32) - 4096, "k_row_gather: too many elements for 32-bit thread indices");
  hipLaunchKernelGGL(k_row_gather, dim3((unsigned)mm_cdiv(N * C, T)), dim3(T), 0, s, vox, ld_v, C, p2v, csr_off, div_count, N,
                     out, ld_o);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_linear_fwd(const float* x, int ld_x, int64_t N, int Cin, int Cout, const float* w, const float* b, float* y, int ld_y,
                  hipStream_t s) {
  if (N == 0) return MM_OK;
  if (linear16_ok(x, ld_x, Cin, Cout)) {
    hipLaunchKernelGGL(k_linear16_fwd, dim3((unsigned)mm_cdiv(N, T)), dim3(T), 0, s, x, ld_x, N, Cout, w, b, y, ld_y);
    MM_LAUNCH_CHECK();
    return MM_OK;
  }
  MM_CHECK_ARG((int64_t)(N * Cout) < (1ll << 32) - 4096, "k_linear_fwd: too many elements for 32-bit thread indices");
  hipLaunchKernelGGL(k_linear_fwd, dim3((unsigned)mm_cdiv(N * Cout, T)), dim3(T), 0, s, x, ld_x, N, Cin, Cout, w, b, y, ld_y);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// dx (nullable), dw [Cout,Cin], db [Cout] (nullable)
int mm_linear_bwd(const float* x, int ld_x, const float* dy, int ld_dy, int64_t N, int Cin, int Cout, const float* w,
                  float* dx, int ld_dx, int accumulate_dx, float* dw, float* db, int accumulate_w, void* ws, size_t ws_bytes,
                  hipStream_t s) {
  MM_CHECK_ARG(Cin > 0 && Cout > 0 && (size_t)64 * (Cin + Cout) * 4 <= 150 * 1024, "linear_bwd: channels too wide");
  if ((size_t)64 * (Cin + Cout) * 4 > 64 * 1024)
    MM_HIP(hipFuncSetAttribute((const void*)k_linear_bwd_w, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * (Cin + Cout) * 4));
  const bool fast = linear16_ok(x, ld_x, Cin, Cout);
  if (dx && N) {
    if (fast && ld_dx % 4 == 0 && ((uintptr_t)dx % 16) == 0)
      hipLaunchKernelGGL(k_linear16_bwd_x, dim3((unsigned)mm_cdiv(N, T)), dim3(T), 0, s, dy, ld_dy, N, Cout, w, dx, ld_dx, accumulate_dx);
    else
      MM_CHECK_ARG((int64_t)(N * Cin) < (1ll << 32) - 4096, "k_linear_bwd_x: too many elements for 32-bit thread indices");
      hipLaunchKernelGGL(k_linear_bwd_x, dim3((unsigned)mm_cdiv(N * Cin, T)), dim3(T), 0, s, dy, ld_dy, N, Cin, Cout, w, dx, ld_dx,
                         accumulate_dx);
  }
  if (dw) {
    int nb = (int)mm_cdiv(N > 0 ? N : 1, fast ? 4096 : 2048);  // the row-per-thread kernel streams 34 rows per thread at 558k rows
    if (nb > MAX_PART) nb = MAX_PART;
    if (fast && nb > 128) nb = 128;
    const int ne = Cout * Cin + Cout;
    if (ws_bytes < (size_t)nb * ne * sizeof(double)) {
      mm_set_error("linear_bwd: workspace too small");
      return MM_ERR_WORKSPACE;
    }
    double* partial = (double*)ws;
    if (fast)
      hipLaunchKernelGGL(k_linear16_bwd_w, dim3(nb), dim3(T), 0, s, x, ld_x, dy, ld_dy, N, Cout, partial);
    else
      hipLaunchKernelGGL(k_linear_bwd_w, dim3(nb), dim3(T), (size_t)64 * (Cin + Cout) * 4, s, x, ld_x, dy, ld_dy, N, Cin, Cout,
                         partial);
    // bias partials live behind the weight partials; when db is null they are summed into a scratch tail of ws
    float* dbp = db ? db : (float*)((char*)ws + (size_t)nb * ne * sizeof(double));
    if (!db && ws_bytes < (size_t)nb * ne * sizeof(double) + Cout * sizeof(float)) {
      mm_set_error("linear_bwd: workspace too small");
      return MM_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)mm_cdiv(ne, 64)), dim3(64), 0, s, partial, nb, ne, dw, Cout * Cin, dbp,
                       accumulate_w);
  }
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
