// Row-wise losses over per-point logits [N, C]:
//   weighted cross entropy, ignore_index, weighted-mean reduction   (/root/reference/lib/losses.py:55-68)
//   cross-modal KL( softmax(target) || softmax(pred) ).sum(1).mean() (/root/reference/.../train.py:157-184)
// plus the 2D->3D lifting gather and its deterministic backward (2d_net/model.py:131-137,166-173)
// and the fused flat AdamW update (K16; train.py:627-636 -> torch.optim.AdamW).
// Block partial sums are fp64 and combined in a fixed order: bit-stable run to run.
#include "common.h"

namespace {
constexpr int T = 256;
constexpr int MAX_PART = 1024;
constexpr int MAXC = 32;

__device__ inline double block_sum(double v, double* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = T / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  return red[0];
}

// partial[b] = (sum w*nll, sum w)
__global__ __launch_bounds__(T) void k_ce_fwd(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                               const float* __restrict__ weight, int64_t N, int C, int64_t ignore,
                                               double* __restrict__ partial) {
  __shared__ double red[T];
  double sl = 0.0, sw = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < N; i += (int64_t)gridDim.x * T) {
    int64_t y = labels[i];
    if (y == ignore || y < 0 || y >= C) continue;
    const float* x = logits + i * ld;
    float m = x[0];
    for (int c = 1; c < C; c++) m = fmaxf(m, x[c]);
    float s = 0.f;
    for (int c = 0; c < C; c++) s += expf(x[c] - m);
    float nll = (m + logf(s)) - x[y];
    float w = weight ? weight[y] : 1.f;
    sl += (double)(w * nll);
    sw += (double)w;
  }
  double a = block_sum(sl, red);
  double b = block_sum(sw, red);
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = a;
    partial[2 * blockIdx.x + 1] = b;
  }
}

// one wave: lanes stride the partials (independent loads), then a fixed shuffle tree - a fixed order, bit-stable.  (Rounds 1-3: one
// thread walked all <= 1024 partials through dependent loads: 18-26 us per call, six calls per step.)
__device__ inline double wave_sum64(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;
}

__global__ __launch_bounds__(64) void k_ce_finalize(const double* __restrict__ partial, int nb, float* __restrict__ out /*[2]: loss, sum_w*/) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) {
    a += partial[2 * i];
    b += partial[2 * i + 1];
  }
  a = wave_sum64(a), b = wave_sum64(b);
  if (threadIdx.x == 0) {
    out[0] = (float)(a / b);  // 0/0 = nan, as torch does when every label is ignored
    out[1] = (float)b;
  }
}

// dlogits = gscale * w[y] * (softmax - onehot) / sum_w
__global__ __launch_bounds__(T) void k_ce_bwd(const float* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                               const float* __restrict__ weight, int64_t N, int C, int64_t ignore,
                                               const float* __restrict__ stats, const float* __restrict__ gout,
                                               float* __restrict__ dlogits, int ld_d) {
  int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
  if (i >= N) return;
  int64_t y = labels[i];
  float* d = dlogits + i * ld_d;
  if (y == ignore || y < 0 || y >= C) {
    for (int c = 0; c < C; c++) d[c] = 0.f;
    return;
  }
  const float* x = logits + i * ld;
  float m = x[0];
  for (int c = 1; c < C; c++) m = fmaxf(m, x[c]);
  float s = 0.f;
  for (int c = 0; c < C; c++) s += expf(x[c] - m);
  float k = gout[0] * (weight ? weight[y] : 1.f) / stats[1];
  float inv = 1.f / s;
  for (int c = 0; c < C; c++) d[c] = k * (expf(x[c] - m) * inv - (c == y ? 1.f : 0.f));
}

// partial[b] = sum_i sum_c q_ic (log q_ic - log p_ic)
__global__ __launch_bounds__(T) void k_kl_fwd(const float* __restrict__ pred, int ld_p, const float* __restrict__ tgt,
                                               int ld_t, int64_t N, int C, double* __restrict__ partial) {
  __shared__ double red[T];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < N; i += (int64_t)gridDim.x * T) {
    const float* p = pred + i * ld_p;
    const float* t = tgt + i * ld_t;
    float mp = p[0], mt = t[0];
    for (int c = 1; c < C; c++) {
      mp = fmaxf(mp, p[c]);
      mt = fmaxf(mt, t[c]);
    }
    float sp = 0.f, st = 0.f;
    for (int c = 0; c < C; c++) {
      sp += expf(p[c] - mp);
      st += expf(t[c] - mt);
    }
    float lp = mp + logf(sp), lt = mt + logf(st);
    float r = 0.f;
    for (int c = 0; c < C; c++) {
      float lq = t[c] - lt;
      r += expf(lq) * (lq - (p[c] - lp));
    }
    acc += (double)r;
  }
  double a = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}

__global__ __launch_bounds__(64) void k_kl_finalize(const double* __restrict__ partial, int nb, int64_t N, float* __restrict__ out) {
  double a = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) a += partial[i];
  a = wave_sum64(a);
  if (threadIdx.x == 0) out[0] = (float)(a / (double)N);
}

// dpred = gscale/N * (softmax(pred) - softmax(tgt))
__global__ __launch_bounds__(T) void k_kl_bwd(const float* __restrict__ pred, int ld_p, const float* __restrict__ tgt,
                                               int ld_t, int64_t N, int C, const float* __restrict__ gout,
                                               float* __restrict__ dpred, int ld_d) {
  int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
  if (i >= N) return;
  const float* p = pred + i * ld_p;
  const float* t = tgt + i * ld_t;
  float mp = p[0], mt = t[0];
  for (int c = 1; c < C; c++) {
    mp = fmaxf(mp, p[c]);
    mt = fmaxf(mt, t[c]);
  }
  float sp = 0.f, st = 0.f;
  for (int c = 0; c < C; c++) {
    sp += expf(p[c] - mp);
    st += expf(t[c] - mt);
  }
  float k = gout[0] / (float)N, ip = 1.f / sp, it = 1.f / st;
  for (int c = 0; c < C; c++) dpred[i * ld_d + c] = k * (expf(p[c] - mp) * ip - expf(t[c] - mt) * it);
}

// ---- lifting: out[p][c] = seg[pix_off[p] + c*sc]
__global__ __launch_bounds__(T) void k_lift_gather(const float* __restrict__ seg, int64_t sc,
                                                    const int64_t* __restrict__ pix_off, int64_t N, int C,
                                                    float* __restrict__ out) {
  int64_t gid = (int64_t)blockIdx.x * T + threadIdx.x;
  int64_t p = gid / C;
  int c = (int)(gid - p * C);
  if (p >= N) return;
  out[p * C + c] = seg[pix_off[p] + c * sc];
}

// dseg[upix_off[u] + c*sc] = sum over the points of unique pixel u (ascending point order)
__global__ __launch_bounds__(T) void k_lift_scatter(const float* __restrict__ dout, int C,
                                                     const int64_t* __restrict__ upix_off,
                                                     const int32_t* __restrict__ csr_off,
                                                     const int32_t* __restrict__ csr_pts, int64_t n_unique, int64_t sc,
                                                     float* __restrict__ dseg) {
  int64_t gid = (int64_t)blockIdx.x * T + threadIdx.x;
  int64_t u = gid / C;
  int c = (int)(gid - u * C);
  if (u >= n_unique) return;
  float s = 0.f;
  for (int e = csr_off[u]; e < csr_off[u + 1]; e++) s += dout[(int64_t)csr_pts[e] * C + c];
  dseg[upix_off[u] + c * sc] = s;
}

// runs of equal pixel keys in the key-sorted point order: the first element of each run sums the run (ascending point
// order inside a run because the sort is stable) - no compaction, so the host never needs the number of unique pixels
__global__ __launch_bounds__(T) void k_lift_scatter_runs(const float* __restrict__ dout, int C, const int64_t* __restrict__ order,
                                                          const unsigned char* __restrict__ first,
                                                          const int64_t* __restrict__ sorted_off, int64_t N, int64_t sc,
                                                          float* __restrict__ dseg) {
  int64_t gid = (int64_t)blockIdx.x * T + threadIdx.x;
  int64_t e = gid / C;
  int c = (int)(gid - e * C);
  if (e >= N || !first[e]) return;
  float s = 0.f;
  int64_t j = e;
  do {
    s += dout[order[j] * C + c];
    j++;
  } while (j < N && !first[j]);
  dseg[sorted_off[e] + c * sc] = s;
}

// ---- evaluation (train.py:297-339): argmax of the 2D logits, of the 3D logits and of the softmax average, three
// confusion matrices [target][pred] over the rows whose label != ignore.  Integer counts: order independent.
__global__ __launch_bounds__(T) void k_eval_confusion(const float* __restrict__ l2, int ld2, const float* __restrict__ l3, int ld3,
                                                       const int64_t* __restrict__ labels, int64_t N, int C, int64_t ignore,
                                                       unsigned long long* __restrict__ cm /*[3][C][C]*/) {
  extern __shared__ unsigned int hist[];  // [3][C][C]
  for (int i = threadIdx.x; i < 3 * C * C; i += T) hist[i] = 0u;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * T + threadIdx.x; i < N; i += (int64_t)gridDim.x * T) {
    int64_t y = labels[i];
    if (y == ignore || y < 0 || y >= C) continue;
    const float* a = l2 + i * ld2;
    const float* b = l3 + i * ld3;
    float ma = a[0], mb = b[0];
    int ia = 0, ib = 0;
    for (int c = 1; c < C; c++) {
      if (a[c] > ma) { ma = a[c]; ia = c; }
      if (b[c] > mb) { mb = b[c]; ib = c; }
    }
    float sa = 0.f, sb = 0.f;
    for (int c = 0; c < C; c++) {
      sa += expf(a[c] - ma);
      sb += expf(b[c] - mb);
    }
    float best = -1.f;
    int ie = 0;
    for (int c = 0; c < C; c++) {
      float e = 0.5f * (expf(a[c] - ma) / sa + expf(b[c] - mb) / sb);
      if (e > best) { best = e; ie = c; }
    }
    atomicAdd(&hist[(0 * C + (int)y) * C + ia], 1u);
    atomicAdd(&hist[(1 * C + (int)y) * C + ib], 1u);
    atomicAdd(&hist[(2 * C + (int)y) * C + ie], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C * C; i += T)
    if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}

// ---- AdamW over flat fp32 arenas (torch.optim.AdamW semantics, amsgrad off; same op order as torch's
// single-tensor path: p*=1-lr*wd; m.lerp_(g,1-b1); v=b2*v+(1-b2)*g*g; p-=step_size*m/(sqrt(v)/sqrt(bc2)+eps))
// Loss-scaled training (mm2d3d_amd/amp.py, the fp16 kind of the 16-bit activation mode): the coefficients of an update live
// on the DEVICE, written by k_amp_prepare from the device-resident loss scale, non-finite flag and step counter, so that a
// skipped step (torch.cuda.amp.GradScaler semantics) needs no read-back: DEV = true reads them, and returns when skip is set.
struct AmpCoef {
  float decay, omb1, beta2, omb2, eps, step_size, bc2_sqrt, grad_scale;
  int skip, pad;
};

template <bool VEC, bool DEV = false>
__global__ __launch_bounds__(T) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t n, float decay, float omb1, float beta2,
                                              float omb2, float eps, float step_size, float bc2_sqrt, float grad_scale,
                                              const AmpCoef* __restrict__ dc = nullptr, const int* __restrict__ skip = nullptr,
                                              int nskip = 0) {
  // skip words (the data-parallel reducer's collective "this step's gradients are invalid" flags, mm2d3d_amd/ddp.py): decided on
  // the device, uniform - the host never reads them before queueing the update
  for (int i = 0; i < nskip; i++)
    if (skip[i]) return;
  if (DEV) {
    if (dc->skip) return;  // uniform
    decay = dc->decay, omb1 = dc->omb1, beta2 = dc->beta2, omb2 = dc->omb2, eps = dc->eps, step_size = dc->step_size;
    bc2_sqrt = dc->bc2_sqrt, grad_scale = dc->grad_scale;
  }
  int64_t i = ((int64_t)blockIdx.x * T + threadIdx.x) * 4;
  if (i >= n) return;
  auto upd = [&](float gj, float& pj, float& mj, float& vj) {
    const float gi = gj * grad_scale;
    const float pi = pj * decay;
    const float mi = mj + omb1 * (gi - mj);
    const float vi = beta2 * vj + omb2 * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pj = pi - step_size * (mi / denom);
    mj = mi;
    vj = vi;
  };
  if (VEC && i + 4 <= n) {  // 16-byte accesses (host: all four pointers 16-B aligned); same arithmetic per element
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 G = *(const f4*)(g + i);
    f4 P = *(f4*)(p + i), M = *(f4*)(m + i), V = *(f4*)(v + i);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float pj = P[j], mj = M[j], vj = V[j];
      upd(G[j], pj, mj, vj);
      P[j] = pj, M[j] = mj, V[j] = vj;
    }
    *(f4*)(p + i) = P;
    *(f4*)(m + i) = M;
    *(f4*)(v + i) = V;
    return;
  }
  for (int j = 0; j < 4 && i + j < n; j++) upd(g[i + j], p[i + j], m[i + j], v[i + j]);  // a thread owns elements [i, i+4)
}

// found[0] |= any element of g is inf / nan (benign race: every writer stores 1)
__global__ __launch_bounds__(T) void k_grad_nonfinite(const float* __restrict__ g, int64_t n, int* __restrict__ found) {
  const int64_t stride = (int64_t)gridDim.x * T * 4;
  bool bad = false;
  for (int64_t i = ((int64_t)blockIdx.x * T + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n && (((uintptr_t)(g + i)) & 15) == 0) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4 G = *(const f4*)(g + i);
#pragma unroll
      for (int j = 0; j < 4; j++) bad |= !(fabsf(G[j]) <= 3.402823466e38f);
    } else {
      for (int j = 0; j < 4 && i + j < n; j++) bad |= !(fabsf(g[i + j]) <= 3.402823466e38f);
    }
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) found[0] = 1;
}

// one thread: the update coefficients of one parameter group from the device state.  t = *step + 1 is the step this update
// would be; it is committed to *step only when the step is taken and ``advance`` is set (first group of an optimiser).
__global__ void k_amp_prepare(const float* __restrict__ scale, const int* __restrict__ found, int nfound, long long* __restrict__ step,
                              int advance, double lr, double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                              AmpCoef* __restrict__ out) {
  // ONE decision for every optimiser of the step (the reference's HybridOptim is one optimiser to Lightning's GradScaler:
  // train.py:627-636 - a non-finite gradient in either network skips both updates) and for the caller's extra skip words
  int skip = 0;
  for (int i = 0; i < nfound; i++) skip |= found[i] != 0;
  const long long t = step[0] + (advance ? 1 : 0);
  if (!skip && advance) step[0] = t;
  const double tt = (double)(t > 0 ? t : 1);
  const double bc1 = 1.0 - pow(beta1, tt), bc2 = 1.0 - pow(beta2, tt);
  AmpCoef c;
  c.decay = (float)(1.0 - lr * weight_decay), c.omb1 = (float)(1.0 - beta1), c.beta2 = (float)beta2, c.omb2 = (float)(1.0 - beta2);
  c.eps = (float)eps, c.step_size = (float)(lr / bc1), c.bc2_sqrt = (float)sqrt(bc2);
  c.grad_scale = (float)(grad_scale / (double)scale[0]);
  c.skip = skip, c.pad = 0;
  *out = c;
}

// GradScaler.update(): found -> scale *= backoff, tracker = 0; else tracker += 1 and scale *= growth every ``interval`` clean steps
__global__ void k_amp_update(float* __restrict__ scale, int* __restrict__ tracker, const int* __restrict__ found, int nfound,
                             float growth, float backoff, int interval) {
  int any = 0;
  for (int i = 0; i < nfound; i++) any |= found[i];
  if (any) {
    scale[0] *= backoff;
    tracker[0] = 0;
  } else if (++tracker[0] >= interval) {
    scale[0] *= growth;
    tracker[0] = 0;
  }
}
}  // namespace

extern "C" {

size_t mm_loss_ws_bytes() { return mm_align((size_t)MAX_PART * 2 * sizeof(double)) + 256; }

static inline int loss_blocks(int64_t N) {
  int64_t nb = mm_cdiv(N > 0 ? N : 1, (int64_t)T * 4);
  return (int)(nb > MAX_PART ? MAX_PART : nb);
}

// stats[0] = loss, stats[1] = sum of weights of the non-ignored rows
int mm_ce_fwd(const float* logits, int ld, const int64_t* labels, const float* weight, int64_t N, int C, int64_t ignore_index,
              float* stats, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(C > 0 && C <= MAXC && ld >= C, "ce: bad C");
  if (ws_bytes < (size_t)MAX_PART * 2 * sizeof(double)) {
    mm_set_error("ce: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  int nb = loss_blocks(N);
  hipLaunchKernelGGL(k_ce_fwd, dim3(nb), dim3(T), 0, s, logits, ld, labels, weight, N, C, ignore_index, (double*)ws);
  hipLaunchKernelGGL(k_ce_finalize, dim3(1), dim3(64), 0, s, (const double*)ws, nb, stats);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_ce_bwd(const float* logits, int ld, const int64_t* labels, const float* weight, int64_t N, int C, int64_t ignore_index,
              const float* stats, const float* grad_out, float* dlogits, int ld_d, hipStream_t s) {
  if (N == 0) return MM_OK;
  hipLaunchKernelGGL(k_ce_bwd, dim3((unsigned)mm_cdiv(N, T)), dim3(T), 0, s, logits, ld, labels, weight, N, C, ignore_index, stats,
                     grad_out, dlogits, ld_d);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_kl_fwd(const float* pred, int ld_p, const float* tgt, int ld_t, int64_t N, int C, float* loss, void* ws, size_t ws_bytes,
              hipStream_t s) {
  MM_CHECK_ARG(C > 0 && C <= MAXC, "kl: bad C");
  if (ws_bytes < (size_t)MAX_PART * sizeof(double)) {
    mm_set_error("kl: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  int nb = loss_blocks(N);
  hipLaunchKernelGGL(k_kl_fwd, dim3(nb), dim3(T), 0, s, pred, ld_p, tgt, ld_t, N, C, (double*)ws);
  hipLaunchKernelGGL(k_kl_finalize, dim3(1), dim3(64), 0, s, (const double*)ws, nb, N, loss);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_kl_bwd(const float* pred, int ld_p, const float* tgt, int ld_t, int64_t N, int C, const float* grad_out, float* dpred,
              int ld_d, hipStream_t s) {
  if (N == 0) return MM_OK;
  hipLaunchKernelGGL(k_kl_bwd, dim3((unsigned)mm_cdiv(N, T)), dim3(T), 0, s, pred, ld_p, tgt, ld_t, N, C, grad_out, dpred, ld_d);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// out[p][c] = seg[pix_off[p] + c*chan_stride]   (pix_off = element offset of channel 0 of the point's pixel)
int mm_lift_gather(const float* seg, int64_t chan_stride, const int64_t* pix_off, int64_t N, int C, float* out, hipStream_t s) {
  if (N == 0) return MM_OK;
  hipLaunchKernelGGL(k_lift_gather, dim3((unsigned)mm_cdiv(N * C, T)), dim3(T), 0, s, seg, chan_stride, pix_off, N, C, out);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// dseg must be zero-filled by the caller; duplicate pixels accumulate in ascending point order
int mm_lift_scatter(const float* dout, int C, const int64_t* upix_off, const int32_t* csr_off, const int32_t* csr_pts,
                    int64_t n_unique, int64_t chan_stride, float* dseg, hipStream_t s) {
  if (n_unique == 0) return MM_OK;
  hipLaunchKernelGGL(k_lift_scatter, dim3((unsigned)mm_cdiv(n_unique * C, T)), dim3(T), 0, s, dout, C, upix_off, csr_off, csr_pts,
                     n_unique, chan_stride, dseg);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// same as mm_lift_scatter without compaction: order = stable argsort of the pixel keys, first[e] = 1 at run starts,
// sorted_off[e] = element offset of channel 0 of the pixel of sorted element e
int mm_lift_scatter_runs(const float* dout, int C, const int64_t* order, const unsigned char* first, const int64_t* sorted_off,
                         int64_t N, int64_t chan_stride, float* dseg, hipStream_t s) {
  if (N == 0) return MM_OK;
  hipLaunchKernelGGL(k_lift_scatter_runs, dim3((unsigned)mm_cdiv(N * C, T)), dim3(T), 0, s, dout, C, order, first, sorted_off, N,
                     chan_stride, dseg);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// cm int64 [3][C][C] (2D, 3D, softmax-average ensemble), accumulated (caller zeroes it at the start of an epoch)
int mm_eval_confusion(const float* logits2d, int ld2, const float* logits3d, int ld3, const int64_t* labels, int64_t N, int C,
                      int64_t ignore_index, int64_t* cm, hipStream_t s) {
  MM_CHECK_ARG(C > 0 && C <= MAXC, "eval_confusion: bad C");
  if (N == 0) return MM_OK;
  int nb = (int)mm_cdiv(N, (int64_t)T * 8);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_eval_confusion, dim3(nb), dim3(T), (size_t)3 * C * C * sizeof(unsigned int), s, logits2d, ld2, logits3d, ld3,
                     labels, N, C, ignore_index, (unsigned long long*)cm);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                  double weight_decay, int64_t step, double grad_scale, const int* skip_dev, int nskip, hipStream_t s) {
  MM_CHECK_ARG(step >= 1, "adamw: step counts from 1");
  MM_CHECK_ARG(nskip >= 0 && nskip <= 16 && (nskip == 0 || skip_dev), "adamw: bad skip words");
  if (n == 0) return MM_OK;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;  // parameter spans may start anywhere
  if (vec)
    hipLaunchKernelGGL(k_adamw<true>, dim3((unsigned)mm_cdiv(n, (int64_t)T * 4)), dim3(T), 0, s, p, g, m, v, n,
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)(lr / bc1), (float)sqrt(bc2), (float)grad_scale, (const AmpCoef*)nullptr, skip_dev, nskip);
  else
    hipLaunchKernelGGL(k_adamw<false>, dim3((unsigned)mm_cdiv(n, (int64_t)T * 4)), dim3(T), 0, s, p, g, m, v, n,
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)(lr / bc1), (float)sqrt(bc2), (float)grad_scale, (const AmpCoef*)nullptr, skip_dev, nskip);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// ---- loss-scaled steps (torch.cuda.amp.GradScaler semantics without a read-back; mm2d3d_amd/amp.py)
int mm_grad_nonfinite(const float* g, int64_t n, int* found_dev, hipStream_t s) {
  if (n == 0) return MM_OK;
  int64_t nb = mm_cdiv(n, (int64_t)T * 4);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_grad_nonfinite, dim3((unsigned)nb), dim3(T), 0, s, g, n, found_dev);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_amp_coef_bytes(void) { return (int)sizeof(AmpCoef); }

int mm_amp_prepare(const float* scale_dev, const int* found_dev, int nfound, int64_t* step_dev, int advance, double lr, double beta1,
                   double beta2, double eps, double weight_decay, double grad_scale, void* coef_dev, hipStream_t s) {
  MM_CHECK_ARG(scale_dev && found_dev && step_dev && coef_dev && nfound >= 1 && nfound <= 32, "amp_prepare: bad argument");
  hipLaunchKernelGGL(k_amp_prepare, dim3(1), dim3(1), 0, s, scale_dev, found_dev, nfound, (long long*)step_dev, advance, lr, beta1, beta2, eps,
                     weight_decay, grad_scale, (AmpCoef*)coef_dev);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// mm_adamw_step with the coefficients of mm_amp_prepare (a step whose coefficients say "skip" leaves p, m, v untouched)
int mm_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const void* coef_dev, hipStream_t s) {
  MM_CHECK_ARG(coef_dev != nullptr, "adamw_step_dev: no coefficients");
  if (n == 0) return MM_OK;
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL((k_adamw<true, true>), dim3((unsigned)mm_cdiv(n, (int64_t)T * 4)), dim3(T), 0, s, p, g, m, v, n, 0.f, 0.f, 0.f, 0.f,
                       0.f, 0.f, 0.f, 0.f, (const AmpCoef*)coef_dev);
  else
    hipLaunchKernelGGL((k_adamw<false, true>), dim3((unsigned)mm_cdiv(n, (int64_t)T * 4)), dim3(T), 0, s, p, g, m, v, n, 0.f, 0.f, 0.f, 0.f,
                       0.f, 0.f, 0.f, 0.f, (const AmpCoef*)coef_dev);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_amp_update(float* scale_dev, int* tracker_dev, const int* found_dev, int nfound, double growth, double backoff, int interval,
                  hipStream_t s) {
  MM_CHECK_ARG(scale_dev && tracker_dev && found_dev && nfound >= 0 && interval >= 1, "amp_update: bad arguments");
  hipLaunchKernelGGL(k_amp_update, dim3(1), dim3(1), 0, s, scale_dev, tracker_dev, found_dev, nfound, (float)growth, (float)backoff, interval);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
