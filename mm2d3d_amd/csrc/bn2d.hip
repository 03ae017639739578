// BatchNorm2d (+ residual add) (+ ReLU) on NHWC bf16 activations, fp32 parameters and statistics
// (torch.nn.BatchNorm2d + ReLU inside the ResNet34 blocks, backbones.py:49-63, and the decoder
// stages, 2d_net/model.py:68-81).  HBM-bound: training forward = one statistics pass + one fused
// normalise/add/activate pass; backward = one reduction pass + one pass that writes dx and the residual gradient.
// Rows = B*H*W pixels, 16-B (8 x bf16) accesses, fp64 combination of the per-block partial sums (bit-stable).
#include "common.h"
#include "fused_bn.h"


#include "h16.h"  // bf16 (default) or IEEE fp16 (-DMM_ACT_FP16) storage: this file is built once for each

namespace {
constexpr int T = 256;
constexpr int MAX_PART = 2048;

__device__ inline void ld8(const u16* p, float (&v)[8]) {
  uint4 t = *(const uint4*)p;
  unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    v[2 * i] = h_lo(w[i]);
    v[2 * i + 1] = h_hi(w[i]);
  }
}
__device__ inline void cvt8(const uint4& t, float (&v)[8]) {
  const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    v[2 * i] = h_lo(w[i]);
    v[2 * i + 1] = h_hi(w[i]);
  }
}
__device__ inline void st8(u16* p, const float (&v)[8]) {
  unsigned w[4];
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = (unsigned)f2bf(v[2 * i]) | ((unsigned)f2bf(v[2 * i + 1]) << 16);
  *(uint4*)p = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- the stems' BatchNorm + ReLU + MaxPool(3, 2, 1) as one forward pass and one backward (round 5).  The stem maps are the largest of
// the net (299 MB each at the bench's size): the separate max-pool read the normalised map again forward, and backward wrote a
// full-resolution gradient map that the batch norm's two backward passes then read.  Backward here: the gradient that reaches a
// full-resolution pixel through the pool is GATHERED from the pooled gradient and the winner indices (<= 4 windows per pixel) inside
// the batch norm's reduce / apply passes - PoolSrc - so that map is never written or read.  The gathered sum is rounded to the 16-bit
// storage format first, as the stored map was: the fused passes compute what the separate kernels computed.
struct PoolSrc {
  const u16* dyp;            // gradient of the pooled map [B][Ho][Wo][C] (pitch ld), NULL: off (the gradient comes as a full map)
  const unsigned char* idx;  // winning tap (kh * 3 + kw) per pooled element [B][Ho][Wo][C]
  int ld, H, W, Ho, Wo, C;
};
__device__ inline uint4 pool_grad8(const PoolSrc& ps, int64_t r, int cv) {
  const unsigned ru = (unsigned)r, t = ru / (unsigned)ps.W;  // 32-bit index arithmetic (host: N < 2^31)
  const int ix = (int)(ru - t * (unsigned)ps.W);
  const int b = (int)(t / (unsigned)ps.H), iy = (int)(t - (unsigned)b * (unsigned)ps.H);
  float sacc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) sacc[i] = 0.f;
  for (int oy = iy / 2; oy <= (iy + 1) / 2; oy++) {  // windows with oy*2-1 <= iy <= oy*2+1 (as k_maxpool_bwd)
    if (oy >= ps.Ho) continue;
    const int kh = iy - (oy * 2 - 1);
    for (int ox = ix / 2; ox <= (ix + 1) / 2; ox++) {
      if (ox >= ps.Wo) continue;
      const int kw = ix - (ox * 2 - 1);
      const int64_t opix = (int64_t)(b * ps.Ho + oy) * ps.Wo + ox;
      const unsigned long long iw = *(const unsigned long long*)(ps.idx + opix * ps.C + cv * 8);
      const uint4 v = *(const uint4*)(ps.dyp + opix * ps.ld + cv * 8);
      const unsigned wv[4] = {v.x, v.y, v.z, v.w};
      const unsigned tap = (unsigned)(kh * 3 + kw);
#pragma unroll
      for (int i = 0; i < 8; i++)
        if (((iw >> (8 * i)) & 0xFFull) == tap) sacc[i] += (i & 1) ? h_hi(wv[i >> 1]) : h_lo(wv[i >> 1]);
    }
  }
  unsigned ow[4];
#pragma unroll
  for (int i = 0; i < 4; i++) ow[i] = (unsigned)f2bf(sacc[2 * i]) | ((unsigned)f2bf(sacc[2 * i + 1]) << 16);
  return make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

// MODE 0: (sum x, sum x^2).  MODE 1: (sum g, sum g*xhat) with g = dy * (yout > 0 if relu).  MODE 2: sum x only.
// POOL: the gradient is gathered from the pooled map (PoolSrc) - a template parameter, not a run-time test of ps.dyp: the test sat
// between the loads of a trip and kept the compiler from issuing them together (round 6: 165 -> see profiles/r06 stream_diag.txt).
// HAS2 / YOUT (MODE 1 without POOL): a second gradient map / the forward output for the ReLU mask are read - compile-time, so that a
// trip's load count is known to the compiler (with run-time tests it waited for vmcnt(0), i.e. also for the loads just issued).
template <int MODE, bool POOL = false, bool HAS2 = false, bool YOUT = false>
__global__ __launch_bounds__(T) void k_bn2d_reduce(const u16* __restrict__ x, int ld_x, const u16* __restrict__ dy, int ld_dy,
                                                    const u16* __restrict__ yout, int ld_y, int relu, int64_t N, int C,
                                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                                    double* __restrict__ partial, int64_t Ns, int nb0,
                                                    const float* __restrict__ weight = nullptr, const float* __restrict__ bias = nullptr,
                                                    const u16* __restrict__ dy2 = nullptr, int ld_dy2 = 0, PoolSrc ps = PoolSrc{}) {
  // rows [0, Ns) are statistics group 0 (blocks [0, nb0)), rows [Ns, N) group 1 (the other blocks): the two domains of a
  // jointly batched training step keep their own batch statistics.  Ns == N: one group.
  __shared__ float red[2][T];
  const int grp = (int)blockIdx.x >= nb0;
  const int lb = grp ? blockIdx.x - nb0 : blockIdx.x, nbg = grp ? gridDim.x - nb0 : nb0;
  const int64_t gbase = grp ? Ns : 0, Ng = grp ? N - Ns : Ns;
  if (MODE == 1) mean += grp * C, invstd += grp * C;
  const int CV = C >> 3;
  const int rs = T / CV;
  const int tid = threadIdx.x;
  const int slot = tid / CV, cv = tid - slot * CV;
  float a[8], b[8], m[8], is[8], sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = b[i] = 0.f;
  // relu without yout: the mask is recomputed from x exactly as the forward decided it (v = fma(x, sc, sh) > 0 in fp32),
  // which saves reading the output map back (layers without a residual input)
  const bool remask = MODE == 1 && relu && yout == nullptr;
  constexpr bool STATIC = MODE == 1 && !POOL;
  const bool has2 = STATIC ? HAS2 : dy2 != nullptr;          // host: HAS2 == (dy2 != NULL), YOUT == (relu && yout != NULL)
  const bool use_y = STATIC ? YOUT : (relu && !remask);
  if (MODE == 1 && slot < rs) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      m[i] = mean[cv * 8 + i];
      is[i] = invstd[cv * 8 + i];
      sc[i] = is[i] * (weight ? weight[cv * 8 + i] : 1.f);
      sh[i] = (bias ? bias[cv * 8 + i] : 0.f) - m[i] * sc[i];
    }
  }
  const int64_t rows_per_block = (Ng + nbg - 1) / nbg;
  const int64_t r0 = gbase + (int64_t)lb * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < gbase + Ng ? r0 + rows_per_block : gbase + Ng;
  if (slot < rs) {
    // Four rows (forward statistics) / two rows (backward reductions) per trip with all their loads issued before any is consumed (round 3: the one-row loop compiled to
    // load -> s_waitcnt vmcnt(0) -> accumulate, one exposed round trip per row and thread).  The rows are accumulated in the
    // same order as before: results are bit-identical.
    auto row_acc = [&](const uint4& tx, const uint4& td, const uint4& t2, const uint4& ty) {
      float xv[8];
      cvt8(tx, xv);
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] += xv[i];
      } else if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          a[i] += xv[i];
          b[i] = fmaf(xv[i], xv[i], b[i]);
        }
      } else {
        float dv[8], yv[8];
        cvt8(td, dv);
        if (has2) {  // second gradient contribution of this map (residual / concat consumer), summed here instead of by an add kernel
          float d2[8];
          cvt8(t2, d2);
#pragma unroll
          for (int i = 0; i < 8; i++) dv[i] += d2[i];
        }
        if (use_y) cvt8(ty, yv);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if (remask) yv[i] = fmaf(xv[i], sc[i], sh[i]);
          float g = (relu && !(yv[i] > 0.f)) ? 0.f : dv[i];
          a[i] += g;
          b[i] = fmaf(g, (xv[i] - m[i]) * is[i], b[i]);
        }
      }
    };
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    constexpr int U = MODE == 1 ? 2 : 4;  // the backward reduction has up to four tensors per row: two rows keep the occupancy
    struct Trip {
      uint4 tx[U], td[U], t2[U], ty[U];
    };
    auto load_trip = [&](Trip& t, int64_t r) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int64_t ru = r + (int64_t)u * rs;
        t.tx[u] = *(const uint4*)(x + ru * ld_x + cv * 8);
        t.td[u] = MODE == 1 ? (POOL ? pool_grad8(ps, ru, cv) : *(const uint4*)(dy + ru * ld_dy + cv * 8)) : z4;
        t.t2[u] = (MODE == 1 && has2) ? *(const uint4*)(dy2 + ru * ld_dy2 + cv * 8) : z4;
        t.ty[u] = (MODE == 1 && use_y) ? *(const uint4*)(yout + ru * ld_y + cv * 8) : z4;
      }
    };
    auto acc_trip = [&](const Trip& t) {
#pragma unroll
      for (int u = 0; u < U; u++) row_acc(t.tx[u], t.td[u], t.t2[u], t.ty[u]);
    };
    int64_t r = r0 + slot;
    const int64_t step = U * (int64_t)rs, last = (U - 1) * (int64_t)rs;
    if (MODE == 1 && !POOL) {
      // software-pipelined (round 6): the next trip's loads are issued before this trip's rows are consumed - the one-deep form
      // drained the memory pipe (s_waitcnt vmcnt(0)) before every trip's arithmetic.  Same rows in the same order: same bits.
      if (r + last < r1) {
        Trip cur;
        load_trip(cur, r);
        r += step;
        while (r + last < r1) {
          Trip nxt;
          load_trip(nxt, r);
          acc_trip(cur);
          cur = nxt;
          r += step;
        }
        acc_trip(cur);
      }
    } else {
      for (; r + last < r1; r += step) {
        Trip t;
        load_trip(t, r);
        acc_trip(t);
      }
    }
    for (; r < r1; r += rs) {
      const uint4 tx = *(const uint4*)(x + r * ld_x + cv * 8);
      const uint4 td = MODE == 1 ? (POOL ? pool_grad8(ps, r, cv) : *(const uint4*)(dy + r * ld_dy + cv * 8)) : z4;
      const uint4 t2 = (MODE == 1 && has2) ? *(const uint4*)(dy2 + r * ld_dy2 + cv * 8) : z4;
      const uint4 ty = (MODE == 1 && use_y) ? *(const uint4*)(yout + r * ld_y + cv * 8) : z4;
      row_acc(tx, td, t2, ty);
    }
  }
  // per-thread fp32 partials cover <= rows_per_block/rs rows (a few hundred); block and grid combination in fp64
#pragma unroll
  for (int i = 0; i < 8; i++) {
    __syncthreads();
    red[0][tid] = a[i];
    red[1][tid] = b[i];
    __syncthreads();
    if (tid < CV) {
      double sa = 0.0, sb = 0.0;
      for (int s = 0; s < rs; s++) {
        sa += (double)red[0][s * CV + tid];
        sb += (double)red[1][s * CV + tid];
      }
      partial[((int64_t)blockIdx.x * 2 + 0) * C + tid * 8 + i] = sa;
      partial[((int64_t)blockIdx.x * 2 + 1) * C + tid * 8 + i] = sb;
    }
  }
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;
}

// sums partial[b][0][c] and partial[b][1][c] over b in [b0, b1) with the 256 threads of the block (fixed tree: lanes stride
// the blocks, shuffle tree per wave, the 4 wave results added in wave order); result valid in thread 0
__device__ inline void block_sum2(const double* __restrict__ partial, int b0, int b1, int C, int c, double& s, double& q) {
  __shared__ double red[2][4];
  double ls = 0.0, lq = 0.0;
  for (int b = b0 + threadIdx.x; b < b1; b += 256) {
    ls += partial[((int64_t)b * 2 + 0) * C + c];
    lq += partial[((int64_t)b * 2 + 1) * C + c];
  }
  ls = wave_sum(ls);
  lq = wave_sum(lq);
  __syncthreads();  // red[] may still be read by the previous call
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = ls, red[1][threadIdx.x >> 6] = lq;
  __syncthreads();
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

// torch semantics: running = (1-momentum)*running + momentum*batch (unbiased variance); with two statistics groups the
// running buffers are updated group 0 first, then group 1 - exactly what two consecutive forward calls do.
__global__ __launch_bounds__(256) void k_bn2d_finalize_fwd(const double* __restrict__ partial, int nb0, int nb1, int64_t Ns, int64_t N,
                                                           int C, float eps, float momentum, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ save_mean,
                                                           float* __restrict__ save_invstd, int64_t* __restrict__ num_batches) {
  const int c = blockIdx.x;
  const int G = nb1 > 0 ? 2 : 1;
  if (c == 0 && threadIdx.x == 0 && num_batches) *num_batches += G;  // nn.BatchNorm2d.num_batches_tracked
  for (int g = 0; g < G; g++) {
    const int b0 = g ? nb0 : 0, b1 = g ? nb0 + nb1 : nb0;
    const int64_t Ng = g ? N - Ns : Ns;
    double s, q;
    block_sum2(partial, b0, b1, C, c, s, q);
    if (threadIdx.x == 0) {
      double mean = Ng > 0 ? s / (double)Ng : 0.0;
      double var = Ng > 0 ? q / (double)Ng - mean * mean : 0.0;
      if (var < 0.0) var = 0.0;
      save_mean[g * C + c] = (float)mean;
      save_invstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        double unbiased = Ng > 1 ? var * (double)Ng / (double)(Ng - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
      }
    }
  }
}

// Batch statistics from the slab the producing convolution filled in its epilogue (csrc/conv2d.hip stats_accum; round 4):
// slab[2 * sub + g][q][C] fp32 = the sums over one 64-pixel sub-block, q = 0: sum, 1: sum of squares.  This kernel adds the slab
// rows of a range of sub-blocks in fp64 in a fixed order (sub-block lanes stride the range, then the lanes in order) and writes
// one row of ``partial`` in k_bn2d_reduce's format, so k_bn2d_finalize_fwd finishes the job: blocks [0, nb0) take group 0's rows,
// [nb0, nb0 + nb1) group 1's; nb1 == 0: one group, both rows of a sub-block are its.  No pass over the map, no atomics.
__global__ __launch_bounds__(256) void k_bn2d_slab_reduce(const float* __restrict__ slab, int64_t nsub, int nb0, int nb1, int C,
                                                          double* __restrict__ partial) {
  __shared__ double red[256 * 4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int g = b >= nb0 ? 1 : 0, nb = g ? nb1 : nb0, r = g ? b - nb0 : b;
  const int64_t per = (nsub + nb - 1) / nb, s0 = r * per, s1 = s0 + per < nsub ? s0 + per : nsub;
  const int n4 = C / 2;                       // float4 columns of one [2][C] slab row
  const int SL = n4 >= 256 ? 1 : 256 / n4;    // sub-block lanes
  const float4* slab4 = (const float4*)slab;
  for (int col0 = 0; col0 < n4; col0 += 256) {
    const int col = col0 + (SL == 1 ? tid : tid % n4), sl = SL == 1 ? 0 : tid / n4;
    const bool act = col < n4 && sl < SL;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (act)
      for (int64_t sub = s0 + sl; sub < s1; sub += SL) {
        const float4 v = slab4[(2 * sub + g) * n4 + col];
        a[0] += (double)v.x, a[1] += (double)v.y, a[2] += (double)v.z, a[3] += (double)v.w;
        if (nb1 == 0) {
          const float4 w = slab4[(2 * sub + 1) * n4 + col];
          a[0] += (double)w.x, a[1] += (double)w.y, a[2] += (double)w.z, a[3] += (double)w.w;
        }
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) red[tid * 4 + k] = a[k];
    __syncthreads();
    if (act && sl == 0) {
      for (int l = 1; l < SL; l++)
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] += red[(l * n4 + col - col0) * 4 + k];
      const int q = (col * 4) / C, c = col * 4 - q * C;
      double* dst = partial + ((int64_t)b * 2 + q) * C + c;
#pragma unroll
      for (int k = 0; k < 4; k++) dst[k] = a[k];
    }
  }
}

// sums[g][0][C] = sum g, sums[g][1][C] = sum g*xhat per statistics group; dweight / dbias are the totals over the groups
__global__ __launch_bounds__(256) void k_bn2d_finalize_bwd(const double* __restrict__ partial, int nb0, int nb1, int C,
                                                           float* __restrict__ sums /*[G][2][C]*/, float* __restrict__ dweight,
                                                           float* __restrict__ dbias, int accumulate) {
  const int c = blockIdx.x;
  const int G = nb1 > 0 ? 2 : 1;
  float ts = 0.f, tq = 0.f;
  for (int g = 0; g < G; g++) {
    const int b0 = g ? nb0 : 0, b1 = g ? nb0 + nb1 : nb0;
    double s, q;
    block_sum2(partial, b0, b1, C, c, s, q);
    if (threadIdx.x == 0) {
      sums[(g * 2 + 0) * C + c] = (float)s;
      sums[(g * 2 + 1) * C + c] = (float)q;
      ts += (float)s, tq += (float)q;  // the order two consecutive backward calls accumulate in
    }
  }
  if (threadIdx.x != 0) return;
  if (dweight) dweight[c] = accumulate ? dweight[c] + tq : tq;
  if (dbias) dbias[c] = accumulate ? dbias[c] + ts : ts;
}

__global__ __launch_bounds__(64) void k_colsum_finalize(const double* __restrict__ partial, int nblk, int C, float* __restrict__ out,
                                                         int accumulate) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) s += partial[((int64_t)b * 2 + 0) * C + c];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// Apply kernels: thread = (row slot, 8-channel group); the per-channel scale/shift live in registers and the thread
// walks rows with a fixed stride, so the only per-row work is the 16-B loads/stores (pure streaming).
constexpr int APPLY_ROWS = 8;  // rows per thread

// y = act((x - mean) * invstd * w + b + res)
__global__ __launch_bounds__(T) void k_bn2d_apply(const u16* __restrict__ x, int ld_x, const u16* __restrict__ res, int ld_r,
                                                   int64_t N, int C, const float* __restrict__ mean,
                                                   const float* __restrict__ invstd, int stat_is_var, float eps,
                                                   const float* __restrict__ weight, const float* __restrict__ bias, int relu,
                                                   u16* __restrict__ y, int ld_y, int64_t Ns, int ab0) {
  const int CV = C >> 3;
  const int rs = T / CV;
  const int slot = threadIdx.x / CV, cv = threadIdx.x - slot * CV;
  if (slot >= rs) return;
  const int grp = (int)blockIdx.x >= ab0;  // statistics group of this block's rows (see k_bn2d_reduce)
  const int64_t gbase = grp ? Ns : 0, gend = grp ? N : Ns;
  mean += grp * C, invstd += grp * C;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    int c = cv * 8 + i;
    float is = stat_is_var ? 1.f / sqrtf(invstd[c] + eps) : invstd[c];
    sc[i] = is * (weight ? weight[c] : 1.f);
    sh[i] = (bias ? bias[c] : 0.f) - mean[c] * sc[i];
  }
  const int64_t r0 = gbase + (int64_t)(grp ? blockIdx.x - ab0 : blockIdx.x) * rs * APPLY_ROWS + slot;
#pragma unroll 4
  for (int k = 0; k < APPLY_ROWS; k++) {
    const int64_t r = r0 + (int64_t)k * rs;
    if (r >= gend) break;
    float xv[8], rv[8], yv[8];
    ld8(x + r * ld_x + cv * 8, xv);
    if (res) ld8(res + r * ld_r + cv * 8, rv);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      float v = fmaf(xv[i], sc[i], sh[i]);
      if (res) v += rv[i];
      yv[i] = (relu && !(v > 0.f)) ? 0.f : v;
    }
    st8(y + r * ld_y + cv * 8, yv);
  }
}

// y = relu((x - mean) * invstd * w + b) written at full resolution AND max-pooled 3x3 / stride 2 / pad 1 in the same pass: thread =
// (pooled pixel, 8-channel group); it normalises the <= 9 pixels of its window (neighbouring windows share pixels: L1 / L2 hits),
// stores the four it owns - (2 oy + {0,1}, 2 ox + {0,1}): H and W even, every pixel has exactly one owner - and keeps the first
// maximum in scan order of the ROUNDED values (what k_maxpool_fwd does on the stored map).  Bs: images [0, Bs) are group 0.
__global__ __launch_bounds__(T) void k_bn2d_apply_pool(const u16* __restrict__ x, int ld_x, int B, int H, int W, int C, int Bs,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        const float* __restrict__ weight, const float* __restrict__ bias,
                                                        u16* __restrict__ y, int ld_y, u16* __restrict__ yp,
                                                        unsigned char* __restrict__ idx, int Ho, int Wo) {
  const int C8 = C >> 3;
  const unsigned gid = blockIdx.x * (unsigned)T + threadIdx.x;  // 32-bit index arithmetic (host: total < 2^32)
  if ((int64_t)gid >= (int64_t)B * Ho * Wo * C8) return;
  const unsigned pixu = gid / (unsigned)C8;
  const int cv = (int)(gid - pixu * (unsigned)C8);
  const unsigned tu = pixu / (unsigned)Wo;
  const int ox = (int)(pixu - tu * (unsigned)Wo);
  const int b = (int)(tu / (unsigned)Ho), oy = (int)(tu - (unsigned)b * (unsigned)Ho);
  const int grp = b >= Bs ? 1 : 0;
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int c = cv * 8 + i;
    sc[i] = invstd[grp * C + c] * (weight ? weight[c] : 1.f);
    sh[i] = (bias ? bias[c] : 0.f) - mean[grp * C + c] * sc[i];
  }
  float best[8];
  unsigned char bi[8];
  bool any = false;
#pragma unroll
  for (int kh = 0; kh < 3; kh++) {
    const int iy = oy * 2 - 1 + kh;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; kw++) {
      const int ix = ox * 2 - 1 + kw;
      if (ix < 0 || ix >= W) continue;
      const int64_t pix = (int64_t)(b * H + iy) * W + ix;
      float xv[8], yv[8];
      ld8(x + pix * ld_x + cv * 8, xv);
      unsigned ow[4];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float v = fmaf(xv[i], sc[i], sh[i]);
        yv[i] = !(v > 0.f) ? 0.f : v;
      }
#pragma unroll
      for (int i = 0; i < 4; i++) ow[i] = (unsigned)f2bf(yv[2 * i]) | ((unsigned)f2bf(yv[2 * i + 1]) << 16);
      if (kh >= 1 && kw >= 1) *(uint4*)(y + pix * ld_y + cv * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float f = (i & 1) ? h_hi(ow[i >> 1]) : h_lo(ow[i >> 1]);  // the stored (rounded) value
        if (!any || f > best[i]) {
          best[i] = f;
          bi[i] = (unsigned char)(kh * 3 + kw);
        }
      }
      any = true;
    }
  }
  unsigned pw[4];
  unsigned long long iw = 0ull;
#pragma unroll
  for (int i = 0; i < 4; i++) pw[i] = (unsigned)f2bf(best[2 * i]) | ((unsigned)f2bf(best[2 * i + 1]) << 16);
#pragma unroll
  for (int i = 0; i < 8; i++) iw |= (unsigned long long)bi[i] << (8 * i);
  *(uint4*)(yp + (int64_t)pixu * C + cv * 8) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
  *(unsigned long long*)(idx + (int64_t)pixu * C + cv * 8) = iw;
}

// g = dy * relu'(yout); dx = w*invstd*(g - sum_g/N - xhat*sum_gx/N); dres = g
__global__ __launch_bounds__(T) void k_bn2d_bwd_apply(const u16* __restrict__ x, int ld_x, const u16* __restrict__ dy, int ld_dy,
                                                       const u16* __restrict__ yout, int ld_y, int relu, int64_t N, int C,
                                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                                       const float* __restrict__ weight, const float* __restrict__ sums,
                                                       u16* __restrict__ dx, int ld_dx, u16* __restrict__ dres, int ld_dr, int64_t Ns,
                                                       int ab0, const float* __restrict__ bias, const u16* __restrict__ dy2, int ld_dy2,
                                                       PoolSrc ps = PoolSrc{}) {
  const int CV = C >> 3;
  const int rs = T / CV;
  const int slot = threadIdx.x / CV, cv = threadIdx.x - slot * CV;
  if (slot >= rs) return;
  const int grp = (int)blockIdx.x >= ab0;
  const int64_t gbase = grp ? Ns : 0, gend = grp ? N : Ns;
  mean += grp * C, invstd += grp * C, sums += grp * 2 * C;
  // dx = a*g + b*x + c0 with a = w*is, b = -w*is^2*sgx/N, c0 = -a*sg/N - b*mean   (N = rows of this statistics group)
  float ka[8], kb[8], kc[8], sc[8], sh[8];
  const float invN = 1.f / (float)(gend - gbase);
  const bool remask = relu && yout == nullptr;  // see k_bn2d_reduce
#pragma unroll
  for (int i = 0; i < 8; i++) {
    int c = cv * 8 + i;
    float is = invstd[c], w = weight ? weight[c] : 1.f;
    sc[i] = is * w;
    sh[i] = (bias ? bias[c] : 0.f) - mean[c] * sc[i];
    ka[i] = w * is;
    kb[i] = -w * is * is * sums[C + c] * invN;
    kc[i] = -ka[i] * sums[c] * invN - kb[i] * mean[c];
  }
  const int64_t r0 = gbase + (int64_t)(grp ? blockIdx.x - ab0 : blockIdx.x) * rs * APPLY_ROWS + slot;
#pragma unroll 4
  for (int k = 0; k < APPLY_ROWS; k++) {
    const int64_t r = r0 + (int64_t)k * rs;
    if (r >= gend) break;
    float xv[8], dv[8], yv[8], ov[8], gv[8];
    ld8(x + r * ld_x + cv * 8, xv);
    if (ps.dyp) cvt8(pool_grad8(ps, r, cv), dv);
    else ld8(dy + r * ld_dy + cv * 8, dv);
    if (dy2) {
      float d2[8];
      ld8(dy2 + r * ld_dy2 + cv * 8, d2);
#pragma unroll
      for (int i = 0; i < 8; i++) dv[i] += d2[i];
    }
    if (relu && !remask) ld8(yout + r * ld_y + cv * 8, yv);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (remask) yv[i] = fmaf(xv[i], sc[i], sh[i]);
      float g = (relu && !(yv[i] > 0.f)) ? 0.f : dv[i];
      gv[i] = g;
      ov[i] = fmaf(ka[i], g, fmaf(kb[i], xv[i], kc[i]));
    }
    st8(dx + r * ld_dx + cv * 8, ov);
    if (dres) st8(dres + r * ld_dr + cv * 8, gv);
  }
}


// ---- single-launch training kernels (shared skeleton: fused_bn.h)
__device__ inline void unpack8(const u32x4 t, float (&v)[8]) {
  const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    v[2 * i] = h_lo(w[i]);
    v[2 * i + 1] = h_hi(w[i]);
  }
}
__device__ inline u32x4 pack8(const float (&v)[8]) {
  u32x4 t;
  t.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
  t.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
  t.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
  t.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
  return t;
}

struct FusedP {
  const u16* x;
  const u16* res;   // forward: residual input; backward: unused
  u16* y;           // forward output
  const u16* dy;
  const u16* dy2;
  const u16* yout;  // backward: forward output for the ReLU mask (NULL: recomputed from x)
  u16* dx;
  u16* dres;
  int ld_x, ld_r, ld_y, ld_dy, ld_dy2, ld_dx, ld_dr;
  int64_t N, Ns;
  int C, relu, G0, G1, R;
  const float *weight, *bias;
  float *running_mean, *running_var;
  int64_t* nbt;
  float eps, momentum;
  float *save_mean, *save_invstd;  // [groups][C]
  float *sums, *dweight, *dbias;   // backward: sums [groups][2][C]
  int accumulate;
  double* partial;  // [G0+G1][2][C]
  unsigned* sync;   // arrival counter at [0], release word at [FUSED_FLAG]
  unsigned* fault;  // the handle's fault word (device address of pinned host memory)
};

// Two problems in one launch (the same layer of the two encoders: one grid barrier pair, one statistics exchange and one launch for
// both): workgroups [0, Ga) run problem a, the others problem b; the barrier counts all of them (a.sync == b.sync).
struct FusedP2 {
  FusedP a, b;
  int Ga;
};

// bid = the workgroup's index within its problem, Gall = workgroups of the whole launch (what the grid barrier counts)
template <int RMAX, bool RES>
__device__ __forceinline__ void bn2d_fused_fwd_body(const FusedP& p, const int bid, const unsigned Gall) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* red = (float*)smem;
  double* red2 = (double*)(smem + FUSED_RED);
  double* outp = red2 + 1024;
  u32x4* rows = (u32x4*)(smem + FUSED_RED + 2 * 8192);
  constexpr int NREG = RMAX > FUSED_NL ? RMAX - FUSED_NL : 1;
  static_assert(RMAX % 4 == 0, "row groups of four");
  const int tid = threadIdx.x;
  const FusedGeom g = fused_geom(p.N, p.Ns, p.C >> 3, p.G0, p.G1, bid);
  const int C = p.C;
  unsigned flag0 = 0;
  if (tid == 0) flag0 = xcd_load(&p.sync[FUSED_FLAG]);
  const int cvc = g.active ? g.cv : 0, slc = g.active ? g.slot : 0;
  const FusedBuf bx = fused_buf(p.x, p.N, p.ld_x, C, slc, cvc), by = fused_buf(p.y, p.N, p.ld_y, C, slc, cvc),
                 br = fused_buf(RES ? p.res : p.x, p.N, RES ? p.ld_r : p.ld_x, C, slc, cvc);
  u32x4 xr[NREG];
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = b[i] = 0.f;
#pragma unroll
  for (int k = 0; k < RMAX; k++) {
    const bool ok = g.active && k < p.R && g.slot + k * g.rs < g.nrows;
    const u32x4 t = fused_ld(bx, g.r0 + (int64_t)k * g.rs, ok);  // zeros where !ok
    if (k < FUSED_NL) rows[k * FT + tid] = t;
    else xr[k < FUSED_NL ? 0 : k - FUSED_NL] = t;
    float xv[8];
    unpack8(t, xv);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      a[i] += xv[i];
      b[i] = fmaf(xv[i], xv[i], b[i]);
    }
    if ((k & 3) == 3) {  // four rows' loads in flight at a time: bounds the live temporaries
      MM_PIN16(a, b);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  fused_block_sums<8>(a, b, red, red2, outp, C, g.CV, g.rs, g.active);
  for (int pr = tid; pr < 2 * C; pr += FT) xcd_store(p.partial + (size_t)bid * 2 * C + pr, outp[pr]);
  const int G = p.G0 + p.G1;
  fused_barrier(p.sync, p.fault, Gall, flag0);
  {
    // statistics: one wave per channel, channels dealt round-robin over the workgroups; both groups by the same wave, the
    // running buffers updated group 0 first, then group 1 (what two consecutive forward calls do)
    const int ngrp = p.G1 > 0 ? 2 : 1;
    if (bid == 0 && tid == 0 && p.nbt) *p.nbt += ngrp;
    for (int c = bid + (tid >> 6) * G; c < C; c += (FT / 64) * G) {
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
          if (p.running_mean) {
            const double unbiased = Ng > 1 ? var * (double)Ng / (double)(Ng - 1) : var;
            p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * (float)mean;
            p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)unbiased;
          }
        }
      }
    }
  }
  fused_barrier(p.sync, p.fault, Gall, flag0 + 1u);
  if (!g.active) return;
  const float *wp = p.weight ? p.weight : p.save_mean, *bp = p.bias ? p.bias : p.save_mean;
  // the kept rows stay PACKED across the barrier: without this the compiler keeps the phase-1 unpacked floats alive instead
  // (common subexpression of the two unpack8 calls), twice the registers
#pragma unroll
  for (int k = 0; k < NREG; k++) asm volatile("" : "+v"(xr[k]));
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int c = g.cv * 8 + i;
    const float is = xcd_load(p.save_invstd + g.grp * C + c), m = xcd_load(p.save_mean + g.grp * C + c);
    const float wv = wp[c], bv = bp[c];  // unconditional loads + selects: no chain of branches on the two null checks
    sc[i] = is * (p.weight ? wv : 1.f);
    sh[i] = (p.bias ? bv : 0.f) - m * sc[i];
  }
  // normalise from the kept rows, four at a time (the residual rows of a group are loaded together).  No break / continue in
  // these loops: the compiler must unroll them completely, or the register arrays are indexed dynamically and land in scratch
#pragma unroll
  for (int k0 = 0; k0 < RMAX; k0 += 4) {
    if (k0 < p.R) {
      u32x4 rv[4];
      if (RES) {
#pragma unroll
        for (int j = 0; j < 4; j++)
          rv[j] = fused_ld(br, g.r0 + (int64_t)(k0 + j) * g.rs, g.active && k0 + j < p.R && g.slot + (k0 + j) * g.rs < g.nrows);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = k0 + j;
        u32x4 t;
        if (k < FUSED_NL) t = rows[k * FT + tid];
        else t = xr[k < FUSED_NL ? 0 : k - FUSED_NL];
        float xv[8], ad[8], yv[8];
        unpack8(t, xv);
        if (RES) unpack8(rv[j], ad);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          float v = fmaf(xv[i], sc[i], sh[i]);
          if (RES) v += ad[i];
          yv[i] = (p.relu && !(v > 0.f)) ? 0.f : v;
        }
        if (k < p.R && g.slot + k * g.rs < g.nrows) fused_st(by, g.r0 + (int64_t)k * g.rs, pack8(yv));
        __builtin_amdgcn_sched_barrier(0);  // one row's arithmetic at a time (the group's loads stay in flight)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int RMAX, bool RES>
__global__ __launch_bounds__(FT) void k_bn2d_fused_fwd(const FusedP p) {
  bn2d_fused_fwd_body<RMAX, RES>(p, (int)blockIdx.x, (unsigned)(p.G0 + p.G1));
}
template <int RMAX, bool RES>
__global__ __launch_bounds__(FT) void k_bn2d_fused_fwd_pair(const FusedP2 pp) {
  const bool second = (int)blockIdx.x >= pp.Ga;
  bn2d_fused_fwd_body<RMAX, RES>(second ? pp.b : pp.a, second ? (int)blockIdx.x - pp.Ga : (int)blockIdx.x, gridDim.x);
}

// Backward: phase 1 keeps x (registers), the first FUSED_NL rows of dy (LDS) and the ReLU mask bits; sums are (sum g, sum g*x), turned
// into sum g*xhat = invstd * (sum g*x - mean * sum g) in fp64 by the last workgroup.
// MASK: 0 = no ReLU, 1 = mask recomputed from x, 2 = mask from the forward output
template <int RMAX, bool DY2, int MASK>
__device__ __forceinline__ void bn2d_fused_bwd_body(const FusedP& p, const int bid, const unsigned Gall) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* red = (float*)smem;
  double* red2 = (double*)(smem + FUSED_RED);
  double* outp = red2 + 1024;
  u32x4* rows = (u32x4*)(smem + FUSED_RED + 2 * 8192);
  constexpr int NREG = RMAX > FUSED_NL ? RMAX - FUSED_NL : 1;
  static_assert(RMAX % 4 == 0, "row groups of four");
  const int tid = threadIdx.x;
  const FusedGeom g = fused_geom(p.N, p.Ns, p.C >> 3, p.G0, p.G1, bid);
  const int C = p.C;
  unsigned flag0 = 0;
  if (tid == 0) flag0 = xcd_load(&p.sync[FUSED_FLAG]);
  const int cvc = g.active ? g.cv : 0, slc = g.active ? g.slot : 0;
  const FusedBuf bx = fused_buf(p.x, p.N, p.ld_x, C, slc, cvc), bd = fused_buf(p.dy, p.N, p.ld_dy, C, slc, cvc),
                 bd2 = fused_buf(DY2 ? p.dy2 : p.dy, p.N, DY2 ? p.ld_dy2 : p.ld_dy, C, slc, cvc),
                 byo = fused_buf(MASK == 2 ? p.yout : p.x, p.N, MASK == 2 ? p.ld_y : p.ld_x, C, slc, cvc),
                 bdx = fused_buf(p.dx, p.N, p.ld_dx, C, slc, cvc),
                 bdr = fused_buf(p.dres ? p.dres : p.dx, p.N, p.dres ? p.ld_dr : p.ld_dx, C, slc, cvc);
  const float* mean = p.save_mean + g.grp * C;
  const float* invstd = p.save_invstd + g.grp * C;
  const float *wp = p.weight ? p.weight : p.save_mean, *bp = p.bias ? p.bias : p.save_mean;
  u32x4 xr[RMAX];
  unsigned mb[RMAX / 4];
#pragma unroll
  for (int i = 0; i < RMAX / 4; i++) mb[i] = 0u;
  float a[8], b[8];
  {
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int c = cvc * 8 + i;
      const float wv = wp[c], bv = bp[c];
      sc[i] = invstd[c] * (p.weight ? wv : 1.f);
      sh[i] = (p.bias ? bv : 0.f) - mean[c] * sc[i];
      a[i] = b[i] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < RMAX; k++) {
      const bool ok = g.active && k < p.R && g.slot + k * g.rs < g.nrows;
      const int64_t row = g.r0 + (int64_t)k * g.rs;
      const u32x4 tx = fused_ld(bx, row, ok);
      const u32x4 td = fused_ld(bd, row, ok);  // zeros where !ok
      float xv[8], dv[8], yv[8];
      unpack8(tx, xv);
      unpack8(td, dv);
      if (DY2) {
        float d2[8];
        unpack8(fused_ld(bd2, row, ok), d2);
#pragma unroll
        for (int i = 0; i < 8; i++) dv[i] = ok ? dv[i] + d2[i] : 0.f;
      }
      if (MASK == 2) unpack8(fused_ld(byo, row, ok), yv);
      unsigned m8 = 0u;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (MASK == 1) yv[i] = fmaf(xv[i], sc[i], sh[i]);
        const bool keep = MASK == 0 || yv[i] > 0.f;
        m8 |= keep ? (1u << i) : 0u;
        const float gg = keep ? dv[i] : 0.f;
        a[i] += gg;
        b[i] = fmaf(gg, xv[i], b[i]);
      }
      mb[k >> 2] |= m8 << (8 * (k & 3));
      xr[k] = tx;
      if (k < FUSED_NL) rows[k * FT + tid] = td;  // rows beyond the LDS budget are read again in phase 2 (L2 / Infinity Cache hits)
      if ((k & 1) == 1) {  // two rows' loads in flight at a time: bounds the live temporaries
        MM_PIN16(a, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  fused_block_sums<8>(a, b, red, red2, outp, C, g.CV, g.rs, g.active);
  for (int pr = tid; pr < 2 * C; pr += FT) xcd_store(p.partial + (size_t)bid * 2 * C + pr, outp[pr]);
  const int G = p.G0 + p.G1;
  fused_barrier(p.sync, p.fault, Gall, flag0);
  {
    const int ngrp = p.G1 > 0 ? 2 : 1;
    for (int c = bid + (tid >> 6) * G; c < C; c += (FT / 64) * G) {
      float ts = 0.f, tq = 0.f;
      for (int gi = 0; gi < ngrp; gi++) {
        double sg, sgx;
        fused_wave_sums(p.partial, gi ? p.G0 : 0, gi ? G : p.G0, C, c, sg, sgx);
        const double m = (double)p.save_mean[gi * C + c], is = (double)p.save_invstd[gi * C + c];
        const float sv = (float)sg, qv = (float)(is * (sgx - m * sg));
        if ((tid & 63) == 0) {
          xcd_store(p.sums + (gi * 2 + 0) * C + c, sv);
          xcd_store(p.sums + (gi * 2 + 1) * C + c, qv);
        }
        ts += sv, tq += qv;  // the order two consecutive backward calls accumulate in
      }
      if ((tid & 63) == 0) {
        if (p.dweight) p.dweight[c] = p.accumulate ? p.dweight[c] + tq : tq;
        if (p.dbias) p.dbias[c] = p.accumulate ? p.dbias[c] + ts : ts;
      }
    }
  }
  fused_barrier(p.sync, p.fault, Gall, flag0 + 1u);
  if (!g.active) return;
#pragma unroll
  for (int k = 0; k < RMAX; k++) asm volatile("" : "+v"(xr[k]));  // keep the rows packed across the barrier (see the forward kernel)
#pragma unroll
  for (int i = 0; i < RMAX / 4; i++) asm volatile("" : "+v"(mb[i]));  // ... and the mask as bits, not as the compared floats
  // dx = ka*g + kb*x + kc (see k_bn2d_bwd_apply)
  float ka[8], kb[8], kc[8];
  const float invN = 1.f / (float)g.Ng;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int c = g.cv * 8 + i;
    const float wv = wp[c];
    const float is = invstd[c], w = p.weight ? wv : 1.f;
    const float sg = xcd_load(p.sums + (g.grp * 2 + 0) * C + c), sq = xcd_load(p.sums + (g.grp * 2 + 1) * C + c);
    ka[i] = w * is;
    kb[i] = -w * is * is * sq * invN;
    kc[i] = -ka[i] * sg * invN - kb[i] * mean[c];
  }
#pragma unroll
  for (int k0 = 0; k0 < RMAX; k0 += 4) {
    if (k0 < p.R) {
      u32x4 d2[4], d1[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool okj = g.active && k0 + j < p.R && g.slot + (k0 + j) * g.rs < g.nrows;
        if (DY2) d2[j] = fused_ld(bd2, g.r0 + (int64_t)(k0 + j) * g.rs, okj);
        if (k0 + j >= FUSED_NL) d1[j] = fused_ld(bd, g.r0 + (int64_t)(k0 + j) * g.rs, okj);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = k0 + j;
        u32x4 td;
        if (k < FUSED_NL) td = rows[k * FT + tid];
        else td = d1[j];
        float xv[8], dv[8], ov[8], gv[8];
        unpack8(xr[k], xv);
        unpack8(td, dv);
        if (DY2) {
          float e[8];
          unpack8(d2[j], e);
#pragma unroll
          for (int i = 0; i < 8; i++) dv[i] += e[i];
        }
        const unsigned m8 = mb[k >> 2] >> (8 * (k & 3));
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const float gg = ((m8 >> i) & 1u) ? dv[i] : 0.f;
          gv[i] = gg;
          ov[i] = fmaf(ka[i], gg, fmaf(kb[i], xv[i], kc[i]));
        }
        if (k < p.R && g.slot + k * g.rs < g.nrows) {
          fused_st(bdx, g.r0 + (int64_t)k * g.rs, pack8(ov));
          if (p.dres) fused_st(bdr, g.r0 + (int64_t)k * g.rs, pack8(gv));
        }
        __builtin_amdgcn_sched_barrier(0);  // one row's arithmetic at a time (the group's loads stay in flight)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int RMAX, bool DY2, int MASK>
__global__ __launch_bounds__(FT) void k_bn2d_fused_bwd(const FusedP p) {
  bn2d_fused_bwd_body<RMAX, DY2, MASK>(p, (int)blockIdx.x, (unsigned)(p.G0 + p.G1));
}
template <int RMAX, bool DY2, int MASK>
__global__ __launch_bounds__(FT) void k_bn2d_fused_bwd_pair(const FusedP2 pp) {
  const bool second = (int)blockIdx.x >= pp.Ga;
  bn2d_fused_bwd_body<RMAX, DY2, MASK>(second ? pp.b : pp.a, second ? (int)blockIdx.x - pp.Ga : (int)blockIdx.x, gridDim.x);
}

inline unsigned apply_blocks(int64_t N, int C) {
  int rs = T / (C / 8);
  return (unsigned)mm_cdiv(N, (int64_t)rs * APPLY_ROWS);
}

inline int stat_blocks(int64_t N, int C) {
  int rs = T / (C / 8);
  int64_t nb = mm_cdiv(N, (int64_t)rs * 16);
  if (nb < 1) nb = 1;
  if (nb > MAX_PART) nb = MAX_PART;
  return (int)nb;
}
}  // namespace

static const void* const k_fused_fns[] = {
    (const void*)k_bn2d_fused_fwd<8, false>,     (const void*)k_bn2d_fused_fwd<8, true>,      (const void*)k_bn2d_fused_fwd<20, false>,
    (const void*)k_bn2d_fused_fwd<20, true>,     (const void*)k_bn2d_fused_fwd<36, false>,    (const void*)k_bn2d_fused_fwd<36, true>,
    (const void*)k_bn2d_fused_bwd<8, false, 0>,  (const void*)k_bn2d_fused_bwd<8, false, 1>,  (const void*)k_bn2d_fused_bwd<8, false, 2>,
    (const void*)k_bn2d_fused_bwd<8, true, 0>,   (const void*)k_bn2d_fused_bwd<8, true, 1>,   (const void*)k_bn2d_fused_bwd<8, true, 2>,
    (const void*)k_bn2d_fused_bwd<20, false, 0>, (const void*)k_bn2d_fused_bwd<20, false, 1>, (const void*)k_bn2d_fused_bwd<20, false, 2>,
    (const void*)k_bn2d_fused_bwd<20, true, 0>,  (const void*)k_bn2d_fused_bwd<20, true, 1>,  (const void*)k_bn2d_fused_bwd<20, true, 2>,
    (const void*)k_bn2d_fused_bwd<36, false, 0>, (const void*)k_bn2d_fused_bwd<36, false, 1>, (const void*)k_bn2d_fused_bwd<36, false, 2>,
    (const void*)k_bn2d_fused_bwd<36, true, 0>,  (const void*)k_bn2d_fused_bwd<36, true, 1>,  (const void*)k_bn2d_fused_bwd<36, true, 2>};
constexpr int k_fused_nfns = (int)(sizeof(k_fused_fns) / sizeof(k_fused_fns[0]));
// the two-problem forms (maps of the encoders' layers 2-4: R <= 36 at half the CUs per problem)
static const void* const k_fused_pair_fns[] = {
    (const void*)k_bn2d_fused_fwd_pair<8, false>,     (const void*)k_bn2d_fused_fwd_pair<8, true>,      (const void*)k_bn2d_fused_fwd_pair<20, false>,
    (const void*)k_bn2d_fused_fwd_pair<20, true>,     (const void*)k_bn2d_fused_fwd_pair<36, false>,    (const void*)k_bn2d_fused_fwd_pair<36, true>,
    (const void*)k_bn2d_fused_bwd_pair<8, false, 0>,  (const void*)k_bn2d_fused_bwd_pair<8, false, 1>,  (const void*)k_bn2d_fused_bwd_pair<8, false, 2>,
    (const void*)k_bn2d_fused_bwd_pair<8, true, 0>,   (const void*)k_bn2d_fused_bwd_pair<8, true, 1>,   (const void*)k_bn2d_fused_bwd_pair<8, true, 2>,
    (const void*)k_bn2d_fused_bwd_pair<20, false, 0>, (const void*)k_bn2d_fused_bwd_pair<20, false, 1>, (const void*)k_bn2d_fused_bwd_pair<20, false, 2>,
    (const void*)k_bn2d_fused_bwd_pair<20, true, 0>,  (const void*)k_bn2d_fused_bwd_pair<20, true, 1>,  (const void*)k_bn2d_fused_bwd_pair<20, true, 2>,
    (const void*)k_bn2d_fused_bwd_pair<36, false, 0>, (const void*)k_bn2d_fused_bwd_pair<36, false, 1>, (const void*)k_bn2d_fused_bwd_pair<36, false, 2>,
    (const void*)k_bn2d_fused_bwd_pair<36, true, 0>,  (const void*)k_bn2d_fused_bwd_pair<36, true, 1>,  (const void*)k_bn2d_fused_bwd_pair<36, true, 2>};
constexpr int k_fused_pair_nfns = (int)(sizeof(k_fused_pair_fns) / sizeof(k_fused_pair_fns[0]));

extern "C" {

// Which path runs is the HANDLE's choice (mm_set_option(h, MM_OPT_BN2D_FUSED, mask): bit 0 = mm_bn2d_fwd_train, bit 1 =
// mm_bn2d_bwd single-launch; 0 = always the reduce / finalize / apply kernels) - no process-wide switch.
#ifdef MM_ACT_FP16
constexpr int BN2D_UNIT = 2;
constexpr int BN2D_PAIR_UNIT = 4;
#else
constexpr int BN2D_UNIT = 1;
constexpr int BN2D_PAIR_UNIT = 3;
#endif

size_t MM_SYM(mm_bn2d_ws_bytes)(int C) { return mm_align((size_t)MAX_PART * 2 * C * sizeof(double)) + mm_align(4 * C * sizeof(float)) + 256; }

// block counts of the two statistics groups (rows [0,Ns) and [Ns,N)); Ns == N or Ns == 0: a single group
static void split_blocks(int64_t N, int64_t& Ns, int C, bool stats, int& b0, int& b1) {
  if (Ns <= 0 || Ns >= N) Ns = N;
  if (stats) {
    b0 = stat_blocks(Ns, C);
    b1 = Ns < N ? stat_blocks(N - Ns, C) : 0;
    if (b1 > 0) {  // both groups share the MAX_PART partial slots
      if (b0 > MAX_PART / 2) b0 = MAX_PART / 2;
      if (b1 > MAX_PART / 2) b1 = MAX_PART / 2;
    }
  } else {
    b0 = (int)apply_blocks(Ns, C);
    b1 = Ns < N ? (int)apply_blocks(N - Ns, C) : 0;
  }
}

// 1 when mm_bn2d_fwd_train / mm_bn2d_bwd (backward != 0) would take the single-launch kernels for N rows of C channels under the
// handle's present options (shape rule only: pitches and the 2 GiB buffer limit are checked at the call).  A producer uses it to
// decide whether to file statistics for mm_bn2d_fwd_train_pre (maps too large for one launch) or to leave the map to the
// single-launch kernel, which reads it once anyway.
int MM_SYM(mm_bn2d_single_launch)(void* h, int64_t N, int64_t Ns, int C, int backward) {
  MMHandle* H = (MMHandle*)h;
  if (!H || H->magic != MM_HANDLE_MAGIC) return 0;
  if (!H->fused_ok || !(H->opt[MM_OPT_BN2D_FUSED] & (backward ? 2 : 1)) || N <= 0 || C % 8 != 0 || C > FT || C < 8) return 0;
  if (Ns <= 0 || Ns >= N) Ns = N;
  int g0, g1;
  return fused_shape(H->cus, N, Ns, C, 8, &g0, &g1) <= 36 ? 1 : 0;
}

// y = act(BN_train(x) + res); x,res,y NHWC bf16 [N rows, C]; momentum = torch's (0.1).
// Ns: rows [0,Ns) and [Ns,N) are normalised with their OWN batch statistics (the source and target halves of a jointly
// batched step, train.py:186-292 calls the net once per domain); Ns = N (or 0) is the ordinary single-batch case.
// save_mean / save_invstd: fp32 [G][C], G = 2 when split.
int MM_SYM(mm_bn2d_fwd_train)(void* h, const void* x, int ld_x, const void* res, int ld_r, int64_t N, int64_t Ns, int C, const float* weight,
                      const float* bias, float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps,
                      float momentum, int relu, void* y, int ld_y, float* save_mean, float* save_invstd, void* ws, size_t ws_bytes,
                      hipStream_t s) {
  MM_CHECK_HANDLE(h);
  MM_CHECK_ARG(C % 8 == 0 && C / 8 <= T && ld_x % 8 == 0 && ld_y % 8 == 0, "bn2d: C must be a multiple of 8, <= 2048");
  if (ws_bytes < (size_t)MAX_PART * 2 * C * sizeof(double)) {
    mm_set_error("bn2d: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  double* partial = (double*)ws;
  int nb0, nb1, ab0, ab1;
  if (Ns <= 0 || Ns >= N) Ns = N;
  FusedPlan pl;
  int rc = fused_plan(H, MM_OPT_BN2D_FUSED, BN2D_UNIT, N, Ns, C, 8, 36, false, k_fused_fns, k_fused_nfns, s, &pl);
  if (rc) return rc;
  const int64_t ldmax_f = std::max(std::max(ld_x, ld_y), res ? ld_r : 0);
  if (pl.ok && (!res || ld_r % 8 == 0) && N * ldmax_f * 2 < (1ll << 31)) {  // 32-bit buffer offsets
    FusedP p = {};
    p.x = (const u16*)x, p.res = (const u16*)res, p.y = (u16*)y;
    p.ld_x = ld_x, p.ld_r = ld_r, p.ld_y = ld_y;
    p.N = N, p.Ns = Ns, p.C = C, p.relu = relu, p.G0 = pl.G0, p.G1 = pl.G1, p.R = pl.R;
    p.weight = weight, p.bias = bias, p.running_mean = running_mean, p.running_var = running_var, p.nbt = num_batches_tracked;
    p.eps = eps, p.momentum = momentum, p.save_mean = save_mean, p.save_invstd = save_invstd;
    p.partial = partial, p.sync = pl.sync, p.fault = pl.fault;
    const dim3 grid(pl.G0 + pl.G1), blk(FT);
#define MM_FWD(RM)                                                                                  \
  do {                                                                                              \
    if (res) hipLaunchKernelGGL((k_bn2d_fused_fwd<RM, true>), grid, blk, FUSED_LDS, s, p);          \
    else hipLaunchKernelGGL((k_bn2d_fused_fwd<RM, false>), grid, blk, FUSED_LDS, s, p);             \
  } while (0)
    if (pl.R <= 8) MM_FWD(8);
    else if (pl.R <= 20) MM_FWD(20);
    else MM_FWD(36);
#undef MM_FWD
    MM_LAUNCH_CHECK();
    return MM_OK;
  }
  split_blocks(N, Ns, C, true, nb0, nb1);
  hipLaunchKernelGGL(k_bn2d_reduce<0>, dim3(nb0 + nb1), dim3(T), 0, s, (const u16*)x, ld_x, nullptr, 0, nullptr, 0, 0, N, C, nullptr,
                     nullptr, partial, Ns, nb0);
  hipLaunchKernelGGL(k_bn2d_finalize_fwd, dim3(C), dim3(256), 0, s, partial, nb0, nb1, Ns, N, C, eps, momentum, running_mean, running_var,
                     save_mean, save_invstd, num_batches_tracked);
  if (N > 0) {
    split_blocks(N, Ns, C, false, ab0, ab1);
    hipLaunchKernelGGL(k_bn2d_apply, dim3(ab0 + ab1), dim3(T), 0, s, (const u16*)x, ld_x, (const u16*)res, ld_r, N, C, save_mean,
                       save_invstd, 0, eps, weight, bias, relu, (u16*)y, ld_y, Ns, ab0);
  }
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// mm_bn2d_fwd_train with the batch statistics taken from the slab that the producing convolution filled in its epilogue
// (mm_conv2d_3x3s1 / mm_conv2d_gemm ``stats``; slab_rows = mm_conv2d_*_stat_rows): slab reduce + finalize (both small) + ONE
// streaming apply pass - no statistics pass over the map, no grid barrier (so no handle and no residency rule: safe beside any
// other stream).  ws as for mm_bn2d_fwd_train.
int MM_SYM(mm_bn2d_fwd_train_pre)(const void* x, int ld_x, const void* res, int ld_r, int64_t N, int64_t Ns, int C, const float* weight,
                          const float* bias, float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps,
                          float momentum, int relu, void* y, int ld_y, float* save_mean, float* save_invstd, const float* slab,
                          int64_t slab_rows, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(C % 8 == 0 && C / 8 <= T && ld_x % 8 == 0 && ld_y % 8 == 0, "bn2d: C must be a multiple of 8, <= 2048");
  MM_CHECK_ARG(slab != nullptr && slab_rows > 0 && slab_rows % 2 == 0 && ((uintptr_t)slab % 16) == 0, "bn2d_fwd_train_pre: no statistics slab");
  MM_CHECK_ARG(ws_bytes >= MM_SYM(mm_bn2d_ws_bytes)(C), "bn2d: workspace too small");
  if (Ns <= 0 || Ns >= N) Ns = N;
  const int64_t nsub = slab_rows / 2;
  int nb = (int)(nsub / 64 < 1 ? 1 : nsub / 64);  // >= 64 sub-blocks (4096 pixels) per block
  if (nb > 256) nb = 256;
  if (nb > MAX_PART / 2) nb = MAX_PART / 2;
  const int nb0 = nb, nb1 = Ns < N ? nb : 0;
  double* partial = (double*)ws;
  hipLaunchKernelGGL(k_bn2d_slab_reduce, dim3(nb0 + nb1), dim3(256), 0, s, slab, nsub, nb0, nb1, C, partial);
  hipLaunchKernelGGL(k_bn2d_finalize_fwd, dim3(C), dim3(256), 0, s, partial, nb0, nb1, Ns, N, C, eps, momentum, running_mean, running_var,
                     save_mean, save_invstd, num_batches_tracked);
  if (N > 0) {
    int ab0, ab1;
    split_blocks(N, Ns, C, false, ab0, ab1);
    hipLaunchKernelGGL(k_bn2d_apply, dim3(ab0 + ab1), dim3(T), 0, s, (const u16*)x, ld_x, (const u16*)res, ld_r, N, C, save_mean,
                       save_invstd, 0, eps, weight, bias, relu, (u16*)y, ld_y, Ns, ab0);
  }
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// mm_bn2d_fwd_train_pre (+ReLU, no residual) on a [B][H][W][C] map, H and W even, FUSED with MaxPool2d(3, 2, 1): y (pitch ld_y) receives
// the normalised map, ypool [B][H/2][W/2][C] its pooled version and idx the winning taps (mm_maxpool3x3s2_fwd's outputs), in ONE pass
// over x after the slab reduce + finalize launches (the stems: 2d_net/backbones.py:43-47).  Bs: images [0, Bs) are statistics group 0.
int MM_SYM(mm_bn2d_fwd_train_pre_pool)(const void* x, int ld_x, int B, int H, int W, int Bs, int C, const float* weight, const float* bias,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps, float momentum, void* y,
                               int ld_y, void* ypool, void* idx, float* save_mean, float* save_invstd, const float* slab, int64_t slab_rows,
                               void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(C % 8 == 0 && C / 8 <= T && ld_x % 8 == 0 && ld_y % 8 == 0, "bn2d_pool: C must be a multiple of 8, <= 2048");
  MM_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "bn2d_pool: H and W must be even");
  MM_CHECK_ARG(slab != nullptr && slab_rows > 0 && slab_rows % 2 == 0 && ((uintptr_t)slab % 16) == 0, "bn2d_pool: no statistics slab");
  MM_CHECK_ARG(ws_bytes >= MM_SYM(mm_bn2d_ws_bytes)(C), "bn2d: workspace too small");
  const int64_t N = (int64_t)B * H * W;
  MM_CHECK_ARG(N * (C / 8) / 4 < (1ll << 32) && N < (1ll << 31), "bn2d_pool: map too large for 32-bit indices");
  int64_t Ns = (Bs <= 0 || Bs >= B) ? N : (int64_t)Bs * H * W;
  const int64_t nsub = slab_rows / 2;
  int nb = (int)(nsub / 64 < 1 ? 1 : nsub / 64);
  if (nb > 256) nb = 256;
  if (nb > MAX_PART / 2) nb = MAX_PART / 2;
  const int nb0 = nb, nb1 = Ns < N ? nb : 0;
  double* partial = (double*)ws;
  hipLaunchKernelGGL(k_bn2d_slab_reduce, dim3(nb0 + nb1), dim3(256), 0, s, slab, nsub, nb0, nb1, C, partial);
  hipLaunchKernelGGL(k_bn2d_finalize_fwd, dim3(C), dim3(256), 0, s, partial, nb0, nb1, Ns, N, C, eps, momentum, running_mean, running_var,
                     save_mean, save_invstd, num_batches_tracked);
  const int Ho = H / 2, Wo = W / 2;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(k_bn2d_apply_pool, dim3((unsigned)mm_cdiv(total, T)), dim3(T), 0, s, (const u16*)x, ld_x, B, H, W, C, Ns < N ? Bs : B,
                     save_mean, save_invstd, weight, bias, (u16*)y, ld_y, (u16*)ypool, (unsigned char*)idx, Ho, Wo);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// Its backward: mm_bn2d_bwd (ReLU mask recomputed from x, no residual output) whose incoming gradient is the pooled map's gradient
// dyp [B][H/2][W/2][C] (pitch ld_dyp) + the winning taps idx (gathered per full-resolution pixel inside the reduce and apply passes:
// PoolSrc) + optionally a second full-resolution contribution dy2 (the decoder's concat slice).  Three launches, no grid barrier.
int MM_SYM(mm_bn2d_bwd_pool)(const void* x, int ld_x, const void* dyp, int ld_dyp, const void* idx, int B, int H, int W, int Bs, const void* dy2,
                     int ld_dy2, int C, const float* weight, const float* bias, const float* save_mean, const float* save_invstd, void* dx,
                     int ld_dx, float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(C % 8 == 0 && C / 8 <= T && ld_x % 8 == 0 && ld_dyp % 8 == 0 && ld_dx % 8 == 0 && (!dy2 || ld_dy2 % 8 == 0), "bn2d_bwd_pool: bad pitch");
  MM_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && x && dyp && idx && dx, "bn2d_bwd_pool: bad arguments");
  const int64_t N = (int64_t)B * H * W;
  MM_CHECK_ARG(N < (1ll << 31), "bn2d_bwd_pool: map too large for 32-bit indices");
  size_t need = mm_align((size_t)MAX_PART * 2 * C * sizeof(double));
  if (ws_bytes < need + 4 * C * sizeof(float)) {
    mm_set_error("bn2d_bwd: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  double* partial = (double*)ws;
  float* sums = (float*)((char*)ws + need);
  int64_t Ns = (Bs <= 0 || Bs >= B) ? N : (int64_t)Bs * H * W;
  PoolSrc ps{(const u16*)dyp, (const unsigned char*)idx, ld_dyp, H, W, H / 2, W / 2, C};
  int nb0, nb1, ab0, ab1;
  split_blocks(N, Ns, C, true, nb0, nb1);
  hipLaunchKernelGGL((k_bn2d_reduce<1, true>), dim3(nb0 + nb1), dim3(T), 0, s, (const u16*)x, ld_x, (const u16*)nullptr, 0, (const u16*)nullptr, 0, 1, N, C,
                     save_mean, save_invstd, partial, Ns, nb0, weight, bias, (const u16*)dy2, ld_dy2, ps);
  hipLaunchKernelGGL(k_bn2d_finalize_bwd, dim3(C), dim3(256), 0, s, partial, nb0, nb1, C, sums, dweight, dbias, accumulate);
  split_blocks(N, Ns, C, false, ab0, ab1);
  hipLaunchKernelGGL(k_bn2d_bwd_apply, dim3(ab0 + ab1), dim3(T), 0, s, (const u16*)x, ld_x, (const u16*)nullptr, 0, (const u16*)nullptr, 0, 1, N, C,
                     save_mean, save_invstd, weight, sums, (u16*)dx, ld_dx, (u16*)nullptr, 0, Ns, ab0, bias, (const u16*)dy2, ld_dy2, ps);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// Two mm_bn2d_fwd_train problems of one shape (N, Ns, C, eps, momentum, relu shared; everything else per problem) - the same layer of
// the two encoders, EXP/2d_net/model.py:43-46 - in ONE single-launch kernel where the maps allow it (half the CUs per problem, one
// pair of grid barriers and one statistics exchange for both): the small maps of layers 2-4 are bound by exactly those fixed costs.
// Otherwise the two problems run one after the other as mm_bn2d_fwd_train would run them.  ws as for mm_bn2d_fwd_train.
typedef struct {
  const void* x; int ld_x; const void* res; int ld_r; const float* weight; const float* bias; float* running_mean; float* running_var;
  int64_t* num_batches_tracked; void* y; int ld_y; float* save_mean; float* save_invstd;
} MMBn2dFwdArgs;

int MM_SYM(mm_bn2d_fwd_train_pair)(void* h, const MMBn2dFwdArgs* a, const MMBn2dFwdArgs* b, int64_t N, int64_t Ns, int C, float eps, float momentum,
                           int relu, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_HANDLE(h);
  MM_CHECK_ARG(a && b && C % 8 == 0 && C / 8 <= T, "bn2d_fwd_train_pair: bad arguments");
  if (ws_bytes < MM_SYM(mm_bn2d_ws_bytes)(C)) {
    mm_set_error("bn2d: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  if (Ns <= 0 || Ns >= N) Ns = N;
  const MMBn2dFwdArgs* q[2] = {a, b};
  bool same = (a->res == nullptr) == (b->res == nullptr);
  int64_t ldmax = 0;
  for (int i = 0; i < 2; i++) {
    same = same && q[i]->ld_x % 8 == 0 && q[i]->ld_y % 8 == 0 && (!q[i]->res || q[i]->ld_r % 8 == 0);
    ldmax = std::max(ldmax, (int64_t)std::max(std::max(q[i]->ld_x, q[i]->ld_y), q[i]->res ? q[i]->ld_r : 0));
  }
  FusedPlan pl;
  pl.ok = false;
  if (same && N * ldmax * 2 < (1ll << 31)) {
    int rc = fused_plan(H, MM_OPT_BN2D_FUSED, BN2D_PAIR_UNIT, N, Ns, C, 8, 36, false, k_fused_pair_fns, k_fused_pair_nfns, s, &pl, 2);
    if (rc) return rc;
  }
  if (!pl.ok) {
    for (int i = 0; i < 2; i++) {
      int rc = MM_SYM(mm_bn2d_fwd_train)(h, q[i]->x, q[i]->ld_x, q[i]->res, q[i]->ld_r, N, Ns, C, q[i]->weight, q[i]->bias, q[i]->running_mean,
                                 q[i]->running_var, q[i]->num_batches_tracked, eps, momentum, relu, q[i]->y, q[i]->ld_y, q[i]->save_mean,
                                 q[i]->save_invstd, ws, ws_bytes, s);
      if (rc) return rc;
    }
    return MM_OK;
  }
  FusedP2 pp = {};
  const int Gp = pl.G0 + pl.G1;
  FusedP* fp[2] = {&pp.a, &pp.b};
  for (int i = 0; i < 2; i++) {
    FusedP& p = *fp[i];
    p.x = (const u16*)q[i]->x, p.res = (const u16*)q[i]->res, p.y = (u16*)q[i]->y;
    p.ld_x = q[i]->ld_x, p.ld_r = q[i]->ld_r, p.ld_y = q[i]->ld_y;
    p.N = N, p.Ns = Ns, p.C = C, p.relu = relu, p.G0 = pl.G0, p.G1 = pl.G1, p.R = pl.R;
    p.weight = q[i]->weight, p.bias = q[i]->bias, p.running_mean = q[i]->running_mean, p.running_var = q[i]->running_var;
    p.nbt = q[i]->num_batches_tracked;
    p.eps = eps, p.momentum = momentum, p.save_mean = q[i]->save_mean, p.save_invstd = q[i]->save_invstd;
    p.partial = (double*)ws + (size_t)i * Gp * 2 * C, p.sync = pl.sync, p.fault = pl.fault;
  }
  pp.Ga = Gp;
  const dim3 grid(2 * Gp), blk(FT);
  const bool res = a->res != nullptr;
#define MM_FWD2(RM)                                                                                    \
  do {                                                                                                 \
    if (res) hipLaunchKernelGGL((k_bn2d_fused_fwd_pair<RM, true>), grid, blk, FUSED_LDS, s, pp);       \
    else hipLaunchKernelGGL((k_bn2d_fused_fwd_pair<RM, false>), grid, blk, FUSED_LDS, s, pp);          \
  } while (0)
  if (pl.R <= 8) MM_FWD2(8);
  else if (pl.R <= 20) MM_FWD2(20);
  else MM_FWD2(36);
#undef MM_FWD2
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int MM_SYM(mm_bn2d_fwd_eval)(const void* x, int ld_x, const void* res, int ld_r, int64_t N, int C, const float* weight, const float* bias,
                     const float* running_mean, const float* running_var, float eps, int relu, void* y, int ld_y, hipStream_t s) {
  MM_CHECK_ARG(C % 8 == 0, "bn2d: C must be a multiple of 8");
  if (N == 0) return MM_OK;
  const int ab = (int)apply_blocks(N, C);
  hipLaunchKernelGGL(k_bn2d_apply, dim3(ab), dim3(T), 0, s, (const u16*)x, ld_x, (const u16*)res, ld_r, N, C, running_mean, running_var,
                     1, eps, weight, bias, relu, (u16*)y, ld_y, N, ab);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// dx (and dres = relu-masked dy when dres != NULL), dweight, dbias; Ns and the [G][C] statistics as in mm_bn2d_fwd_train.
// dy2 != NULL: the incoming gradient is dy + dy2 (the map had two consumers; summed here in fp32 instead of by an add kernel).
// yout == NULL with relu != 0 (only valid when the forward had no residual input): the ReLU mask is recomputed from x,
// weight, bias and the saved statistics instead of reading the output map.
int MM_SYM(mm_bn2d_bwd)(void* h, const void* x, int ld_x, const void* dy, int ld_dy, const void* dy2, int ld_dy2, const void* yout, int ld_y, int relu,
                int64_t N, int64_t Ns, int C,
                const float* weight, const float* bias, const float* save_mean, const float* save_invstd, void* dx, int ld_dx, void* dres, int ld_dr,
                float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_HANDLE(h);
  MM_CHECK_ARG(C % 8 == 0 && C / 8 <= T, "bn2d: C must be a multiple of 8, <= 2048");
  size_t need = mm_align((size_t)MAX_PART * 2 * C * sizeof(double));
  if (ws_bytes < need + 4 * C * sizeof(float)) {
    mm_set_error("bn2d_bwd: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  double* partial = (double*)ws;
  float* sums = (float*)((char*)ws + need);
  int nb0, nb1, ab0, ab1;
  if (Ns <= 0 || Ns >= N) Ns = N;
  FusedPlan pl;
  int rc = fused_plan(H, MM_OPT_BN2D_FUSED, BN2D_UNIT, N, Ns, C, 8, 36, true, k_fused_fns, k_fused_nfns, s, &pl);
  if (rc) return rc;
  const int64_t ldmax_b = std::max(std::max(std::max(ld_x, ld_dy), std::max(ld_dx, dy2 ? ld_dy2 : 0)), std::max(yout ? ld_y : 0, dres ? ld_dr : 0));
  if (pl.ok && ld_x % 8 == 0 && ld_dy % 8 == 0 && ld_dx % 8 == 0 && (!dy2 || ld_dy2 % 8 == 0) && (!yout || ld_y % 8 == 0) &&
      (!dres || ld_dr % 8 == 0) && N * ldmax_b * 2 < (1ll << 31)) {  // 32-bit buffer offsets
    FusedP p = {};
    p.x = (const u16*)x, p.dy = (const u16*)dy, p.dy2 = (const u16*)dy2, p.yout = (const u16*)yout, p.dx = (u16*)dx, p.dres = (u16*)dres;
    p.ld_x = ld_x, p.ld_dy = ld_dy, p.ld_dy2 = ld_dy2, p.ld_y = ld_y, p.ld_dx = ld_dx, p.ld_dr = ld_dr;
    p.N = N, p.Ns = Ns, p.C = C, p.relu = relu, p.G0 = pl.G0, p.G1 = pl.G1, p.R = pl.R;
    p.weight = weight, p.bias = bias, p.save_mean = (float*)save_mean, p.save_invstd = (float*)save_invstd;
    p.sums = sums, p.dweight = dweight, p.dbias = dbias, p.accumulate = accumulate;
    p.partial = partial, p.sync = pl.sync, p.fault = pl.fault;
    const dim3 grid(pl.G0 + pl.G1), blk(FT);
    const int mask = !relu ? 0 : (yout ? 2 : 1);
#define MM_BWD(RM, D2)                                                                                        \
  do {                                                                                                        \
    if (mask == 0) hipLaunchKernelGGL((k_bn2d_fused_bwd<RM, D2, 0>), grid, blk, FUSED_LDS, s, p);             \
    else if (mask == 1) hipLaunchKernelGGL((k_bn2d_fused_bwd<RM, D2, 1>), grid, blk, FUSED_LDS, s, p);        \
    else hipLaunchKernelGGL((k_bn2d_fused_bwd<RM, D2, 2>), grid, blk, FUSED_LDS, s, p);                       \
  } while (0)
    if (pl.R <= 8) {
      if (dy2) MM_BWD(8, true);
      else MM_BWD(8, false);
    } else if (pl.R <= 20) {
      if (dy2) MM_BWD(20, true);
      else MM_BWD(20, false);
    } else {
      if (dy2) MM_BWD(36, true);
      else MM_BWD(36, false);
    }
#undef MM_BWD
    MM_LAUNCH_CHECK();
    return MM_OK;
  }
  split_blocks(N, Ns, C, true, nb0, nb1);
#define MM_RED1(H2, YO)                                                                                                                \
  hipLaunchKernelGGL((k_bn2d_reduce<1, false, H2, YO>), dim3(nb0 + nb1), dim3(T), 0, s, (const u16*)x, ld_x, (const u16*)dy, ld_dy,    \
                     (const u16*)yout, ld_y, relu, N, C, save_mean, save_invstd, partial, Ns, nb0, weight, bias, (const u16*)dy2, ld_dy2, \
                     PoolSrc{})
  {
    const bool yo = relu && yout != nullptr;
    if (dy2) {
      if (yo) MM_RED1(true, true);
      else MM_RED1(true, false);
    } else {
      if (yo) MM_RED1(false, true);
      else MM_RED1(false, false);
    }
  }
#undef MM_RED1
  hipLaunchKernelGGL(k_bn2d_finalize_bwd, dim3(C), dim3(256), 0, s, partial, nb0, nb1, C, sums, dweight, dbias, accumulate);
  if (N > 0) {
    split_blocks(N, Ns, C, false, ab0, ab1);
    hipLaunchKernelGGL(k_bn2d_bwd_apply, dim3(ab0 + ab1), dim3(T), 0, s, (const u16*)x, ld_x, (const u16*)dy, ld_dy, (const u16*)yout,
                       ld_y, relu, N, C, save_mean, save_invstd, weight, sums, (u16*)dx, ld_dx, (u16*)dres, ld_dr, Ns, ab0, bias, (const u16*)dy2,
                       ld_dy2);
  }
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// Two mm_bn2d_bwd problems of one shape in one single-launch kernel (see mm_bn2d_fwd_train_pair); relu, N, Ns, C, accumulate shared.
typedef struct {
  const void* x; int ld_x; const void* dy; int ld_dy; const void* dy2; int ld_dy2; const void* yout; int ld_y;
  const float* weight; const float* bias; const float* save_mean; const float* save_invstd; void* dx; int ld_dx; void* dres; int ld_dr;
  float* dweight; float* dbias;
} MMBn2dBwdArgs;

int MM_SYM(mm_bn2d_bwd_pair)(void* h, const MMBn2dBwdArgs* a, const MMBn2dBwdArgs* b, int relu, int64_t N, int64_t Ns, int C, int accumulate, void* ws,
                     size_t ws_bytes, hipStream_t s) {
  MM_CHECK_HANDLE(h);
  MM_CHECK_ARG(a && b && C % 8 == 0 && C / 8 <= T, "bn2d_bwd_pair: bad arguments");
  if (ws_bytes < MM_SYM(mm_bn2d_ws_bytes)(C)) {
    mm_set_error("bn2d_bwd: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  if (Ns <= 0 || Ns >= N) Ns = N;
  const MMBn2dBwdArgs* q[2] = {a, b};
  bool same = (a->dy2 == nullptr) == (b->dy2 == nullptr) && (a->yout == nullptr) == (b->yout == nullptr);
  int64_t ldmax = 0;
  for (int i = 0; i < 2; i++) {
    const MMBn2dBwdArgs& r = *q[i];
    same = same && r.ld_x % 8 == 0 && r.ld_dy % 8 == 0 && r.ld_dx % 8 == 0 && (!r.dy2 || r.ld_dy2 % 8 == 0) && (!r.yout || r.ld_y % 8 == 0) &&
           (!r.dres || r.ld_dr % 8 == 0);
    ldmax = std::max(ldmax, (int64_t)std::max(std::max(std::max(r.ld_x, r.ld_dy), std::max(r.ld_dx, r.dy2 ? r.ld_dy2 : 0)),
                                             std::max(r.yout ? r.ld_y : 0, r.dres ? r.ld_dr : 0)));
  }
  FusedPlan pl;
  pl.ok = false;
  if (same && N * ldmax * 2 < (1ll << 31)) {
    int rc = fused_plan(H, MM_OPT_BN2D_FUSED, BN2D_PAIR_UNIT, N, Ns, C, 8, 36, true, k_fused_pair_fns, k_fused_pair_nfns, s, &pl, 2);
    if (rc) return rc;
  }
  if (!pl.ok) {
    for (int i = 0; i < 2; i++) {
      const MMBn2dBwdArgs& r = *q[i];
      int rc = MM_SYM(mm_bn2d_bwd)(h, r.x, r.ld_x, r.dy, r.ld_dy, r.dy2, r.ld_dy2, r.yout, r.ld_y, relu, N, Ns, C, r.weight, r.bias, r.save_mean,
                           r.save_invstd, r.dx, r.ld_dx, r.dres, r.ld_dr, r.dweight, r.dbias, accumulate, ws, ws_bytes, s);
      if (rc) return rc;
    }
    return MM_OK;
  }
  FusedP2 pp = {};
  const int Gp = pl.G0 + pl.G1;
  FusedP* fp[2] = {&pp.a, &pp.b};
  for (int i = 0; i < 2; i++) {
    FusedP& p = *fp[i];
    const MMBn2dBwdArgs& r = *q[i];
    p.x = (const u16*)r.x, p.dy = (const u16*)r.dy, p.dy2 = (const u16*)r.dy2, p.yout = (const u16*)r.yout, p.dx = (u16*)r.dx, p.dres = (u16*)r.dres;
    p.ld_x = r.ld_x, p.ld_dy = r.ld_dy, p.ld_dy2 = r.ld_dy2, p.ld_y = r.ld_y, p.ld_dx = r.ld_dx, p.ld_dr = r.ld_dr;
    p.N = N, p.Ns = Ns, p.C = C, p.relu = relu, p.G0 = pl.G0, p.G1 = pl.G1, p.R = pl.R;
    p.weight = r.weight, p.bias = r.bias, p.save_mean = (float*)r.save_mean, p.save_invstd = (float*)r.save_invstd;
    p.dweight = r.dweight, p.dbias = r.dbias, p.accumulate = accumulate;
    p.partial = (double*)ws + (size_t)i * Gp * 2 * C;
    p.sums = (float*)((double*)ws + (size_t)2 * Gp * 2 * C) + (size_t)i * 4 * C;  // behind both problems' partial rows (MAX_PART of them fit)
    p.sync = pl.sync, p.fault = pl.fault;
  }
  pp.Ga = Gp;
  const dim3 grid(2 * Gp), blk(FT);
  const int mask = !relu ? 0 : (a->yout ? 2 : 1);
  const bool d2 = a->dy2 != nullptr;
#define MM_BWD2(RM, D2)                                                                                          \
  do {                                                                                                           \
    if (mask == 0) hipLaunchKernelGGL((k_bn2d_fused_bwd_pair<RM, D2, 0>), grid, blk, FUSED_LDS, s, pp);          \
    else if (mask == 1) hipLaunchKernelGGL((k_bn2d_fused_bwd_pair<RM, D2, 1>), grid, blk, FUSED_LDS, s, pp);     \
    else hipLaunchKernelGGL((k_bn2d_fused_bwd_pair<RM, D2, 2>), grid, blk, FUSED_LDS, s, pp);                    \
  } while (0)
  if (pl.R <= 8) {
    if (d2) MM_BWD2(8, true);
    else MM_BWD2(8, false);
  } else if (pl.R <= 20) {
    if (d2) MM_BWD2(20, true);
    else MM_BWD2(20, false);
  } else {
    if (d2) MM_BWD2(36, true);
    else MM_BWD2(36, false);
  }
#undef MM_BWD2
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// out[c] (+)= sum over the N rows of x[:, c]  (conv bias gradient: torch's dy.sum((0, 2, 3)))
int MM_H(mm_colsum)(const void* x, int ld_x, int64_t N, int C, float* out, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(C % 8 == 0 && C / 8 <= T && ld_x % 8 == 0, "colsum: C must be a multiple of 8, <= 2048");
  if (ws_bytes < (size_t)MAX_PART * 2 * C * sizeof(double)) {
    mm_set_error("colsum: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  int nb = stat_blocks(N, C);
  hipLaunchKernelGGL(k_bn2d_reduce<2>, dim3(nb), dim3(T), 0, s, (const u16*)x, ld_x, nullptr, 0, nullptr, 0, 0, N, C, nullptr, nullptr,
                     (double*)ws, N, nb);
  hipLaunchKernelGGL(k_colsum_finalize, dim3(C), dim3(64), 0, s, (const double*)ws, nb, C, out, accumulate);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
