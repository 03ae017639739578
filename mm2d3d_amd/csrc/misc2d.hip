// Memory-bound glue of the 2D branch on NHWC bf16:
//   channel concat / split          torch.cat([depth, up, rgb], 1) in the decoder (2d_net/model.py:107,112,117,122)
//   max-pool 3x3 s2 p1 (+ backward)  backbones.py:53
//   segmentation heads               AvgPool2d(5,1,2) -> Conv2d 1x1 64->C, twice (model.py:59-60,129-130; aux :158,163-164)
// The two heads share their input, and a 1x1 convolution commutes with the (linear) box filter, so both heads are
// computed as ONE 64 -> 2C projection per pixel followed by a 5x5 box filter on 2C channels (10x less filter traffic);
// the bias is added after the filter, which is exact for count_include_pad=True zero padding.
#include "common.h"

#include "h16.h"  // bf16 (default) or IEEE fp16 (-DMM_ACT_FP16) storage: this file is built once for each

namespace {
constexpr int T = 256;
constexpr int MAXJ = 32;


// dst[r][0..C) = src[r][0..C), 16-B vectors (C multiple of 8 elements of 2 bytes)
__global__ __launch_bounds__(T) void k_copy_rows(const u16* __restrict__ src, int64_t ld_s, u16* __restrict__ dst, int64_t ld_d,
                                                  int64_t N, int C8) {
  const unsigned gid = blockIdx.x * (unsigned)T + threadIdx.x;  // 32-bit index arithmetic (host: N * C8 < 2^32)
  const unsigned r = gid / (unsigned)C8;
  const int c = (int)(gid - r * (unsigned)C8);
  if ((int64_t)r >= N) return;
  *(uint4*)(dst + (int64_t)r * ld_d + c * 8) = *(const uint4*)(src + (int64_t)r * ld_s + c * 8);
}

// Channel concat / split of up to 4 NHWC bf16 maps in ONE launch: dst[r] = [src0[r] | src1[r] | ...] (SPLIT: the reverse).
// A thread owns one 16-B chunk column of the wide row and walks ROWS rows, so the wide rows are written (read) as whole
// contiguous lines and there is no per-element division.
struct CatP {
  const u16* src[4];  // SPLIT: destinations (cast away const at the store)
  int c8[4];          // chunks (8 channels) per part
  int n;
  u16* wide;
  int ct8;            // chunks per wide row
  int64_t N;
};
template <bool SPLIT>
__global__ __launch_bounds__(T) void k_concat(CatP p) {
  constexpr int ROWS = 8;
  const int rows_per_pass = T / p.ct8;  // host guarantees ct8 <= T
  const int rl = threadIdx.x / p.ct8, c = threadIdx.x - rl * p.ct8;
  if (rl >= rows_per_pass) return;
  int part = 0, cc = c;
  while (part < p.n - 1 && cc >= p.c8[part]) cc -= p.c8[part], part++;
  u16* narrow = (u16*)p.src[part];
  const int ldn = p.c8[part] * 8;
  int64_t r = (int64_t)blockIdx.x * rows_per_pass * ROWS + rl;
#pragma unroll
  for (int k = 0; k < ROWS; k++, r += rows_per_pass) {
    if (r >= p.N) break;
    uint4* w = (uint4*)(p.wide + (r * p.ct8 + c) * 8);
    uint4* q = (uint4*)(narrow + r * ldn + cc * 8);
    if (SPLIT) *q = *w;
    else *w = *q;
  }
}

// ---- max-pool 3x3 stride 2 pad 1, NHWC bf16; idx = winning tap (kh*3+kw), first maximum in scan order (torch).
// thread = 8 consecutive channels of one pixel (16-B loads/stores, 8-B index vectors)
__global__ __launch_bounds__(T) void k_maxpool_fwd(const u16* __restrict__ x, int ldx, int B, int H, int W, int C, u16* __restrict__ y,
                                                    unsigned char* __restrict__ idx, int Ho, int Wo) {
  const int C8 = C >> 3;
  // 32-bit index arithmetic (host: total < 2^32): three 64-bit div/mod pairs were ~300 instructions per thread, several times
  // the thread's memory work
  const unsigned gid = blockIdx.x * (unsigned)T + threadIdx.x;
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  if ((int64_t)gid >= total) return;
  const unsigned pixu = gid / (unsigned)C8;
  const int c8 = (int)(gid - pixu * (unsigned)C8);
  const unsigned tu = pixu / (unsigned)Wo;
  const int ox = (int)(pixu - tu * (unsigned)Wo);
  const int b = (int)(tu / (unsigned)Ho), oy = (int)(tu - (unsigned)b * (unsigned)Ho);
  const int64_t pix = pixu;
  float best[8];
  unsigned char bi[8];
  bool any = false;
  for (int kh = 0; kh < 3; kh++) {
    int iy = oy * 2 - 1 + kh;
    if (iy < 0 || iy >= H) continue;
    for (int kw = 0; kw < 3; kw++) {
      int ix = ox * 2 - 1 + kw;
      if (ix < 0 || ix >= W) continue;
      uint4 v = *(const uint4*)(x + ((int64_t)(b * H + iy) * W + ix) * ldx + c8 * 8);
      unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        float f = (i & 1) ? h_hi(wv[i >> 1]) : h_lo(wv[i >> 1]);
        if (!any || f > best[i]) {
          best[i] = f;
          bi[i] = (unsigned char)(kh * 3 + kw);
        }
      }
      any = true;
    }
  }
  unsigned ow[4];
  unsigned long long iw = 0ull;
#pragma unroll
  for (int i = 0; i < 4; i++) ow[i] = (unsigned)f2bf(best[2 * i]) | ((unsigned)f2bf(best[2 * i + 1]) << 16);
#pragma unroll
  for (int i = 0; i < 8; i++) iw |= (unsigned long long)bi[i] << (8 * i);
  *(uint4*)(y + pix * C + c8 * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  *(unsigned long long*)(idx + pix * C + c8 * 8) = iw;
}

// dy2 != NULL: the pooled map had two consumers (layer1.0's conv1 and its residual add, backbones.py:31-33); their gradients are
// summed here in fp32 (pixel pitches ld_dy / ld_dy2) instead of by an add kernel
__global__ __launch_bounds__(T) void k_maxpool_bwd(const u16* __restrict__ dy, int ld_dy, const u16* __restrict__ dy2, int ld_dy2,
                                                    const unsigned char* __restrict__ idx, int B, int H, int W, int C, int Ho, int Wo,
                                                    u16* __restrict__ dx) {
  const int C8 = C >> 3;
  const unsigned gid = blockIdx.x * (unsigned)T + threadIdx.x;  // 32-bit index arithmetic (host: total < 2^32)
  const int64_t total = (int64_t)B * H * W * C8;
  if ((int64_t)gid >= total) return;
  const unsigned pixu = gid / (unsigned)C8;
  const int c8 = (int)(gid - pixu * (unsigned)C8);
  const unsigned tu = pixu / (unsigned)W;
  const int ix = (int)(pixu - tu * (unsigned)W);
  const int b = (int)(tu / (unsigned)H), iy = (int)(tu - (unsigned)b * (unsigned)H);
  const int64_t pix = pixu;
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; i++) s[i] = 0.f;
  for (int oy = iy / 2; oy <= (iy + 1) / 2; oy++) {  // windows with oy*2-1 <= iy <= oy*2+1
    if (oy >= Ho) continue;
    int kh = iy - (oy * 2 - 1);
    if (kh < 0 || kh > 2) continue;
    for (int ox = ix / 2; ox <= (ix + 1) / 2; ox++) {
      if (ox >= Wo) continue;
      int kw = ix - (ox * 2 - 1);
      if (kw < 0 || kw > 2) continue;
      const int64_t opix = (int64_t)(b * Ho + oy) * Wo + ox;
      unsigned long long iw = *(const unsigned long long*)(idx + opix * C + c8 * 8);
      uint4 v = *(const uint4*)(dy + opix * ld_dy + c8 * 8);
      unsigned wv[4] = {v.x, v.y, v.z, v.w};
      const unsigned tap = (unsigned)(kh * 3 + kw);
#pragma unroll
      for (int i = 0; i < 8; i++)
        if (((iw >> (8 * i)) & 0xFFull) == tap) s[i] += (i & 1) ? h_hi(wv[i >> 1]) : h_lo(wv[i >> 1]);
      if (dy2) {
        const uint4 v2 = *(const uint4*)(dy2 + opix * ld_dy2 + c8 * 8);
        const unsigned w2[4] = {v2.x, v2.y, v2.z, v2.w};
#pragma unroll
        for (int i = 0; i < 8; i++)
          if (((iw >> (8 * i)) & 0xFFull) == tap) s[i] += (i & 1) ? h_hi(w2[i >> 1]) : h_lo(w2[i >> 1]);
      }
    }
  }
  unsigned ow[4];
#pragma unroll
  for (int i = 0; i < 4; i++) ow[i] = (unsigned)f2bf(s[2 * i]) | ((unsigned)f2bf(s[2 * i + 1]) << 16);
  *(uint4*)(dx + pix * C + c8 * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

// ---- heads: z[pix][j] = sum_c x[pix][c] * Wj[j][c]   (x NHWC bf16 with row pitch; region h x w of an Hp x Wp map)
template <int MJ>
__global__ __launch_bounds__(T) void k_head_proj(const u16* __restrict__ x, int Hp, int Wp, int ld, int B, int h, int w, int C,
                                                  const float* __restrict__ Wj, int NJ, float* __restrict__ z) {
  extern __shared__ float ws[];  // [NJ][C]
  for (int i = threadIdx.x; i < NJ * C; i += T) ws[i] = Wj[i];
  __syncthreads();
  const unsigned pix = blockIdx.x * (unsigned)T + threadIdx.x;  // host guarantees B*h*w < 2^31
  if (pix >= (unsigned)B * h * w) return;
  const unsigned t = pix / (unsigned)w;
  const int xx = (int)(pix - t * w), b = (int)(t / (unsigned)h), yy = (int)(t - (unsigned)b * h);
  const u16* row = x + ((int64_t)(b * Hp + yy) * Wp + xx) * ld;
  float acc[MJ];
#pragma unroll
  for (int j = 0; j < MJ; j++) acc[j] = 0.f;
  for (int c0 = 0; c0 < C; c0 += 8) {
    uint4 v = *(const uint4*)(row + c0);
    unsigned wv[4] = {v.x, v.y, v.z, v.w};
    float xv[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      xv[2 * i] = h_lo(wv[i]);
      xv[2 * i + 1] = h_hi(wv[i]);
    }
#pragma unroll
    for (int j = 0; j < MJ; j++)
      if (j < NJ) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[j] = fmaf(xv[i], ws[j * C + c0 + i], acc[j]);
      }
  }
#pragma unroll
  for (int j = 0; j < MJ; j++)
    if (j < NJ) z[(int64_t)pix * NJ + j] = acc[j];
}

// MFMA version for C = 64, NJ <= 16 (the model's heads): D[j][pixel] = W[j][:] . x[pixel][:] on v_mfma_f32_16x16x32_bf16;
// x is bf16 already, the fp32 weights enter as two bf16 terms (hi + lo: 2^-17 relative), so this is the fp32-weight
// product to fp32 accumulation noise.  A lane loads 16 B of one pixel row per MFMA (whole 128-B rows per 4 lanes) and
// stores 4 consecutive outputs of its pixel (16 B): a pure streaming kernel.
typedef h16x8 hbf16x8;
typedef float hf32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(T) void k_head_proj_mfma(const u16* __restrict__ x, int Hp, int Wp, int ld, int B, int h, int w,
                                                       const float* __restrict__ Wj, int NJ, float* __restrict__ z, int groups) {
  const int lane = threadIdx.x & 63, jl = lane & 15, sl = lane >> 4;
  hbf16x8 wh[2], wl[2];  // A operand: W[j = jl][c = 32*half + 8*sl + t]
#pragma unroll
  for (int half = 0; half < 2; half++)
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const float v = jl < NJ ? Wj[jl * 64 + 32 * half + 8 * sl + t] : 0.f;
      const h16 hh = (h16)v;
      wh[half][t] = hh;
      wl[half][t] = (h16)(v - (float)hh);
    }
  const unsigned total = (unsigned)B * h * w;
  const int gw = (blockIdx.x * (T / 64) + (threadIdx.x >> 6));  // global wave index
  const int nw = gridDim.x * (T / 64);
  for (int g = gw; g < groups; g += nw) {
    const unsigned pix = (unsigned)g * 16 + jl;  // this lane's pixel (B operand column / D column)
    const bool ok = pix < total;
    const unsigned t = pix / (unsigned)w;
    const int xx = (int)(pix - t * w), b = (int)(t / (unsigned)h), yy = (int)(t - (unsigned)b * h);
    const u16* row = x + ((int64_t)(b * Hp + yy) * Wp + xx) * ld + 8 * sl;
    uint4 v0 = make_uint4(0u, 0u, 0u, 0u), v1 = v0;
    if (ok) v0 = *(const uint4*)row, v1 = *(const uint4*)(row + 32);
    const hbf16x8 x0 = __builtin_bit_cast(hbf16x8, v0), x1 = __builtin_bit_cast(hbf16x8, v1);
    hf32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = MM_MFMA_16x16x32(wl[0], x0, acc);
    acc = MM_MFMA_16x16x32(wl[1], x1, acc);
    acc = MM_MFMA_16x16x32(wh[0], x0, acc);
    acc = MM_MFMA_16x16x32(wh[1], x1, acc);
    if (ok) {  // D row 4*sl + r = output j, column jl = pixel
      float* o = z + (int64_t)pix * NJ + 4 * sl;
      if (4 * sl + 3 < NJ && (NJ & 3) == 0) *(hf32x4*)o = acc;
      else
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (4 * sl + r < NJ) o[r] = acc[r];
    }
  }
}

// 5x5 box / 25 with zero padding on NHWC fp32 maps [B,h,w,NJ] (+ bias): thread = (pixel, j), j fastest, so both the
// 25 reads and the write are coalesced across the wave
__global__ __launch_bounds__(T) void k_box5(const float* __restrict__ in, int B, int h, int w, int NJ, const float* __restrict__ bias,
                                             float* __restrict__ out) {
  const unsigned gid = blockIdx.x * (unsigned)T + threadIdx.x;  // 32-bit index arithmetic (host: total < 2^32)
  const int64_t total = (int64_t)B * h * w * NJ;
  if ((int64_t)gid >= total) return;
  const unsigned t1 = gid / (unsigned)NJ;
  const int j = (int)(gid - t1 * (unsigned)NJ);
  const unsigned t2 = t1 / (unsigned)w;
  const int xx = (int)(t1 - t2 * (unsigned)w);
  const int b = (int)(t2 / (unsigned)h), yy = (int)(t2 - (unsigned)b * (unsigned)h);
  float s = 0.f;
  for (int dy = -2; dy <= 2; dy++) {
    int y2 = yy + dy;
    if (y2 < 0 || y2 >= h) continue;
    const float* rowp = in + ((int64_t)(b * h + y2) * w) * NJ + j;
    for (int dx = -2; dx <= 2; dx++) {
      int x2 = xx + dx;
      if (x2 >= 0 && x2 < w) s += rowp[(int64_t)x2 * NJ];
    }
  }
  out[gid] = s * (1.f / 25.f) + (bias ? bias[j] : 0.f);
}

// 4 maps per thread (NJ % 4 == 0) as a sliding window (round 4): a thread owns (image, column, 4 maps) and a run of BOX_ROWS rows; it forms the horizontal
// 5-sums H[y] once per row (5 loads) and keeps the last five in registers: out[y] = (H[y-2] + ... + H[y+2]) / 25 - 6.25 loads per
// output instead of 25 (the 25-load form ran at the L1 rate: 135 us for a 112 MB map).  Sums: left to right, then top to bottom.
constexpr int BOX_ROWS = 16;
__global__ __launch_bounds__(T) void k_box5_slide(const float* __restrict__ in, int B, int h, int w, int NJ4, const float* __restrict__ bias,
                                                   float* __restrict__ out) {
  const unsigned gid = blockIdx.x * (unsigned)T + threadIdx.x;
  const int nrun = (h + BOX_ROWS - 1) / BOX_ROWS;
  const int64_t total = (int64_t)B * nrun * w * NJ4;
  if ((int64_t)gid >= total) return;
  const unsigned t1 = gid / (unsigned)NJ4;
  const int j4 = (int)(gid - t1 * (unsigned)NJ4);
  const unsigned t2 = t1 / (unsigned)w;
  const int xx = (int)(t1 - t2 * (unsigned)w);
  const int b = (int)(t2 / (unsigned)nrun), run = (int)(t2 - (unsigned)b * (unsigned)nrun);
  const int y0 = run * BOX_ROWS, y1 = y0 + BOX_ROWS < h ? y0 + BOX_ROWS : h;
  const int x0 = xx - 2 < 0 ? 0 : xx - 2, x1 = xx + 2 >= w ? w - 1 : xx + 2;
  const float4* base = (const float4*)in + (int64_t)b * h * w * NJ4 + j4;
  auto hsum = [&](int y) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y >= 0 && y < h) {
      const float4* rowp = base + (int64_t)y * w * NJ4;
      for (int x2 = x0; x2 <= x1; x2++) {
        const float4 v = rowp[(int64_t)x2 * NJ4];
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
      }
    }
    return s;
  };
  const float4 bb = bias ? *(const float4*)(bias + 4 * j4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 h0 = hsum(y0 - 2), h1 = hsum(y0 - 1), h2 = hsum(y0), h3 = hsum(y0 + 1);
  for (int y = y0; y < y1; y++) {
    const float4 h4 = hsum(y + 2);
    float4 r;
    r.x = ((((h0.x + h1.x) + h2.x) + h3.x) + h4.x) * (1.f / 25.f) + bb.x;
    r.y = ((((h0.y + h1.y) + h2.y) + h3.y) + h4.y) * (1.f / 25.f) + bb.y;
    r.z = ((((h0.z + h1.z) + h2.z) + h3.z) + h4.z) * (1.f / 25.f) + bb.z;
    r.w = ((((h0.w + h1.w) + h2.w) + h3.w) + h4.w) * (1.f / 25.f) + bb.w;
    ((float4*)out)[((int64_t)(b * h + y) * w + xx) * NJ4 + j4] = r;
    h0 = h1, h1 = h2, h2 = h3, h3 = h4;
  }
}

// dx[pix][c] = sum_j dz[pix][j] * Wj[j][c]  (bf16, zero outside the h x w region), partial dW[j][c] = sum_pix dz*x.
// C = 64: a thread owns 4 channels of one pixel (8-byte loads/stores), 16 threads per pixel, 4 pixels per wave; the
// per-block dW partial is folded over the wave's 4 pixel slots with two shuffles and over the 4 waves through LDS.
// PART 0: both results in one pass (234 registers at 12 outputs: two waves per SIMD); PART 1: dx only (reads dz, writes dx), PART 2:
// the dW partials only (reads x and dz) - half the registers each, twice the occupancy.  Measured at the heads' 12 outputs (round 4):
// 200 + 165 us against 308 us for the one pass - the loop is bound by its per-thread chain of dependent trips, not by registers -
// so the split only serves output counts whose single pass would spill (> 12).
template <int MJ, int PART>
__global__ __launch_bounds__(T) void k_head_bwd(const u16* __restrict__ x, int Hp, int Wp, int ld, int B, int h, int w, int C,
                                                 const float* __restrict__ Wj, int NJ, const float* __restrict__ dz,
                                                 u16* __restrict__ dx, float* __restrict__ partial, int64_t pix_per_block) {
  extern __shared__ float sm[];  // [T/64][NJ][C]
  const int c4 = (threadIdx.x & 15) * 4, slot = threadIdx.x >> 4;
  constexpr int NSLOT = T / 16;
  float wc[MJ][4], acc[MJ][4];
#pragma unroll
  for (int j = 0; j < MJ; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      wc[j][i] = (PART != 2 && j < NJ) ? Wj[j * C + c4 + i] : 0.f;
      acc[j][i] = 0.f;
    }
  const int64_t total = (int64_t)B * Hp * Wp;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
  const int64_t p1 = p0 + pix_per_block < total ? p0 + pix_per_block : total;
  // UN pixels per thread and iteration, all loads issued before the arithmetic (the loop is latency-bound otherwise)
  constexpr int UN = PART == 0 ? 4 : 2;  // (the light passes: fewer pixels in flight per thread, twice the waves per SIMD)
  for (int64_t pb = p0 + slot; pb < p1; pb += NSLOT * UN) {
    uint2 xv2[UN];
    float gj[UN][MJ];
    bool in[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int64_t pp = pb + (int64_t)u * NSLOT;
      const unsigned t = (unsigned)pp / (unsigned)Wp;  // host guarantees B*Hp*Wp < 2^31
      const int xx = (int)((unsigned)pp - t * (unsigned)Wp), b = (int)(t / (unsigned)Hp), yy = (int)(t - (unsigned)b * (unsigned)Hp);
      in[u] = pp < p1 && yy < h && xx < w;
      xv2[u] = make_uint2(0u, 0u);
      if (in[u]) {
        if (PART != 1) xv2[u] = *(const uint2*)(x + pp * ld + c4);
        const float* g = dz + ((int64_t)(b * h + yy) * w + xx) * NJ;
        if (MJ % 4 == 0 && NJ == MJ && !((uintptr_t)dz & 15)) {  // the pixel's NJ floats as 16-byte loads (a quarter of the load instructions)
#pragma unroll
          for (int q = 0; q < MJ / 4; q++) {
            const float4 v = ((const float4*)g)[q];
            gj[u][4 * q] = v.x, gj[u][4 * q + 1] = v.y, gj[u][4 * q + 2] = v.z, gj[u][4 * q + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int j = 0; j < MJ; j++) gj[u][j] = j < NJ ? g[j] : 0.f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int64_t pp = pb + (int64_t)u * NSLOT;
      if (pp >= p1) break;
      float o[4] = {0.f, 0.f, 0.f, 0.f};
      if (in[u]) {
        const float xv[4] = {h_lo(xv2[u].x), h_hi(xv2[u].x), h_lo(xv2[u].y), h_hi(xv2[u].y)};
#pragma unroll
        for (int j = 0; j < MJ; j++)
          if (j < NJ) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
              if (PART != 2) o[i] = fmaf(gj[u][j], wc[j][i], o[i]);
              if (PART != 1) acc[j][i] = fmaf(gj[u][j], xv[i], acc[j][i]);
            }
          }
      }
      if (PART != 2) {
        uint2 r;
        r.x = (unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16);
        r.y = (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16);
        *(uint2*)(dx + pp * ld + c4) = r;
      }
    }
  }
  if (PART == 1) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < MJ; j++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float a = acc[j][i];
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      if (j < NJ && lane < 16) sm[(wave * NJ + j) * C + c4 + i] = a;
    }
  __syncthreads();
  for (int e = threadIdx.x; e < NJ * C; e += T) {
    float s = 0.f;
    for (int wv = 0; wv < T / 64; wv++) s += sm[wv * NJ * C + e];
    partial[(int64_t)blockIdx.x * NJ * C + e] = s;
  }
}

// one wave per output element: lanes stride over the block partials (fp64), fixed shuffle tree
__global__ __launch_bounds__(64) void k_sum_partials_f(const float* __restrict__ partial, int nblk, int ne, float* __restrict__ out) {
  const int e = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) s += (double)partial[(int64_t)b * ne + e];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if (threadIdx.x == 0) out[e] = (float)s;
}
}  // namespace

static void launch_box5(const float* in, int B, int h, int w, int NJ, const float* bias, float* out, hipStream_t s) {
  int64_t npix = (int64_t)B * h * w;
  // (the kernels index their threads with 32 bits: the callers' maps are far below 2^32 elements, mm_head_* check it)
  if (NJ % 4 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0))
    hipLaunchKernelGGL(k_box5_slide, dim3((unsigned)mm_cdiv((int64_t)B * mm_cdiv(h, BOX_ROWS) * w * (NJ / 4), T)), dim3(T), 0, s, in, B, h, w,
                       NJ / 4, bias, out);
  else
    hipLaunchKernelGGL(k_box5, dim3((unsigned)mm_cdiv(npix * NJ, T)), dim3(T), 0, s, in, B, h, w, NJ, bias, out);
}

extern "C" {

#ifndef MM_ACT_FP16  // the copies move 2-byte elements whatever they encode: one build serves both storage formats
// strided 2-byte-element row copy (channel concat / split of NHWC tensors); C multiple of 8
int mm_copy_rows_bf16(const void* src, int64_t ld_s, void* dst, int64_t ld_d, int64_t N, int C, hipStream_t s) {
  MM_CHECK_ARG(C % 8 == 0 && ld_s % 8 == 0 && ld_d % 8 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
               "copy_rows: C and pitches must be multiples of 8 elements");
  if (N == 0) return MM_OK;
  MM_CHECK_ARG(N * (C / 8) < (1ll << 32) - 4096, "copy_rows: too many elements for 32-bit thread indices");
  hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)mm_cdiv(N * (C / 8), T)), dim3(T), 0, s, (const u16*)src, ld_s, (u16*)dst, ld_d, N, C / 8);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// wide[r] = [parts[0][r] | parts[1][r] | ...] (split != 0: parts[i][r] = the i-th channel slice of wide[r]); all maps are
// dense NHWC bf16 rows, channel counts multiples of 8, 1 <= nparts <= 4
int mm_concat_bf16(void* const* parts, const int* channels, int nparts, void* wide, int64_t N, int split, hipStream_t s) {
  MM_CHECK_ARG(nparts >= 1 && nparts <= 4 && parts && channels && wide, "concat: 1..4 parts");
  CatP p = {};
  int ct = 0;
  for (int i = 0; i < nparts; i++) {
    MM_CHECK_ARG(channels[i] > 0 && channels[i] % 8 == 0 && ((uintptr_t)parts[i] % 16) == 0, "concat: channels must be multiples of 8");
    p.src[i] = (const u16*)parts[i];
    p.c8[i] = channels[i] / 8;
    ct += channels[i];
  }
  MM_CHECK_ARG(ct / 8 <= T && ((uintptr_t)wide % 16) == 0, "concat: at most 2048 channels");
  p.n = nparts, p.wide = (u16*)wide, p.ct8 = ct / 8, p.N = N;
  if (N == 0) return MM_OK;
  const int rows_per_block = (T / p.ct8) * 8;
  const unsigned nb = (unsigned)mm_cdiv(N, rows_per_block);
  if (split) hipLaunchKernelGGL(k_concat<true>, dim3(nb), dim3(T), 0, s, p);
  else hipLaunchKernelGGL(k_concat<false>, dim3(nb), dim3(T), 0, s, p);
  MM_LAUNCH_CHECK();
  return MM_OK;
}
#endif  // MM_ACT_FP16

int MM_SYM(mm_maxpool3x3s2_fwd)(const void* x, int ldx, int B, int H, int W, int C, void* y, void* idx, hipStream_t s) {
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  MM_CHECK_ARG(C % 8 == 0 && ldx % 8 == 0 && ldx >= C, "maxpool: C and the input pitch must be multiples of 8");
  int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  if (total == 0) return MM_OK;
  MM_CHECK_ARG(total < (1ll << 32) - 4096, "maxpool: too many elements for 32-bit thread indices");
  hipLaunchKernelGGL(k_maxpool_fwd, dim3((unsigned)mm_cdiv(total, T)), dim3(T), 0, s, (const u16*)x, ldx, B, H, W, C, (u16*)y,
                     (unsigned char*)idx, Ho, Wo);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int MM_SYM(mm_maxpool3x3s2_bwd)(const void* dy, int ld_dy, const void* dy2, int ld_dy2, const void* idx, int B, int H, int W, int C, void* dx,
                        hipStream_t s) {
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  MM_CHECK_ARG(C % 8 == 0 && ld_dy % 8 == 0 && (!dy2 || ld_dy2 % 8 == 0), "maxpool: C and the pitches must be multiples of 8");
  int64_t total = (int64_t)B * H * W * (C / 8);
  if (total == 0) return MM_OK;
  MM_CHECK_ARG(total < (1ll << 32) - 4096, "maxpool: too many elements for 32-bit thread indices");
  hipLaunchKernelGGL(k_maxpool_bwd, dim3((unsigned)mm_cdiv(total, T)), dim3(T), 0, s, (const u16*)dy, ld_dy, (const u16*)dy2, ld_dy2,
                     (const unsigned char*)idx, B, H, W, C, Ho, Wo, (u16*)dx);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

size_t MM_SYM(mm_head_ws_bytes)(int B, int h, int w, int Hp, int Wp, int C, int NJ) {
  size_t z = mm_align((size_t)B * h * w * NJ * sizeof(float));
  size_t part = mm_align((size_t)8192 * NJ * C * sizeof(float));
  return z + part + 256;
}

// out [B,h,w,NJ] fp32 (NHWC) = box5x5( x[.., :h, :w, :] . Wj^T ) + bias      (x: NHWC bf16 [B,Hp,Wp,C] with pitch ld)
int MM_SYM(mm_head_fwd)(const void* x, int B, int Hp, int Wp, int ld, int h, int w, int C, const float* Wj, const float* bias, int NJ,
                float* out, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(NJ > 0 && NJ <= MAXJ && C % 8 == 0 && (size_t)NJ * C * 4 <= 60 * 1024, "head: bad NJ/C");
  size_t zb = mm_align((size_t)B * h * w * NJ * sizeof(float));
  if (ws_bytes < zb) {
    mm_set_error("head_fwd: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  float* z = (float*)ws;
  int64_t npix = (int64_t)B * h * w;
  if (npix == 0) return MM_OK;
  MM_CHECK_ARG(npix < (1ll << 31), "head: too many pixels");
#define MM_HEAD_PROJ(MJ)                                                                                                          \
  hipLaunchKernelGGL(k_head_proj<MJ>, dim3((unsigned)mm_cdiv(npix, T)), dim3(T), (size_t)NJ * C * 4, s, (const u16*)x, Hp, Wp, ld, B, h, \
                     w, C, Wj, NJ, z)
  if (C == 64 && NJ <= 16 && ld % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)z % 16) == 0) {
    const int groups = (int)mm_cdiv(npix, 16);
    int nb = (int)mm_cdiv(groups, (T / 64) * 4);  // ~4 pixel groups per wave
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_head_proj_mfma, dim3(nb), dim3(T), 0, s, (const u16*)x, Hp, Wp, ld, B, h, w, Wj, NJ, z, groups);
  } else if (NJ <= 8) MM_HEAD_PROJ(8);
  else if (NJ <= 12) MM_HEAD_PROJ(12);
  else if (NJ <= 20) MM_HEAD_PROJ(20);
  else MM_HEAD_PROJ(32);
#undef MM_HEAD_PROJ
  MM_CHECK_ARG((int64_t)B * h * w * NJ < (1ll << 32) - 4096, "head: too many elements for 32-bit thread indices");
  launch_box5(z, B, h, w, NJ, bias, out, s);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// dout [B,h,w,NJ] fp32 (NHWC) -> dx NHWC bf16 [B,Hp,Wp,C] (zero outside h x w), dWj [NJ,C]
int MM_SYM(mm_head_bwd)(const void* x, int B, int Hp, int Wp, int ld, int h, int w, int C, const float* Wj, int NJ, const float* dout,
                void* dx, float* dWj, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(NJ > 0 && NJ <= MAXJ && C == 64, "head_bwd: C must be 64");
  MM_CHECK_ARG((int64_t)B * Hp * Wp < (1ll << 31), "head_bwd: too many pixels");
  size_t zb = mm_align((size_t)B * h * w * NJ * sizeof(float));
  const int64_t total = (int64_t)B * Hp * Wp;
  int nblk = (int)mm_cdiv(total, 128);
  if (nblk > 2048) nblk = 2048;
  if (nblk < 1) nblk = 1;
  const int64_t ppb = mm_cdiv(total, nblk);
  if (ws_bytes < zb + (size_t)nblk * NJ * C * sizeof(float)) {
    mm_set_error("head_bwd: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  float* dz = (float*)ws;
  float* partial = (float*)((char*)ws + zb);
  int64_t npix = (int64_t)B * h * w;
  MM_CHECK_ARG(npix * NJ < (1ll << 32) - 4096, "head: too many elements for 32-bit thread indices");
  if (npix) launch_box5(dout, B, h, w, NJ, nullptr, dz, s);
#define MM_HEAD_BWD(MJ, PART)                                                                                                         \
  hipLaunchKernelGGL((k_head_bwd<MJ, PART>), dim3(nblk), dim3(T), (size_t)(T / 64) * NJ * C * 4, s, (const u16*)x, Hp, Wp, ld, B, h, w, C, \
                     Wj, NJ, dz, (u16*)dx, partial, ppb)
  if (NJ <= 8) {
    MM_HEAD_BWD(8, 0);
  } else if (NJ <= 12) {  // (the two heads of the net.  Two light passes measured 200 + 165 us against 308 for the one pass: not used)
    MM_HEAD_BWD(12, 0);
  } else if (NJ <= 20) {
    MM_HEAD_BWD(20, 1);
    MM_HEAD_BWD(20, 2);
  } else {
    MM_HEAD_BWD(32, 1);
    MM_HEAD_BWD(32, 2);
  }
#undef MM_HEAD_BWD
  hipLaunchKernelGGL(k_sum_partials_f, dim3(NJ * C), dim3(64), 0, s, partial, nblk, NJ * C, dWj);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
