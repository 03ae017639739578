// Sparse convolution engines for gfx950 (Appendix A.2-A.4).
//
// One rulebook format serves SubmanifoldConvolution, Convolution and Deconvolution, forward and backward:
//   k-major rule lists (src[r], dst[r]), bucket offsets[K+1], and a CSR over destination rows.
// Engines:
//   G  gather-GEMM   : per bucket k, rows in[src[r]] (coalesced 16-B row gathers straight into MFMA operands)
//                      times W[k] (staged in LDS in MFMA-fragment order) -> tmp[r] (contiguous) or out[dst[r]]
//                      when every destination row has exactly one rule (Deconvolution fwd, Convolution dX).
//   R  CSR reduce    : out[o] = sum over the rules of row o, ascending k (canonical order A.8 iv) - plain
//                      stores, no float atomics, bit-stable run to run.
//   dW               : per bucket k, in[src]^T . dout[dst] on fp32 MFMA from LDS-staged row tiles, partial
//                      slabs per block + ordered reduce (deterministic).
// Arithmetic is exact fp32: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain (no reduced precision).
// Generic VALU kernels cover channel counts that are not multiples of 16 (the 3-channel stem).
#include <stdlib.h>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAXK = 32;
struct KSeg {
  int32_t blk_start[MAXK + 1];
  int32_t rule_off[MAXK + 1];
};

constexpr int TR = 256;    // rules per workgroup, engine G
constexpr int TRW = 1024;  // rules per workgroup, dW

__device__ inline int find_k(const KSeg& seg, int b, int K) {
  int k = 0;
  while (k + 1 < K && b >= seg.blk_start[k + 1]) k++;
  return k;
}

// ------------------------------------------------------------------------------------------------ engine G (MFMA)
// Weights are first packed (k_pack_frag, one small launch per call) into MFMA-fragment order
//   Wf[k][q][cb][lane][j] = W[kk][ci = 16q + 4(lane>>4) + j][co = 16cb + (lane&15)]   (0 beyond Cin / Cout)
// so a workgroup stages W[k] with coalesced 16-B copies and every lane reads its A operand with one ds_read_b128.
__global__ __launch_bounds__(256) void k_pack_frag(const float* __restrict__ W, int64_t w_kstride, int s_ci, int s_co, int kflip,
                                                    int K, int Cin, int Cout, int nq, int ncb, float* __restrict__ Wf) {
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t total = (int64_t)K * nq * ncb * 256;
  if (e >= total) return;
  int j = (int)(e & 3), lane = (int)((e >> 2) & 63);
  int64_t t = e >> 8;
  int cb = (int)(t % ncb);
  t /= ncb;
  int q = (int)(t % nq), k = (int)(t / nq);
  int ci = 16 * q + 4 * (lane >> 4) + j, co = 16 * cb + (lane & 15);
  float v = 0.f;
  if (ci < Cin && co < Cout) v = W[(int64_t)(kflip ? K - 1 - k : k) * w_kstride + (int64_t)ci * s_ci + (int64_t)co * s_co];
  Wf[e] = v;
}

// EDGE: Cin or Cout not a multiple of 16 (the 3-channel stem and its dX): guarded scalar row loads / stores
template <int NCB, bool EDGE>
__global__ __launch_bounds__(256) void k_gather_gemm(const float* __restrict__ in, int ld_in,
                                                      const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                      float* __restrict__ out, int ld_out, const float* __restrict__ Wf,
                                                      int ncb_tot, int K, int Cin, int Cout, int tr, KSeg seg) {
  extern __shared__ float wl[];  // [nq][NCB][64 lanes][4]
  const int tid = threadIdx.x;
  const int k = find_k(seg, blockIdx.x, K);
  const int r_begin = seg.rule_off[k] + (blockIdx.x - seg.blk_start[k]) * tr;
  const int r_end = min(seg.rule_off[k + 1], r_begin + tr);
  const int cb0 = blockIdx.y * NCB;
  const int co_base = cb0 * 16;
  const int nq = (Cin + 15) >> 4;
  const int wave = tid >> 6, lane = tid & 63, rl = lane & 15, sl = lane >> 4;
  {  // W[k] slice -> LDS by LDS-DMA, one 1-KiB (q, cb) fragment block per wave instruction, all in flight together
    const char* Wk = (const char*)((const f32x4*)Wf + ((int64_t)k * nq * ncb_tot + cb0) * 64) + lane * 16;
    for (int b = wave; b < nq * NCB; b += 4) {
      const int q = b / NCB, r = b - q * NCB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Wk + ((int64_t)q * ncb_tot + r) * 1024),
                                       (__attribute__((address_space(3))) void*)((char*)wl + b * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  for (int g = r_begin + wave * 16; g < r_end; g += 64) {
    const int r = g + rl;
    const bool valid = r < r_end;
    const int sidx = valid ? src[r] : 0;
    const float* row = in + (int64_t)sidx * ld_in + sl * 4;
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < nq; q++) {
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        if (EDGE) {
          const int c0 = q * 16 + sl * 4;
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (c0 + j < Cin) x[j] = row[q * 16 + j];
        } else {
          x = *(const f32x4*)(row + q * 16);
        }
      }
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
        f32x4 w = *(const f32x4*)&wl[((q * NCB + cb) * 64 + lane) * 4];
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, acc[cb], 0, 0, 0);
      }
    }
    if (valid) {
      const int64_t orow = dst ? (int64_t)dst[r] : (int64_t)r;
      float* o = out + orow * ld_out + co_base + sl * 4;
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
        if (EDGE) {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (co_base + cb * 16 + sl * 4 + j < Cout) o[cb * 16 + j] = acc[cb][j];
        } else {
          *(f32x4*)(o + cb * 16) = acc[cb];
        }
      }
    }
  }
}

// Small rulebooks (coarse levels): staging W[k] per workgroup costs more than the rows it multiplies.  Here every wave owns
// one 16-rule group and reads the packed fragments straight from L2 (1 KiB coalesced per wave-instruction): no LDS, no
// barrier, one group per wave -> thousands of independent waves even at a few thousand rules.
template <int NCB>
__global__ __launch_bounds__(256) void k_gather_gemm_direct(const float* __restrict__ in, int ld_in,
                                                             const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                             float* __restrict__ out, int ld_out, const float* __restrict__ Wf,
                                                             int ncb_tot, int K, int Cin, KSeg seg) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 15, sl = lane >> 4;
  const int k = find_k(seg, blockIdx.x, K);
  const int g = seg.rule_off[k] + (blockIdx.x - seg.blk_start[k]) * 64 + wave * 16;
  const int r_end = seg.rule_off[k + 1];
  if (g >= r_end) return;
  const int cb0 = blockIdx.y * NCB;
  const int nq = Cin >> 4;
  const int r = g + rl;
  const bool valid = r < r_end;
  const int sidx = valid ? src[r] : 0;
  const float* row = in + (int64_t)sidx * ld_in + sl * 4;
  const f32x4* Wk = (const f32x4*)Wf + ((int64_t)k * nq * ncb_tot + cb0) * 64 + lane;
  f32x4 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; cb++) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int q = 0; q < nq; q++) {
    f32x4 x = valid ? *(const f32x4*)(row + q * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 w[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) w[cb] = Wk[((int64_t)q * ncb_tot + cb) * 64];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cb].x, x.x, acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cb].y, x.y, acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cb].z, x.z, acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[cb].w, x.w, acc[cb], 0, 0, 0);
    }
  }
  if (valid) {
    const int64_t orow = dst ? (int64_t)dst[r] : (int64_t)r;
    float* o = out + orow * ld_out + cb0 * 16 + sl * 4;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) *(f32x4*)(o + cb * 16) = acc[cb];
  }
}

// ------------------------------------------------------------------------------------- engine G, split-bf16 products
// For Cin >= 64 the gather-GEMM is bound by the fp32 matrix rate (v_mfma_f32_16x16x4_f32: 256 FLOP/clk/CU; the layers
// measure 46-55 TFLOP/s), not by HBM.  Here every fp32 operand is split in registers into NT bf16 terms,
// x = x1 + x2 (+ x3), x1 = bf16_rne(x), x2 = bf16_rne(x - x1), x3 = bf16_rne(x - x1 - x2); the weights are pre-split by
// k_pack_frag_s3, and x.w is accumulated in fp32 on v_mfma_f32_16x16x32_bf16 from the partial products whose magnitude
// reaches the target precision:
//   NT = 3 (default): x1w1 + x1w2 + x2w1 + x2w2 + x1w3 + x3w1   - everything down to 2^-18 |x.w|; the three 8-bit terms
//            carry the whole 24-bit significand, the dropped products are <= 2^-26 |x.w|: fp32-faithful (6 x 16 cycles
//            per 16x16x32 block instead of 8 x 32 on the fp32 matrix instruction: 2.7x the rate)
//   NT = 2 (MM_SPCONV_SPLIT=2): x1w1 + x1w2 + x2w1, product error <= ~2^-16 |x.w| (3 x 16 cycles: 5.3x the rate)
// MM_SPCONV_FP32=1 keeps the plain fp32 engine.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(32))) const float g_zero8s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
__device__ __attribute__((aligned(64))) const float g_zero128[128] = {0.f};  // zero line for absent rules: 16 lanes + up to 4 blocks of 16

template <int NT>
__device__ inline void split8(const f32x4& a, const f32x4& b, bf16x8 (&t)[NT]) {
#ifdef MM_DIAG_FAKESPLIT  // diagnostic build only (tools/diag_lib.sh): no VALU split - what operands stored pre-split would cost
#pragma unroll
  for (int n = 0; n < NT; n++) t[n] = __builtin_bit_cast(bf16x8, (n & 1) ? b : a);
  return;
#endif
  float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float r = v[i];
#pragma unroll
    for (int n = 0; n < NT; n++) {
      const __bf16 h = (__bf16)r;
      t[n][i] = h;
      r -= (float)h;
    }
  }
}

// acc += x . w from the split terms (a: MFMA A operand, b: MFMA B operand), largest products last is not needed: fp32 adds
// IEEE fp16 rows (16-bit activation mode, fp16 kind): the values ARE fp16, one exact term, v_mfma_f32_16x16x32_f16
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NT>
__device__ inline void split8(const f32x4& a, const f32x4& b, f16x8 (&t)[NT]) {
  static_assert(NT == 1, "fp16 rows take one term");
  t[0] = f16x8{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
}
template <int NT>
__device__ inline f32x4 mfma_split(const f16x8 (&a)[NT], const f16x8 (&b)[NT], f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
}
template <int NT>
__device__ inline f32x4 mfma_split(const bf16x8 (&a)[NT], const bf16x8 (&b)[NT], f32x4 acc) {
  if constexpr (NT == 1) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  }
  if (NT == 3) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NT > 1 ? 1 : 0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[NT > 1 ? 1 : 0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  return acc;
}

// Wf3[k][q][cb][term][lane][0..7] = bf16 terms of W[kk][ci = 32q + 8(lane>>4) + j][co = 16cb + (lane&15)]  (0 beyond Cin):
// the layout of the output-stationary engine's fragments (csrc/osconv.hip pack_one), so that a caller that keeps those per
// optimiser step (mm_spconv_os_pack_batch) can hand them to mm_spconv_apply_packed and skip this launch
template <int NT>
__global__ __launch_bounds__(256) void k_pack_frag_s3(const float* __restrict__ W, int64_t w_kstride, int s_ci, int s_co, int kflip,
                                                       int K, int Cin, int Cout, int nq, int ncb, __bf16* __restrict__ Wf) {
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t total = (int64_t)K * nq * ncb * 64 * 8;
  if (e >= total) return;
  int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
  int64_t t = e >> 9;
  int cb = (int)(t % ncb);
  t /= ncb;
  int q = (int)(t % nq), k = (int)(t / nq);
  int ci = 32 * q + 8 * (lane >> 4) + j, co = 16 * cb + (lane & 15);
  float r = 0.f;
  if (ci < Cin && co < Cout) r = W[(int64_t)(kflip ? K - 1 - k : k) * w_kstride + (int64_t)ci * s_ci + (int64_t)co * s_co];
  const int64_t base = (e >> 9) * (512 * NT) + lane * 8 + j;
#pragma unroll
  for (int n = 0; n < NT; n++) {
    const __bf16 h = (__bf16)r;
    Wf[base + 512 * n] = h;
    r -= (float)h;
  }
}

// LDSW: W[k] (the NT term fragments of the NCB cout blocks) staged in LDS once per workgroup and reused over its `tr`
// rules; !LDSW (small rulebooks): one 16-rule group per wave, fragments straight from L2, no barrier.
// Cin, Cout multiples of 16; D[co][rule] as in k_gather_gemm.
template <int NCB, bool LDSW, int NT>
__global__ __launch_bounds__(256) void k_gather_gemm_s3(const float* __restrict__ in, int ld_in, const int32_t* __restrict__ src,
                                                         const int32_t* __restrict__ dst, float* __restrict__ out, int ld_out,
                                                         const __bf16* __restrict__ Wf, int ncb_tot, int K, int Cin, int tr,
                                                         KSeg seg, int preload) {
  extern __shared__ __attribute__((aligned(16))) char wlds[];  // [nq][NCB][NT][64 lanes][16 B]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 15, sl = lane >> 4;
  const int k = find_k(seg, blockIdx.x, K);
  const int r_begin = seg.rule_off[k] + (blockIdx.x - seg.blk_start[k]) * tr;
  const int r_end = min(seg.rule_off[k + 1], r_begin + tr);
  const int cb0 = blockIdx.y * NCB;
  const int nq = (Cin + 31) >> 5;
  const bf16x8* Wk = (const bf16x8*)Wf + ((int64_t)k * nq * ncb_tot + cb0) * 64 * NT;
  if (LDSW) {
    // W[k] slice -> LDS by LDS-DMA: the fragment blocks ([64 lanes] x 16 B = 1 KiB, one per (q, cb, term)) are exactly what one
    // wave instruction writes, all of a wave's blocks are in flight together and nothing passes through registers.  (The
    // round-2 copy loop compiled to load -> s_waitcnt vmcnt(0) -> ds_write per 16-byte piece: up to 19 serial L2 round trips
    // per workgroup before the first MFMA.)
    for (int b = wave; b < nq * NCB * NT; b += 4) {
      const int q = b / (NCB * NT), r = b - q * (NCB * NT);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const char*)Wk + ((int64_t)q * ncb_tot * NT + r) * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(wlds + b * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // Round 3: the row loads of a rule group no longer sit inside the channel-chunk loop (one exposed gather latency per 32
  // channels: 1 + nq dependent round trips per 16 rules, and the SQ counters showed the waves parked 67 % of their cycles).
  // All chunks of the group are requested at once (NQMAX x 2 sixteen-byte loads in flight per lane, absent chunks read a
  // zero line), and the source index of the NEXT group is fetched before this group is multiplied.
  constexpr int NQMAX = 7;  // Cin <= 224
  int g = r_begin + wave * 16;
  int sidx_next = (g + rl < r_end) ? src[g + rl] : 0;
  for (; g < r_end; g += 64) {
    const int r = g + rl;
    const bool valid = r < r_end;
    const int sidx = sidx_next;
    if (g + 64 < r_end) sidx_next = (g + 64 + rl < r_end) ? src[g + 64 + rl] : 0;  // uniform branch
    const float* row = in + (int64_t)sidx * ld_in + sl * 8;
    f32x4 xq[NQMAX][2];
    const bool pre = preload && nq <= NQMAX;  // uniform (MM_SPCONV_G_PRELOAD=0: the round-2 loop, for A/B runs)
    if (pre) {
      // no branch per chunk: with one, the compiler moved every chunk's loads down next to that chunk's multiply again; a chunk
      // beyond Cin (or an absent rule) reads the zero line - a data select, so all loads issue back to back
#pragma unroll
      for (int q = 0; q < NQMAX; q++) {
        typedef const __attribute__((address_space(1))) f32x4* gp4;  // keep the global address space through the select (else: flat loads)
        const gp4 pq = (valid && q * 32 + sl * 8 < Cin) ? (gp4)(row + q * 32) : (gp4)g_zero8s;  // Cin % 16 == 0: 8 channels in or out together
        xq[q][0] = pq[0];
        xq[q][1] = pq[1];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto chunk = [&](int q, const f32x4& x0, const f32x4& x1) {
      bf16x8 xt[NT];
      split8<NT>(x0, x1, xt);
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
        const bf16x8* w8 = LDSW ? (const bf16x8*)wlds + (q * NCB + cb) * 64 * NT + lane
                                : Wk + ((int64_t)q * ncb_tot + cb) * 64 * NT + lane;
        bf16x8 wt[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) wt[n] = w8[n * 64];
        acc[cb] = mfma_split<NT>(wt, xt, acc[cb]);
      }
    };
    if (pre) {
#pragma unroll
      for (int q = 0; q < NQMAX; q++)
        if (q < nq) chunk(q, xq[q][0], xq[q][1]);
    } else {  // wider than the preload window: one chunk at a time (the round-2 loop)
      for (int q = 0; q < nq; q++) {
        f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
        if (valid && q * 32 + sl * 8 < Cin) {
          x0 = *(const f32x4*)(row + q * 32);
          x1 = *(const f32x4*)(row + q * 32 + 4);
        }
        chunk(q, x0, x1);
      }
    }
    if (valid) {
      const int64_t orow = dst ? (int64_t)dst[r] : (int64_t)r;
      float* o = out + orow * ld_out + cb0 * 16 + sl * 4;
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) *(f32x4*)(o + cb * 16) = acc[cb];
    }
  }
}

// Narrow inputs (the 3-channel stem, Cin <= 4): output-stationary, thread = one output row with all Cout (<= 32)
// accumulators; the rules of the row come from the CSR in ascending k; W (K*Cin*Cout floats) sits in LDS.
__global__ __launch_bounds__(256) void k_rows_narrow(const float* __restrict__ in, int ld_in, const int32_t* __restrict__ src,
                                                      KSeg seg, int K, const int32_t* __restrict__ csr_off,
                                                      const int32_t* __restrict__ csr_pos, int64_t n_out, float* __restrict__ out,
                                                      int ld_out, const float* __restrict__ W, int64_t w_kstride, int s_ci, int s_co,
                                                      int kflip, int Cin, int Cout) {
  extern __shared__ float wsm[];  // [K][Cin][Cout]
  for (int e = threadIdx.x; e < K * Cin * Cout; e += 256) {
    int co = e % Cout, t = e / Cout, ci = t % Cin, k = t / Cin;
    wsm[e] = W[(int64_t)(kflip ? K - 1 - k : k) * w_kstride + (int64_t)ci * s_ci + (int64_t)co * s_co];
  }
  __syncthreads();
  int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n_out) return;
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; c++) acc[c] = 0.f;
  for (int e = csr_off[row]; e < csr_off[row + 1]; e++) {
    const int pos = csr_pos[e];
    int k = 0;
    while (k + 1 < K && pos >= seg.rule_off[k + 1]) k++;
    const float* x = in + (int64_t)src[pos] * ld_in;
    for (int ci = 0; ci < Cin; ci++) {
      const float xv = x[ci];
      const float* wr = wsm + (k * Cin + ci) * Cout;
#pragma unroll
      for (int c = 0; c < 32; c++)
        if (c < Cout) acc[c] = fmaf(xv, wr[c], acc[c]);
    }
  }
  float* o = out + row * ld_out;
#pragma unroll
  for (int c = 0; c < 32; c++)
    if (c < Cout) o[c] = acc[c];
}

// ------------------------------------------------------------------------------------------------ engine R
__global__ __launch_bounds__(256) void k_csr_reduce(const float* __restrict__ tmp, int ld_tmp,
                                                     const int32_t* __restrict__ csr_off,
                                                     const int32_t* __restrict__ csr_pos, int64_t n_out,
                                                     float* __restrict__ out, int ld_out, int C4) {
  // 32-bit index arithmetic (host: n_out * C4 < 2^32): the 64-bit division this replaced was ~100 instructions per thread,
  // more than the thread's actual work (six 16-byte loads on average)
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  const unsigned row = gid / (unsigned)C4;
  const int c4 = (int)(gid - row * (unsigned)C4);
  if ((int64_t)row >= n_out) return;
  int a = csr_off[row], b = csr_off[row + 1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // four rules at a time: the four positions, then the four tmp rows, are independent loads (a rule-by-rule loop is a chain
  // of two dependent memory round trips per rule); rules beyond the row's last one re-read it and are not added.  The sum
  // stays in ascending rule order = ascending k.
  // (round 3) the positions of the next four rules are requested before the current four tmp rows are added: one exposed round
  // trip per iteration instead of two (positions, then rows)
  int posn[4];
  if (a < b) {
#pragma unroll
    for (int j = 0; j < 4; j++) posn[j] = csr_pos[a + j < b ? a + j : b - 1];
  }
  for (int e = a; e < b; e += 4) {
    int pos[4];
#pragma unroll
    for (int j = 0; j < 4; j++) pos[j] = posn[j];
    if (e + 4 < b) {
#pragma unroll
      for (int j = 0; j < 4; j++) posn[j] = csr_pos[e + 4 + j < b ? e + 4 + j : b - 1];
    }
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = *(const f32x4*)(tmp + (int64_t)pos[j] * ld_tmp + c4 * 4);
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (e + j < b) acc += v[j];
  }
  *(f32x4*)(out + (int64_t)row * ld_out + c4 * 4) = acc;
}

__global__ __launch_bounds__(256) void k_csr_reduce_scalar(const float* __restrict__ tmp, int ld_tmp,
                                                            const int32_t* __restrict__ csr_off,
                                                            const int32_t* __restrict__ csr_pos, int64_t n_out,
                                                            float* __restrict__ out, int ld_out, int C) {
  int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t row = gid / C;
  int c = (int)(gid - row * C);
  if (row >= n_out) return;
  float acc = 0.f;
  for (int e = csr_off[row]; e < csr_off[row + 1]; e++) acc += tmp[(int64_t)csr_pos[e] * ld_tmp + c];
  out[row * ld_out + c] = acc;
}

// ------------------------------------------------------------------------------------------------ generic VALU engines
__device__ inline int k_of_pos(const int32_t* __restrict__ offsets, int K, int pos) {
  int k = 0;
  while (k + 1 < K && pos >= offsets[k + 1]) k++;
  return k;
}

// output-stationary: thread = (out row, co); rules of the row visited in ascending k
__global__ __launch_bounds__(256) void k_generic_rows(const float* __restrict__ in, int ld_in,
                                                       const int32_t* __restrict__ src,
                                                       const int32_t* __restrict__ offsets, int K,
                                                       const int32_t* __restrict__ csr_off,
                                                       const int32_t* __restrict__ csr_pos, int64_t n_out,
                                                       float* __restrict__ out, int ld_out, const float* __restrict__ W,
                                                       int64_t w_kstride, int s_ci, int s_co, int kflip, int Cin,
                                                       int Cout) {
  int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t row = gid / Cout;
  int co = (int)(gid - row * Cout);
  if (row >= n_out) return;
  float acc = 0.f;
  for (int e = csr_off[row]; e < csr_off[row + 1]; e++) {
    int pos = csr_pos[e];
    int k = k_of_pos(offsets, K, pos);
    const float* Wk = W + (int64_t)(kflip ? K - 1 - k : k) * w_kstride + (int64_t)co * s_co;
    const float* x = in + (int64_t)src[pos] * ld_in;
    for (int ci = 0; ci < Cin; ci++) acc = fmaf(x[ci], Wk[(int64_t)ci * s_ci], acc);
  }
  out[row * ld_out + co] = acc;
}

// unique destinations: thread = (rule, co)
__global__ __launch_bounds__(256) void k_generic_rules(const float* __restrict__ in, int ld_in,
                                                        const int32_t* __restrict__ src,
                                                        const int32_t* __restrict__ dst,
                                                        const int32_t* __restrict__ offsets, int K, int64_t n_rules,
                                                        float* __restrict__ out, int ld_out, const float* __restrict__ W,
                                                        int64_t w_kstride, int s_ci, int s_co, int kflip, int Cin,
                                                        int Cout) {
  int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t r = gid / Cout;
  int co = (int)(gid - r * Cout);
  if (r >= n_rules) return;
  int k = k_of_pos(offsets, K, (int)r);
  const float* Wk = W + (int64_t)(kflip ? K - 1 - k : k) * w_kstride + (int64_t)co * s_co;
  const float* x = in + (int64_t)src[r] * ld_in;
  float acc = 0.f;
  for (int ci = 0; ci < Cin; ci++) acc = fmaf(x[ci], Wk[(int64_t)ci * s_ci], acc);
  out[(int64_t)dst[r] * ld_out + co] = acc;
}

// ------------------------------------------------------------------------------------------------ dW (MFMA)
// dW[k] = in[src]^T . dout[dst] over the rules of bucket k.  One workgroup = `chunk` rules of one bucket and a
// TI x TJ rectangle of 16x16 output blocks (blockIdx.y); its four waves interleave 4-rule MFMA steps and feed the
// operands straight from global memory (16 lanes = one contiguous 64-B row segment, no LDS staging, no barriers
// in the loop), then combine their accumulators through LDS in wave order (deterministic) into one partial slab.
template <int TI, int TJ>
__global__ __launch_bounds__(256) void k_dw_direct(const float* __restrict__ in, int ld_in,
                                                    const float* __restrict__ dout, int ld_do,
                                                    const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                    int Cin, int Cout, int K, KSeg seg, int chunk,
                                                    float* __restrict__ partial) {
  extern __shared__ float red[];  // [3 waves][TI*TJ][64 lanes] f32x4
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 15, sl = lane >> 4;
  const int k = find_k(seg, blockIdx.x, K);
  const int r_begin = seg.rule_off[k] + (blockIdx.x - seg.blk_start[k]) * chunk;
  const int r_end = min(seg.rule_off[k + 1], r_begin + chunk);
  const int ncib = Cin >> 4, ncob = Cout >> 4;
  const int nrj = (ncob + TJ - 1) / TJ;
  const int ci0 = (blockIdx.y / nrj) * TI, co0 = (blockIdx.y % nrj) * TJ;
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int r0 = r_begin + wave * 4; r0 < r_end; r0 += 16) {
    const int r = r0 + sl;
    const bool valid = r < r_end;
    const int si = valid ? src[r] : 0, di = valid ? dst[r] : 0;
    const float* ap = in + (int64_t)si * ld_in + rl;
    const float* bp = dout + (int64_t)di * ld_do + rl;
    float a[TI], b[TJ];
#pragma unroll
    for (int i = 0; i < TI; i++) a[i] = (valid && ci0 + i < ncib) ? ap[(ci0 + i) * 16] : 0.f;
#pragma unroll
    for (int j = 0; j < TJ; j++) b[j] = (valid && co0 + j < ncob) ? bp[(co0 + j) * 16] : 0.f;
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) *(f32x4*)&red[(((wave - 1) * TI * TJ + i * TJ + j) * 64 + lane) * 4] = acc[i][j];
  }
  __syncthreads();
  if (wave == 0) {
    float* P = partial + (int64_t)blockIdx.x * Cin * Cout;
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) {
        f32x4 v = acc[i][j];
#pragma unroll
        for (int w = 0; w < 3; w++) v += *(const f32x4*)&red[((w * TI * TJ + i * TJ + j) * 64 + lane) * 4];
        if (ci0 + i < ncib && co0 + j < ncob) {
#pragma unroll
          for (int r = 0; r < 4; r++) P[(int64_t)((ci0 + i) * 16 + sl * 4 + r) * Cout + (co0 + j) * 16 + rl] = v[r];
        }
      }
  }
}

// Split-bf16 variant (see engine G): 32 rules per v_mfma_f32_16x16x32_bf16; lane (rl = channel, sl) gathers the 8 rules
// 8*sl .. 8*sl+7 of the group for its channel of every block, splits them into NT terms, and each (ci block, co block)
// pair accumulates the partial products of mfma_split.
// E = __bf16 (16-bit activation mode, NT = 1): the gathered values are already bf16, nothing is split.
template <int TI, int TJ, int NT, typename E = float>
__global__ __launch_bounds__(256) void k_dw_direct_s3(const E* __restrict__ in, int ld_in, const E* __restrict__ dout,
                                                       int ld_do, const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                       int Cin, int Cout, int K, KSeg seg, int chunk, float* __restrict__ partial) {
  extern __shared__ float red[];  // [3 waves][TI*TJ][64 lanes] f32x4
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 15, sl = lane >> 4;
  const int k = find_k(seg, blockIdx.x, K);
  const int r_begin = seg.rule_off[k] + (blockIdx.x - seg.blk_start[k]) * chunk;
  const int r_end = min(seg.rule_off[k + 1], r_begin + chunk);
  const int ncib = Cin >> 4, ncob = Cout >> 4;
  const int nrj = (ncob + TJ - 1) / TJ;
  const int ci0 = (blockIdx.y / nrj) * TI, co0 = (blockIdx.y % nrj) * TJ;
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Round 3: an iteration of the round-2 loop was two exposed round trips - sixteen dword index loads, a wait, then the value
  // loads.  The eight rule indices of a lane are contiguous (rules 8 sl .. 8 sl + 7): two 16-byte loads per index array,
  // and the indices of the NEXT 32-rule group are requested before this group is gathered, split and multiplied.  (Also
  // tried: unconditional value loads with the absent rules selected to zero afterwards instead of one exec-masked block per
  // load - the scheduler then hoists the 64-bit address arithmetic of all 48 loads, runs out of registers and issues them
  // in waited batches of 12: dW 2.14 -> 3.44 ms.  The branchy form keeps one address pair live at a time.)
  typedef int i32x4 __attribute__((ext_vector_type(4), aligned(4)));  // bucket offsets are arbitrary: dword alignment only
  auto load_idx = [&](int r0, int (&si)[8], int (&di)[8]) {
    const int rb = r0 + 8 * sl;
    if (rb + 8 <= r_end) {  // the usual case: 32 contiguous bytes of each index array (4-byte aligned is enough)
      const i32x4 s0 = *(const i32x4*)(src + rb), s1 = *(const i32x4*)(src + rb + 4);
      const i32x4 d0 = *(const i32x4*)(dst + rb), d1 = *(const i32x4*)(dst + rb + 4);
#pragma unroll
      for (int t = 0; t < 4; t++) si[t] = s0[t], si[4 + t] = s1[t], di[t] = d0[t], di[4 + t] = d1[t];
    } else {  // the tail of the chunk
#pragma unroll
      for (int t = 0; t < 8; t++) {
        const int r = rb + t;
        const bool valid = r < r_end;
        const int rc = valid ? r : r_end - 1;
        const int sv = src[rc], dv = dst[rc];
        si[t] = valid ? sv : -1;
        di[t] = valid ? dv : 0;
      }
    }
  };
  int sn[8], dn[8];  // indices of the NEXT group (prefetched)
  int r0 = r_begin + wave * 32;
  if (r0 < r_end) load_idx(r0, sn, dn);
  for (; r0 < r_end; r0 += 128) {
    // row pointers of the eight rules, once per group (the round-2 loop redid the 64-bit row multiply for every one of the
    // 8 (TI + TJ) loads, each inside its own exec-masked block: ~6 vector and 4 scalar instructions per loaded dword).  An
    // absent rule points at a zero line: the loads are unconditional, carry the block as an immediate offset and need no
    // select afterwards.
    typedef const __attribute__((address_space(1))) E* gptr;  // global address space kept through the select (else: flat loads)
    gptr pa[8], pb[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const bool live = sn[t] >= 0;
      pa[t] = live ? (gptr)(in + (int64_t)sn[t] * ld_in + ci0 * 16 + rl) : (gptr)((const E*)g_zero128 + rl);
      pb[t] = live ? (gptr)(dout + (int64_t)dn[t] * ld_do + co0 * 16 + rl) : (gptr)((const E*)g_zero128 + rl);
    }
    if (r0 + 128 < r_end) load_idx(r0 + 128, sn, dn);  // uniform: in flight while this group is gathered, split and multiplied
    typedef std::conditional_t<std::is_same_v<E, _Float16>, f16x8, bf16x8> HV;
    HV at[TI][NT], bt[TJ][NT];
#pragma unroll
    for (int i = 0; i < TI; i++) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (ci0 + i < ncib) {  // uniform
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const float x = (float)pa[t][i * 16];
          if (t < 4) v0[t] = x;
          else v1[t - 4] = x;
        }
      }
      split8<NT>(v0, v1, at[i]);
    }
#pragma unroll
    for (int j = 0; j < TJ; j++) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (co0 + j < ncob) {  // uniform
#pragma unroll
        for (int t = 0; t < 8; t++) {
          const float x = (float)pb[t][j * 16];
          if (t < 4) v0[t] = x;
          else v1[t - 4] = x;
        }
      }
      split8<NT>(v0, v1, bt[j]);
    }
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) acc[i][j] = mfma_split<NT>(at[i], bt[j], acc[i][j]);
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) *(f32x4*)&red[(((wave - 1) * TI * TJ + i * TJ + j) * 64 + lane) * 4] = acc[i][j];
  }
  __syncthreads();
  if (wave == 0) {
    float* P = partial + (int64_t)blockIdx.x * Cin * Cout;
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) {
        f32x4 v = acc[i][j];
#pragma unroll
        for (int w = 0; w < 3; w++) v += *(const f32x4*)&red[((w * TI * TJ + i * TJ + j) * 64 + lane) * 4];
        if (ci0 + i < ncib && co0 + j < ncob) {
#pragma unroll
          for (int r = 0; r < 4; r++) P[(int64_t)((ci0 + i) * 16 + sl * 4 + r) * Cout + (co0 + j) * 16 + rl] = v[r];
        }
      }
  }
}

// dW over 16-BIT rows (round 6; BASELINE.json configs[4]): whole rows through LDS and the hardware transpose read.  k_dw_direct_s3
// gathers one ELEMENT per rule and lane - the MFMA's K dimension (the rules) lies along a lane's register vector - i.e. 2-byte loads
// for 16-bit rows: 8 (TI + TJ) load instructions per 32 rules and wave, the instructions that bound the kernel; halving the row bytes
// did not shorten it (2.5 TB/s of the 16-bit algorithmic bytes on configs[4], against 3.2-4.1 of the fp32 ones on configs[1]).  Here a
// wave gathers the 32 rows of a group with 16-byte loads (TI + TJ instructions), writes them as a [32 rules][channels] tile of its
// own LDS region and reads the MFMA operands back transposed with ds_read_b64_tr_b16 (4 rules x 16 channels per 16 lanes: two reads
// per 16-channel block) - no split (the rows are 16-bit already), no workgroup barrier in the loop (a wave's LDS operations execute in
// order), the next group's rows in flight under the current group's MFMAs.  Tile rows are padded to an odd multiple of 32 bytes: the
// eight rows a 32-lane half reads then cover all 64 banks.  Same channel tiles, slabs, cross-wave sum and summation order per slab
// as k_dw_direct_s3<.., 1, E> (the 32 rules of a group are summed by ONE MFMA either way): the slab sums are unchanged.
template <int TI, int TJ, typename E>
__global__ __launch_bounds__(256) void k_dw_tr16(const E* __restrict__ in, int ld_in, const E* __restrict__ dout, int ld_do,
                                                  const int32_t* __restrict__ src, const int32_t* __restrict__ dst, int Cin, int Cout, int K,
                                                  KSeg seg, int chunk, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char ldsb[];
  constexpr int SA = ((TI * 32 + 31) / 32 | 1) * 32, SB = ((TJ * 32 + 31) / 32 | 1) * 32;  // row pitch (bytes): odd multiples of 32
  constexpr int WREG = 32 * (SA + SB);                                                     // bytes of one wave's two tiles
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef std::conditional_t<std::is_same_v<E, _Float16>, f16x8, bf16x8> HV;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 15, sl = lane >> 4;
  const int k = find_k(seg, blockIdx.x, K);
  const int r_begin = seg.rule_off[k] + (blockIdx.x - seg.blk_start[k]) * chunk;
  const int r_end = min(seg.rule_off[k + 1], r_begin + chunk);
  const int ncib = Cin >> 4, ncob = Cout >> 4;
  const int nrj = (ncob + TJ - 1) / TJ;
  const int ci0 = (blockIdx.y / nrj) * TI, co0 = (blockIdx.y % nrj) * TJ;
  char* const At = ldsb + wave * WREG;
  char* const Bt = At + 32 * SA;
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // gather geometry: a row of the input tile is 2 TI 16-byte chunks, lanes walk (rule, chunk) pairs: pair e = lane + 64 it
  constexpr int CA = 2 * TI, CB = 2 * TJ, ITA = CA / 2, ITB = CB / 2;  // 32 CA / 64 iterations
  typedef const __attribute__((address_space(1))) u4* gq;
  auto load_idx = [&](int r0, int& si, int& di) {  // lane l (and l + 32) holds rule r0 + (l & 31)
    const int r = r0 + (lane & 31);
    const bool ok = r < r_end;
    si = ok ? src[ok ? r : r_begin] : -1;
    di = ok ? dst[ok ? r : r_begin] : 0;
  };
  u4 ra[ITA], rb[ITB];  // the gathered chunks of one group
  auto gather = [&](int si, int di) {
#pragma unroll
    for (int it = 0; it < ITA; it++) {
      const int e = lane + 64 * it, rule = e / CA, ch = e % CA;
      const int s = __shfl(si, rule, 64);
      const bool live = s >= 0 && ci0 * 16 + ch * 8 < Cin;
      const gq pq = live ? (gq)(in + (int64_t)s * ld_in + ci0 * 16 + ch * 8) : (gq)(const void*)g_zero128;
      ra[it] = *pq;
    }
#pragma unroll
    for (int it = 0; it < ITB; it++) {
      const int e = lane + 64 * it, rule = e / CB, ch = e % CB;
      const int s = __shfl(si, rule, 64), d = __shfl(di, rule, 64);
      const bool live = s >= 0 && co0 * 16 + ch * 8 < Cout;
      const gq pq = live ? (gq)(dout + (int64_t)d * ld_do + co0 * 16 + ch * 8) : (gq)(const void*)g_zero128;
      rb[it] = *pq;
    }
  };
  // transpose-read address of this lane inside a tile: lane group g = sl reads rules 8 g + 4 h + q, piece pp (8 bytes) of a 32-byte block
  const int q = (lane >> 2) & 3, pp = lane & 3;
  const int ta = (8 * sl + q) * SA + pp * 8, tb = (8 * sl + q) * SB + pp * 8;
  int r0 = r_begin + wave * 32;
  int si = -1, di = 0, sn = -1, dn = 0;
  if (r0 < r_end) {
    load_idx(r0, si, di);
    gather(si, di);
    if (r0 + 128 < r_end) load_idx(r0 + 128, sn, dn);
  }
  for (; r0 < r_end; r0 += 128) {
    // this group's rows -> the wave's tiles (the previous group's transpose reads have returned: their MFMAs were issued)
#pragma unroll
    for (int it = 0; it < ITA; it++) {
      const int e = lane + 64 * it;
      *(u4*)(At + (e / CA) * SA + (e % CA) * 16) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < ITB; it++) {
      const int e = lane + 64 * it;
      *(u4*)(Bt + (e / CB) * SB + (e % CB) * 16) = rb[it];
    }
    // the next group's rows are requested now and land under this group's reads and MFMAs; its successor's indices behind them
    const bool more = r0 + 128 < r_end;
    if (more) {
      gather(sn, dn);
      if (r0 + 256 < r_end) load_idx(r0 + 256, sn, dn);
    }
    HV at[TI], bt[TJ];
#pragma unroll
    for (int i = 0; i < TI; i++) {
      const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(At + ta + i * 32));
      const s16x4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(At + ta + 4 * SA + i * 32));
      const s16x8 o = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
      at[i] = __builtin_bit_cast(HV, o);
    }
#pragma unroll
    for (int j = 0; j < TJ; j++) {
      const s16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Bt + tb + j * 32));
      const s16x4 w = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Bt + tb + 4 * SB + j * 32));
      const s16x8 o = {u.x, u.y, u.z, u.w, w.x, w.y, w.z, w.w};
      bt[j] = __builtin_bit_cast(HV, o);
    }
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) {
        if constexpr (std::is_same_v<E, _Float16>) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(at[i], bt[j], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[i], bt[j], acc[i][j], 0, 0, 0);
      }
  }
  // cross-wave sum in wave order (k_dw_direct_s3's), through the same LDS: every wave has left its tiles first
  float* red = (float*)ldsb;
  __syncthreads();
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) *(f32x4*)&red[(((wave - 1) * TI * TJ + i * TJ + j) * 64 + lane) * 4] = acc[i][j];
  }
  __syncthreads();
  if (wave == 0) {
    float* P = partial + (int64_t)blockIdx.x * Cin * Cout;
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
      for (int j = 0; j < TJ; j++) {
        f32x4 v = acc[i][j];
#pragma unroll
        for (int w = 0; w < 3; w++) v += *(const f32x4*)&red[((w * TI * TJ + i * TJ + j) * 64 + lane) * 4];
        if (ci0 + i < ncib && co0 + j < ncob) {
#pragma unroll
          for (int r = 0; r < 4; r++) P[(int64_t)((ci0 + i) * 16 + sl * 4 + r) * Cout + (co0 + j) * 16 + rl] = v[r];
        }
      }
  }
}

// (Measured and dropped, round 6: the same for fp32 rows - 16-byte row gathers into a per-wave fp32 tile, the eight rules of a lane's
// channel read back with ds_read_b32, then split8 / mfma_split as k_dw_direct_s3: identical slabs, 1.0-1.8 x its time on the bench's
// layers - the LDS round trip of 4-byte operands costs more than the 4-byte gathers it replaces, and the
// tiles cap the occupancy at two workgroups per CU.)
// generic dW: thread = (ci, co) pairs strided over the block; rules staged in LDS 64 at a time
__global__ __launch_bounds__(256) void k_dw_generic(const float* __restrict__ in, int ld_in,
                                                     const float* __restrict__ dout, int ld_do,
                                                     const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                                     int Cin, int Cout, int K, KSeg seg, float* __restrict__ partial) {
  extern __shared__ float sm[];
  float* As = sm;             // [64][Cin]
  float* Bs = sm + 64 * Cin;  // [64][Cout]
  const int tid = threadIdx.x;
  const int k = find_k(seg, blockIdx.x, K);
  const int r_begin = seg.rule_off[k] + (blockIdx.x - seg.blk_start[k]) * TRW;
  const int r_end = min(seg.rule_off[k + 1], r_begin + TRW);
  const int ne = Cin * Cout;
  float* P = partial + (int64_t)blockIdx.x * ne;
  for (int e0 = 0; e0 < ne; e0 += 256) {  // usually one pass (3x16 = 48 elements)
    const int e = e0 + tid;
    const int ci = e / Cout, co = e - ci * Cout;
    float acc = 0.f;
    for (int t0 = r_begin; t0 < r_end; t0 += 64) {
      __syncthreads();
      for (int x = tid; x < 64 * Cin; x += 256) {
        int rr = x / Cin, c = x - rr * Cin, r = t0 + rr;
        As[x] = r < r_end ? in[(int64_t)src[r] * ld_in + c] : 0.f;
      }
      for (int x = tid; x < 64 * Cout; x += 256) {
        int rr = x / Cout, c = x - rr * Cout, r = t0 + rr;
        Bs[x] = r < r_end ? dout[(int64_t)dst[r] * ld_do + c] : 0.f;
      }
      __syncthreads();
      if (e < ne)
        for (int rr = 0; rr < 64; rr++) acc = fmaf(As[rr * Cin + ci], Bs[rr * Cout + co], acc);
    }
    if (e < ne) P[e] = acc;
  }
}

__global__ __launch_bounds__(256) void k_dw_reduce(const float* __restrict__ partial, int ne, int K, KSeg seg,
                                                    float* __restrict__ dW, int accumulate) {
  // block = 32 consecutive elements x 8 chunk slices; fp64 accumulation, slices combined in a fixed order
  __shared__ double red[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int k = blockIdx.y;
  const int e = blockIdx.x * 32 + el;
  double acc = 0.0;
  if (e < ne)
    for (int b = seg.blk_start[k] + sl; b < seg.blk_start[k + 1]; b += 8) acc += (double)partial[(int64_t)b * ne + e];
  red[sl][el] = acc;
  __syncthreads();
  if (sl == 0 && e < ne) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) t += red[i][el];
    float* d = dW + (int64_t)k * ne + e;
    *d = accumulate ? *d + (float)t : (float)t;
  }
}

// k_dw_reduce for every layer of a backward pass in one launch (mm_spconv_dw_reduce_batch): a workgroup finds its layer by
// its block index, then does exactly what k_dw_reduce does for (32 elements, kernel offset k) - same slices, same fp64 sums.
struct DwRedD {
  const float* partial;
  float* dW;
  int32_t ne, K, accumulate, blk_first;
  int32_t blk_start[MAXK + 1];
};

// Work per thread instead of threads per element (the first form - k_dw_reduce's 8 slices x 32 elements per workgroup - was
// 100k workgroups of five loads per thread: 149 us for ~200 MB, bound by workgroup dispatch): a thread owns VEC consecutive
// elements of one kernel offset and walks ALL its slabs with eight loads in flight, keeping k_dw_reduce's eight slice sums
// (slab i of the offset goes to slice i mod 8) in eight accumulators, then adds them in slice order: the same fp64 sums in the
// same order, no LDS, no barrier.
__host__ __device__ inline int dw_red_vec(int ne) { return (ne & 3) ? 1 : 4; }
__host__ __device__ inline int dw_red_blocks_k(int ne) { const int v = dw_red_vec(ne); return (ne / v + 255) >> 8; }

template <int VEC>
__device__ inline void dw_reduce_thread(const DwRedD& d, int k, int g) {
  const int ne = d.ne;
  const int e = g * VEC;
  if (e >= ne) return;
  typedef float fv __attribute__((ext_vector_type(VEC)));
  const float* __restrict__ base = d.partial + e;
  const int begin = d.blk_start[k], end = d.blk_start[k + 1];
  double acc[8][VEC];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int c = 0; c < VEC; c++) acc[i][c] = 0.0;
  int s = begin;
  for (; s + 8 <= end; s += 8) {
    fv v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = *(const fv*)(base + (int64_t)(s + i) * ne);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int c = 0; c < VEC; c++) acc[i][c] += (double)v[i][c];
  }
#pragma unroll
  for (int i = 0; i < 8; i++)  // the last, partial round of slices
    if (s + i < end) {
      const fv v = *(const fv*)(base + (int64_t)(s + i) * ne);
#pragma unroll
      for (int c = 0; c < VEC; c++) acc[i][c] += (double)v[c];
    }
  float* o = d.dW + (int64_t)k * ne + e;
#pragma unroll
  for (int c = 0; c < VEC; c++) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) t += acc[i][c];
    o[c] = d.accumulate ? o[c] + (float)t : (float)t;
  }
}

__global__ __launch_bounds__(256) void k_dw_reduce_batch(const DwRedD* __restrict__ descs, int n) {
  int li = 0;
  while (li + 1 < n && (int)blockIdx.x >= descs[li + 1].blk_first) li++;  // uniform; n is a few dozen
  const DwRedD& d = descs[li];
  const int nbk = dw_red_blocks_k(d.ne);
  const int b = blockIdx.x - d.blk_first, k = b / nbk, g = (b - k * nbk) * 256 + threadIdx.x;
  if (dw_red_vec(d.ne) == 4) dw_reduce_thread<4>(d, k, g);
  else dw_reduce_thread<1>(d, k, g);
}

int make_seg(const int32_t* offsets_host, int K, int rules_per_block, KSeg* seg) {
  int nb = 0;
  for (int k = 0; k < K; k++) {
    seg->blk_start[k] = nb;
    seg->rule_off[k] = offsets_host[k];
    nb += (int)mm_cdiv(offsets_host[k + 1] - offsets_host[k], rules_per_block);
  }
  for (int k = K; k <= MAXK; k++) {
    seg->blk_start[k] = nb;
    seg->rule_off[k] = offsets_host[K];
  }
  return nb;
}

template <int NCB, bool EDGE>
int launch_g(int nb, int nchunk, const float* in, int ld_in, const int32_t* src, const int32_t* dst, float* out, int ld_out,
             const float* Wf, int ncb_tot, int K, int Cin, int Cout, int tr, const KSeg& seg, hipStream_t s) {
  size_t lds = (size_t)((Cin + 15) / 16) * 16 * NCB * 16 * sizeof(float);
  if (lds > 64 * 1024)
    MM_HIP(hipFuncSetAttribute((const void*)k_gather_gemm<NCB, EDGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_gather_gemm<NCB, EDGE>), dim3(nb, nchunk), dim3(256), lds, s, in, ld_in, src, dst, out, ld_out, Wf, ncb_tot, K,
                     Cin, Cout, tr, seg);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // namespace

// ``mode`` of the fp32-row entry points (include/mm2d3d.h MM_SPCONV_*): bits 0-1 = bf16 terms per fp32 operand in the
// split-product engines - 0: three (fp32-faithful, default), 1: none (plain fp32 engines), 2: two (faster, fails the gradient
// parity bar: diagnostics only); bit 2: dW keeps the <= 4 x 4 channel tiles.  The caller passes it with every call - the library
// reads no environment variable and keeps no switch.
static int split_terms(int mode) {
  const int m = mode & 3;
  return m == 0 ? 3 : (m == 1 ? 0 : 2);
}

// smallest input width that takes the split-product engines: 32 for fwd / dX (below that the fp32 engine streams as fast),
// 16 for dW (its fp32 variant is bound by 4-byte gathers + 32-cycle MFMAs already at 16 channels).  Measured on the
// bench step: dW 2.42 -> 2.17 ms, fwd + dX 5.04 -> 4.95 ms against a 64-channel threshold for both.
static constexpr int split_min_cin(bool dw) { return dw ? 16 : 32; }

template <int N, int NT>
static int launch_s3(bool small, int nb, int nch, size_t lds, const float* in, int ld_in, const int32_t* src, const int32_t* d, float* tgt,
                     int ld_t, const __bf16* Wf3, int ncb, int K, int Cin, int tr, const KSeg& sg, hipStream_t s) {
  constexpr int preload = 1;
  if (small) {
    hipLaunchKernelGGL((k_gather_gemm_s3<N, false, NT>), dim3(nb, nch), dim3(256), 0, s, in, ld_in, src, d, tgt, ld_t, Wf3, ncb, K, Cin, tr,
                       sg, preload);
  } else {
    if (lds > 64 * 1024) MM_HIP(hipFuncSetAttribute((const void*)k_gather_gemm_s3<N, true, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_gather_gemm_s3<N, true, NT>), dim3(nb, nch), dim3(256), lds, s, in, ld_in, src, d, tgt, ld_t, Wf3, ncb, K, Cin, tr,
                       sg, preload);
  }
  return MM_OK;
}

extern "C" {

static inline size_t frag_floats(int K, int Cin, int Cout) {
  return (size_t)K * ((Cin + 15) / 16) * ((Cout + 15) / 16) * 256;
}

// bytes of the workspace (tmp rows + packed weight fragments) one mm_spconv_apply call needs
size_t mm_spconv_ws_bytes(int64_t n_rules, int Cin, int Cout, int K) {
  const size_t frag3 = (size_t)K * ((Cin + 31) / 32) * ((Cout + 15) / 16) * 64 * 48;  // split-bf16 fragments (3 terms, Cin padded to 32)
  size_t frag = frag_floats(K, Cin, Cout) * sizeof(float);
  if (frag3 > frag) frag = frag3;
  return mm_align((size_t)n_rules * Cout * sizeof(float)) + mm_align(frag) + 512;
}

// out[dst] (+)= in[src] . W[k]   over a k-major rulebook.
//   unique_dst != 0 : every destination row has exactly one rule -> direct row writes, no reduction
//   unique_dst == 0 : destinations reduced through the CSR (csr_off/csr_pos over n_out rows), k ascending
//   weight element (k, ci, co) is W[kk*w_kstride + ci*s_ci + co*s_co], kk = kflip ? K-1-k : k
//   rows of out without a rule are written as zeros (CSR path) or left untouched (unique path: caller pre-zeros if needed)
//   Wpk (nullable): the three-term fragments of the same weights (same strides, kflip) as written by mm_spconv_os_pack /
//   mm_spconv_os_pack_batch; used by the split-product kernels in place of their own per-call pack
int mm_spconv_apply_packed(const float* in, int ld_in, int Cin, float* out, int ld_out, int Cout, int64_t n_out,
                           const int32_t* src, const int32_t* dst, const int32_t* offsets_dev, const int32_t* offsets_host,
                           int K, const int32_t* csr_off, const int32_t* csr_pos, int unique_dst, const float* W,
                           int64_t w_kstride, int s_ci, int s_co, int kflip, const void* Wpk, int mode, void* ws, size_t ws_bytes,
                           hipStream_t s) {
  MM_CHECK_ARG(K > 0 && K <= MAXK && Cin > 0 && Cout > 0 && ld_in >= Cin && ld_out >= Cout, "spconv_apply: bad shape");
  MM_CHECK_ARG(unique_dst || (csr_off && csr_pos), "spconv_apply: the row CSR is required unless every destination is unique");
  const int64_t R = offsets_host[K];
  if (n_out == 0) return MM_OK;
  const bool edge = (Cin % 16 != 0) || (Cout % 16 != 0) || (ld_in % 4 != 0) || (ld_out % 4 != 0) ||
                    (((uintptr_t)in | (uintptr_t)out) % 16 != 0);
  const int nq = (Cin + 15) / 16, ncb = (Cout + 15) / 16;
  int nchunk = 1;
  if (ncb > 8) {
    nchunk = 0;
    for (int d = 2; d <= ncb; d++)
      if (ncb % d == 0 && ncb / d <= 8) {
        nchunk = d;
        break;
      }
  }
  const size_t tmp_bytes = unique_dst ? 0 : mm_align((size_t)R * Cout * sizeof(float));
  const size_t need = tmp_bytes + frag_floats(K, Cin, Cout) * sizeof(float);
  const bool lds_ok = nchunk && (size_t)nq * 16 * (ncb / (nchunk ? nchunk : 1)) * 64 <= 150 * 1024;
  if (!lds_ok) {  // very wide layers: plain VALU kernels
    if (unique_dst) {
      if (R) hipLaunchKernelGGL(k_generic_rules, dim3((unsigned)mm_cdiv(R * Cout, 256)), dim3(256), 0, s, in, ld_in, src, dst,
                                offsets_dev, K, R, out, ld_out, W, w_kstride, s_ci, s_co, kflip, Cin, Cout);
    } else {
      hipLaunchKernelGGL(k_generic_rows, dim3((unsigned)mm_cdiv(n_out * Cout, 256)), dim3(256), 0, s, in, ld_in, src,
                         offsets_dev, K, csr_off, csr_pos, n_out, out, ld_out, W, w_kstride, s_ci, s_co, kflip, Cin, Cout);
    }
    MM_LAUNCH_CHECK();
    return MM_OK;
  }
  if (!unique_dst && Cin <= 4 && Cout <= 32 && (size_t)K * Cin * Cout * 4 <= 48 * 1024) {  // the stem: no tmp, one pass
    KSeg sg;
    make_seg(offsets_host, K, TR, &sg);
    hipLaunchKernelGGL(k_rows_narrow, dim3((unsigned)mm_cdiv(n_out, 256)), dim3(256), (size_t)K * Cin * Cout * 4, s, in, ld_in, src, sg, K,
                       csr_off, csr_pos, n_out, out, ld_out, W, w_kstride, s_ci, s_co, kflip, Cin, Cout);
    MM_LAUNCH_CHECK();
    return MM_OK;
  }
  if (ws_bytes < need) {
    mm_set_error("spconv_apply: workspace too small (%zu < %zu)", ws_bytes, need);
    return MM_ERR_WORKSPACE;
  }
  MM_CHECK_ARG(((uintptr_t)ws % 16) == 0, "spconv_apply: workspace must be 16-B aligned");
  float* Wf = (float*)((char*)ws + tmp_bytes);
  const int nt = split_terms(mode);
  if (nt && !edge && Cin >= split_min_cin(false) && (unique_dst || Cout % 4 == 0) && R > 0) {  // matrix-rate-bound widths: split-bf16 products
    const int nq3 = (Cin + 31) / 32;
    MM_CHECK_ARG(ws_bytes >= tmp_bytes + (size_t)K * nq3 * ncb * 64 * 16 * nt, "spconv_apply: workspace too small for the split fragments");
    const __bf16* Wf3 = (const __bf16*)Wf;
    if (Wpk && nt == 3 && ((uintptr_t)Wpk % 16) == 0)
      Wf3 = (const __bf16*)Wpk;
    else if (nt == 3)
      hipLaunchKernelGGL(k_pack_frag_s3<3>, dim3((unsigned)mm_cdiv((int64_t)K * nq3 * ncb * 512, 256)), dim3(256), 0, s, W, w_kstride,
                         s_ci, s_co, kflip, K, Cin, Cout, nq3, ncb, (__bf16*)Wf);
    else
      hipLaunchKernelGGL(k_pack_frag_s3<2>, dim3((unsigned)mm_cdiv((int64_t)K * nq3 * ncb * 512, 256)), dim3(256), 0, s, W, w_kstride,
                         s_ci, s_co, kflip, K, Cin, Cout, nq3, ncb, (__bf16*)Wf);
    float* tgt = out;
    int ld_t = ld_out;
    const int32_t* d = dst;
    if (!unique_dst) tgt = (float*)ws, ld_t = Cout, d = nullptr;
    int nch = 1;  // cout blocks per workgroup <= 8 and the staged W[k] slice <= 80 KiB (two workgroups per CU)
    while (nch < ncb && (ncb % nch != 0 || ncb / nch > 8 || (size_t)nq3 * (ncb / nch) * 64 * 16 * nt > 80 * 1024)) nch++;
    if ((size_t)nq3 * (ncb / nch) * 64 * 16 * nt > 150 * 1024) {
      mm_set_error("spconv_apply: %d input channels are too wide for the split-product engine", Cin);
      return MM_ERR_UNSUPPORTED;
    }
    const int ncbw = ncb / nch;
    constexpr int64_t small_r = 0;  // round 3: the LDS-staged form everywhere - with the W slice arriving by LDS-DMA it wins at every size (112->112 on 88k rules: 96 -> 57 us); the unstaged form re-reads the whole slice from L2 per wave
    const bool small = R < small_r;
    int tr = small ? 64 : TR;
    while (!small && tr > 64 && mm_cdiv(R, tr) * nch < 1024) tr >>= 1;
    KSeg sg;
    const int nb = make_seg(offsets_host, K, tr, &sg);
    const size_t lds = small ? 0 : (size_t)nq3 * ncbw * 64 * 16 * nt;
    int rc = MM_OK;
    switch (ncbw) {
#define SCASE(N)                                                                                                          \
  case N:                                                                                                                 \
    rc = nt == 3 ? launch_s3<N, 3>(small, nb, nch, lds, in, ld_in, src, d, tgt, ld_t, Wf3, ncb, K, Cin, tr, sg, s)          \
                 : launch_s3<N, 2>(small, nb, nch, lds, in, ld_in, src, d, tgt, ld_t, Wf3, ncb, K, Cin, tr, sg, s);         \
    break;
      SCASE(1) SCASE(2) SCASE(3) SCASE(4) SCASE(5) SCASE(6) SCASE(7) SCASE(8)
#undef SCASE
      default:
        mm_set_error("spconv_apply: unsupported channel-block count %d", ncbw);
        return MM_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    MM_LAUNCH_CHECK();
    if (!unique_dst) {
      MM_CHECK_ARG(n_out * (Cout / 4) < (1ll << 32) - 256, "spconv_apply: too many output elements for the 32-bit reduce index");
      hipLaunchKernelGGL(k_csr_reduce, dim3((unsigned)mm_cdiv(n_out * (Cout / 4), 256)), dim3(256), 0, s, tgt, Cout, csr_off, csr_pos,
                         n_out, out, ld_out, Cout / 4);
      MM_LAUNCH_CHECK();
    }
    return MM_OK;
  }
  hipLaunchKernelGGL(k_pack_frag, dim3((unsigned)mm_cdiv((int64_t)frag_floats(K, Cin, Cout), 256)), dim3(256), 0, s, W, w_kstride, s_ci,
                     s_co, kflip, K, Cin, Cout, nq, ncb, Wf);
  // small layers: fewer rules per workgroup and channel-block splitting so that the grid still covers the 256 CUs
  int tr = TR;
  while (tr > 64 && mm_cdiv(R, tr) * nchunk < 1024) tr >>= 1;
  while (mm_cdiv(R, tr) * nchunk < 512 && (ncb / nchunk) % 2 == 0 && ncb / nchunk >= 2) nchunk *= 2;
  KSeg seg;
  int nb = make_seg(offsets_host, K, tr, &seg);
  float* tgt = out;
  int ld_t = ld_out;
  const int32_t* d = dst;
  if (!unique_dst) {
    tgt = (float*)ws;
    ld_t = Cout;
    d = nullptr;
  }
  const bool e2 = edge || (!unique_dst && (Cout % 4 != 0));
  if (!e2 && R > 0 && R < 200000 && ncb <= 8) {  // coarse levels: direct-from-L2 weights, one 16-rule group per wave
    KSeg sg;
    const int nbd = make_seg(offsets_host, K, 64, &sg);
    switch (ncb) {
#define DCASE(N)                                                                                                         \
  case N:                                                                                                                \
    hipLaunchKernelGGL(k_gather_gemm_direct<N>, dim3(nbd, 1), dim3(256), 0, s, in, ld_in, src, d, tgt, ld_t, Wf, ncb, K, Cin, sg); \
    break;
      DCASE(1) DCASE(2) DCASE(3) DCASE(4) DCASE(5) DCASE(6) DCASE(7) DCASE(8)
#undef DCASE
    }
    MM_LAUNCH_CHECK();
  } else if (nb > 0) {
    int rc = MM_OK;
    switch (ncb / nchunk) {
#define CASE(N)                                                                                                          \
  case N:                                                                                                                \
    rc = e2 ? launch_g<N, true>(nb, nchunk, in, ld_in, src, d, tgt, ld_t, Wf, ncb, K, Cin, Cout, tr, seg, s)              \
            : launch_g<N, false>(nb, nchunk, in, ld_in, src, d, tgt, ld_t, Wf, ncb, K, Cin, Cout, tr, seg, s);            \
    break;
      CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
      default:
        mm_set_error("spconv_apply: unsupported Cout %d", Cout);
        return MM_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
  }
  if (!unique_dst) {
    if (Cout % 4 == 0 && ld_out % 4 == 0 && ((uintptr_t)out % 16) == 0 && n_out * (Cout / 4) < (1ll << 32) - 256)
      hipLaunchKernelGGL(k_csr_reduce, dim3((unsigned)mm_cdiv(n_out * (Cout / 4), 256)), dim3(256), 0, s, tgt, Cout, csr_off, csr_pos,
                         n_out, out, ld_out, Cout / 4);
    else
      hipLaunchKernelGGL(k_csr_reduce_scalar, dim3((unsigned)mm_cdiv(n_out * Cout, 256)), dim3(256), 0, s, tgt, Cout, csr_off, csr_pos,
                         n_out, out, ld_out, Cout);
    MM_LAUNCH_CHECK();
  }
  return MM_OK;
}

static inline int dw_tile(int n) { return (n + ((n + 3) / 4) - 1) / ((n + 3) / 4); }  // balanced split, <= 4

// Channel tile (ti x tj blocks of 16) of one dW workgroup.  A rule's input row is gathered once per OUTPUT-channel tile and
// its gradient row once per INPUT-channel tile - with 4-byte gathers, the instructions that bound the kernel.  Up to 4 x 4
// blocks every tile shape keeps three or more workgroups per CU; on the wide layers (more than four blocks on a side) the
// balanced <= 4 split gathers 2 - 2.3x the minimum (160 -> 80: 42 block gathers per rule against 15), so there (round 3) the tile
// may grow to 6 blocks on a side, at most 25 accumulator blocks (5 x 5: 155 VGPRs + 100 AGPRs, still two workgroups per CU like
// 4 x 4): the shape with the fewest gathers per rule is taken, the smaller area on a tie (160 -> 80: 5 x 5, 20 gathers).
// Measured per layer on the bench step (same box, us): 96->48 169 -> 126, 160->80 227 -> 139, 192->96 107 -> 91, 80->80 141 -> 86;
// rule lists below ~200k rules (the strided layers of the coarse levels) lose 3-5 us each with the fewer, fatter workgroups and
// keep the <= 4 split.  Element sums and their order do not depend on the tile shape.
constexpr int DW_WIDE_MIN_RULES = 200000;
static void dw_tiles(int nci, int nco, bool wide_ok, int* ti, int* tj) {
  *ti = dw_tile(nci), *tj = dw_tile(nco);
  if (!wide_ok || (nci <= 4 && nco <= 4)) return;
  int best = (int)(mm_cdiv(nci, *ti) * mm_cdiv(nco, *tj)) * (*ti + *tj), area = *ti * *tj;
  for (int a = 1; a <= nci; a++) {
    const int i = (int)mm_cdiv(nci, a);
    if (i > 6) continue;
    for (int b = 1; b <= nco; b++) {
      const int j = (int)mm_cdiv(nco, b);
      if (j > 6 || i * j > 25 || (i <= 4 && j <= 4)) continue;
      const int cost = a * b * (i + j);
      if (cost < best || (cost == best && i * j < area)) best = cost, area = i * j, *ti = i, *tj = j;
    }
  }
}

// rules per workgroup for dW: aim at ~1024 workgroups, 16-rule granularity
static int dw_chunk(int64_t R, int Cin, int Cout, bool wide_ok) {
  if (Cin % 16 || Cout % 16) return TRW;
  int ti, tj;
  dw_tiles(Cin / 16, Cout / 16, wide_ok, &ti, &tj);
  int ny = (int)(mm_cdiv(Cin / 16, ti) * mm_cdiv(Cout / 16, tj));
  int64_t c = mm_cdiv(R, 1024 / ny > 0 ? 1024 / ny : 1);
  c = mm_cdiv(c, 16) * 16;
  if (c < 64) c = 64;
  if (c > 8192) c = 8192;
  return (int)c;
}

int mm_spconv_apply(const float* in, int ld_in, int Cin, float* out, int ld_out, int Cout, int64_t n_out,
                    const int32_t* src, const int32_t* dst, const int32_t* offsets_dev, const int32_t* offsets_host,
                    int K, const int32_t* csr_off, const int32_t* csr_pos, int unique_dst, const float* W,
                    int64_t w_kstride, int s_ci, int s_co, int kflip, int mode, void* ws, size_t ws_bytes, hipStream_t s) {
  return mm_spconv_apply_packed(in, ld_in, Cin, out, ld_out, Cout, n_out, src, dst, offsets_dev, offsets_host, K, csr_off, csr_pos,
                                unique_dst, W, w_kstride, s_ci, s_co, kflip, nullptr, mode, ws, ws_bytes, s);
}

}  // extern "C"

namespace {

// the partial slabs of one layer: partial[slab b][ci][co], slab b = rules [chunk b) of kernel offset find_k(b)
int dw_partial(int bf, int mode, const void* in, int ld_in, int Cin, const void* dout, int ld_do, int Cout, const int32_t* src,
               const int32_t* dst, const int32_t* offsets_host, int K, void* ws, size_t ws_bytes, KSeg* seg_out, hipStream_t s) {
  MM_CHECK_ARG(K > 0 && K <= MAXK && Cin > 0 && Cout > 0, "spconv_dw: bad shape");
  MM_CHECK_ARG(bf >= 0 && bf <= 2, "spconv_dw: row kind must be 0 (fp32), 1 (bf16) or 2 (fp16)");
  MM_CHECK_ARG(!bf || (Cin % 16 == 0 && Cout % 16 == 0), "spconv_dw (16-bit rows): channels must be multiples of 16");
  KSeg& seg = *seg_out;
  const bool mfma_ok = (Cin % 16 == 0) && (Cout % 16 == 0);
  const int nt = bf ? -bf : (mfma_ok && Cin >= split_min_cin(true) ? split_terms(mode) : 0);  // matrix-rate-bound widths, as in mm_spconv_apply
  // the wide tiles exist for the split kernels (they save 4-byte gathers); 16-bit rows are gathered whole (k_dw_tr16): <= 4 x 4 there
  const bool wide_ok = nt > 0 && !(mode & 4) && offsets_host[K] >= DW_WIDE_MIN_RULES;
  const int chunk = dw_chunk(offsets_host[K], Cin, Cout, wide_ok);
  int nb = make_seg(offsets_host, K, chunk, &seg);
  const int ne = Cin * Cout;
  if ((size_t)(nb > 0 ? nb : 1) * ne * sizeof(float) > ws_bytes) {
    mm_set_error("spconv_dw: workspace too small");
    return MM_ERR_WORKSPACE;
  }
  float* partial = (float*)ws;
  if (nb == 0) return MM_OK;
  if (mfma_ok) {
    int ti, tj;
    dw_tiles(Cin / 16, Cout / 16, wide_ok, &ti, &tj);
    const int ny = (int)(mm_cdiv(Cin / 16, ti) * mm_cdiv(Cout / 16, tj));
    const size_t lds = (size_t)3 * ti * tj * 64 * 4 * sizeof(float);
#define DWCASE(I, J)                                                                                                          \
  if (ti == I && tj == J) {                                                                                                   \
    constexpr size_t l16 = (size_t)4 * 32 * ((((I * 32 + 31) / 32 | 1) + ((J * 32 + 31) / 32 | 1)) * 32);                     \
    const size_t lds16 = l16 > lds ? l16 : lds;                                                                               \
    if (nt == -1 && (mode & 8))                                                                                               \
      hipLaunchKernelGGL((k_dw_direct_s3<I, J, 1, __bf16>), dim3(nb, ny), dim3(256), lds, s, (const __bf16*)in, ld_in,        \
                         (const __bf16*)dout, ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                            \
    else if (nt == -2 && (mode & 8))                                                                                          \
      hipLaunchKernelGGL((k_dw_direct_s3<I, J, 1, _Float16>), dim3(nb, ny), dim3(256), lds, s, (const _Float16*)in, ld_in,    \
                         (const _Float16*)dout, ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                          \
    else if (nt == -1)                                                                                                        \
      hipLaunchKernelGGL((k_dw_tr16<I, J, __bf16>), dim3(nb, ny), dim3(256), lds16, s, (const __bf16*)in, ld_in,              \
                         (const __bf16*)dout, ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                            \
    else if (nt == -2)                                                                                                        \
      hipLaunchKernelGGL((k_dw_tr16<I, J, _Float16>), dim3(nb, ny), dim3(256), lds16, s, (const _Float16*)in, ld_in,          \
                         (const _Float16*)dout, ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                          \
    else if (nt == 3)                                                                                                         \
      hipLaunchKernelGGL((k_dw_direct_s3<I, J, 3>), dim3(nb, ny), dim3(256), lds, s, (const float*)in, ld_in, (const float*)dout, \
                         ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                                                 \
    else if (nt == 2)                                                                                                         \
      hipLaunchKernelGGL((k_dw_direct_s3<I, J, 2>), dim3(nb, ny), dim3(256), lds, s, (const float*)in, ld_in, (const float*)dout, \
                         ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                                                 \
    else                                                                                                                      \
      hipLaunchKernelGGL((k_dw_direct<I, J>), dim3(nb, ny), dim3(256), lds, s, (const float*)in, ld_in, (const float*)dout,   \
                         ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                                                 \
  }
    DWCASE(1, 1) DWCASE(1, 2) DWCASE(1, 3) DWCASE(1, 4) DWCASE(2, 1) DWCASE(2, 2) DWCASE(2, 3) DWCASE(2, 4)
    DWCASE(3, 1) DWCASE(3, 2) DWCASE(3, 3) DWCASE(3, 4) DWCASE(4, 1) DWCASE(4, 2) DWCASE(4, 3) DWCASE(4, 4)
#undef DWCASE
    // the wide tiles (split / 16-bit kernels only); more than 64 KB of LDS for the cross-wave sums from 22 blocks up
#define DWWIDE(I, J)                                                                                                          \
  if (ti == I && tj == J) {                                                                                                   \
    if (nt == -1) {                                                                                                           \
      if (lds > 64 * 1024) MM_HIP(hipFuncSetAttribute((const void*)k_dw_direct_s3<I, J, 1, __bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((k_dw_direct_s3<I, J, 1, __bf16>), dim3(nb, ny), dim3(256), lds, s, (const __bf16*)in, ld_in,        \
                         (const __bf16*)dout, ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                            \
    } else if (nt == -2) {                                                                                                    \
      if (lds > 64 * 1024) MM_HIP(hipFuncSetAttribute((const void*)k_dw_direct_s3<I, J, 1, _Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((k_dw_direct_s3<I, J, 1, _Float16>), dim3(nb, ny), dim3(256), lds, s, (const _Float16*)in, ld_in,    \
                         (const _Float16*)dout, ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                          \
    } else if (nt == 3) {                                                                                                     \
      if (lds > 64 * 1024) MM_HIP(hipFuncSetAttribute((const void*)k_dw_direct_s3<I, J, 3, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((k_dw_direct_s3<I, J, 3>), dim3(nb, ny), dim3(256), lds, s, (const float*)in, ld_in, (const float*)dout, \
                         ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                                                 \
    } else {                                                                                                                  \
      if (lds > 64 * 1024) MM_HIP(hipFuncSetAttribute((const void*)k_dw_direct_s3<I, J, 2, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((k_dw_direct_s3<I, J, 2>), dim3(nb, ny), dim3(256), lds, s, (const float*)in, ld_in, (const float*)dout, \
                         ld_do, src, dst, Cin, Cout, K, seg, chunk, partial);                                                 \
    }                                                                                                                         \
  }
    DWWIDE(5, 1) DWWIDE(5, 2) DWWIDE(5, 3) DWWIDE(5, 4) DWWIDE(5, 5) DWWIDE(6, 1) DWWIDE(6, 2) DWWIDE(6, 3) DWWIDE(6, 4)
    DWWIDE(1, 5) DWWIDE(2, 5) DWWIDE(3, 5) DWWIDE(4, 5) DWWIDE(1, 6) DWWIDE(2, 6) DWWIDE(3, 6) DWWIDE(4, 6)
#undef DWWIDE
  } else {
    size_t lds = (size_t)64 * (Cin + Cout) * sizeof(float);
    MM_CHECK_ARG(lds <= 150 * 1024, "spconv_dw: channels too wide for the generic kernel");
    if (lds > 64 * 1024)
      MM_HIP(hipFuncSetAttribute((const void*)k_dw_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_dw_generic, dim3(nb), dim3(256), lds, s, (const float*)in, ld_in, (const float*)dout, ld_do, src, dst, Cin,
                       Cout, K, seg, partial);
  }
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // namespace

extern "C" {

size_t mm_spconv_dw_ws_bytes(const int32_t* offsets_host, int K, int Cin, int Cout) {
  KSeg seg;
  int nb = make_seg(offsets_host, K, dw_chunk(offsets_host[K], Cin, Cout, true), &seg);
  const int nb2 = make_seg(offsets_host, K, dw_chunk(offsets_host[K], Cin, Cout, false), &seg);  // plain-fp32 kernels: <= 4 x 4 tiles
  if (nb2 > nb) nb = nb2;
  return mm_align((size_t)(nb > 0 ? nb : 1) * Cin * Cout * sizeof(float)) + 256;
}

// dW[k][ci][co] (+)= sum over rules r of bucket k of in[src[r]][ci] * dout[dst[r]][co]
int mm_spconv_dw(const float* in, int ld_in, int Cin, const float* dout, int ld_do, int Cout, const int32_t* src,
                 const int32_t* dst, const int32_t* offsets_host, int K, float* dW, int accumulate, int mode, void* ws,
                 size_t ws_bytes, hipStream_t s) {
  KSeg seg;
  int rc = dw_partial(0, mode, in, ld_in, Cin, dout, ld_do, Cout, src, dst, offsets_host, K, ws, ws_bytes, &seg, s);
  if (rc != MM_OK) return rc;
  const int ne = Cin * Cout;
  hipLaunchKernelGGL(k_dw_reduce, dim3((unsigned)mm_cdiv(ne, 32), K), dim3(256), 0, s, (const float*)ws, ne, K, seg, dW, accumulate);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// 16-bit activation mode: in / dout are bf16 rows (ld in elements), dW stays fp32.  Cin, Cout multiples of 16.
int mm_spconv_dw_bf16(const void* in, int ld_in, int Cin, const void* dout, int ld_do, int Cout, const int32_t* src,
                      const int32_t* dst, const int32_t* offsets_host, int K, float* dW, int accumulate, int mode, void* ws,
                      size_t ws_bytes, hipStream_t s) {
  KSeg seg;
  int rc = dw_partial(1, mode, in, ld_in, Cin, dout, ld_do, Cout, src, dst, offsets_host, K, ws, ws_bytes, &seg, s);
  if (rc != MM_OK) return rc;
  const int ne = Cin * Cout;
  hipLaunchKernelGGL(k_dw_reduce, dim3((unsigned)mm_cdiv(ne, 32), K), dim3(256), 0, s, (const float*)ws, ne, K, seg, dW, accumulate);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// the same over IEEE fp16 rows
int mm_spconv_dw_f16(const void* in, int ld_in, int Cin, const void* dout, int ld_do, int Cout, const int32_t* src,
                     const int32_t* dst, const int32_t* offsets_host, int K, float* dW, int accumulate, int mode, void* ws,
                     size_t ws_bytes, hipStream_t s) {
  KSeg seg;
  int rc = dw_partial(2, mode, in, ld_in, Cin, dout, ld_do, Cout, src, dst, offsets_host, K, ws, ws_bytes, &seg, s);
  if (rc != MM_OK) return rc;
  const int ne = Cin * Cout;
  hipLaunchKernelGGL(k_dw_reduce, dim3((unsigned)mm_cdiv(ne, 32), K), dim3(256), 0, s, (const float*)ws, ne, K, seg, dW, accumulate);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// The weight gradient in two calls (round 3): the partial slabs of a layer now, the slab sums of EVERY layer of a backward
// pass later in one launch (mm_spconv_dw_reduce_batch) - 26 small reduce launches and as many dependent-launch gaps per step
// become one.  ``partial`` (mm_spconv_dw_ws_bytes) must stay untouched until the batched reduce has run; ``blk_start_host``
// receives the MAXK + 1 = 33 slab offsets per kernel offset that the reduce needs (a row of its descriptor table).
// bf16 = 1 / 2: in / dout are bf16 / IEEE fp16 rows.  Same kernels, same slabs, same summation order as mm_spconv_dw: bit-identical.
int mm_spconv_dw_partial(int bf16, const void* in, int ld_in, int Cin, const void* dout, int ld_do, int Cout, const int32_t* src,
                         const int32_t* dst, const int32_t* offsets_host, int K, int mode, void* partial, size_t partial_bytes,
                         int32_t* blk_start_host, hipStream_t s) {
  MM_CHECK_ARG(blk_start_host != nullptr, "spconv_dw_partial: no descriptor row");
  KSeg seg;
  int rc = dw_partial(bf16, mode, in, ld_in, Cin, dout, ld_do, Cout, src, dst, offsets_host, K, partial, partial_bytes, &seg, s);
  if (rc != MM_OK) return rc;
  for (int k = 0; k <= MAXK; k++) blk_start_host[k] = seg.blk_start[k];
  return MM_OK;
}

int mm_spconv_dw_desc_bytes(void) { return (int)sizeof(DwRedD); }
int64_t mm_spconv_dw_reduce_blocks(int ne, int K) { return (int64_t)dw_red_blocks_k(ne) * K; }

// descs (device): n rows of DwRedD {partial, dW, ne, K, accumulate, blk_first, blk_start[33]}; blk_first = prefix sum of
// mm_spconv_dw_reduce_blocks(ne, K) over the preceding rows, total_blocks = the sum over all rows.
int mm_spconv_dw_reduce_batch(const void* descs_dev, int n, int64_t total_blocks, hipStream_t s) {
  MM_CHECK_ARG(n >= 0 && total_blocks >= 0 && total_blocks < (1ll << 31), "spconv_dw_reduce_batch: bad table");
  if (n == 0 || total_blocks == 0) return MM_OK;
  hipLaunchKernelGGL(k_dw_reduce_batch, dim3((unsigned)total_blocks), dim3(256), 0, s, (const DwRedD*)descs_dev, n);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
