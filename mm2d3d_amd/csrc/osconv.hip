// Engine F: output-stationary sparse convolution for gfx950 (Appendix A.2-A.4, A.8 iv).
//
// Replaces the gather-GEMM -> tmp[rule] -> CSR-reduce pair of spconv.hip for SubmanifoldConvolution / Convolution /
// Deconvolution forward and data gradients (reference call sites scn_unet.py:43,45,52,68-70,75-77,114).
//
// The destination rows of a level are ordered by their neighbour bitmask (csrc/ostable.hip) and cut into tiles of
// NW*16 rows.  One workgroup (NW waves) owns a tile; wave w owns one 16-row sub-block and keeps its output
//   D[co][row] (v_mfma_f32_16x16x32_bf16, lane = 4 consecutive output channels of one row)
// in registers from the first offset k to the last.  Offsets are visited in ascending k (canonical order A.8 iv) and only
// those present in the tile's mask; a sub-block none of whose 16 rows has neighbour k skips the k altogether.  Per
// (k, chunk of up to three 32-input-channel blocks) the workgroup stages the packed weight fragments once in LDS (double buffered, one barrier
// per step) and every wave gathers its rows' input channels straight into the MFMA B operand (two 16-B loads per lane).
// No tmp[rule x Cout] round trip, no reduction kernel, no destination indices per rule: HBM sees the gathered input rows,
// 4 B of neighbour index per (row, present k), and one store per output row.
//
// Arithmetic: every fp32 operand is split into three bf16 terms that together carry its 24-bit significand
// (x = x1 + x2 + x3); x.w is accumulated in fp32 from the six partial products down to 2^-18 |x.w|
// (x3w1, x1w3, x2w2, x2w1, x1w2, x1w1 - the same products and order as the split engines of spconv.hip): fp32-faithful.
// The weights are split once per optimiser step by k_os_pack (one batched launch for every layer of the net).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int FRAG_B = 3072;   // bytes of one (k, q, cb) fragment block: 3 terms x 64 lanes x 16 B

__device__ __attribute__((aligned(32))) const float g_zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

struct OsP {
  const float* in;
  float* out;
  const u32x4* Wf;  // [K][nq][ncb_tot][3 terms][64 lanes] x 16 B
  const int32_t* dst;
  const int32_t* nbrp;
  const uint32_t* tmask;
  int64_t npad;
  int ld_in, Cin, ld_out, nq, ncb_tot;
  int cb_first;  // first output-channel block of this launch (k-parallel form: one launch per group of <= 4 blocks)
  int dbg;  // bring-up switches (MM_OS_DBG): 1 = no MFMA, 2 = no row gathers (row 0), 4 = no split
};

struct Cur {  // (k, q) cursor over the present offsets of a tile; wave-uniform
  uint32_t m;
  int k, q;
};
__device__ inline void cur_init(Cur& c, uint32_t m) {
  c.m = m;
  c.k = m ? __ffs(m) - 1 : -1;
  c.q = 0;
}
__device__ inline void cur_next(Cur& c, int nq) {
  if (c.k < 0) return;
  if (++c.q == nq) {
    c.q = 0;
    c.m &= c.m - 1;
    c.k = c.m ? __ffs(c.m) - 1 : -1;
  }
}

__device__ inline void split3(const f32x4& a, const f32x4& b, bf16x8 (&t)[3]) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float r = v[i];
#pragma unroll
    for (int n = 0; n < 3; n++) {
      const __bf16 h = (__bf16)r;
      t[n][i] = h;
      r -= (float)h;
    }
  }
}

template <int NCB, int QS, int NW>
__global__ __launch_bounds__(NW * 64) void k_osconv(OsP p) {
  constexpr int MT = NW * 16;                       // rows per workgroup tile: one 16-row MFMA sub-block per wave
  constexpr int QPC = NCB * (FRAG_B / 16);          // 16-B pieces per input-channel block of a staged chunk
  constexpr int PIECES = QS * QPC;                  // 16-B pieces of one staged W chunk (QS input-channel blocks)
  constexpr int NP = (PIECES + NW * 64 - 1) / (NW * 64);
  constexpr int LBUF = NP * NW * 64 * 16;           // bytes of one (padded) LDS weight buffer
  extern __shared__ __attribute__((aligned(16))) char lds[];  // 2 x LBUF
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 15, sl = lane >> 4;
  // tiles are sorted by neighbour mask, the heaviest (most offsets) last: dispatch them first so the light ones fill the tail
  const int t = gridDim.x - 1 - blockIdx.x, cb0 = blockIdx.y * NCB;
  const int nq = p.nq, nqc = (nq + QS - 1) / QS;    // an item = (offset k, chunk of QS input-channel blocks)
  const uint32_t tm = p.tmask[t];
  const int64_t j = (int64_t)t * MT + wave * 16 + rl;

  f32x4 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; cb++) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // cursors: item being multiplied (c0), two ahead (cW: weights), G ahead (cG: row gathers), G+D ahead (cI: neighbour ids)
  constexpr int G = QS == 1 ? 3 : 1, D = 3;  // fat steps are long: one step ahead is enough for the rows
  Cur c0, cW, cG, cI;
  cur_init(c0, tm);
  cW = c0;
  cur_next(cW, nqc);
  cur_next(cW, nqc);
  cG = c0;
#pragma unroll
  for (int i = 0; i < G; i++) cur_next(cG, nqc);
  cI = cG;
#pragma unroll
  for (int i = 0; i < D; i++) cur_next(cI, nqc);

  // Every load below is UNCONDITIONAL (clamped addresses, values selected afterwards): conditional loads make hipcc branch
  // around each of them and wait vmcnt(0) in between, which drains the whole software pipeline.
  auto load_id = [&](const Cur& c) {
    const int kk = c.k >= 0 ? c.k : 0;
    const int v = p.nbrp[(int64_t)kk * p.npad + j];
    return c.k >= 0 ? v : -1;
  };
  auto load_w = [&](const Cur& c, u32x4 (&w)[NP]) {
    const int kk = c.k >= 0 ? c.k : 0;
    const u32x4* src = p.Wf + (((int64_t)kk * nq + c.q * QS) * p.ncb_tot + cb0) * (FRAG_B / 16);
#pragma unroll
    for (int i = 0; i < NP; i++) {
      int e = tid + i * NW * 64;
      e = e < PIECES ? e : PIECES - 1;
      int qq = e / QPC;  // compile-time divisor
      const int r = e - qq * QPC;
      qq = c.q * QS + qq < nq ? qq : 0;  // last chunk of an offset: the missing blocks re-read block 0 (never multiplied)
      w[i] = src[(int64_t)qq * p.ncb_tot * (FRAG_B / 16) + r];
    }
  };
  auto store_w = [&](int buf, const u32x4 (&w)[NP]) {  // the LDS buffers are padded to NP * NW * 64 pieces
    u32x4* d = (u32x4*)(lds + buf * LBUF);
#pragma unroll
    for (int i = 0; i < NP; i++) d[tid + i * NW * 64] = w[i];
  };
  auto gather = [&](const Cur& c, int id, f32x4 (&x)[QS][2], bool& pr) {
    pr = __ballot(id >= 0) != 0ull;
    const float* row = p.in + (int64_t)((id >= 0 && !(p.dbg & 2)) ? id : 0) * p.ld_in;
#pragma unroll
    for (int qq = 0; qq < QS; qq++) {
      const int ci = (c.q * QS + qq) * 32 + sl * 8;
      const bool ok = id >= 0 && ci < p.Cin;
      const float* r = row + (ci < p.Cin ? ci : 0);
      const f32x4 a = *(const f32x4*)r, b = *(const f32x4*)(r + 4);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      x[qq][0] = ok ? a : z;
      x[qq][1] = ok ? b : z;
    }
  };

  // Software pipeline over the items of the tile.  At step i: the neighbour id of item i+G+D is requested, the row of item
  // i+G is gathered (id requested D steps earlier), the weights of item i+2 are requested, item i is multiplied and the
  // weights of item i+1 are published in the other LDS buffer behind one barrier.  Row gathers have G steps to land (HBM /
  // Infinity Cache), the L2-resident weights two.  The loop body holds four steps with the register sets rotated at compile
  // time, so nothing is copied and every wait is counted (s_waitcnt vmcnt(N)).
  f32x4 x[G + 1][QS][2];
  bool pr[G + 1];
  int ids[D + 1];       // ring: the id of item j lives in slot (j - G) mod (D+1)
  u32x4 wreg[2][NP];    // weights of items i+1 / i+2 on their way to LDS
  {
    Cur c = c0;
#pragma unroll
    for (int g = 0; g < G; g++) {  // prologue: items 0 .. G-1
      const int id0 = load_id(c);
      gather(c, id0, x[g], pr[g]);
      cur_next(c, nqc);
    }
    c = cG;
#pragma unroll
    for (int d = 0; d < D; d++) {  // ids of items G .. G+D-1
      ids[d] = load_id(c);
      cur_next(c, nqc);
    }
    load_w(c0, wreg[0]);
    store_w(0, wreg[0]);
    Cur c1 = c0;
    cur_next(c1, nqc);
    load_w(c1, wreg[1]);
  }
  __syncthreads();
  int buf = 0;
  auto step = [&](auto slot) {
    constexpr int S = decltype(slot)::value;  // i mod 4
    constexpr int SC = S % (G + 1);           // ring slot of item i
    constexpr int SG = (S + G) % (G + 1);     // ring slot of item i+G
    ids[(S + D) % (D + 1)] = load_id(cI);     // item i+G+D
    load_w(cW, wreg[S & 1]);                  // item i+2; wreg[(S+1)&1] holds item i+1 (requested one step ago)
    gather(cG, ids[S % (D + 1)], x[SG], pr[SG]);
    __builtin_amdgcn_sched_barrier(0);
    if (pr[SC] && !(p.dbg & 1)) {  // none of this wave's 16 rows has neighbour k: only the staging and the barrier
      const char* wb = lds + buf * LBUF + lane * 16;
#pragma unroll
      for (int qq = 0; qq < QS; qq++) {
        if (c0.q * QS + qq < nq) {
          bf16x8 xt[3];
          if (p.dbg & 4) {
#pragma unroll
            for (int n = 0; n < 3; n++) xt[n] = __builtin_bit_cast(bf16x8, n & 1 ? x[SC][qq][1] : x[SC][qq][0]);
          } else {
            split3(x[SC][qq][0], x[SC][qq][1], xt);
          }
#pragma unroll
          for (int cb = 0; cb < NCB; cb++) {
            bf16x8 wt[3];
#pragma unroll
            for (int n = 0; n < 3; n++) wt[n] = *(const bf16x8*)(wb + ((qq * NCB + cb) * 3 + n) * 1024);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[2], xt[0], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[0], xt[2], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[1], xt[1], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[1], xt[0], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[0], xt[1], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[0], xt[0], acc[cb], 0, 0, 0);
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    store_w(buf ^ 1, wreg[(S + 1) & 1]);
    __syncthreads();
    buf ^= 1;
    cur_next(c0, nqc);
    cur_next(cW, nqc);
    cur_next(cG, nqc);
    cur_next(cI, nqc);
  };
  static_assert(4 % (G + 1) == 0 && D == 3, "the loop body below is unrolled for rings that divide 4");
  while (c0.k >= 0) {
    step(std::integral_constant<int, 0>{});
    if (c0.k < 0) break;
    step(std::integral_constant<int, 1>{});
    if (c0.k < 0) break;
    step(std::integral_constant<int, 2>{});
    if (c0.k < 0) break;
    step(std::integral_constant<int, 3>{});
  }
  const int d = p.dst[j];
  if (d >= 0) {
    float* o = p.out + (int64_t)d * p.ld_out + cb0 * 16 + sl * 4;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) *(f32x4*)(o + cb * 16) = acc[cb];
  }
}

// ---------------------------------------------------------------------------------------------- engine F, k-parallel form
// The pipelined kernel above walks the offsets of a tile one after the other: a tile whose rows have all 27 neighbours is a
// serial chain of 27 x nq barrier-separated steps, and such tiles (the dense cores of a scene, sorted to the end) set the
// kernel time.  Here the four waves of a workgroup (tile = 64 rows = 4 sub-blocks) take DIFFERENT offsets at the same time:
// in round r wave w multiplies offset k = (4r + w)-th present offset of the tile for all four sub-blocks, with the W[k]
// fragments read straight from L2 into registers (one wave uses them for the whole tile, so nothing is staged or shared),
// then publishes its four partial sub-block results in LDS; after one barrier wave w adds the four partials of sub-block w
// in wave order = ascending k (canonical order A.8 iv; bit-identical to "tmp[rule] then CSR reduce") into its final
// accumulators.  Two barriers per FOUR offsets and no dependent-latency chain longer than one offset's q loop.
// BF: the 16-bit activation mode - rows are bf16 (in and out), the weights one bf16 term per
// element ([K][nq][ncb][64 lanes] x 16 B fragments), one MFMA per block and no operand splitting; accumulation stays fp32.
template <int NCB, bool BF>
__global__ __launch_bounds__(256) void k_osconv4(OsP p) {
  constexpr int NW = 4, MT = 64;
  constexpr int NTW = BF ? 1 : 3;             // bf16 terms per weight
  constexpr int FRB = NTW * 1024 / 16;        // 16-B pieces per (k, q, cb) fragment block
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [NW producers][4 sub-blocks][NCB][64 lanes] f32x4
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 15, sl = lane >> 4;
  const int t = gridDim.x - 1 - blockIdx.x, cb0 = p.cb_first;
  const int nq = p.nq;
  const int64_t j0 = (int64_t)t * MT + rl;
  f32x4 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; cb++) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint32_t m = p.tmask[t];
  while (m) {
    uint32_t mm = m;
    for (int i = 0; i < wave; i++) mm &= mm - 1;
    const int myk = mm ? __ffs(mm) - 1 : -1;  // wave-uniform
#pragma unroll
    for (int i = 0; i < NW; i++) m &= m - 1;
    f32x4 accP[4][NCB];
#pragma unroll
    for (int sb = 0; sb < 4; sb++)
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) accP[sb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (myk >= 0) {
      int ids[4];
      bool pres[4];
#pragma unroll
      for (int sb = 0; sb < 4; sb++) ids[sb] = p.nbrp[(int64_t)myk * p.npad + j0 + sb * 16];
#pragma unroll
      for (int sb = 0; sb < 4; sb++) pres[sb] = __ballot(ids[sb] >= 0) != 0ull;
      const u32x4* wk = p.Wf + ((int64_t)myk * nq * p.ncb_tot + cb0) * FRB + lane;
      // rows without this neighbour read a zero line instead (pointer select: cheaper than zeroing eight loaded values)
      constexpr int ES = BF ? 2 : 4;  // bytes per input element
      const char* rows[4];
#pragma unroll
      for (int sb = 0; sb < 4; sb++)
        rows[sb] = ids[sb] >= 0 ? (const char*)p.in + ((int64_t)ids[sb] * p.ld_in + sl * 8) * ES : nullptr;
      for (int q = 0; q < nq; q++) {
        bf16x8 wt[NCB][NTW];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
          for (int n = 0; n < NTW; n++) wt[cb][n] = __builtin_bit_cast(bf16x8, wk[((int64_t)q * p.ncb_tot + cb) * FRB + n * 64]);
        const bool in_c = q * 32 + sl * 8 < p.Cin;  // Cin % 16 == 0: the 8 channels are inside or outside together
        f32x4 x[4][BF ? 1 : 2];
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {  // unconditional loads
          const char* r = (rows[sb] && in_c) ? rows[sb] + q * 32 * ES : (const char*)g_zero8;
          x[sb][0] = *(const f32x4*)r;
          if (!BF) x[sb][BF ? 0 : 1] = *(const f32x4*)(r + 16);
        }
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
          if (pres[sb]) {
            if constexpr (BF) {
              const bf16x8 xb = __builtin_bit_cast(bf16x8, x[sb][0]);
#pragma unroll
              for (int cb = 0; cb < NCB; cb++)
                accP[sb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[cb][0], xb, accP[sb][cb], 0, 0, 0);
              continue;
            }
            bf16x8 xt[3];
            split3(x[sb][0], x[sb][BF ? 0 : 1], xt);
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
              f32x4 a = accP[sb][cb];
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[cb][NTW - 1], xt[0], a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[cb][0], xt[2], a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[cb][NTW / 2], xt[1], a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[cb][NTW / 2], xt[0], a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[cb][0], xt[1], a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[cb][0], xt[0], a, 0, 0, 0);
              accP[sb][cb] = a;
            }
          }
        }
      }
    }
    f32x4* slot = (f32x4*)lds;
#pragma unroll
    for (int sb = 0; sb < 4; sb++)
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) slot[((wave * 4 + sb) * NCB + cb) * 64 + lane] = accP[sb][cb];
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; w++)
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) acc[cb] += slot[((w * 4 + wave) * NCB + cb) * 64 + lane];
    __syncthreads();
  }
  const int d = p.dst[j0 + wave * 16];
  if (d >= 0) {
    if constexpr (BF) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      __bf16* o = (__bf16*)p.out + (int64_t)d * p.ld_out + cb0 * 16 + sl * 4;
#pragma unroll
      for (int cb = 0; cb < NCB; cb++)
        *(bf16x4*)(o + cb * 16) = bf16x4{(__bf16)acc[cb].x, (__bf16)acc[cb].y, (__bf16)acc[cb].z, (__bf16)acc[cb].w};
    } else {
      float* o = p.out + (int64_t)d * p.ld_out + cb0 * 16 + sl * 4;
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) *(f32x4*)(o + cb * 16) = acc[cb];
    }
  }
}

// ---------------------------------------------------------------------------------------------- weight fragments
// Wf[k][q][cb][term][lane][j] = bf16 term of W[kk][ci = 32q + 8(lane>>4) + j][co = 16cb + (lane&15)]   (0 beyond Cin / Cout)
// weight element (k, ci, co) is W[kk*w_kstride + ci*s_ci + co*s_co], kk = kflip ? K-1-k : k
struct PackD {  // 12 x int64: the layout of the device descriptor table of mm_spconv_os_pack_batch
  int64_t W, Wf, K, Cin, Cout, nq, ncb, w_kstride, s_ci, s_co, kflip, blk_end;
};

template <int NT>
__device__ inline void pack_one(const PackD& d, int64_t e) {
  const int64_t total = d.K * d.nq * d.ncb * 512;
  if (e >= total) return;
  const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
  int64_t t = e >> 9;
  const int cb = (int)(t % d.ncb);
  t /= d.ncb;
  const int q = (int)(t % d.nq), k = (int)(t / d.nq);
  const int ci = 32 * q + 8 * (lane >> 4) + j, co = 16 * cb + (lane & 15);
  float r = 0.f;
  if (ci < d.Cin && co < d.Cout)
    r = ((const float*)d.W)[(d.kflip ? d.K - 1 - k : k) * d.w_kstride + (int64_t)ci * d.s_ci + (int64_t)co * d.s_co];
  __bf16* o = (__bf16*)d.Wf + (e >> 9) * (512 * NT) + lane * 8 + j;
#pragma unroll
  for (int n = 0; n < NT; n++) {
    const __bf16 h = (__bf16)r;
    o[n * 512] = h;
    r -= (float)h;
  }
}

template <int NT>
__global__ __launch_bounds__(256) void k_os_pack(PackD d) { pack_one<NT>(d, (int64_t)blockIdx.x * 256 + threadIdx.x); }

template <int NT>
__global__ __launch_bounds__(256) void k_os_pack_batch(const PackD* __restrict__ descs, int n) {
  int i = 0;
  while (i + 1 < n && (int64_t)blockIdx.x >= descs[i].blk_end) i++;
  const PackD d = descs[i];
  const int64_t blk0 = i ? descs[i - 1].blk_end : 0;
  pack_one<NT>(d, ((int64_t)blockIdx.x - blk0) * 256 + threadIdx.x);
}

template <int NCB, int QS, int NW>
int launch_os(const OsP& p, int64_t n_tiles, int nchunk, hipStream_t s) {
  constexpr int PIECES = QS * NCB * (FRAG_B / 16);
  constexpr int LBUF = ((PIECES + NW * 64 - 1) / (NW * 64)) * NW * 64 * 16;
  if (2 * LBUF > 64 * 1024)
    MM_HIP(hipFuncSetAttribute((const void*)k_osconv<NCB, QS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LBUF));
  hipLaunchKernelGGL((k_osconv<NCB, QS, NW>), dim3((unsigned)n_tiles, nchunk), dim3(NW * 64), 2 * LBUF, s, p);
  return MM_OK;
}

// QS input-channel blocks per barrier: as many as the offset has, while the staged chunk stays <= 36 KB (QS * NCB <= 12)
template <int NCB, int NW>
int dispatch_qs(const OsP& p, int64_t n_tiles, int nchunk, hipStream_t s) {
  constexpr int QMAX = 12 / NCB >= 3 ? 3 : 12 / NCB >= 2 ? 2 : 1;
  const int qs = p.nq < QMAX ? p.nq : QMAX;
  if (QMAX >= 3 && qs == 3) return launch_os<NCB, (QMAX >= 3 ? 3 : 1), NW>(p, n_tiles, nchunk, s);
  if (QMAX >= 2 && qs == 2) return launch_os<NCB, (QMAX >= 2 ? 2 : 1), NW>(p, n_tiles, nchunk, s);
  return launch_os<NCB, 1, NW>(p, n_tiles, nchunk, s);
}

template <int NW>
int dispatch_os(int ncbw, const OsP& p, int64_t n_tiles, int nchunk, hipStream_t s) {
  switch (ncbw) {
    case 1: return dispatch_qs<1, NW>(p, n_tiles, nchunk, s);
    case 2: return dispatch_qs<2, NW>(p, n_tiles, nchunk, s);
    case 3: return dispatch_qs<3, NW>(p, n_tiles, nchunk, s);
    case 4: return dispatch_qs<4, NW>(p, n_tiles, nchunk, s);
    case 5: return dispatch_qs<5, NW>(p, n_tiles, nchunk, s);
    case 6: return dispatch_qs<6, NW>(p, n_tiles, nchunk, s);
    case 7: return dispatch_qs<7, NW>(p, n_tiles, nchunk, s);
    case 8: return dispatch_qs<8, NW>(p, n_tiles, nchunk, s);
  }
  mm_set_error("spconv_os_apply: unsupported channel-block count %d", ncbw);
  return MM_ERR_UNSUPPORTED;
}

template <int NCB>
int launch_os4(const OsP& p, int64_t n_tiles, int nchunk, int bf, hipStream_t s) {
  constexpr int LDSB = 4 * 4 * NCB * 1024;
  if (bf) hipLaunchKernelGGL((k_osconv4<NCB, true>), dim3((unsigned)n_tiles, nchunk), dim3(256), LDSB, s, p);
  else hipLaunchKernelGGL((k_osconv4<NCB, false>), dim3((unsigned)n_tiles, nchunk), dim3(256), LDSB, s, p);
  return MM_OK;
}

}  // namespace

extern "C" {

// bytes of the packed fragments of one weight tensor [K][Cin][Cout]
size_t mm_spconv_os_pack_bytes(int K, int Cin, int Cout) {
  return (size_t)K * ((Cin + 31) / 32) * ((Cout + 15) / 16) * FRAG_B;
}

// int64 fields of one descriptor of mm_spconv_os_pack_batch, blocks (256 threads) it needs
int mm_spconv_os_pack_desc_fields(void) { return (int)(sizeof(PackD) / sizeof(int64_t)); }
int64_t mm_spconv_os_pack_blocks(int K, int Cin, int Cout) {
  return mm_cdiv((int64_t)K * ((Cin + 31) / 32) * ((Cout + 15) / 16) * 512, 256);
}

int mm_spconv_os_pack(const float* W, int64_t w_kstride, int s_ci, int s_co, int kflip, int K, int Cin, int Cout, void* Wf,
                      hipStream_t s) {
  MM_CHECK_ARG(K > 0 && K <= 32 && Cin > 0 && Cout > 0 && W && Wf, "spconv_os_pack: bad arguments");
  PackD d{(int64_t)W, (int64_t)Wf, K, Cin, Cout, (Cin + 31) / 32, (Cout + 15) / 16, w_kstride, s_ci, s_co, kflip, 0};
  hipLaunchKernelGGL(k_os_pack<3>, dim3((unsigned)mm_spconv_os_pack_blocks(K, Cin, Cout)), dim3(256), 0, s, d);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// one bf16 term per weight (the 16-bit activation mode): [K][nq][ncb][64 lanes] x 16 B
size_t mm_spconv_os_pack_bytes_bf16(int K, int Cin, int Cout) {
  return (size_t)K * ((Cin + 31) / 32) * ((Cout + 15) / 16) * 1024;
}
int mm_spconv_os_pack_bf16(const float* W, int64_t w_kstride, int s_ci, int s_co, int kflip, int K, int Cin, int Cout, void* Wf,
                           hipStream_t s) {
  MM_CHECK_ARG(K > 0 && K <= 32 && Cin > 0 && Cout > 0 && W && Wf, "spconv_os_pack_bf16: bad arguments");
  PackD d{(int64_t)W, (int64_t)Wf, K, Cin, Cout, (Cin + 31) / 32, (Cout + 15) / 16, w_kstride, s_ci, s_co, kflip, 0};
  hipLaunchKernelGGL(k_os_pack<1>, dim3((unsigned)mm_spconv_os_pack_blocks(K, Cin, Cout)), dim3(256), 0, s, d);
  MM_LAUNCH_CHECK();
  return MM_OK;
}
int mm_spconv_os_pack_batch_bf16(const int64_t* descs_dev, int n_desc, int64_t total_blocks, hipStream_t s) {
  MM_CHECK_ARG(descs_dev && n_desc > 0 && total_blocks > 0, "spconv_os_pack_batch_bf16: bad arguments");
  hipLaunchKernelGGL(k_os_pack_batch<1>, dim3((unsigned)total_blocks), dim3(256), 0, s, (const PackD*)descs_dev, n_desc);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// descs_dev: [n_desc][12] int64 on the device = {W, Wf, K, Cin, Cout, nq, ncb, w_kstride, s_ci, s_co, kflip, blk_end}, blk_end =
// running sum of mm_spconv_os_pack_blocks; one launch packs every weight of the net
int mm_spconv_os_pack_batch(const int64_t* descs_dev, int n_desc, int64_t total_blocks, hipStream_t s) {
  MM_CHECK_ARG(descs_dev && n_desc > 0 && total_blocks > 0, "spconv_os_pack_batch: bad arguments");
  hipLaunchKernelGGL(k_os_pack_batch<3>, dim3((unsigned)total_blocks), dim3(256), 0, s, (const PackD*)descs_dev, n_desc);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// out[dst[j]] = sum over the present offsets k (ascending) of in[nbrp[k][j]] . W[k]    for every position j of the table
//   dst / nbrp / tmask: the tile table of mm_os_table_build (tile_rows in {64, 128, 256})
//   Wf: fragments of mm_spconv_os_pack(_batch) for (K, Cin, Cout); Cin, Cout multiples of 16; in / out 16-B aligned, ld % 4 == 0
}  // extern "C"

static int os_apply(int bf, const void* in_, int ld_in, int Cin, void* out_, int ld_out, int Cout, const void* Wf, int K,
                       const int32_t* dst, const int32_t* nbrp, const uint32_t* tmask, int64_t n_tiles, int tile_rows,
                       hipStream_t s) {
  const float* in = (const float*)in_;
  float* out = (float*)out_;
  const int ea = bf ? 8 : 4;  // elements per 16 bytes
  MM_CHECK_ARG(K > 0 && K <= 32 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0, "spconv_os_apply: channels must be multiples of 16");
  MM_CHECK_ARG(ld_in >= Cin && ld_out >= Cout && ld_in % ea == 0 && ld_out % 4 == 0 && (((uintptr_t)in | (uintptr_t)out | (uintptr_t)Wf) % 16) == 0,
               "spconv_os_apply: rows must be 16-B aligned");
  MM_CHECK_ARG(tile_rows == 64 || tile_rows == 128, "spconv_os_apply: tile_rows must be 64 or 128");
  if (n_tiles == 0) return MM_OK;
  const int ncb = Cout / 16;
  int nchunk = 1;
  while (ncb % nchunk != 0 || ncb / nchunk > 8) nchunk++;
  // small levels: split the output channels over more workgroups (each re-gathers its rows) until the grid covers the chip
  while (n_tiles * nchunk < 512 && (ncb / nchunk) % 2 == 0) nchunk *= 2;
  OsP p;
  p.in = in, p.out = out, p.Wf = (const u32x4*)Wf, p.dst = dst, p.nbrp = nbrp, p.tmask = tmask;
  p.cb_first = 0;
  p.dbg = getenv("MM_OS_DBG") ? atoi(getenv("MM_OS_DBG")) : 0;
  p.npad = n_tiles * tile_rows, p.ld_in = ld_in, p.Cin = Cin, p.ld_out = ld_out, p.nq = (Cin + 31) / 32, p.ncb_tot = ncb;
  int rc;
  static const int v3 = getenv("MM_OS_V3") ? atoi(getenv("MM_OS_V3")) : 0;
  MM_CHECK_ARG(!bf || tile_rows == 64, "spconv_os_apply_bf16: tile_rows must be 64");
  if (tile_rows == 64 && (!v3 || bf)) {  // k-parallel form: at most 4 output-channel blocks per workgroup (64 KB of LDS partials)
    // at most THREE blocks per launch: the four-block instance needs 64 accumulator + 48 fragment registers and loses the
    // occupancy that hides the gathers (64 output channels from 32: 184 us as one launch of four, 124 us as two of two)
    static const int maxw = getenv("MM_OS_MAXW") ? atoi(getenv("MM_OS_MAXW")) : 3;
    int parts = (ncb + maxw - 1) / maxw;
    // small levels: more, narrower launches (each re-gathers its rows) until the grid covers the chip
    while (n_tiles * parts < 1024 && parts < ncb && (ncb + parts) / (parts + 1) >= 2) parts++;
    rc = MM_OK;
    for (int i = 0, cb = 0; i < parts && rc == MM_OK; i++) {
      const int w = (ncb - cb + (parts - i) - 1) / (parts - i);  // near-equal parts, the wider ones first
      p.cb_first = cb;
      switch (w) {
        case 1: rc = launch_os4<1>(p, n_tiles, 1, bf, s); break;
        case 2: rc = launch_os4<2>(p, n_tiles, 1, bf, s); break;
        case 3: rc = launch_os4<3>(p, n_tiles, 1, bf, s); break;
        default: rc = launch_os4<4>(p, n_tiles, 1, bf, s); break;
      }
      cb += w;
    }
  } else if (tile_rows == 128) rc = dispatch_os<8>(ncb / nchunk, p, n_tiles, nchunk, s);
  else rc = dispatch_os<4>(ncb / nchunk, p, n_tiles, nchunk, s);
  if (rc) return rc;
  MM_LAUNCH_CHECK();
  return MM_OK;
}


extern "C" {

int mm_spconv_os_apply(const float* in, int ld_in, int Cin, float* out, int ld_out, int Cout, const void* Wf, int K,
                       const int32_t* dst, const int32_t* nbrp, const uint32_t* tmask, int64_t n_tiles, int tile_rows,
                       hipStream_t s) {
  return os_apply(0, in, ld_in, Cin, out, ld_out, Cout, Wf, K, dst, nbrp, tmask, n_tiles, tile_rows, s);
}
// 16-bit activation mode: in / out are bf16 rows (ld in elements, multiples of 8), Wf from mm_spconv_os_pack(_batch)_bf16,
// fp32 accumulation in ascending k, tile_rows = 64
int mm_spconv_os_apply_bf16(const void* in, int ld_in, int Cin, void* out, int ld_out, int Cout, const void* Wf, int K,
                            const int32_t* dst, const int32_t* nbrp, const uint32_t* tmask, int64_t n_tiles, int tile_rows,
                            hipStream_t s) {
  return os_apply(1, in, ld_in, Cin, out, ld_out, Cout, Wf, K, dst, nbrp, tmask, n_tiles, tile_rows, s);
}

}  // extern "C"
