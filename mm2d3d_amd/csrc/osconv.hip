// Engine F: output-stationary sparse convolution for gfx950 (Appendix A.2-A.4, A.8 iv).
//
// Replaces the gather-GEMM -> tmp[rule] -> CSR-reduce pair of spconv.hip for SubmanifoldConvolution / Convolution /
// Deconvolution forward and data gradients (reference call sites scn_unet.py:43,45,52,68-70,75-77,114).
//
// The destination rows of a level are ordered by their neighbour bitmask (csrc/ostable.hip) and cut into tiles of
// 64 rows.  One 4-wave workgroup owns a tile (four 16-row sub-blocks); the waves take DIFFERENT present offsets k at the same
// time, each multiplies its k for all four sub-blocks,
//   D[co][row] (v_mfma_f32_16x16x32_bf16, lane = 4 consecutive output channels of one row),
// with the W[k] fragments straight from L2 and the rows' input channels gathered into the MFMA B operand (two 16-B loads per
// lane); the partial results meet in LDS and are added in ascending k (canonical order A.8 iv) - see k_osconv4 below.  Only
// the offsets present in the tile's mask are visited.
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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
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
};

__device__ inline void split3(const f32x4& a, const f32x4& b, bf16x8 (&t)[3]) {
#ifdef MM_DIAG_FAKESPLIT  // diagnostic build only (tools/diag_lib.sh): no VALU split - what operands stored pre-split would cost
  t[0] = __builtin_bit_cast(bf16x8, a), t[1] = __builtin_bit_cast(bf16x8, b), t[2] = t[0];
  return;
#endif
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

// ---------------------------------------------------------------------------------------------- engine F, k-parallel form
// (An earlier LDS-staged form walked the offsets of a tile one after the other: a tile whose rows have all 27 neighbours was
// a serial chain of 27 x nq barrier-separated steps, and such tiles - the dense cores of a scene - set the kernel time.)
// The four waves of a workgroup (tile = 64 rows = 4 sub-blocks) take DIFFERENT offsets at the same time:
// in round r wave w multiplies offset k = (4r + w)-th present offset of the tile for all four sub-blocks, with the W[k]
// fragments read straight from L2 into registers (one wave uses them for the whole tile, so nothing is staged or shared),
// then publishes its four partial sub-block results in LDS; after one barrier wave w adds the four partials of sub-block w
// in wave order = ascending k (canonical order A.8 iv; bit-identical to "tmp[rule] then CSR reduce") into its final
// accumulators.  Two barriers per FOUR offsets and no dependent-latency chain longer than one offset's q loop.
// MODE 1 / 2: the 16-bit activation modes - rows are bf16 / IEEE fp16 (in and out), the weights one
// 16-bit term per element ([K][nq][ncb][64 lanes] x 16 B fragments), one MFMA per block (v_mfma_f32_16x16x32_bf16 / _f16) and no
// operand splitting; accumulation stays fp32.  MODE 0: fp32 rows, three-term split.
// NW (round 6): waves per tile.  4 everywhere until now; a tile with all 27 offsets is then a chain of 7 rounds of (neighbour ids ->
// row gathers -> MFMAs -> LDS exchange), and on the small deep levels - a few hundred tiles for 256 CUs - that chain, not bandwidth, is
// the layer's time (53 us for 34 MB at 112 -> 112 channels).  There 8 or 16 waves share a tile's offsets: 4 or 2 rounds.
template <int NCB, int MODE, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_osconv4(OsP p) {
  constexpr bool BF = MODE != 0;
  constexpr int MT = 64;
  constexpr int NTW = BF ? 1 : 3;             // bf16 terms per weight
  constexpr int FRB = NTW * 1024 / 16;        // 16-B pieces per (k, q, cb) fragment block
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [NW producers][4 sub-blocks][NCB][64 lanes] f32x4
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rl = lane & 15, sl = lane >> 4;
  const int t = gridDim.x - 1 - blockIdx.x, cb0 = p.cb_first + blockIdx.y * NCB;  // blockIdx.y: further groups of NCB output-channel blocks
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
      // (round 3) global address space kept through the pointer selects: a select between a row pointer and the zero line as
      // plain `const char*` made every row load a FLAT load (both address paths, both counters)
      typedef const __attribute__((address_space(1))) char* gptr;
      gptr rows[4];
#pragma unroll
      for (int sb = 0; sb < 4; sb++)
        rows[sb] = ids[sb] >= 0 ? (gptr)((const char*)p.in + ((int64_t)ids[sb] * p.ld_in + sl * 8) * ES) : (gptr) nullptr;
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
          gptr r = (rows[sb] && in_c) ? rows[sb] + q * 32 * ES : (gptr)(const char*)g_zero8;
          x[sb][0] = *(const __attribute__((address_space(1))) f32x4*)r;
          if (!BF) x[sb][BF ? 0 : 1] = *(const __attribute__((address_space(1))) f32x4*)(r + 16);
        }
#pragma unroll
        for (int sb = 0; sb < 4; sb++) {
          if (pres[sb]) {
            if constexpr (MODE == 1) {
              const bf16x8 xb = __builtin_bit_cast(bf16x8, x[sb][0]);
#pragma unroll
              for (int cb = 0; cb < NCB; cb++)
                accP[sb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[cb][0], xb, accP[sb][cb], 0, 0, 0);
              continue;
            }
            if constexpr (MODE == 2) {
              const f16x8 xb = __builtin_bit_cast(f16x8, x[sb][0]);
#pragma unroll
              for (int cb = 0; cb < NCB; cb++)
                accP[sb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wt[cb][0]), xb, accP[sb][cb], 0, 0, 0);
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
    if (wave < 4) {  // wave w adds the partials of sub-block w in wave order = ascending k
#pragma unroll
      for (int w = 0; w < NW; w++)
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb] += slot[((w * 4 + wave) * NCB + cb) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave >= 4) return;
  const int d = p.dst[j0 + wave * 16];
  if (d >= 0) {
    if constexpr (BF) {
      typedef std::conditional_t<MODE == 1, __bf16, _Float16> H;
      typedef H hx4 __attribute__((ext_vector_type(4)));
      H* o = (H*)p.out + (int64_t)d * p.ld_out + cb0 * 16 + sl * 4;
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) *(hx4*)(o + cb * 16) = hx4{(H)acc[cb].x, (H)acc[cb].y, (H)acc[cb].z, (H)acc[cb].w};
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

// thread = (fragment block, lane): the lane's eight elements (ci = 32q + 8(lane>>4) + j) are read (16 lanes = 64 contiguous
// bytes of a weight row when co is the fast axis), split, and stored as ONE 16-byte piece per term (the first version wrote
// 2-byte pieces 1 KB apart, one thread per element: 63 us per step for 16 MB of fragments; this form is bound by the 11 MB
// of weights it reads)
template <int NT, typename H = __bf16>
__device__ inline void pack_one(const PackD& d, int64_t e) {
  const int64_t total = d.K * d.nq * d.ncb * 64;
  if (e >= total) return;
  const int lane = (int)(e & 63);
  int64_t t = e >> 6;
  const int64_t blk = t;
  const int cb = (int)(t % d.ncb);
  t /= d.ncb;
  const int q = (int)(t % d.nq), k = (int)(t / d.nq);
  const int co = 16 * cb + (lane & 15);
  const float* wsrc = (const float*)d.W + (d.kflip ? d.K - 1 - k : k) * d.w_kstride + (int64_t)co * d.s_co;
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int ci = 32 * q + 8 * (lane >> 4) + j;
    r[j] = (ci < d.Cin && co < d.Cout) ? wsrc[(int64_t)ci * d.s_ci] : 0.f;
  }
  typedef H hx8 __attribute__((ext_vector_type(8)));
  hx8* o = (hx8*)((H*)d.Wf + blk * (512 * NT)) + lane;
#pragma unroll
  for (int n = 0; n < NT; n++) {
    hx8 h;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      h[j] = (H)r[j];
      r[j] -= (float)h[j];
    }
    o[n * 64] = h;
  }
}

template <int NT, typename H = __bf16>
__global__ __launch_bounds__(256) void k_os_pack(PackD d) { pack_one<NT, H>(d, (int64_t)blockIdx.x * 256 + threadIdx.x); }

template <int NT, typename H = __bf16>
__global__ __launch_bounds__(256) void k_os_pack_batch(const PackD* __restrict__ descs, int n) {
  int i = 0;
  while (i + 1 < n && (int64_t)blockIdx.x >= descs[i].blk_end) i++;
  const PackD d = descs[i];
  const int64_t blk0 = i ? descs[i - 1].blk_end : 0;
  pack_one<NT, H>(d, ((int64_t)blockIdx.x - blk0) * 256 + threadIdx.x);
}

template <int NCB, int NW>
int launch_os4w(const OsP& p, int64_t n_tiles, int nchunk, int bf, hipStream_t s) {
  constexpr int LDSB = NW * 4 * NCB * 1024;
  static_assert(LDSB <= 65536, "LDS partials of one workgroup");
  if (bf == 2) hipLaunchKernelGGL((k_osconv4<NCB, 2, NW>), dim3((unsigned)n_tiles, nchunk), dim3(64 * NW), LDSB, s, p);
  else if (bf) hipLaunchKernelGGL((k_osconv4<NCB, 1, NW>), dim3((unsigned)n_tiles, nchunk), dim3(64 * NW), LDSB, s, p);
  else hipLaunchKernelGGL((k_osconv4<NCB, 0, NW>), dim3((unsigned)n_tiles, nchunk), dim3(64 * NW), LDSB, s, p);
  return MM_OK;
}
// nw: waves per tile (4, or 8 / 16 on small levels: only the one- and two-block instances exist in those forms)
template <int NCB>
int launch_os4(const OsP& p, int64_t n_tiles, int nchunk, int bf, int nw, hipStream_t s) {
  if constexpr (NCB <= 2) {
    if (nw == 8) return launch_os4w<NCB, 8>(p, n_tiles, nchunk, bf, s);
  }
  if constexpr (NCB == 1) {
    if (nw == 16) return launch_os4w<NCB, 16>(p, n_tiles, nchunk, bf, s);
  }
  return launch_os4w<NCB, 4>(p, n_tiles, nchunk, bf, s);
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
  return mm_cdiv((int64_t)K * ((Cin + 31) / 32) * ((Cout + 15) / 16) * 64, 256);
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
// the same with one IEEE fp16 term per weight (fragment sizes as mm_spconv_os_pack_bytes_bf16)
int mm_spconv_os_pack_f16(const float* W, int64_t w_kstride, int s_ci, int s_co, int kflip, int K, int Cin, int Cout, void* Wf,
                          hipStream_t s) {
  MM_CHECK_ARG(K > 0 && K <= 32 && Cin > 0 && Cout > 0 && W && Wf, "spconv_os_pack_f16: bad arguments");
  PackD d{(int64_t)W, (int64_t)Wf, K, Cin, Cout, (Cin + 31) / 32, (Cout + 15) / 16, w_kstride, s_ci, s_co, kflip, 0};
  hipLaunchKernelGGL((k_os_pack<1, _Float16>), dim3((unsigned)mm_spconv_os_pack_blocks(K, Cin, Cout)), dim3(256), 0, s, d);
  MM_LAUNCH_CHECK();
  return MM_OK;
}
int mm_spconv_os_pack_batch_f16(const int64_t* descs_dev, int n_desc, int64_t total_blocks, hipStream_t s) {
  MM_CHECK_ARG(descs_dev && n_desc > 0 && total_blocks > 0, "spconv_os_pack_batch_f16: bad arguments");
  hipLaunchKernelGGL((k_os_pack_batch<1, _Float16>), dim3((unsigned)total_blocks), dim3(256), 0, s, (const PackD*)descs_dev, n_desc);
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
//   dst / nbrp / tmask: the tile table of mm_os_table_build (tile_rows = 64)
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
  MM_CHECK_ARG(tile_rows == 64, "spconv_os_apply: tile_rows must be 64");
  if (n_tiles == 0) return MM_OK;
  const int ncb = Cout / 16;
  OsP p;
  p.in = in, p.out = out, p.Wf = (const u32x4*)Wf, p.dst = dst, p.nbrp = nbrp, p.tmask = tmask;
  p.cb_first = 0;
  p.npad = n_tiles * tile_rows, p.ld_in = ld_in, p.Cin = Cin, p.ld_out = ld_out, p.nq = (Cin + 31) / 32, p.ncb_tot = ncb;
  // at most 4 output-channel blocks per workgroup (64 KB of LDS partials), and at most THREE per launch: the four-block
  // instance needs 64 accumulator + 48 fragment registers and loses the occupancy that hides the gathers (64 output channels
  // from 32: 184 us as one launch of four, 124 us as two of two)
  constexpr int maxw = 3;
  int parts = (ncb + maxw - 1) / maxw;
  // small levels: more, narrower launches (each re-gathers its rows) until the grid covers the chip
  while (n_tiles * parts < 1024 && parts < ncb && (ncb + parts) / (parts + 1) >= 2) parts++;
  // ... and more waves per tile (k_osconv4 NW): a level that cannot fill the chip with tiles is bound by the chain of rounds per tile
  int nw = 4;
  if (n_tiles * parts < 2048) {
    while (n_tiles * parts < 2048 && parts < ncb) parts++;  // one-block parts: the 16-wave form exists for them
    nw = (ncb + parts - 1) / parts == 1 ? (K > 8 ? 16 : 8) : ((ncb + parts - 1) / parts == 2 ? 8 : 4);
  }
  // Parts of equal width share ONE launch (blockIdx.y = part).  Round 6: as one launch per part a small level paid its latency chain
  // (7 rounds of neighbour ids -> row gathers -> MFMAs -> LDS exchange per tile, ~15 us however few tiles there are) once per part, one
  // after the other: 112 -> 112 channels on a 72k-rule level = 7 launches = 114 us for 34 MB (tools/sparse_layers.py, configs[4]:
  // the 16-bit mode runs every level on this engine).
  int rc = MM_OK;
  for (int i = 0, cb = 0; i < parts && rc == MM_OK;) {
    const int w = (ncb - cb + (parts - i) - 1) / (parts - i);  // near-equal parts, the wider ones first
    int cnt = 1, cbn = cb + w;
    while (i + cnt < parts && (ncb - cbn + (parts - i - cnt) - 1) / (parts - i - cnt) == w) cnt++, cbn += w;
    p.cb_first = cb;
    switch (w) {
      case 1: rc = launch_os4<1>(p, n_tiles, cnt, bf, nw, s); break;
      case 2: rc = launch_os4<2>(p, n_tiles, cnt, bf, nw, s); break;
      case 3: rc = launch_os4<3>(p, n_tiles, cnt, bf, nw, s); break;
      default: rc = launch_os4<4>(p, n_tiles, cnt, bf, nw, s); break;
    }
    cb = cbn;
    i += cnt;
  }
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
// the same over IEEE fp16 rows, Wf from mm_spconv_os_pack(_batch)_f16 (v_mfma_f32_16x16x32_f16)
int mm_spconv_os_apply_f16(const void* in, int ld_in, int Cin, void* out, int ld_out, int Cout, const void* Wf, int K,
                           const int32_t* dst, const int32_t* nbrp, const uint32_t* tmask, int64_t n_tiles, int tile_rows,
                           hipStream_t s) {
  return os_apply(2, in, ld_in, Cin, out, ld_out, Cout, Wf, K, dst, nbrp, tmask, n_tiles, tile_rows, s);
}

}  // extern "C"
