// Shared pieces of the single-launch batch-norm training kernels (csrc/bn2d.hip: NHWC bf16 maps; csrc/bn.hip: fp32 sparse rows).
// Included inside each translation unit: the device helpers are inline; the host-side state (barrier words, fault word, switches)
// lives in the caller's handle (common.h MMHandle, include/mm2d3d.h mm_create).
#pragma once

#include <algorithm>

#include "common.h"

namespace {
// Single-launch training kernels for maps that fit in the chip's registers + LDS (everything but the full-resolution maps).
//
// The three-kernel path above reads x twice in the forward pass and x, dy twice in the backward pass, and pays two kernel
// boundaries per call; on the 9..75 MB maps of the encoder/decoder it reaches 0.6-2.5 TB/s of its single-pass traffic.
// Here ONE workgroup per CU (512 threads) loads its rows once, keeps them (FUSED_NL rows per thread in LDS, the rest
// in VGPRs), publishes its fp64 partial sums, and meets the other workgroups at a grid barrier; the statistics of
// channel c are then combined by one wave (channels dealt over the workgroups, fixed order: bit-stable) and, after a second
// barrier, every workgroup normalises straight from its registers.  x is read once, dy once.
//
// Grid barrier: every workgroup must be resident at the same time, so the grid never exceeds the CU count and a
// workgroup needs more than half of a CU's registers/LDS (one per CU).  Another PROCESS running the same kernel on the
// same GPU can starve both grids: the wait is bounded (FUSED_TIMEOUT_TICKS of the 100 MHz wall clock); a barrier that runs
// out of time raises a fault word in pinned host memory, releases the grid and lets the launch finish with invalid outputs
// (no trap: the HIP context survives); the host polls the word (mm_fault_poll; the trainer does it every step and raises), which
// also switches the handle to the three-kernel path.  mm_set_option(h, MM_OPT_BN2D_FUSED, 0) selects that path up front.
// A kernel of another stream that holds LDS on some CUs makes the grid wait for it, and one that spin-waits across its own
// workgroups (decoupled look-back scan / Onesweep sort, an RCCL collective) can DEADLOCK with it - each holds CUs the other's
// missing workgroups need (tools/barrier_stress.py reproduces this with torch.cumsum on a second stream): the data-parallel trainer therefore takes the three-kernel path (ddp.py), and so do the directions that share the GPU
// with an optional second stream of the trainer (train.py: overlap_branches, overlap_metadata).
constexpr int FT = 512;       // 8 waves: 256 VGPRs per thread, and the per-thread constants are paid by half as many threads
constexpr int FUSED_NL = 13;  // rows per thread kept in LDS: 13 x 16 B x 512 threads = 104 KB
constexpr size_t FUSED_RED = (size_t)2 * FT * 8 * 4;  // float red[2][FT][8]
constexpr size_t FUSED_LDS = FUSED_RED + 2 * 8192 + (size_t)FUSED_NL * FT * 16;  // red | double red2[1024] | double out[1024] | rows
constexpr int FUSED_FLAG = 64;  // the release word sits 256 B after the arrival counter: pollers and arrivals on different lines
constexpr unsigned long long FUSED_TIMEOUT_TICKS = 1000000000ull;  // 10 s

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Where a grid barrier that ran out of time reports it: a word of pinned host memory owned by the handle the launch came
// through (MMHandle::fault_dev, passed with the kernel parameters).  The launch then runs out with invalid outputs instead of
// killing the HIP context with a trap; the host polls the word (mm_fault_poll), which switches the handle's single-launch kernels
// off and re-arms its barrier words.


__device__ inline double fused_wave_sum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;
}

// Values exchanged between workgroups INSIDE a launch go through agent-scope atomic loads / stores (sc1: they bypass the
// per-XCD L2's non-coherent lines) and a wait for the stores' acknowledgements, NOT through __threadfence(): an agent-scope
// release fence writes back the whole 4 MB L2 of the XCD, and with 256 workgroups doing that the barrier cost ~100 us.
template <typename V>
__device__ inline void xcd_store(V* p, V v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename V>
__device__ inline V xcd_load(const V* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void stores_acked() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Pins the 16 accumulators at this point of the instruction stream.  Instruction selection is free to sink pure arithmetic
// below every later load (only memory operations are ordered), and did: all rows' unpacked values then stay live until the end
// of the load phase.  An (empty) volatile asm is ordered with the loads and needs its inputs computed.
#define MM_PIN8(a, b) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]))
#define MM_PIN16(a, b)                                                                                                       \
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), \
               "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]))

// Sums of a[i] / b[i] over the row slots of the workgroup -> out[p] (LDS), pair p = q*C + c (q = 0: a, 1: b), p < 2C.
// Fixed order: FT/(2C) threads per pair stride the slots, their fp64 results are added in thread order.
template <int VEC>
__device__ inline void fused_block_sums(const float (&a)[VEC], const float (&b)[VEC], float* red, double* red2, double* out, int C, int CV,
                                        int rs, bool active) {
  const int tid = threadIdx.x;
  if (active) {
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      red[(size_t)tid * VEC + i] = a[i];
      red[((size_t)FT + tid) * VEC + i] = b[i];
    }
  }
  __syncthreads();
  const int np = 2 * C;
  if (np <= FT) {
    const int nparts = FT / np;
    const int pair = tid % np, part = tid / np;
    double acc = 0.0;
    if (part < nparts) {
      const int q = pair / C, c = pair - q * C;
      const float* src = red + ((size_t)q * FT + c / VEC) * VEC + c % VEC;
      for (int sl = part; sl < rs; sl += nparts) acc += (double)src[(size_t)sl * CV * VEC];
    }
    red2[tid] = acc;
    __syncthreads();
    if (tid < np) {
      double tot = 0.0;
      for (int j = 0; j < nparts; j++) tot += red2[j * np + tid];
      out[tid] = tot;
    }
  } else {
    for (int pair = tid; pair < np; pair += FT) {
      const int q = pair / C, c = pair - q * C;
      const float* src = red + ((size_t)q * FT + c / VEC) * VEC + c % VEC;
      double acc = 0.0;
      for (int sl = 0; sl < rs; sl++) acc += (double)src[(size_t)sl * CV * VEC];
      out[pair] = acc;
    }
  }
  __syncthreads();
}

// (sum over the workgroups [b0, b1) of partial[b][c], ... of partial[b][C + c]) by ONE wave: lanes stride the workgroups
// (at most 4 each, independent loads), then a fixed shuffle tree.  Result valid in lane 0.
__device__ inline void fused_wave_sums(const double* partial, int b0, int b1, int C, int c, double& s, double& q) {
  const int lane = threadIdx.x & 63;
  double vs[4], vq[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int bb = b0 + lane + 64 * j;
    const int bc = bb < b1 ? bb : b0;  // unconditional loads; masked below
    vs[j] = xcd_load(partial + (size_t)bc * 2 * C + c);
    vq[j] = xcd_load(partial + (size_t)bc * 2 * C + C + c);
    if (bb >= b1) vs[j] = 0.0, vq[j] = 0.0;
  }
  s = fused_wave_sum((vs[0] + vs[1]) + (vs[2] + vs[3]));
  q = fused_wave_sum((vq[0] + vq[1]) + (vq[2] + vq[3]));
}

// Grid barrier (every workgroup of the launch is resident, see the header), in two levels (round 6).  Agent-scope atomics execute at
// the memory side, one after the other per address (~12 ns each): 256 arrivals on ONE word are ~3 us of every barrier, and a
// single-launch batch norm has two of them, ~370 times per step.  The workgroups therefore arrive in eight groups (blockIdx & 7 - the
// blocks that share an XCD under round-robin dispatch; any grouping is correct) on eight counters, each on a line of its own; the last
// arrival of a group (its increment wraps the group's counter to 0) arrives at the top counter, and the last of those advances the
// release word from flag_old to flag_old + 1; everybody else polls the release word.  G = gridDim.x.
constexpr int FUSED_GROUP0 = 128, FUSED_GROUP_STRIDE = 32;  // group counters at sync[128 + 32 g], g = 0..7
__device__ inline void fused_barrier(unsigned* sync, unsigned* fault, unsigned G, unsigned flag_old) {
  stores_acked();   // this thread's xcd_store()s have reached the coherence point
  __syncthreads();  // ... and so have the whole workgroup's
  if (threadIdx.x == 0) {
    // wrapping increments: the n-th arrival (old value n - 1) sets its counter back to 0 in the same atomic
    const unsigned grp = blockIdx.x & 7u, ng = (G - grp + 7u) >> 3, ngroups = G < 8u ? G : 8u;
    bool last = atomicInc(&sync[FUSED_GROUP0 + FUSED_GROUP_STRIDE * grp], ng - 1u) == ng - 1u;
    if (last) last = atomicInc(&sync[0], ngroups - 1u) == ngroups - 1u;
    if (last) {
      xcd_store(&sync[FUSED_FLAG], flag_old + 1u);
    } else {
      const unsigned long long t0 = wall_clock64();
      while (xcd_load(&sync[FUSED_FLAG]) == flag_old) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > FUSED_TIMEOUT_TICKS) {
          // the grid is not co-resident (see the header): report it, release everybody, let the launch run out
          if (fault) __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          xcd_store(&sync[FUSED_FLAG], flag_old + 1u);
          break;
        }
      }
    }
  }
  __syncthreads();
}

// Row geometry of the fused kernels: thread (slot, cv) of a workgroup that owns rows [r0, r1) visits rows r0 + slot + k*rs;
// the row base r0 + k*rs is uniform.  Visits beyond r1 (or with k >= R) still load (see FusedBuf) and are masked afterwards:
// no branch around any load.
struct FusedGeom {
  int CV, rs, slot, cv, grp;
  bool active;
  int64_t r0, r1, Ng;
  int nrows;  // r1 - r0
};

// bid: the workgroup's index within its problem (blockIdx.x unless the launch carries two problems, csrc/bn2d.hip pair kernels)
__device__ inline FusedGeom fused_geom(int64_t N, int64_t Ns, int CV, int G0, int G1, int bid = -1) {
  if (bid < 0) bid = (int)blockIdx.x;
  FusedGeom g;
  g.CV = CV;
  g.rs = FT / g.CV;
  g.slot = threadIdx.x / g.CV;
  g.cv = threadIdx.x - g.slot * g.CV;
  g.active = g.slot < g.rs;
  g.grp = bid >= G0;
  const int lb = g.grp ? bid - G0 : bid, nbg = g.grp ? G1 : G0;
  const int64_t gbase = g.grp ? Ns : 0;
  g.Ng = g.grp ? N - Ns : Ns;
  const int64_t rpb = (g.Ng + nbg - 1) / nbg;
  g.r0 = gbase + (int64_t)lb * rpb;
  g.r1 = g.r0 + rpb < gbase + g.Ng ? g.r0 + rpb : gbase + g.Ng;
  if (g.r1 < g.r0) g.r1 = g.r0;
  g.nrows = (int)(g.r1 - g.r0);
  return g;
}

// Buffer addressing: resource descriptors in scalar registers and one 32-bit per-thread offset per tensor; the hardware
// bounds check makes an out-of-range visit return zeros (loads) or vanish (stores).
// The hardware's range check covers the per-thread offset only, NOT the scalar offset (raw buffers): it tests
// voffset < num_records.  The uniform row base travels as the scalar offset, so a visit past the tensor's last row would read
// whatever lies behind the allocation (up to RMAX * rs rows).  Every load therefore takes the caller's row-validity predicate:
// a lane whose visit is not a row of this workgroup gets a per-thread offset beyond num_records, and the hardware drops its
// access and returns zeros - no memory is touched, whatever the scalar offset is.  Costs one select per load (and saves the
// selects that used to zero the loaded value); the descriptor stays loop-invariant (building a descriptor per access - either
// rebased or with the remaining bytes as num_records - measured +0.6 ms per step on the 79 BatchNorm2d layer shapes).
// Stores are only issued under the same predicate by the callers.
struct FusedBuf {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff;
  int ld2;  // row pitch in bytes
};
constexpr unsigned FUSED_VOFF_OUT = 0x7FFFFFF0u;  // >= any num_records (tensors are below 2^31 bytes: host guard)
__device__ inline FusedBuf fused_buf_bytes(const void* base, int64_t N, int ld, int C, int slot, int cv, int esize, int vec) {
  FusedBuf b;
  b.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(((N - 1) * (long long)ld + C) * esize), 0x00020000);
  b.voff = (unsigned)(slot * ld + cv * vec) * (unsigned)esize;
  b.ld2 = ld * esize;
  return b;
}
__device__ inline FusedBuf fused_buf(const void* base, int64_t N, int ld, int C, int slot, int cv) {
  return fused_buf_bytes(base, N, ld, C, slot, cv, 2, 8);
}
// ok = this lane's row (row base + slot) belongs to the workgroup's row range; otherwise zeros come back without an access
__device__ inline u32x4 fused_ld(const FusedBuf& b, int64_t row, bool ok) {
  return __builtin_amdgcn_raw_buffer_load_b128(b.rsrc, (int)(ok ? b.voff : FUSED_VOFF_OUT), (int)((unsigned)row * (unsigned)b.ld2), 0);
}
__device__ inline void fused_st(const FusedBuf& b, int64_t row, const u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, b.rsrc, (int)b.voff, (int)((unsigned)row * (unsigned)b.ld2), 0);
}

}  // namespace

// ---- host side of the single-launch kernels: all state lives in the caller's handle (common.h MMHandle)
struct FusedPlan {
  bool ok;
  int G0, G1, R;
  unsigned* sync;
  unsigned* fault;
};

constexpr int FUSED_ROWS_TARGET = 6;  // about six rows per thread (measured on the 79 BatchNorm2d shapes of the bench step)

// Grid and rows per thread for a map of N rows (Ns in the first statistics group) and C channels; ok = false: use the
// three-kernel path (map too large to keep on chip, channel count outside the layout, or the handle's switch).
// vec: channels per 16-byte access (8 x 16 bit / 4 x fp32); fns: the translation unit's fused kernels (their dynamic LDS limit is
// raised once per handle: bit ``unit`` of MMHandle::attr_done); opt: MM_OPT_BN2D_FUSED / MM_OPT_BN3D_FUSED
// workgroups per statistics group and row slices per thread of a single-launch kernel on N rows of C channels (shape rule only)
static inline int64_t fused_shape(int cus, int64_t N, int64_t Ns, int C, int vec, int* G0_, int* G1_) {
  const int rs = FT / (C / vec);
  const bool two = Ns > 0 && Ns < N;
  int64_t G = mm_cdiv(N, (int64_t)rs * FUSED_ROWS_TARGET);
  if (G > cus) G = cus;
  if (G < (two ? 2 : 1)) G = two ? 2 : 1;
  int G0 = (int)G, G1 = 0;
  if (two) {
    G0 = (int)((double)G * (double)Ns / (double)N + 0.5);
    if (G0 < 1) G0 = 1;
    if (G0 > (int)G - 1) G0 = (int)G - 1;
    G1 = (int)G - G0;
  }
  const int64_t rpb0 = mm_cdiv(two ? Ns : N, G0), rpb1 = two ? mm_cdiv(N - Ns, G1) : 0;
  *G0_ = G0, *G1_ = G1;
  return mm_cdiv(rpb0 > rpb1 ? rpb0 : rpb1, rs);
}

// cus_div = 2: the plan of ONE of two problems that share a launch (half the CUs each; csrc/bn2d.hip pair entry points)
static inline int fused_plan(MMHandle* H, int opt, int unit, int64_t N, int64_t Ns, int C, int vec, int rmax, bool backward,
                             const void* const* fns, int nfns, hipStream_t s, FusedPlan* pl, int cus_div = 1) {
  pl->ok = false;
  if (!H->fused_ok || !(H->opt[opt] & (backward ? 2 : 1)) || N <= 0 || C % vec != 0 || C > FT || C < vec) return MM_OK;
  if (!(H->attr_done & (1u << unit))) {
    bool good = true;
    for (int i = 0; good && i < nfns; i++)
      good = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS) == hipSuccess;
    if (!good) {
      (void)hipGetLastError();
      H->fused_ok = 0;
      return MM_OK;
    }
    H->attr_done |= 1u << unit;
  }
  int slot = -1;
  for (int i = 0; i < H->nstream; i++)
    if (H->streams[i] == s) slot = i;
  if (slot < 0) {
    if (H->nstream >= MM_SYNC_SLOTS) return MM_OK;  // more streams than barrier slots: three-kernel path
    slot = H->nstream++;
    H->streams[slot] = s;
  }
  int G0, G1;
  const int64_t R = fused_shape(H->cus / cus_div, N, Ns, C, vec, &G0, &G1);
  if (R > rmax) return MM_OK;
  pl->ok = true;
  pl->G0 = G0, pl->G1 = G1, pl->R = (int)R;
  pl->sync = H->sync + slot * (MM_SYNC_SLOT_BYTES / 4);
  pl->fault = H->fault_dev;
  return MM_OK;
}
