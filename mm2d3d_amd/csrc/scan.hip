// Exclusive prefix sum (int32) used by the voxel-dedupe and rulebook compaction kernels.
// Default: rocPRIM's single-pass scan (decoupled look-back: one state-init launch + one scan launch, each element read and
// written once).  Its workgroups wait for their predecessors' partial sums, so - like the Onesweep sort of csrc/ostable.hip -
// it must not run beside a grid-barrier kernel of another stream; a build on a side stream selects the three-launch
// reduce-then-scan below (per-block sums, one-block scan of the sums, per-block rescan: no workgroup waits for another)
// with mm_os_table_set_sort(1).  Same result either way (integer sums).
#include <stdarg.h>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "common.h"

static thread_local char g_err[512] = "";
void mm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mm_last_error() { return g_err; }

namespace {
constexpr int SCAN_T = 256;
constexpr int SCAN_PER = 8;
constexpr int SCAN_BLK = SCAN_T * SCAN_PER;  // items per block

__device__ inline int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// block-wide exclusive scan of one int per thread (256 threads); returns exclusive prefix, total in *tot
__device__ inline int block_excl_scan(int v, int* tot) {
  __shared__ int wsum[SCAN_T / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = wave_incl_scan(v);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int i = 0; i < SCAN_T / 64; i++) {
    int x = wsum[i];
    if (i < w) base += x;
    total += x;
  }
  __syncthreads();
  *tot = total;
  return base + inc - v;
}

// FLAG: the scanned value is (in[i] >= 0) instead of in[i] (mm_exclusive_scan_nonneg_i32)
template <bool FLAG>
__global__ __launch_bounds__(SCAN_T) void k_block_sums(const int32_t* __restrict__ in, int32_t* __restrict__ sums,
                                                        int64_t n) {
  int64_t base = (int64_t)blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_PER;
  int v = 0;
#pragma unroll
  for (int i = 0; i < SCAN_PER; i++)
    if (base + i < n) v += FLAG ? (in[base + i] >= 0 ? 1 : 0) : in[base + i];
  int tot;
  block_excl_scan(v, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_T) void k_scan_sums(int32_t* __restrict__ sums, int64_t nb,
                                                       int32_t* __restrict__ total_out) {
  int carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += SCAN_T) {
    int64_t i = b0 + threadIdx.x;
    int v = i < nb ? sums[i] : 0;
    int tot;
    int ex = block_excl_scan(v, &tot);
    if (i < nb) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <bool FLAG>
__global__ __launch_bounds__(SCAN_T) void k_rescan(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                    const int32_t* __restrict__ sums, int64_t n) {
  int64_t base = (int64_t)blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_PER;
  int x[SCAN_PER];
  int v = 0;
#pragma unroll
  for (int i = 0; i < SCAN_PER; i++) {
    x[i] = (base + i < n) ? (FLAG ? (in[base + i] >= 0 ? 1 : 0) : in[base + i]) : 0;
    v += x[i];
  }
  int tot;
  int ex = block_excl_scan(v, &tot) + sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_PER; i++) {
    if (base + i < n) out[base + i] = ex;
    ex += x[i];
  }
}
struct NonNeg {  // i -> (in[i] >= 0), 0 past the end: the look-back scan runs over n + 1 elements to produce the total
  const int32_t* in;
  int64_t n;
  __host__ __device__ int32_t operator()(int64_t i) const { return i < n ? (in[i] >= 0 ? 1 : 0) : 0; }
};
}  // namespace


static size_t lookback_bytes(int64_t n) {
  size_t b = 0;
  (void)rocprim::exclusive_scan(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t)0, (size_t)(n > 0 ? n : 1),
                                rocprim::plus<int32_t>(), (hipStream_t)0);
  return b;
}

size_t mm_scan_ws_bytes(int64_t n) {
  return mm_align((size_t)(mm_cdiv(n, SCAN_BLK) + 1) * sizeof(int32_t)) + mm_align(lookback_bytes(n + 1)) + 256;
}

int mm_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total_out, void* ws, size_t ws_bytes,
                          hipStream_t s, int no_spin) {
  if (n <= 0) {
    if (total_out) MM_HIP(hipMemsetAsync(total_out, 0, sizeof(int32_t), s));
    return MM_OK;
  }
  if (ws_bytes < mm_scan_ws_bytes(n)) {
    mm_set_error("scan workspace too small: %zu < %zu", ws_bytes, mm_scan_ws_bytes(n));
    return MM_ERR_WORKSPACE;
  }
  int32_t* sums = (int32_t*)ws;
  int64_t nb = mm_cdiv(n, SCAN_BLK);
  if (!no_spin && total_out) {  // rocPRIM's single-pass scan: its workgroups spin on their predecessors (decoupled look-back)
    // in and out hold n + 1 elements (common.h): scan n + 1 of them, out[n] = the total (in[n] is read, its value does not
    // enter out[0..n])
    size_t tb = lookback_bytes(n + 1);
    char* tmp = (char*)ws + mm_align((size_t)(nb + 1) * sizeof(int32_t));
    if ((size_t)(tmp - (char*)ws) + tb <= ws_bytes) {
      MM_HIP(rocprim::exclusive_scan((void*)tmp, tb, in, out, (int32_t)0, (size_t)(n + 1), rocprim::plus<int32_t>(), s));
      if (total_out != out + n) MM_HIP(hipMemcpyAsync(total_out, out + n, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
      return MM_OK;
    }
  }
  hipLaunchKernelGGL(k_block_sums<false>, dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, sums, n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_T), 0, s, sums, nb, total_out);
  hipLaunchKernelGGL(k_rescan<false>, dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, out, sums, n);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// out[i] = number of j < i with in[j] >= 0, out[n] = the count (out holds n + 1 elements; in exactly n, may alias nothing).
// The rulebook compaction scans the neighbour table itself: no flag array is written and read back (round 4).
int mm_exclusive_scan_nonneg_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes, hipStream_t s, int no_spin) {
  if (n <= 0) {
    MM_HIP(hipMemsetAsync(out, 0, sizeof(int32_t), s));
    return MM_OK;
  }
  if (ws_bytes < mm_scan_ws_bytes(n)) {
    mm_set_error("scan workspace too small: %zu < %zu", ws_bytes, mm_scan_ws_bytes(n));
    return MM_ERR_WORKSPACE;
  }
  int32_t* sums = (int32_t*)ws;
  int64_t nb = mm_cdiv(n, SCAN_BLK);
  if (!no_spin) {
    size_t tb = lookback_bytes(n + 1);
    char* tmp = (char*)ws + mm_align((size_t)(nb + 1) * sizeof(int32_t));
    if ((size_t)(tmp - (char*)ws) + tb <= ws_bytes) {
      auto it = rocprim::make_transform_iterator(rocprim::counting_iterator<int64_t>(0), NonNeg{in, n});
      MM_HIP(rocprim::exclusive_scan((void*)tmp, tb, it, out, (int32_t)0, (size_t)(n + 1), rocprim::plus<int32_t>(), s));
      return MM_OK;
    }
  }
  hipLaunchKernelGGL(k_block_sums<true>, dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, sums, n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_T), 0, s, sums, nb, out + n);
  hipLaunchKernelGGL(k_rescan<true>, dim3((unsigned)nb), dim3(SCAN_T), 0, s, in, out, sums, n);
  MM_LAUNCH_CHECK();
  return MM_OK;
}
