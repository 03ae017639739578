// Tile tables of the output-stationary sparse convolution engine (csrc/osconv.hip).
//
// Input: the dense neighbour table nbr[K][n] of a rulebook (csrc/meta.hip: k_subm_nbr for SubmanifoldConvolution, k_down_nbr
// for Convolution, k_up_nbr here for Deconvolution / the data gradient of Convolution), nbr[k][o] = source row of
// destination row o at offset k or -1.  Output, for tiles of `tile_rows` destination rows:
//   dst[npad]      destination rows ordered by their neighbour bitmask (bit k = row has a neighbour at offset k; stable, so
//                  rows with equal masks stay in id order), -1 in the padding of the last tile
//   nbrp[K][npad]  nbr permuted the same way
//   tmask[nt]      OR of the masks of a tile = the offsets the tile has to visit
// Rows with the same neighbour pattern end up in the same 16-row MFMA sub-block, so an offset that is visited is (nearly)
// full: 1.25-1.5x the MFMA work of a perfectly dense rule list on LiDAR-shaped scenes, against 3-3.4x in id order.
// Nothing here changes the canonical ids or rulebook orders (A.8): the permutation only decides which workgroup computes
// which output row.
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.h"


namespace {

constexpr int T = 256;

// (also clears the tile masks k_os_fill ORs into: nt <= n, one launch less than a memset)
__global__ __launch_bounds__(T) void k_row_mask(const int32_t* __restrict__ nbr, int K, int64_t n, uint32_t* __restrict__ mask,
                                                 uint32_t* __restrict__ tmask, int64_t nt) {
  const int64_t o = (int64_t)blockIdx.x * T + threadIdx.x;
  if (o < nt) tmask[o] = 0u;
  if (o >= n) return;
  uint32_t m = 0;
  for (int k = 0; k < K; k++) m |= (nbr[(int64_t)k * n + o] >= 0 ? 1u : 0u) << k;
  mask[o] = m;
}

// thread = position j of the sorted order; tile_rows is a multiple of 64, so a wave lies inside one tile
__global__ __launch_bounds__(T) void k_os_fill(const int32_t* __restrict__ nbr, int K, int64_t n, int64_t npad, int tile_rows,
                                                const uint32_t* __restrict__ mask_sorted, const int32_t* __restrict__ perm,
                                                int32_t* __restrict__ dst, int32_t* __restrict__ nbrp, uint32_t* __restrict__ tmask) {
  const int64_t j = (int64_t)blockIdx.x * T + threadIdx.x;
  if (j >= npad) return;
  const int p = j < n ? perm[j] : -1;
  const uint32_t m = j < n ? mask_sorted[j] : 0u;
  dst[j] = p;
  for (int k = 0; k < K; k++) nbrp[(int64_t)k * npad + j] = ((m >> k) & 1u) ? nbr[(int64_t)k * n + p] : -1;
  uint32_t wm = m;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wm |= __shfl_xor(wm, off);
  if ((threadIdx.x & 63) == 0 && wm) atomicOr(&tmask[j / tile_rows], wm);
}

// Deconvolution / d(Convolution)/d(input): destination = fine row, its single rule is (octant k, parent)
__global__ __launch_bounds__(T) void k_up_nbr(const int32_t* __restrict__ vc_fine, int64_t n_fine, const int32_t* __restrict__ fine2coarse,
                                               int32_t* __restrict__ nbr) {
  const int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
  if (i >= n_fine) return;
  const int kk = ((vc_fine[i * 4 + 0] & 1) * 2 + (vc_fine[i * 4 + 1] & 1)) * 2 + (vc_fine[i * 4 + 2] & 1);
  const int par = fine2coarse[i];
#pragma unroll
  for (int k = 0; k < 8; k++) nbr[(int64_t)k * n_fine + i] = k == kk ? par : -1;
}

// rocPRIM's default switches to a merge sort below 2^20 keys: ~17 launch-bound passes (100 us) for the 60k..800k-row tables
// here, where Onesweep needs one histogram + one scatter pass per 8 key bits (K = 8: a single pass).
using SortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 16384>;
// The same sort as a merge sort at every size: block sorts + merge passes, workgroups independent of each other.  Onesweep's
// workgroups spin on their predecessors' partial sums (decoupled look-back); beside a kernel with a grid barrier (the single-launch
// batch norms) on another stream that can deadlock, so a build that runs on a side stream selects this one (mm_os_table_set_sort).
using SortCfgMerge = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, ((size_t)1 << 40)>;

template <typename Cfg>
hipError_t sort_masks(void* tmp, size_t& bytes, const uint32_t* mask, uint32_t* mask_sorted, int32_t* perm, int64_t n, int K, hipStream_t s) {
  return rocprim::radix_sort_pairs<Cfg>(tmp, bytes, mask, mask_sorted, rocprim::counting_iterator<int32_t>(0), perm, (size_t)(n > 0 ? n : 1),
                                        0u, (unsigned)K, s);
}

size_t sort_tmp_bytes(int64_t n, int K) {
  size_t a = 0, b = 0;
  (void)sort_masks<SortCfg>(nullptr, a, nullptr, nullptr, nullptr, n, K, (hipStream_t)0);
  (void)sort_masks<SortCfgMerge>(nullptr, b, nullptr, nullptr, nullptr, n, K, (hipStream_t)0);
  return a > b ? a : b;
}

}  // namespace

extern "C" {

size_t mm_os_table_ws_bytes(int64_t n, int K) {
  return 3 * mm_align((size_t)(n + 1) * 4) + mm_align(sort_tmp_bytes(n, K)) + 1024;
}

int mm_up_neighbors(const int32_t* vox_coords_fine, int64_t n_fine, const int32_t* fine2coarse, int32_t* nbr, hipStream_t s) {
  MM_CHECK_ARG(n_fine >= 0, "up_neighbors: bad args");
  if (n_fine == 0) return MM_OK;
  hipLaunchKernelGGL(k_up_nbr, dim3((unsigned)mm_cdiv(n_fine, T)), dim3(T), 0, s, vox_coords_fine, n_fine, fine2coarse, nbr);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// nbr [K][n] -> dst [npad], nbrp [K][npad], tmask [nt];  nt = ceil(n / tile_rows), npad = nt * tile_rows
// sort_merge: 0 = rocPRIM Onesweep radix sort (default); 1 = merge sort - no workgroup of the sort waits for another, safe beside
// grid-barrier kernels of other streams.  Same result (both stable).
int mm_os_table_build(const int32_t* nbr, int K, int64_t n, int tile_rows, int sort_merge, int32_t* dst, int32_t* nbrp, uint32_t* tmask,
                      void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(K > 0 && K <= 32 && n >= 0 && tile_rows > 0 && tile_rows % 64 == 0, "os_table_build: bad arguments");
  if (n == 0) return MM_OK;
  const int64_t nt = mm_cdiv(n, tile_rows), npad = nt * tile_rows;
  MMArena ar(ws, ws_bytes);
  uint32_t* mask = ar.take<uint32_t>(n + 1);
  uint32_t* mask_sorted = ar.take<uint32_t>(n + 1);
  int32_t* perm = ar.take<int32_t>(n + 1);
  size_t tmp_bytes = sort_tmp_bytes(n, K);
  char* tmp = ar.take<char>(tmp_bytes);
  if (!mask || !mask_sorted || !perm || !tmp) {
    mm_set_error("os_table_build: workspace too small (%zu < %zu)", ws_bytes, mm_os_table_ws_bytes(n, K));
    return MM_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(k_row_mask, dim3((unsigned)mm_cdiv(n, T)), dim3(T), 0, s, nbr, K, n, mask, tmask, nt);
  if (sort_merge) MM_HIP(sort_masks<SortCfgMerge>(tmp, tmp_bytes, mask, mask_sorted, perm, n, K, s));
  else MM_HIP(sort_masks<SortCfg>(tmp, tmp_bytes, mask, mask_sorted, perm, n, K, s));
  hipLaunchKernelGGL(k_os_fill, dim3((unsigned)mm_cdiv(npad, T)), dim3(T), 0, s, nbr, K, n, npad, tile_rows, mask_sorted, perm, dst,
                     nbrp, tmask);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
