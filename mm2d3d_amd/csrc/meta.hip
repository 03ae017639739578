// Voxel hash, active-set dedupe and rulebook construction on the GPU.
//
// Replaces the host-side std::hash-map work SparseConvNet does for
//   scn.InputLayer(3, full_scale, mode=4)   (/root/reference/.../3d_net/scn_unet.py:113,121)
//   scn.SubmanifoldConvolution rulebooks    (scn_unet.py:43,45,52,114)
//   scn.Convolution / Deconvolution rules   (scn_unet.py:68-70,75-77)
// Canonical orders (A.8): ids = first occurrence in scan order; bucket pairs sorted by out id.
#include "common.h"

namespace {

constexpr unsigned long long EMPTY_KEY = 0xFFFFFFFFFFFFFFFFull;
constexpr int T = 256;

__device__ inline unsigned long long pack_key(int x, int y, int z, int b) {
  return ((unsigned long long)(unsigned)b << 48) | ((unsigned long long)(unsigned)x << 32) |
         ((unsigned long long)(unsigned)y << 16) | (unsigned long long)(unsigned)z;
}

__device__ inline unsigned long long mix64(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

__device__ inline int hash_find(const unsigned long long* __restrict__ tkeys, const int32_t* __restrict__ tvals,
                                long long mask, unsigned long long key) {
  long long slot = (long long)(mix64(key) & (unsigned long long)mask);
  for (long long probe = 0; probe <= mask; probe++) {
    unsigned long long k = tkeys[slot];
    if (k == key) return tvals[slot];
    if (k == EMPTY_KEY) return -1;
    slot = (slot + 1) & mask;
  }
  return -1;
}

template <typename CT>
__global__ __launch_bounds__(T) void k_insert(const CT* __restrict__ coords, int64_t n_bound,
                                               const int32_t* __restrict__ n_dev, int shift,
                                               unsigned long long* __restrict__ tkeys, int32_t* __restrict__ tvals,
                                               long long mask, int32_t* __restrict__ slot_of, int32_t* __restrict__ err) {
  int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
  int64_t n = n_dev ? (int64_t)*n_dev : n_bound;
  if (p >= n) return;
  long long x = (long long)coords[p * 4 + 0], y = (long long)coords[p * 4 + 1], z = (long long)coords[p * 4 + 2],
            b = (long long)coords[p * 4 + 3];
  if (x < 0 || y < 0 || z < 0 || b < 0 || x > 65535 || y > 65535 || z > 65535 || b > 65535) {
    atomicExch(err, 1);
    slot_of[p] = -1;
    return;
  }
  unsigned long long key = pack_key((int)(x >> shift), (int)(y >> shift), (int)(z >> shift), (int)b);
  long long slot = (long long)(mix64(key) & (unsigned long long)mask);
  for (long long probe = 0; probe <= mask; probe++) {
    unsigned long long prev = atomicCAS(&tkeys[slot], EMPTY_KEY, key);
    if (prev == EMPTY_KEY || prev == key) {
      atomicMin(&tvals[slot], (int32_t)p);
      slot_of[p] = (int32_t)slot;
      return;
    }
    slot = (slot + 1) & mask;
  }
  atomicExch(err, 2);  // table full (cannot happen with cap >= 2n)
  slot_of[p] = -1;
}

__global__ __launch_bounds__(T) void k_flag_first(const int32_t* __restrict__ slot_of, const int32_t* __restrict__ tvals,
                                                   int64_t n_bound, const int32_t* __restrict__ n_dev,
                                                   int32_t* __restrict__ flag) {
  int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
  if (p >= n_bound) return;
  int64_t n = n_dev ? (int64_t)*n_dev : n_bound;
  int f = 0;
  if (p < n) {
    int s = slot_of[p];
    f = (s >= 0 && tvals[s] == (int32_t)p) ? 1 : 0;
  }
  flag[p] = f;
}

template <typename CT>
__global__ __launch_bounds__(T) void k_assign(const CT* __restrict__ coords, int64_t n_bound,
                                               const int32_t* __restrict__ n_dev, int shift,
                                               const int32_t* __restrict__ slot_of, const int32_t* __restrict__ tvals,
                                               const int32_t* __restrict__ rank, int32_t* __restrict__ item2vox,
                                               int32_t* __restrict__ vox_coords, int32_t* __restrict__ cnt) {
  int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
  int64_t n = n_dev ? (int64_t)*n_dev : n_bound;
  if (p >= n) return;
  int s = slot_of[p];
  if (s < 0) {
    item2vox[p] = -1;
    return;
  }
  int first = tvals[s];
  int v = rank[first];
  item2vox[p] = v;
  atomicAdd(&cnt[v], 1);
  if (first == (int32_t)p) {
    vox_coords[(int64_t)v * 4 + 0] = (int32_t)((long long)coords[p * 4 + 0] >> shift);
    vox_coords[(int64_t)v * 4 + 1] = (int32_t)((long long)coords[p * 4 + 1] >> shift);
    vox_coords[(int64_t)v * 4 + 2] = (int32_t)((long long)coords[p * 4 + 2] >> shift);
    vox_coords[(int64_t)v * 4 + 3] = (int32_t)coords[p * 4 + 3];
  }
}

// also clears flag[p]: the array is free from here on and serves as the per-voxel cursor of k_fill_lists
__global__ __launch_bounds__(T) void k_store_ids(int64_t n_bound, const int32_t* __restrict__ n_dev,
                                                  const int32_t* __restrict__ slot_of, int32_t* __restrict__ flag,
                                                  const int32_t* __restrict__ rank, int32_t* __restrict__ tvals) {
  int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
  if (p > n_bound) return;
  int64_t n = n_dev ? (int64_t)*n_dev : n_bound;
  if (p < n && flag[p]) tvals[slot_of[p]] = rank[p];
  flag[p] = 0;
}

// one launch instead of three fills: empty hash table (keys all ones, values 0x7F7F7F7F) and zero counters
__global__ __launch_bounds__(T) void k_dedupe_init(unsigned long long* __restrict__ tkeys, int32_t* __restrict__ tvals, int64_t cap,
                                                    int32_t* __restrict__ cnt, int64_t ncnt) {
  int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
  if (i < cap) {
    tkeys[i] = ~0ull;
    tvals[i] = 0x7F7F7F7F;
  }
  if (i < ncnt) cnt[i] = 0;
}

__global__ __launch_bounds__(T) void k_fill_lists(int64_t n_bound, const int32_t* __restrict__ n_dev,
                                                   const int32_t* __restrict__ item2vox,
                                                   const int32_t* __restrict__ csr_off, int32_t* __restrict__ cursor,
                                                   int32_t* __restrict__ csr_items) {
  int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
  int64_t n = n_dev ? (int64_t)*n_dev : n_bound;
  if (p >= n) return;
  int v = item2vox[p];
  if (v < 0) return;
  int j = atomicAdd(&cursor[v], 1);
  csr_items[csr_off[v] + j] = (int32_t)p;
}

// each voxel's list is tiny (1-3 points, <= 8 children): insertion sort makes the order ascending,
// which fixes the floating-point summation order of the mean / segmented sums (bit-stable run to run).
__global__ __launch_bounds__(T) void k_sort_lists(int64_t n_bound, const int32_t* __restrict__ n_active,
                                                   const int32_t* __restrict__ csr_off, int32_t* __restrict__ csr_items) {
  int64_t v = (int64_t)blockIdx.x * T + threadIdx.x;
  if (v >= n_bound || v >= (int64_t)*n_active) return;
  int a = csr_off[v], b = csr_off[v + 1];
  for (int i = a + 1; i < b; i++) {
    int x = csr_items[i];
    int j = i - 1;
    while (j >= a && csr_items[j] > x) {
      csr_items[j + 1] = csr_items[j];
      j--;
    }
    csr_items[j + 1] = x;
  }
}

// ---- submanifold 3^3 neighbour table, k-major: nbr[k*n + o] = id of the active site at coord(o)+off(k), or -1.
// The relation is symmetric (o' = nbr[k][o]  <=>  o = nbr[26-k][o']), so only offsets 0..12 are probed in the hash table and
// each hit also fills its mirror entry; planes 14..26 are preset to -1 by the host wrapper.  Every entry still has exactly
// one writer.  Halves the random 8-byte probes that bound this kernel.
__global__ __launch_bounds__(T) void k_subm_nbr(const int32_t* __restrict__ vc, int64_t n, int spatial,
                                                 const unsigned long long* __restrict__ tkeys,
                                                 const int32_t* __restrict__ tvals, long long mask,
                                                 int32_t* __restrict__ nbr) {
  int64_t o = (int64_t)blockIdx.x * T + threadIdx.x;
  int k = blockIdx.y;  // 0..13
  if (o >= n) return;
  if (k == 13) {
    nbr[(int64_t)13 * n + o] = (int)o;
    return;
  }
  int x = vc[o * 4 + 0] + (k / 9 - 1), y = vc[o * 4 + 1] + ((k / 3) % 3 - 1), z = vc[o * 4 + 2] + (k % 3 - 1);
  int b = vc[o * 4 + 3];
  int r = -1;
  if (x >= 0 && y >= 0 && z >= 0 && x < spatial && y < spatial && z < spatial) r = hash_find(tkeys, tvals, mask, pack_key(x, y, z, b));
  nbr[(int64_t)k * n + o] = r;
  if (r >= 0) nbr[(int64_t)(26 - k) * n + r] = (int)o;
}

// ---- strided 2^3 table: nbr[k*n_coarse + parent] = child
__global__ __launch_bounds__(T) void k_down_nbr(const int32_t* __restrict__ vc_fine, int64_t n_fine,
                                                 const int32_t* __restrict__ fine2coarse, int64_t n_coarse,
                                                 int32_t* __restrict__ nbr) {
  int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
  if (i >= n_fine) return;
  int k = ((vc_fine[i * 4 + 0] & 1) * 2 + (vc_fine[i * 4 + 1] & 1)) * 2 + (vc_fine[i * 4 + 2] & 1);
  nbr[(int64_t)k * n_coarse + fine2coarse[i]] = (int32_t)i;
}

__global__ __launch_bounds__(T) void k_emit_rules(const int32_t* __restrict__ nbr, const int32_t* __restrict__ pos,
                                                   int64_t n_out, int K, int32_t* __restrict__ rin,
                                                   int32_t* __restrict__ rout, int32_t* __restrict__ offsets,
                                                   const int32_t* __restrict__ total) {
  int64_t o = (int64_t)blockIdx.x * T + threadIdx.x;
  int k = blockIdx.y;
  if (o == 0) {
    offsets[k] = pos[(int64_t)k * n_out];
    if (k == K - 1) offsets[K] = *total;
  }
  if (o >= n_out) return;
  int i = nbr[(int64_t)k * n_out + o];
  if (i >= 0) {
    int p = pos[(int64_t)k * n_out + o];
    rin[p] = i;
    rout[p] = (int32_t)o;
  }
}

__global__ __launch_bounds__(T) void k_row_counts(const int32_t* __restrict__ nbr, int64_t n_out, int K,
                                                   int32_t* __restrict__ cnt) {
  int64_t o = (int64_t)blockIdx.x * T + threadIdx.x;
  if (o >= n_out) return;
  int c = 0;
  for (int k = 0; k < K; k++) c += nbr[(int64_t)k * n_out + o] >= 0;
  cnt[o] = c;
}

__global__ __launch_bounds__(T) void k_row_fill(const int32_t* __restrict__ nbr, const int32_t* __restrict__ pos,
                                                 int64_t n_out, int K, const int32_t* __restrict__ csr_off,
                                                 int32_t* __restrict__ csr_pos) {
  int64_t o = (int64_t)blockIdx.x * T + threadIdx.x;
  if (o >= n_out) return;
  int e = csr_off[o];
  for (int k = 0; k < K; k++)
    if (nbr[(int64_t)k * n_out + o] >= 0) csr_pos[e++] = pos[(int64_t)k * n_out + o];
}

__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

inline unsigned nblk(int64_t n) { return (unsigned)mm_cdiv(n > 0 ? n : 1, T); }

// first row whose batch index (column 3 of the [n,4] voxel coordinates) is >= split; rows are in first-occurrence order of a
// batch-sorted point list, hence batch-sorted themselves.  One thread: a 20-step binary search.
__global__ void k_batch_lower_bound(const int32_t* __restrict__ coords, const int32_t* __restrict__ n_dev, int32_t split,
                                    int32_t* __restrict__ out) {
  int lo = 0, hi = *n_dev;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (coords[(int64_t)mid * 4 + 3] < split) lo = mid + 1;
    else hi = mid;
  }
  *out = lo;
}
}  // namespace

extern "C" {

int64_t mm_hash_capacity(int64_t n_items) {
  int64_t c = 1024;
  while (c < 2 * n_items) c <<= 1;
  return c;
}

size_t mm_dedupe_ws_bytes(int64_t n) { return 4 * mm_align((size_t)(n + 1) * 4) + mm_scan_ws_bytes(n + 1) + 1024; }

int mm_voxel_dedupe(const void* coords, int coords_is_i64, int64_t n_bound, const int32_t* n_dev, int shift,
                    uint64_t* tkeys, int32_t* tvals, int64_t cap, int32_t* item2vox, int32_t* vox_coords,
                    int32_t* csr_off, int32_t* csr_items, int32_t* n_active_dev, int32_t* err_dev, int no_spin, void* ws,
                    size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(n_bound >= 0 && cap >= 2 * n_bound && (cap & (cap - 1)) == 0, "dedupe: cap must be pow2 >= 2n (n=%lld cap=%lld)",
               (long long)n_bound, (long long)cap);
  MM_CHECK_ARG(shift >= 0 && shift < 16, "dedupe: bad shift %d", shift);
  MMArena ar(ws, ws_bytes);
  int32_t* slot_of = ar.take<int32_t>(n_bound + 1);
  int32_t* flag = ar.take<int32_t>(n_bound + 1);
  int32_t* rank = ar.take<int32_t>(n_bound + 1);
  int32_t* cnt = ar.take<int32_t>(n_bound + 1);
  size_t sws = mm_scan_ws_bytes(n_bound + 1);
  char* scan_ws = ar.take<char>(sws);
  if (!slot_of || !flag || !rank || !cnt || !scan_ws) {
    mm_set_error("dedupe: workspace too small (%zu < %zu)", ws_bytes, mm_dedupe_ws_bytes(n_bound));
    return MM_ERR_WORKSPACE;
  }
  {
    const int64_t ni = cap > n_bound + 1 ? cap : n_bound + 1;
    hipLaunchKernelGGL(k_dedupe_init, dim3(nblk(ni)), dim3(T), 0, s, (unsigned long long*)tkeys, tvals, cap, cnt, n_bound + 1);
  }
  if (n_bound == 0) {
    MM_HIP(hipMemsetAsync(n_active_dev, 0, 4, s));
    MM_HIP(hipMemsetAsync(csr_off, 0, 4, s));
    return MM_OK;
  }
  const long long mask = cap - 1;
  unsigned g = nblk(n_bound);
  if (coords_is_i64)
    hipLaunchKernelGGL(k_insert<int64_t>, dim3(g), dim3(T), 0, s, (const int64_t*)coords, n_bound, n_dev, shift,
                       (unsigned long long*)tkeys, tvals, mask, slot_of, err_dev);
  else
    hipLaunchKernelGGL(k_insert<int32_t>, dim3(g), dim3(T), 0, s, (const int32_t*)coords, n_bound, n_dev, shift,
                       (unsigned long long*)tkeys, tvals, mask, slot_of, err_dev);
  hipLaunchKernelGGL(k_flag_first, dim3(g), dim3(T), 0, s, slot_of, tvals, n_bound, n_dev, flag);
  int rc = mm_exclusive_scan_i32(flag, rank, n_bound, n_active_dev, scan_ws, sws, s, no_spin);
  if (rc) return rc;
  if (coords_is_i64)
    hipLaunchKernelGGL(k_assign<int64_t>, dim3(g), dim3(T), 0, s, (const int64_t*)coords, n_bound, n_dev, shift, slot_of,
                       tvals, rank, item2vox, vox_coords, cnt);
  else
    hipLaunchKernelGGL(k_assign<int32_t>, dim3(g), dim3(T), 0, s, (const int32_t*)coords, n_bound, n_dev, shift, slot_of,
                       tvals, rank, item2vox, vox_coords, cnt);
  hipLaunchKernelGGL(k_store_ids, dim3(nblk(n_bound + 1)), dim3(T), 0, s, n_bound, n_dev, slot_of, flag, rank, tvals);
  // csr over voxels (bound n_bound); csr_off[n_bound] = number of items
  rc = mm_exclusive_scan_i32(cnt, csr_off, n_bound, csr_off + n_bound, scan_ws, sws, s, no_spin);
  if (rc) return rc;
  hipLaunchKernelGGL(k_fill_lists, dim3(g), dim3(T), 0, s, n_bound, n_dev, item2vox, csr_off, flag, csr_items);
  hipLaunchKernelGGL(k_sort_lists, dim3(g), dim3(T), 0, s, n_bound, n_active_dev, csr_off, csr_items);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_subm_neighbors(const int32_t* vox_coords, int64_t n, int32_t spatial_size, const uint64_t* tkeys,
                      const int32_t* tvals, int64_t cap, int32_t* nbr, hipStream_t s) {
  MM_CHECK_ARG(n >= 0 && (cap & (cap - 1)) == 0, "subm_neighbors: bad args");
  if (n == 0) return MM_OK;
  MM_HIP(hipMemsetAsync(nbr + 14 * n, 0xFF, (size_t)13 * n * 4, s));
  hipLaunchKernelGGL(k_subm_nbr, dim3(nblk(n), 14), dim3(T), 0, s, vox_coords, n, (int)spatial_size,
                     (const unsigned long long*)tkeys, tvals, (long long)(cap - 1), nbr);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_down_neighbors(const int32_t* vox_coords_fine, int64_t n_fine, const int32_t* fine2coarse, int64_t n_coarse,
                      int32_t* nbr, hipStream_t s) {
  MM_CHECK_ARG(n_fine >= 0 && n_coarse >= 0, "down_neighbors: bad args");
  if (n_coarse == 0) return MM_OK;
  hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)mm_cdiv(8 * n_coarse, 256)), dim3(256), 0, s, nbr, 8 * n_coarse, -1);
  if (n_fine) hipLaunchKernelGGL(k_down_nbr, dim3(nblk(n_fine)), dim3(T), 0, s, vox_coords_fine, n_fine, fine2coarse, n_coarse, nbr);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

size_t mm_rulebook_ws_bytes(int64_t n_out, int K) {
  return mm_align((size_t)(K * n_out + 1) * 4) + mm_align((size_t)(n_out + 1) * 4) + mm_scan_ws_bytes(K * n_out + 1) + 1024;
}

// nbr [K][n_out] -> k-major rule lists (rin/rout, capacity K*n_out), offsets[K+1], CSR over out rows.
// csr_off == csr_pos == NULL: only the rule lists (levels served by the output-stationary engine never read the CSR; a
// caller that turns out to need it builds it later with mm_rulebook_csr from the same nbr table).
int mm_rulebook_compact(const int32_t* nbr, int K, int64_t n_out, int32_t* rin, int32_t* rout, int32_t* offsets,
                        int32_t* csr_off, int32_t* csr_pos, int no_spin, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(K > 0 && K <= 64 && n_out >= 0 && (csr_off == nullptr) == (csr_pos == nullptr), "rulebook_compact: bad args");
  if (n_out == 0) {
    MM_HIP(hipMemsetAsync(offsets, 0, (size_t)(K + 1) * 4, s));
    if (csr_off) MM_HIP(hipMemsetAsync(csr_off, 0, 4, s));
    return MM_OK;
  }
  MMArena ar(ws, ws_bytes);
  int64_t total = (int64_t)K * n_out;
  int32_t* pos = ar.take<int32_t>(total + 1);
  int32_t* cnt = ar.take<int32_t>(n_out + 1);
  size_t sws = mm_scan_ws_bytes(total + 1);
  char* scan_ws = ar.take<char>(sws);
  if (!pos || !cnt || !scan_ws) {
    mm_set_error("rulebook_compact: workspace too small (%zu < %zu)", ws_bytes, mm_rulebook_ws_bytes(n_out, K));
    return MM_ERR_WORKSPACE;
  }
  int rc = mm_exclusive_scan_nonneg_i32(nbr, pos, total, scan_ws, sws, s, no_spin);  // pos[f] = rules before table entry f
  if (rc) return rc;
  if (rin) hipLaunchKernelGGL(k_emit_rules, dim3(nblk(n_out), K), dim3(T), 0, s, nbr, pos, n_out, K, rin, rout, offsets, pos + total);
  if (csr_off) {
    hipLaunchKernelGGL(k_row_counts, dim3(nblk(n_out)), dim3(T), 0, s, nbr, n_out, K, cnt);
    rc = mm_exclusive_scan_i32(cnt, csr_off, n_out, csr_off + n_out, scan_ws, sws, s, no_spin);
    if (rc) return rc;
    hipLaunchKernelGGL(k_row_fill, dim3(nblk(n_out)), dim3(T), 0, s, nbr, pos, n_out, K, csr_off, csr_pos);
  }
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// The CSR half of mm_rulebook_compact alone (same nbr table, same workspace size): csr_off [n_out+1], csr_pos [n_rules].
int mm_rulebook_csr(const int32_t* nbr, int K, int64_t n_out, int32_t* csr_off, int32_t* csr_pos, int no_spin, void* ws, size_t ws_bytes,
                    hipStream_t s) {
  MM_CHECK_ARG(csr_off && csr_pos, "rulebook_csr: null output");
  return mm_rulebook_compact(nbr, K, n_out, nullptr, nullptr, nullptr, csr_off, csr_pos, no_spin, ws, ws_bytes, s);
}

// out[0] = number of active rows whose batch index < split (their row ids are 0 .. out[0]-1); n_dev: device row count
int mm_batch_lower_bound(const int32_t* vox_coords, const int32_t* n_dev, int32_t split, int32_t* out, hipStream_t s) {
  MM_CHECK_ARG(vox_coords && n_dev && out, "batch_lower_bound: null pointer");
  hipLaunchKernelGGL(k_batch_lower_bound, dim3(1), dim3(1), 0, s, vox_coords, n_dev, split, out);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
