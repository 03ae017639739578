// Point -> pixel index of the 2D -> 3D lifting (reference: 2d_net/model.py:131-137, 166-173 gathers
// segm.permute(0,2,3,1)[i][rows, cols] per sample, whose backward is index_put_(accumulate=True)).
// One build per batch, entirely on the device: pixel key of every point -> stable radix sort of (key, point) -> the gather
// reads the map at the decoded key, the scatter sums each run of equal keys in ascending point order (no float atomics: bit-stable).
// Round 4: this replaces ~25 torch launches (repeat_interleave, arange, key arithmetic, sort, run flags, offset tables) by three.
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.h"

namespace {
constexpr int T = 256;
constexpr int MAXB = 1024;  // scenes per batch

// key[p] = (b * H + row) * W + col, b = the scene whose point range holds p (counts[] = points per scene)
__global__ __launch_bounds__(T) void k_lift_keys(const int64_t* __restrict__ rc, const int64_t* __restrict__ counts, int nb, int64_t n, int H,
                                                  int W, int32_t* __restrict__ key, int32_t* __restrict__ err) {
  __shared__ int64_t ends[MAXB];
  if (threadIdx.x == 0) {
    int64_t acc = 0;
    for (int b = 0; b < nb; b++) {
      acc += counts[b];
      ends[b] = acc;
    }
  }
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
  if (p >= n) return;
  int lo = 0, hi = nb - 1;  // first scene whose end exceeds p
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (ends[mid] > p) hi = mid;
    else lo = mid + 1;
  }
  int64_t r = rc[2 * p], c = rc[2 * p + 1];
  if (r < 0 || r >= H || c < 0 || c >= W) {
    err[0] = 1;  // benign race: every writer stores 1
    r = r < 0 ? 0 : (r >= H ? H - 1 : r);
    c = c < 0 ? 0 : (c >= W ? W - 1 : c);  // whatever the caller passed, the gather / scatter only ever see pixels of the map
  }
  key[p] = (int32_t)(((int64_t)lo * H + r) * W + c);
}

__device__ inline int64_t key_offset(int32_t key, int HW, int W, int64_t sb, int64_t sy, int64_t sx) {
  const unsigned k = (unsigned)key, b = k / (unsigned)HW, rem = k - b * (unsigned)HW, r = rem / (unsigned)W, c = rem - r * (unsigned)W;
  return (int64_t)b * sb + (int64_t)r * sy + (int64_t)c * sx;
}

__global__ __launch_bounds__(T) void k_lift_gather_key(const float* __restrict__ seg, int64_t sb, int64_t sy, int64_t sx, int64_t sc,
                                                        const int32_t* __restrict__ key, int64_t N, int C, int HW, int W,
                                                        float* __restrict__ out) {
  const unsigned gid = blockIdx.x * T + threadIdx.x;  // host: N * C < 2^31
  const unsigned p = gid / (unsigned)C, c = gid - p * (unsigned)C;
  if (p >= N) return;
  out[(int64_t)p * C + c] = seg[key_offset(key[p], HW, W, sb, sy, sx) + (int64_t)c * sc];
}

// sorted element e opens a run iff its key differs from its predecessor's: it sums the run (ascending point order inside a run
// because the sort is stable); pixels without points are not touched (the caller zero-fills dseg)
__global__ __launch_bounds__(T) void k_lift_scatter_key(const float* __restrict__ dout, int C, const int32_t* __restrict__ order,
                                                         const int32_t* __restrict__ skey, int64_t N, int HW, int W, int64_t sb, int64_t sy,
                                                         int64_t sx, int64_t sc, float* __restrict__ dseg) {
  const unsigned gid = blockIdx.x * T + threadIdx.x;
  const unsigned e = gid / (unsigned)C, c = gid - e * (unsigned)C;
  if (e >= N) return;
  const int32_t k = skey[e];
  if (e > 0 && skey[e - 1] == k) return;
  float s = 0.f;
  int64_t j = e;
  do {
    s += dout[(int64_t)order[j] * C + c];
    j++;
  } while (j < N && skey[j] == k);
  dseg[key_offset(k, HW, W, sb, sy, sx) + (int64_t)c * sc] = s;
}

// (rocPRIM's default switches to a merge sort below 2^20 keys: 20 launch-bound passes for the 140k points of a batch; Onesweep needs a
// histogram + one scatter pass per 8 key bits - see csrc/ostable.hip)
using SortCfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 16384>;
using SortCfgMerge = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, ((size_t)1 << 40)>;
template <typename Cfg>
hipError_t sort_keys(void* tmp, size_t& bytes, const int32_t* key, int32_t* skey, int32_t* order, int64_t n, unsigned bits, hipStream_t s) {
  return rocprim::radix_sort_pairs<Cfg>(tmp, bytes, (const uint32_t*)key, (uint32_t*)skey, rocprim::counting_iterator<int32_t>(0), order,
                                        (size_t)(n > 0 ? n : 1), 0u, bits, s);
}
size_t sort_bytes(int64_t n) {
  size_t a = 0, b = 0;
  (void)sort_keys<SortCfg>(nullptr, a, nullptr, nullptr, nullptr, n, 32, (hipStream_t)0);
  (void)sort_keys<SortCfgMerge>(nullptr, b, nullptr, nullptr, nullptr, n, 32, (hipStream_t)0);
  return a > b ? a : b;
}
}  // namespace

extern "C" {

size_t mm_lift_index_ws_bytes(int64_t n) { return mm_align(sort_bytes(n)) + 256; }

// rc: device int64 [n, 2] (row, col) of every point, scenes concatenated; counts: device int64 [nb] points per scene.
// key [n] = pixel id (b*H + row)*W + col (rows / cols clamped into the map, err[0] set to 1 if one was outside);
// skey [n] / order [n] = keys ascending and the point of each sorted position (stable).  no_spin: see mm_voxel_dedupe.
int mm_lift_index(const int64_t* rc, const int64_t* counts, int nb, int64_t n, int H, int W, int no_spin, int32_t* key, int32_t* skey,
                  int32_t* order, int32_t* err, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(nb >= 0 && nb <= MAXB && n >= 0 && H > 0 && W > 0 && (int64_t)nb * H * W < (1ll << 31) && n < (1ll << 31),
               "lift_index: bad shape (nb=%d n=%lld H=%d W=%d)", nb, (long long)n, H, W);
  if (n == 0) return MM_OK;
  size_t tb = sort_bytes(n);
  if (ws_bytes < tb) {
    mm_set_error("lift_index: workspace too small (%zu < %zu)", ws_bytes, tb);
    return MM_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(k_lift_keys, dim3((unsigned)mm_cdiv(n, T)), dim3(T), 0, s, rc, counts, nb, n, H, W, key, err);
  unsigned bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < (int64_t)nb * H * W) bits++;
  if (no_spin) MM_HIP(sort_keys<SortCfgMerge>(ws, tb, key, skey, order, n, bits, s));
  else MM_HIP(sort_keys<SortCfg>(ws, tb, key, skey, order, n, bits, s));
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// out[p][c] = seg[b*sb + row*sy + col*sx + c*sc] at the pixel of point p
int mm_lift_gather_key(const float* seg, int64_t sb, int64_t sy, int64_t sx, int64_t sc, const int32_t* key, int64_t N, int C, int H, int W,
                       float* out, hipStream_t s) {
  MM_CHECK_ARG(N * C < (1ll << 31) && C > 0, "lift_gather: too many elements");
  if (N == 0) return MM_OK;
  hipLaunchKernelGGL(k_lift_gather_key, dim3((unsigned)mm_cdiv(N * C, T)), dim3(T), 0, s, seg, sb, sy, sx, sc, key, N, C, H * W, W, out);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// dseg (zero-filled by the caller) [pixel of run][c] = sum of dout over the run's points, ascending point order
int mm_lift_scatter_key(const float* dout, int C, const int32_t* order, const int32_t* skey, int64_t N, int H, int W, int64_t sb, int64_t sy,
                        int64_t sx, int64_t sc, float* dseg, hipStream_t s) {
  MM_CHECK_ARG(N * C < (1ll << 31) && C > 0, "lift_scatter: too many elements");
  if (N == 0) return MM_OK;
  hipLaunchKernelGGL(k_lift_scatter_key, dim3((unsigned)mm_cdiv(N * C, T)), dim3(T), 0, s, dout, C, order, skey, N, H * W, W, sb, sy, sx,
                     sc, dseg);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
