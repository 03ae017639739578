// GPU-side sample preparation: the loader-side numpy code of the reference as HIP kernels over a
// whole batch of scenes, bit-exact with the host restatement (mm2d3d_amd/voxelize.py, projection.py) and through it with
// the golden vectors generated from the reference (tests/golden/voxelize.npz).
//   a1  augment_and_scale_3d + int cast + in-range mask   lib/utils/augmentation_3d.py:83-158,
//                                                          lib/dataset/nuscenes_dataloader.py:323-332
//   a16 pixel indices, sparse depth / 2D label maps (last write wins), fliplr, RGB point features
//                                                          lib/dataset/nuscenes_dataloader.py:262-283,291-297,361-364
//   a2  collate: batch index appended as last coordinate column, per-point arrays concatenated
//                                                          lib/dataset/__init__.py:63-68,91-96
// The random draws (rotation matrix, flips, translation fractions) stay on the host in the reference's RNG order
// (mm2d3d_amd/dataprep.py); everything that touches a point runs here.
//
// Arithmetic is pinned, not approximated.  numpy evaluates the float32 product points.dot(rot) through OpenBLAS sgemm, whose
// x86 kernels accumulate the three terms with fused multiply-adds in k order: y_j = fma(p2, r2j, fma(p1, r1j, p0 * r0j))
// (checked against numpy on 35k points: every bit; the unfused left-to-right sum differs on 26 % of the elements).  The
// rest follows numpy's dtype rules: float32 for coords, `full_scale - max - 0.001` in float32, the translation in
// float64 (`clip(..) * rand(3)`), `coords += offset` as float32(double(c) + offset), astype(int64) = truncation.
#pragma clang fp contract(off)
#include "common.h"

namespace {
constexpr int T = 256;

__device__ inline float fatomic_min(float* addr, float v) {  // exact (min/max are order-independent)
  int* a = (int*)addr;
  int old = *a;
  while (__int_as_float(old) > v) {
    int assumed = old;
    old = atomicCAS(a, assumed, __float_as_int(v));
    if (old == assumed) break;
  }
  return __int_as_float(old);
}
__device__ inline float fatomic_max(float* addr, float v) {
  int* a = (int*)addr;
  int old = *a;
  while (__int_as_float(old) < v) {
    int assumed = old;
    old = atomicCAS(a, assumed, __float_as_int(v));
    if (old == assumed) break;
  }
  return __int_as_float(old);
}

// blockIdx.y = scene; cf = (points . rot) * scale; per-scene min / max of cf
__global__ __launch_bounds__(T) void k_vox_transform(const float* __restrict__ pts, const int32_t* __restrict__ scene_off,
                                                      const float* __restrict__ rot, float scale, float* __restrict__ cf,
                                                      float* __restrict__ minv, float* __restrict__ maxv) {
  const int b = blockIdx.y;
  const int64_t lo = scene_off[b], hi = scene_off[b + 1];
  const float* r = rot + b * 9;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = lo + (int64_t)blockIdx.x * T + threadIdx.x; i < hi; i += (int64_t)gridDim.x * T) {
    const float p0 = pts[i * 3 + 0], p1 = pts[i * 3 + 1], p2 = pts[i * 3 + 2];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const float y = __builtin_fmaf(p2, r[6 + j], __builtin_fmaf(p1, r[3 + j], p0 * r[j]));
      const float c = y * scale;
      cf[i * 3 + j] = c;
      mn[j] = fminf(mn[j], c);
      mx[j] = fmaxf(mx[j], c);
    }
  }
  __shared__ float smn[3][T], smx[3][T];
#pragma unroll
  for (int j = 0; j < 3; j++) smn[j][threadIdx.x] = mn[j], smx[j][threadIdx.x] = mx[j];
  __syncthreads();
  for (int s = T / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int j = 0; j < 3; j++) {
        smn[j][threadIdx.x] = fminf(smn[j][threadIdx.x], smn[j][threadIdx.x + s]);
        smx[j][threadIdx.x] = fmaxf(smx[j][threadIdx.x], smx[j][threadIdx.x + s]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 3 && hi > lo) {
    fatomic_min(&minv[b * 3 + threadIdx.x], smn[threadIdx.x][0]);
    fatomic_max(&maxv[b * 3 + threadIdx.x], smx[threadIdx.x][0]);
  }
}

// offset[b][j] = transl ? clip(float32(full_scale - (max - min)) - float32(0.001), 0) * u[b][j] : 0   (float64)
__global__ void k_vox_offset(const float* __restrict__ minv, const float* __restrict__ maxv, const double* __restrict__ u, int transl,
                             int full_scale, int B, double* __restrict__ offset) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * B) return;
  double off = 0.0;
  if (transl) {
    const float mxs = maxv[i] - minv[i];          // (coords - min).max(0): the subtraction is monotone
    float t = (float)full_scale - mxs;             // python int with a float32 array: float32
    t = t - 0.001f;                                // python float with a float32 array: float32
    t = t > 0.f ? t : 0.f;                         // np.clip(a_min=0); NaN cannot occur
    off = (double)t * u[i];
  }
  offset[i] = off;
}

// integer voxel coordinates + in-range flag of every point (blockIdx.y = scene)
__global__ __launch_bounds__(T) void k_vox_flags(const float* __restrict__ cf, const int32_t* __restrict__ scene_off,
                                                  const float* __restrict__ minv, const double* __restrict__ offset, int transl,
                                                  int full_scale, int32_t* __restrict__ ic, int32_t* __restrict__ flag) {
  const int b = blockIdx.y;
  const int64_t lo = scene_off[b], hi = scene_off[b + 1];
  for (int64_t i = lo + (int64_t)blockIdx.x * T + threadIdx.x; i < hi; i += (int64_t)gridDim.x * T) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float c = cf[i * 3 + j] - minv[b * 3 + j];
      if (transl) c = (float)((double)c + offset[b * 3 + j]);
      const long long v = (long long)c;  // astype(int64): truncation toward zero
      ok = ok && v >= 0 && v < full_scale;
      ic[i * 3 + j] = (int32_t)(v < -2147483647LL ? -2147483647LL : v > 2147483647LL ? 2147483647LL : v);
    }
    flag[i] = ok ? 1 : 0;
  }
}

// order-preserving compaction: locs[pos] = (x, y, z, scene), keep[pos] = original row; counts[b] = kept rows of scene b
__global__ __launch_bounds__(T) void k_vox_emit(const int32_t* __restrict__ ic, const int32_t* __restrict__ flag,
                                                 const int32_t* __restrict__ pos, const int32_t* __restrict__ scene_off, int B,
                                                 int64_t n_total, int64_t* __restrict__ locs, int32_t* __restrict__ keep,
                                                 int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * T + threadIdx.x;
  if (i < B) {
    const int a = scene_off[i], e = scene_off[i + 1];
    const int pa = a < n_total ? pos[a] : pos[n_total];  // pos has n_total + 1 entries (the total in the last one)
    const int pe = e < n_total ? pos[e] : pos[n_total];
    counts[i] = pe - pa;
    if (i == 0) counts[B] = pos[n_total];
  }
  if (i >= n_total || !flag[i]) return;
  int b = 0;
  while (b + 1 < B && i >= scene_off[b + 1]) b++;
  const int p = pos[i];
  locs[(int64_t)p * 4 + 0] = ic[i * 3 + 0];
  locs[(int64_t)p * 4 + 1] = ic[i * 3 + 1];
  locs[(int64_t)p * 4 + 2] = ic[i * 3 + 2];
  locs[(int64_t)p * 4 + 3] = b;
  keep[p] = (int32_t)i;
}

// ---- projection (a16)
// pixel indices = int64(points_img) (truncation), optional fliplr of the column; winner[pixel] = largest point index that
// lands there (numpy fancy assignment: the last write wins)
__global__ __launch_bounds__(T) void k_proj_index(const float* __restrict__ pimg, const int32_t* __restrict__ scene_off, int H, int W,
                                                   const uint8_t* __restrict__ flip, int64_t* __restrict__ img_indices,
                                                   int32_t* __restrict__ winner, int32_t* __restrict__ err) {
  const int b = blockIdx.y;
  const int64_t lo = scene_off[b], hi = scene_off[b + 1];
  for (int64_t i = lo + (int64_t)blockIdx.x * T + threadIdx.x; i < hi; i += (int64_t)gridDim.x * T) {
    const long long r = (long long)pimg[i * 2 + 0];
    long long c = (long long)pimg[i * 2 + 1];
    if (r < 0 || c < 0 || r >= H || c >= W) {  // the reference asserts this (nuscenes_dataloader.py:279-283)
      atomicExch(err, 1);
      img_indices[i * 2 + 0] = 0, img_indices[i * 2 + 1] = 0;
      continue;
    }
    if (flip && flip[b]) c = W - 1 - c;
    img_indices[i * 2 + 0] = r;
    img_indices[i * 2 + 1] = c;
    atomicMax(&winner[((int64_t)b * H + r) * W + c], (int32_t)i);
  }
}

__global__ __launch_bounds__(T) void k_proj_fill(const int32_t* __restrict__ winner, int64_t npix, const float* __restrict__ depth_vals,
                                                  const int64_t* __restrict__ labels, float* __restrict__ depth,
                                                  double* __restrict__ seg2d) {
  const int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
  if (p >= npix) return;
  const int w = winner[p];
  depth[p] = w >= 0 ? depth_vals[w] : 0.f;
  if (seg2d) seg2d[p] = w >= 0 ? (double)labels[w] : -100.0;
}

// the per-point arrays of the kept rows: image indices, labels, RGB features img[b, :, r, c]
__global__ __launch_bounds__(T) void k_collect(const int32_t* __restrict__ keep, const int32_t* __restrict__ n_keep, int64_t n_bound,
                                                const int64_t* __restrict__ locs, const int64_t* __restrict__ img_indices,
                                                const int64_t* __restrict__ labels, const float* __restrict__ image, int C, int H, int W,
                                                const float* __restrict__ points, int64_t* __restrict__ idx_out,
                                                int64_t* __restrict__ lab_out, float* __restrict__ feats, float* __restrict__ pts_out) {
  const int64_t p = (int64_t)blockIdx.x * T + threadIdx.x;
  if (p >= n_bound || p >= (int64_t)*n_keep) return;
  const int i = keep[p];
  const int64_t r = img_indices[(int64_t)i * 2 + 0], c = img_indices[(int64_t)i * 2 + 1];
  idx_out[p * 2 + 0] = r;
  idx_out[p * 2 + 1] = c;
  if (lab_out) lab_out[p] = labels[i];
  if (pts_out) {
    pts_out[p * 3 + 0] = points[(int64_t)i * 3 + 0];
    pts_out[p * 3 + 1] = points[(int64_t)i * 3 + 1];
    pts_out[p * 3 + 2] = points[(int64_t)i * 3 + 2];
  }
  if (feats) {
    const int64_t b = locs[p * 4 + 3];
    for (int ch = 0; ch < C; ch++) feats[p * C + ch] = image[((b * C + ch) * H + r) * W + c];
  }
}

inline unsigned scene_blocks(const int32_t* off_host, int B) {
  int64_t mx = 1;
  for (int b = 0; b < B; b++) mx = off_host[b + 1] - off_host[b] > mx ? off_host[b + 1] - off_host[b] : mx;
  int64_t g = mm_cdiv(mx, (int64_t)T * 4);
  return (unsigned)(g < 1 ? 1 : g > 1024 ? 1024 : g);
}
}  // namespace

extern "C" {

size_t mm_voxelize_ws_bytes(int64_t n_total, int B) {
  return mm_align((size_t)n_total * 3 * 4) * 2 + mm_align((size_t)(n_total + 1) * 4) * 2 + mm_align((size_t)B * 3 * 4) + mm_scan_ws_bytes(n_total + 1) + 1024;
}

// points [n_total][3] fp32 (the scenes of a batch back to back, scene b = rows scene_off[b] .. scene_off[b+1]),
// rot [B][9] fp32 row-major, u [B][3] fp64 = the np.random.rand(3) draw of each scene (ignored unless transl)
// -> locs int64 [kept][4] (x, y, z, scene; original order), keep [kept] = original row of each kept point,
//    counts [B+1] = kept points per scene and their total, min_value [B][3] fp32, offset [B][3] fp64
int mm_voxelize_batch(const float* points, const int32_t* scene_off_dev, const int32_t* scene_off_host, int B, const float* rot,
                      const double* u, int transl, float scale, int full_scale, int64_t* locs, int32_t* keep, int32_t* counts,
                      float* min_value, double* offset, void* ws, size_t ws_bytes, hipStream_t s) {
  MM_CHECK_ARG(B > 0 && B <= 65535 && full_scale > 0 && points && rot && locs && keep && counts && min_value && offset,
               "voxelize_batch: bad arguments");
  const int64_t n = scene_off_host[B];
  MMArena ar(ws, ws_bytes);
  float* cf = ar.take<float>(n * 3 + 1);
  int32_t* ic = ar.take<int32_t>(n * 3 + 1);
  int32_t* flag = ar.take<int32_t>(n + 1);
  int32_t* pos = ar.take<int32_t>(n + 1);
  float* maxv = ar.take<float>(B * 3);
  const size_t sws = mm_scan_ws_bytes(n + 1);
  char* scan_ws = ar.take<char>(sws);
  if (!cf || !ic || !flag || !pos || !maxv || !scan_ws) {
    mm_set_error("voxelize_batch: workspace too small (%zu < %zu)", ws_bytes, mm_voxelize_ws_bytes(n, B));
    return MM_ERR_WORKSPACE;
  }
  // +inf / -inf as the identities of min / max
  MM_HIP(hipMemsetD32Async((hipDeviceptr_t)min_value, 0x7F800000, (size_t)B * 3, s));
  MM_HIP(hipMemsetD32Async((hipDeviceptr_t)maxv, 0xFF800000, (size_t)B * 3, s));
  const unsigned g = scene_blocks(scene_off_host, B);
  if (n > 0)
    hipLaunchKernelGGL(k_vox_transform, dim3(g, B), dim3(T), 0, s, points, scene_off_dev, rot, scale, cf, min_value, maxv);
  hipLaunchKernelGGL(k_vox_offset, dim3((unsigned)mm_cdiv(3 * B, 64)), dim3(64), 0, s, min_value, maxv, u, transl, full_scale, B, offset);
  if (n > 0)
    hipLaunchKernelGGL(k_vox_flags, dim3(g, B), dim3(T), 0, s, cf, scene_off_dev, min_value, offset, transl, full_scale, ic, flag);
  int rc = mm_exclusive_scan_i32(flag, pos, n, pos + n, scan_ws, sws, s);
  if (rc) return rc;
  const int64_t nthreads = n > B ? n : B;
  hipLaunchKernelGGL(k_vox_emit, dim3((unsigned)mm_cdiv(nthreads, T)), dim3(T), 0, s, ic, flag, pos, scene_off_dev, B, n, locs, keep, counts);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// points_img [n_total][2] fp32 (row, col) already scaled to the network image, depth_vals [n_total] (camera z), labels may be
// NULL -> img_indices int64 [n_total][2], depth fp32 [B][H][W], seg2d fp64 [B][H][W] (optional); flip [B] bytes (optional):
// fliplr of that scene's image (the caller flips the image itself); winner: int32 [B*H*W] scratch; err: 1 = point outside
int mm_project_batch(const float* points_img, const float* depth_vals, const int64_t* labels, const int32_t* scene_off_dev,
                     const int32_t* scene_off_host, int B, int H, int W, const uint8_t* flip, int64_t* img_indices, float* depth,
                     double* seg2d, int32_t* winner, int32_t* err, hipStream_t s) {
  MM_CHECK_ARG(B > 0 && H > 0 && W > 0 && points_img && depth_vals && img_indices && depth && winner && err, "project_batch: bad arguments");
  MM_CHECK_ARG(!seg2d || labels, "project_batch: seg2d needs labels");
  const int64_t npix = (int64_t)B * H * W;
  MM_HIP(hipMemsetAsync(winner, 0xFF, (size_t)npix * 4, s));
  MM_HIP(hipMemsetAsync(err, 0, 4, s));
  if (scene_off_host[B] > 0)
    hipLaunchKernelGGL(k_proj_index, dim3(scene_blocks(scene_off_host, B), B), dim3(T), 0, s, points_img, scene_off_dev, H, W, flip,
                       img_indices, winner, err);
  hipLaunchKernelGGL(k_proj_fill, dim3((unsigned)mm_cdiv(npix, T)), dim3(T), 0, s, winner, npix, depth_vals, labels, depth, seg2d);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

// rows of the kept points (n_keep_dev = counts + B of mm_voxelize_batch, n_bound >= that value): image indices, labels,
// original points and feats[p][c] = image[scene(p)][c][row][col]  (image fp32 [B][C][H][W]); any output may be NULL
int mm_collect_points(const int32_t* keep, const int32_t* n_keep_dev, int64_t n_bound, const int64_t* locs, const int64_t* img_indices,
                      const int64_t* labels, const float* image, int C, int H, int W, const float* points, int64_t* img_indices_out,
                      int64_t* labels_out, float* feats_out, float* points_out, hipStream_t s) {
  MM_CHECK_ARG(keep && n_keep_dev && locs && img_indices && img_indices_out, "collect_points: bad arguments");
  MM_CHECK_ARG(!feats_out || image, "collect_points: feats need the image");
  if (n_bound == 0) return MM_OK;
  hipLaunchKernelGGL(k_collect, dim3((unsigned)mm_cdiv(n_bound, T)), dim3(T), 0, s, keep, n_keep_dev, n_bound, locs, img_indices, labels,
                     image, C, H, W, points, img_indices_out, labels_out, feats_out, points_out);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
