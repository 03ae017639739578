/* mm2d3d.h - C ABI of libmm2d3d_hip.so: the MI355X (gfx950) hot path of CVLAB-Unibo/MM2D3D.
 *
 * The reference has no FFI of its own: its 3D arithmetic is reached through the Python module
 * `sparseconvnet` (un-vendored dependency, /root/reference/environment.yml:37) and its 2D arithmetic through
 * torch.nn.  This header declares the entry points a binding for that path uses; each cites the reference
 * call site it replaces (EXP = /root/reference/experiments_USA_SING/rgbd_rgbxyz_sigmoid_for_rgb).
 * INTEGRATION.md shows the ctypes stub (mm2d3d_amd/_lib.py is the shipped one).
 *
 * Conventions
 *   - every pointer named *_dev or documented "device" is a HIP device pointer; *_host is host memory;
 *   - feature matrices are row-major fp32 with an explicit row stride `ld_*` in floats;
 *   - the library never allocates device memory: outputs and workspaces are supplied by the caller
 *     (size queries: mm_*_ws_bytes); all work is enqueued on `stream` and returns immediately;
 *   - no process-wide mutable state: what outlives a call (grid-barrier words, fault word, mode switches of the
 *     single-launch batch norms) lives in a per-device HANDLE (mm_create) over memory the caller supplied; entry points
 *     that use it take the handle first; modes of the stateless engines travel as explicit arguments; no entry point
 *     reads an environment variable (the Python layer reads them once and passes them on).  One host thread per handle
 *     at a time.  [The only process-wide memory are per-device "done" bits of idempotent hipFuncSetAttribute calls.]
 *   - return value 0 = ok, negative = error (MM_ERR_*), message via mm_last_error() (thread-local);
 *   - integer results (ids, rulebooks) are deterministic and follow the canonical orders of SURVEY.md A.8;
 *     floating-point reductions run in a fixed order (no float atomics): results are bit-stable run to run.
 */
#ifndef MM2D3D_H
#define MM2D3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mm_stream_t; /* hipStream_t */

#define MM_OK 0
#define MM_ERR_ARG (-1)
#define MM_ERR_HIP (-2)
#define MM_ERR_WORKSPACE (-3)
#define MM_ERR_UNSUPPORTED (-4)

const char* mm_last_error(void);

/* ---------------------------------------------------------------- per-device handle (csrc/handle.hip)
 * The reference keeps the equivalent state inside torch / cuDNN / SparseConvNet handles; this ABI exposes it. */
typedef void* mm_handle_t;
size_t mm_handle_sync_bytes(void);  /* device memory, zero-filled by the caller: 64 barrier slots of 2 KB (one per stream) */
size_t mm_handle_fault_bytes(void); /* pinned, device-mapped host memory (hipHostMalloc / torch pinned), zero-filled */
int mm_create(int device_id, void* sync_dev, size_t sync_bytes, void* fault_host, size_t fault_bytes, mm_handle_t* out);
int mm_destroy(mm_handle_t h);
#define MM_OPT_BN2D_FUSED 0   /* single-launch BatchNorm2d kernels: bit 0 = forward, bit 1 = backward; default 3 */
#define MM_OPT_BN3D_FUSED 1   /* the same for the sparse rows (mm_bn_*); default 3 */
#define MM_OPT_OS_SORT 2      /* reserved (the tile-table sort is an argument of mm_os_table_build) */
#define MM_OPT_SPCONV_TERMS 3 /* reserved (the sparse engines take their mode as an argument, MM_SPCONV_*) */
#define MM_OPT_DW_WIDE 4      /* reserved */
/* returns the previous value (>= 0) or MM_ERR_ARG.  The single-launch batch-norm kernels need every CU at once: use 0 when
 * several PROCESSES share one GPU, and clear the bit of a direction whose launches overlap kernels of another stream that
 * spin-wait across their own workgroups (collectives, look-back scans; csrc/fused_bn.h). */
int mm_set_option(mm_handle_t h, int option, int value);
int mm_get_option(mm_handle_t h, int option);
/* 1 if a single-launch batch-norm kernel launched through h gave up at its grid barrier since the last call (its grid was not
 * co-resident).  That launch's outputs are invalid; the handle's single-launch kernels are switched off (three-kernel path) and
 * its barrier words re-armed.  A host-memory read when nothing happened. */
int mm_fault_poll(mm_handle_t h);

/* ---------------------------------------------------------------- active sets and rulebooks (csrc/meta.hip)
 * Replaces the host hash-map work of scn.InputLayer(3, full_scale, mode=4) (EXP/3d_net/scn_unet.py:113,121)
 * and the rulebook construction of scn.SubmanifoldConvolution / Convolution / Deconvolution
 * (scn_unet.py:43,45,52,114 / :68-70 / :75-77). */

/* power-of-two capacity >= 2*n_items for the open-addressing table */
int64_t mm_hash_capacity(int64_t n_items);
size_t mm_dedupe_ws_bytes(int64_t n_bound);

/* Dedupe items into active sites, ids in order of first occurrence.
 *   coords        device [n,4] (x,y,z,batch), int64 when coords_is_i64 else int32; 0 <= value < 65536
 *   n_bound       rows allocated; n_dev (device int32, may be NULL) = actual row count <= n_bound
 *   shift         x,y,z >> shift before keying (0: InputLayer; 1: parents of a stride-2 Convolution)
 *   tkeys/tvals   device table [cap] (uint64 / int32), cap = mm_hash_capacity(n_bound); afterwards key -> site id
 *   item2vox      device [n]   site id of every item
 *   vox_coords    device [n,4] int32 coords of the sites (first n_active rows)
 *   csr_off/items device [n+1]/[n]  site -> its items, ascending
 *   n_active_dev  device int32 [1];  err_dev device int32 [1] set non-zero on out-of-range coordinates */
int mm_voxel_dedupe(const void* coords, int coords_is_i64, int64_t n_bound, const int32_t* n_dev, int shift,
                    uint64_t* tkeys, int32_t* tvals, int64_t cap, int32_t* item2vox, int32_t* vox_coords,
                    int32_t* csr_off, int32_t* csr_items, int32_t* n_active_dev, int32_t* err_dev, int no_spin, void* ws,
                    size_t ws_bytes, mm_stream_t stream);
/* no_spin != 0 (here and below): only kernels whose workgroups never wait for each other (plain three-launch prefix sums
 * instead of the decoupled look-back scan) - what a build on a side stream beside grid-barrier kernels needs. */

/* nbr[k*n + o] = id of the active site at coord(o) + offset(k), k = ((dx+1)*3 + (dy+1))*3 + (dz+1), or -1 */
/* out[0] = number of active rows with batch index < split.  Rows are in first-occurrence order of a batch-sorted point
 * list (collate_scn_base, lib/dataset/__init__.py:63-67), so those are rows 0 .. out[0]-1: the statistics-group boundary
 * of the batch-norm layers when source and target scenes share one pass. */
int mm_batch_lower_bound(const int32_t* vox_coords, const int32_t* n_dev, int32_t split, int32_t* out, mm_stream_t stream);
int mm_subm_neighbors(const int32_t* vox_coords, int64_t n, int32_t spatial_size, const uint64_t* tkeys,
                      const int32_t* tvals, int64_t cap, int32_t* nbr, mm_stream_t stream);

/* nbr[k*n_coarse + parent] = child with offset k = ((x&1)*2 + (y&1))*2 + (z&1), or -1 */
int mm_down_neighbors(const int32_t* vox_coords_fine, int64_t n_fine, const int32_t* fine2coarse, int64_t n_coarse,
                      int32_t* nbr, mm_stream_t stream);

size_t mm_rulebook_ws_bytes(int64_t n_out, int K);
/* neighbour table -> k-major rule lists rin/rout (capacity K*n_out, pairs sorted by out id inside a bucket),
 * offsets[K+1] (device), CSR over out rows: csr_off[n_out+1], csr_pos[R] = rule positions in ascending k.
 * csr_off = csr_pos = NULL skips the CSR (levels served by the output-stationary engine never read it);
 * mm_rulebook_csr builds it later from the same table (same workspace size). */
int mm_rulebook_compact(const int32_t* nbr, int K, int64_t n_out, int32_t* rin, int32_t* rout, int32_t* offsets,
                        int32_t* csr_off, int32_t* csr_pos, int no_spin, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_rulebook_csr(const int32_t* nbr, int K, int64_t n_out, int32_t* csr_off, int32_t* csr_pos, int no_spin, void* ws,
                    size_t ws_bytes, mm_stream_t stream);

/* ---------------------------------------------------------------- GPU-side sample preparation (csrc/dataprep.hip)
 * The loader-side numpy code of the reference for a whole batch of scenes, bit-exact with it:
 * augment_and_scale_3d + int cast + range mask (lib/utils/augmentation_3d.py:83-158, nuscenes_dataloader.py:323-332),
 * pixel indices / last-write-wins depth and 2D label maps / fliplr / RGB point features (nuscenes_dataloader.py:262-283,
 * 291-297,361-364), collate with the batch index as last coordinate column (lib/dataset/__init__.py:63-68,91-96).
 * The random draws (rotation matrix, translation fractions, flips) are made on the host in the reference's order. */
size_t mm_voxelize_ws_bytes(int64_t n_total, int B);
/* points [n_total][3] fp32 (scene b = rows scene_off[b]..scene_off[b+1]); rot [B][9] fp32; u [B][3] fp64 (rand(3) draws)
 * -> locs int64 [kept][4] (x,y,z,scene), keep int32 [kept] (original rows, order preserved), counts int32 [B+1] (per scene,
 * total), min_value fp32 [B][3], offset fp64 [B][3] */
int mm_voxelize_batch(const float* points, const int32_t* scene_off_dev, const int32_t* scene_off_host, int B, const float* rot,
                      const double* u, int transl, float scale, int full_scale, int64_t* locs, int32_t* keep, int32_t* counts,
                      float* min_value, double* offset, void* ws, size_t ws_bytes, mm_stream_t stream);
/* points_img [n_total][2] fp32 (row, col), depth_vals [n_total] -> img_indices int64 [n_total][2], depth fp32 [B][H][W],
 * seg2d fp64 [B][H][W] (optional, needs labels); flip [B] bytes (optional); winner int32 [B*H*W] scratch; err: 1 = outside */
int mm_project_batch(const float* points_img, const float* depth_vals, const int64_t* labels, const int32_t* scene_off_dev,
                     const int32_t* scene_off_host, int B, int H, int W, const uint8_t* flip, int64_t* img_indices, float* depth,
                     double* seg2d, int32_t* winner, int32_t* err, mm_stream_t stream);
/* per-point arrays of the kept rows; feats[p][c] = image[scene(p)][c][row][col] (image fp32 [B][C][H][W]) */
int mm_collect_points(const int32_t* keep, const int32_t* n_keep_dev, int64_t n_bound, const int64_t* locs, const int64_t* img_indices,
                      const int64_t* labels, const float* image, int C, int H, int W, const float* points, int64_t* img_indices_out,
                      int64_t* labels_out, float* feats_out, float* points_out, mm_stream_t stream);

/* ---------------------------------------------------------------- sparse convolution engines (csrc/spconv.hip)
 * scn.SubmanifoldConvolution / Convolution / Deconvolution forward and backward. */
size_t mm_spconv_ws_bytes(int64_t n_rules, int Cin, int Cout, int K);
/* out[dst[r]] (+)= in[src[r]] . W[k(r)] over a k-major rulebook (offsets_host = host copy of offsets[K+1]).
 *   unique_dst != 0: every destination row has exactly one rule (direct writes);
 *   else destinations are reduced through csr_off/csr_pos in ascending k and rows without rules become 0.
 *   weight element (k,ci,co) = W[kk*w_kstride + ci*s_ci + co*s_co], kk = kflip ? K-1-k : k.
 *   mode (fp32 rows): MM_SPCONV_DEFAULT = fp32-faithful three-term split-bf16 products on the matrix-rate-bound widths;
 *   MM_SPCONV_FP32 = plain fp32 engines everywhere; MM_SPCONV_TWO_TERMS = two terms (faster; fails the gradient parity bar:
 *   diagnostics only); | MM_SPCONV_DW_NARROW: the weight gradient keeps its <= 4 x 4 channel tiles. */
#define MM_SPCONV_DEFAULT 0
#define MM_SPCONV_FP32 1
#define MM_SPCONV_TWO_TERMS 2
#define MM_SPCONV_DW_NARROW 4
#define MM_SPCONV_DW16_ELEM 8 /* 16-bit rows: the weight gradient gathers single elements (the round-3 kernel) instead of whole rows
                              * through LDS and transpose reads (round 6, same slab sums): A/B measurements */
int mm_spconv_apply(const float* in, int ld_in, int Cin, float* out, int ld_out, int Cout, int64_t n_out,
                    const int32_t* src, const int32_t* dst, const int32_t* offsets_dev, const int32_t* offsets_host,
                    int K, const int32_t* csr_off, const int32_t* csr_pos, int unique_dst, const float* W,
                    int64_t w_kstride, int s_ci, int s_co, int kflip, int mode, void* ws, size_t ws_bytes, mm_stream_t stream);
/* The same with the weights' three-term bf16 fragments supplied by the caller (Wpk: written by mm_spconv_os_pack /
 * mm_spconv_os_pack_batch for the same K, Cin, Cout, strides and kflip; NULL = pack inside the call): a net packs every
 * layer once per optimiser step in one launch instead of once per layer call. */
int mm_spconv_apply_packed(const float* in, int ld_in, int Cin, float* out, int ld_out, int Cout, int64_t n_out,
                           const int32_t* src, const int32_t* dst, const int32_t* offsets_dev, const int32_t* offsets_host,
                           int K, const int32_t* csr_off, const int32_t* csr_pos, int unique_dst, const float* W,
                           int64_t w_kstride, int s_ci, int s_co, int kflip, const void* Wpk, int mode, void* ws, size_t ws_bytes,
                           mm_stream_t stream);
size_t mm_spconv_dw_ws_bytes(const int32_t* offsets_host, int K, int Cin, int Cout);
/* dW[k][ci][co] (+)= sum over rules r of bucket k: in[src[r]][ci] * dout[dst[r]][co] */
int mm_spconv_dw(const float* in, int ld_in, int Cin, const float* dout, int ld_do, int Cout, const int32_t* src,
                 const int32_t* dst, const int32_t* offsets_host, int K, float* dW, int accumulate, int mode, void* ws,
                 size_t ws_bytes, mm_stream_t stream);

/* ---------------------------------------------------------------- output-stationary engine (csrc/ostable.hip, csrc/osconv.hip)
 * The same three scn convolutions (scn_unet.py:43,45,52,68-70,75-77,114), forward and data gradient, without the
 * tmp[rule] round trip: destination rows are ordered by neighbour bitmask, a workgroup owns a tile of them and
 * accumulates the offsets k in ascending order in registers. */
/* nbr[8][n_fine] for Deconvolution / d(Convolution)/d(input): nbr[octant(i)][i] = parent of fine row i, else -1 */
int mm_up_neighbors(const int32_t* vox_coords_fine, int64_t n_fine, const int32_t* fine2coarse, int32_t* nbr,
                    mm_stream_t stream);
size_t mm_os_table_ws_bytes(int64_t n, int K);
/* nbr[K][n] -> dst[npad] (rows sorted by neighbour bitmask, -1 = padding), nbrp[K][npad], tmask[nt];
 * nt = ceil(n / tile_rows), npad = nt * tile_rows, tile_rows a multiple of 64 (the engine takes 64) */
/* sort_merge: 0 = rocPRIM Onesweep radix sort (default), 1 = merge sort (identical result; its workgroups do not wait for each
 * other, which makes the build safe on a stream that runs beside grid-barrier kernels, see MM_OPT_BN2D_FUSED) */
int mm_os_table_build(const int32_t* nbr, int K, int64_t n, int tile_rows, int sort_merge, int32_t* dst, int32_t* nbrp,
                      uint32_t* tmask, void* ws, size_t ws_bytes, mm_stream_t stream);
/* three-term bf16 MFMA fragments of a weight tensor: element (k,ci,co) = W[kk*w_kstride + ci*s_ci + co*s_co],
 * kk = kflip ? K-1-k : k */
size_t mm_spconv_os_pack_bytes(int K, int Cin, int Cout);
int64_t mm_spconv_os_pack_blocks(int K, int Cin, int Cout);
int mm_spconv_os_pack_desc_fields(void);
int mm_spconv_os_pack(const float* W, int64_t w_kstride, int s_ci, int s_co, int kflip, int K, int Cin, int Cout, void* Wf,
                      mm_stream_t stream);
/* descs_dev [n_desc][12] int64 (device) = {W, Wf, K, Cin, Cout, nq, ncb, w_kstride, s_ci, s_co, kflip, blk_end}: one launch
 * packs every weight of a net after an optimiser step */
int mm_spconv_os_pack_batch(const int64_t* descs_dev, int n_desc, int64_t total_blocks, mm_stream_t stream);
/* out[dst[j]] = sum over present k (ascending) of in[nbrp[k][j]] . W[k];  Cin, Cout multiples of 16, rows 16-B aligned */
int mm_spconv_os_apply(const float* in, int ld_in, int Cin, float* out, int ld_out, int Cout, const void* Wf, int K,
                       const int32_t* dst, const int32_t* nbrp, const uint32_t* tmask, int64_t n_tiles, int tile_rows,
                       mm_stream_t stream);

/* 16-bit activation mode (BASELINE.json configs[4]): the sparse rows are bf16 (in and out), the
 * weights one bf16 term per element, accumulation fp32 in ascending k.  Same tables as above, tile_rows = 64. */
size_t mm_spconv_os_pack_bytes_bf16(int K, int Cin, int Cout);
int mm_spconv_os_pack_bf16(const float* W, int64_t w_kstride, int s_ci, int s_co, int kflip, int K, int Cin, int Cout, void* Wf,
                           mm_stream_t stream);
int mm_spconv_os_pack_batch_bf16(const int64_t* descs_dev, int n_desc, int64_t total_blocks, mm_stream_t stream);
int mm_spconv_os_apply_bf16(const void* in, int ld_in, int Cin, void* out, int ld_out, int Cout, const void* Wf, int K,
                            const int32_t* dst, const int32_t* nbrp, const uint32_t* tmask, int64_t n_tiles, int tile_rows,
                            mm_stream_t stream);
/* dW[k][ci][co] (+)= sum over the rules of bucket k of in[src][ci] * dout[dst][co]; in / dout bf16 rows, dW fp32;
 * workspace of mm_spconv_dw_ws_bytes */
int mm_spconv_dw_bf16(const void* in, int ld_in, int Cin, const void* dout, int ld_do, int Cout, const int32_t* src,
                      const int32_t* dst, const int32_t* offsets_host, int K, float* dW, int accumulate, int mode, void* ws,
                      size_t ws_bytes, mm_stream_t stream);
/* The fp16 kind of the 16-bit activation mode (BASELINE.json configs[4] "fp16 activations"; the reference trains with
 * ``precision: 16`` = fp16 autocast + GradScaler, train.yaml:11): the sparse rows are IEEE fp16, the weights one fp16 term per
 * element (fragment sizes: mm_spconv_os_pack_bytes_bf16), products on v_mfma_f32_16x16x32_f16, accumulation fp32 in
 * ascending k.  Gradient rows need loss scaling (mm2d3d_amd/amp.py). */
int mm_spconv_os_pack_f16(const float* W, int64_t w_kstride, int s_ci, int s_co, int kflip, int K, int Cin, int Cout, void* Wf,
                          mm_stream_t stream);
int mm_spconv_os_pack_batch_f16(const int64_t* descs_dev, int n_desc, int64_t total_blocks, mm_stream_t stream);
int mm_spconv_os_apply_f16(const void* in, int ld_in, int Cin, void* out, int ld_out, int Cout, const void* Wf, int K,
                           const int32_t* dst, const int32_t* nbrp, const uint32_t* tmask, int64_t n_tiles, int tile_rows,
                           mm_stream_t stream);
int mm_spconv_dw_f16(const void* in, int ld_in, int Cin, const void* dout, int ld_do, int Cout, const int32_t* src,
                     const int32_t* dst, const int32_t* offsets_host, int K, float* dW, int accumulate, int mode, void* ws,
                     size_t ws_bytes, mm_stream_t stream);
/* The weight gradient in two calls: mm_spconv_dw_partial writes the partial slabs of ONE layer (fp32 rows; bf16 rows when
 * bf16 == 1, IEEE fp16 rows when bf16 == 2) into ``partial`` (mm_spconv_dw_ws_bytes; must stay untouched until the reduce) and the 33 slab offsets of the
 * layer into ``blk_start_host``; mm_spconv_dw_reduce_batch sums the slabs of n layers in ONE launch.  descs_dev: n rows of
 * mm_spconv_dw_desc_bytes() bytes {const float* partial; float* dW; int32 ne = Cin*Cout, K, accumulate, blk_first;
 * int32 blk_start[33]}, blk_first = sum of mm_spconv_dw_reduce_blocks(ne, K) over the preceding rows, total_blocks = that sum over all rows.
 * Bit-identical to mm_spconv_dw (same slabs, same order).  Replaces the per-layer tail of scn's ConvolutionFunction /
 * SubmanifoldConvolutionFunction backward (reference call sites scn_unet.py:43-52,68-77,114). */
int mm_spconv_dw_partial(int bf16, const void* in, int ld_in, int Cin, const void* dout, int ld_do, int Cout, const int32_t* src,
                         const int32_t* dst, const int32_t* offsets_host, int K, int mode, void* partial, size_t partial_bytes,
                         int32_t* blk_start_host, mm_stream_t stream);
int mm_spconv_dw_desc_bytes(void);
int64_t mm_spconv_dw_reduce_blocks(int ne, int K);
int mm_spconv_dw_reduce_batch(const void* descs_dev, int n, int64_t total_blocks, mm_stream_t stream);

/* ---------------------------------------------------------------- batch norm + (leaky) ReLU (csrc/bn.hip)
 * scn.BatchNormReLU / BatchNormLeakyReLU (scn_unet.py:42,44,51,66,73,116); momentum = keep fraction (0.9). */
/* Row sets that fit on chip take single-launch training kernels (one workgroup per CU, rows kept on chip across two grid
 * barriers) when the handle allows it: mm_set_option(h, MM_OPT_BN3D_FUSED, mask), bit 0 = forward, bit 1 = backward. */
size_t mm_bn_ws_bytes(int C);
/* Ns: rows [0,Ns) and [Ns,N) (the active sites of the source and of the target scenes of a jointly batched step;
 * train.py:186-292 calls the net once per domain) are normalised with their OWN batch statistics and the running
 * buffers are updated group 0 first, then group 1.  Ns = N (or 0): single batch.  save_mean/save_invstd: [G][C]. */
int mm_bn_fwd_train(mm_handle_t h, const float* x, int ld_x, int64_t N, int64_t Ns, int C, const float* weight, const float* bias,
                    float* running_mean, float* running_var, float eps, float momentum, float leak, float* y, int ld_y,
                    float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn_fwd_eval(const float* x, int ld_x, int64_t N, int C, const float* weight, const float* bias,
                   const float* running_mean, const float* running_var, float eps, float leak, float* y, int ld_y,
                   mm_stream_t stream);
int mm_bn_bwd(mm_handle_t h, const float* x, int ld_x, const float* dy, int ld_dy, int64_t N, int64_t Ns, int C, const float* weight,
              const float* bias, const float* save_mean, const float* save_invstd, float leak, float* dx, int ld_dx,
              float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, mm_stream_t stream);
/* the same three entry points over bf16 rows (16-bit activation mode): x / y / dy / dx bf16 [N, C] (ld in elements),
 * statistics and parameters fp32 */
int mm_bn_fwd_train_bf16(mm_handle_t h, const void* x, int ld_x, int64_t N, int64_t Ns, int C, const float* weight, const float* bias,
                         float* running_mean, float* running_var, float eps, float momentum, float leak, void* y, int ld_y,
                         float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn_fwd_eval_bf16(const void* x, int ld_x, int64_t N, int C, const float* weight, const float* bias,
                        const float* running_mean, const float* running_var, float eps, float leak, void* y, int ld_y,
                        mm_stream_t stream);
int mm_bn_bwd_bf16(mm_handle_t h, const void* x, int ld_x, const void* dy, int ld_dy, int64_t N, int64_t Ns, int C, const float* weight,
                   const float* bias, const float* save_mean, const float* save_invstd, float leak, void* dx, int ld_dx,
                   float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, mm_stream_t stream);
/* ... and over IEEE fp16 rows (the fp16 kind of the 16-bit activation mode: train.yaml:11 ``precision: 16``) */
int mm_bn_fwd_train_f16(mm_handle_t h, const void* x, int ld_x, int64_t N, int64_t Ns, int C, const float* weight, const float* bias,
                        float* running_mean, float* running_var, float eps, float momentum, float leak, void* y, int ld_y,
                        float* save_mean, float* save_invstd, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn_fwd_eval_f16(const void* x, int ld_x, int64_t N, int C, const float* weight, const float* bias,
                       const float* running_mean, const float* running_var, float eps, float leak, void* y, int ld_y,
                       mm_stream_t stream);
int mm_bn_bwd_f16(mm_handle_t h, const void* x, int ld_x, const void* dy, int ld_dy, int64_t N, int64_t Ns, int C, const float* weight,
                  const float* bias, const float* save_mean, const float* save_invstd, float leak, void* dx, int ld_dx,
                  float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, mm_stream_t stream);

/* ---------------------------------------------------------------- per-point rows (csrc/point.hip) */
size_t mm_point_ws_bytes(int Cin, int Cout);
/* y = x * sigmoid(x.w + b), mask = sigmoid(...)   (EXP/3d_net/model.py:46-48) */
int mm_gate_fwd(const float* x, int64_t N, int C, const float* w, const float* b, float* y, float* mask,
                mm_stream_t stream);
int mm_gate_bwd(const float* x, const float* mask, const float* dy, int64_t N, int C, const float* w, float* dx,
                float* dw, float* db, int accumulate, void* ws, size_t ws_bytes, mm_stream_t stream);
/* site rows = mean (or sum) of their items' rows: InputLayer mode 4 / 3 forward, OutputLayer backward */
int mm_segment_reduce(const float* feats, int ld_f, int C, const int32_t* csr_off, const int32_t* csr_items,
                      int64_t n_vox, int mean, float* out, int ld_o, mm_stream_t stream);
/* item rows = their site's row (optionally / count): OutputLayer forward (scn_unet.py:117), InputLayer backward */
int mm_row_gather(const float* vox, int ld_v, int C, const int32_t* p2v, const int32_t* csr_off, int div_count,
                  int64_t N, float* out, int ld_o, mm_stream_t stream);
/* y = x W^T + b with torch nn.Linear layout W [Cout, Cin]  (EXP/3d_net/model.py:50,85) */
int mm_linear_fwd(const float* x, int ld_x, int64_t N, int Cin, int Cout, const float* w, const float* b, float* y,
                  int ld_y, mm_stream_t stream);
int mm_linear_bwd(const float* x, int ld_x, const float* dy, int ld_dy, int64_t N, int Cin, int Cout, const float* w,
                  float* dx, int ld_dx, int accumulate_dx, float* dw, float* db, int accumulate_w, void* ws,
                  size_t ws_bytes, mm_stream_t stream);

/* ---------------------------------------------------------------- losses, lifting, optimiser (csrc/loss.hip) */
size_t mm_loss_ws_bytes(void);
/* weighted cross entropy, ignore_index, weighted-mean reduction (lib/losses.py:55-68 -> F.cross_entropy).
 * stats[0] = loss, stats[1] = sum of the class weights of the counted rows */
int mm_ce_fwd(const float* logits, int ld, const int64_t* labels, const float* weight, int64_t N, int C,
              int64_t ignore_index, float* stats, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_ce_bwd(const float* logits, int ld, const int64_t* labels, const float* weight, int64_t N, int C,
              int64_t ignore_index, const float* stats, const float* grad_out, float* dlogits, int ld_d,
              mm_stream_t stream);
/* mean_i sum_c softmax(tgt)_ic (log softmax(tgt)_ic - log softmax(pred)_ic)  (EXP/train.py:157-184) */
int mm_kl_fwd(const float* pred, int ld_p, const float* tgt, int ld_t, int64_t N, int C, float* loss, void* ws,
              size_t ws_bytes, mm_stream_t stream);
int mm_kl_bwd(const float* pred, int ld_p, const float* tgt, int ld_t, int64_t N, int C, const float* grad_out,
              float* dpred, int ld_d, mm_stream_t stream);
/* 2D->3D lifting (EXP/2d_net/model.py:131-137,166-173): out[p][c] = seg[pix_off[p] + c*chan_stride];
 * backward: dseg[upix_off[u] + c*chan_stride] = sum over the points of unique pixel u (CSR, ascending) */
int mm_lift_gather(const float* seg, int64_t chan_stride, const int64_t* pix_off, int64_t N, int C, float* out,
                   mm_stream_t stream);
int mm_lift_scatter(const float* dout, int C, const int64_t* upix_off, const int32_t* csr_off,
                    const int32_t* csr_pts, int64_t n_unique, int64_t chan_stride, float* dseg, mm_stream_t stream);
/* backward of the lifting without compaction: order = stable argsort of the pixel keys, first[e] marks run starts */
int mm_lift_scatter_runs(const float* dout, int C, const int64_t* order, const unsigned char* first,
                         const int64_t* sorted_off, int64_t N, int64_t chan_stride, float* dseg, mm_stream_t stream);
/* The lifting index built on the device in three launches (csrc/lift.hip; round 4).  rc: device int64 [n, 2] (row, col) of every
 * point, scenes concatenated (img_indices of lib/dataset/__init__.py:74,111); counts: device int64 [nb] points per scene.
 * key [n] = pixel id (b*H + row)*W + col (rows / cols clamped into the map; err[0] = 1 if one was outside - the reference asserts
 * these bounds in its loader, nuscenes_dataloader.py:280-283); skey / order [n] = the keys ascending and the point at each sorted
 * position (stable radix sort).  no_spin: see mm_voxel_dedupe. */
size_t mm_lift_index_ws_bytes(int64_t n);
int mm_lift_index(const int64_t* rc, const int64_t* counts, int nb, int64_t n, int H, int W, int no_spin, int32_t* key, int32_t* skey,
                  int32_t* order, int32_t* err, void* ws, size_t ws_bytes, mm_stream_t stream);
/* out[p][c] = seg[b*sb + row*sy + col*sx + c*sc] at the pixel of point p (2d_net/model.py:131-137) */
int mm_lift_gather_key(const float* seg, int64_t sb, int64_t sy, int64_t sx, int64_t sc, const int32_t* key, int64_t N, int C, int H,
                       int W, float* out, mm_stream_t stream);
/* its backward: dseg (zero-filled by the caller) at every pixel with points = the sum of dout over them, ascending point order */
int mm_lift_scatter_key(const float* dout, int C, const int32_t* order, const int32_t* skey, int64_t N, int H, int W, int64_t sb,
                        int64_t sy, int64_t sx, int64_t sc, float* dseg, mm_stream_t stream);
/* evaluation (EXP/train.py:297-339): confusion matrices [3][C][C] int64 of argmax(2D), argmax(3D), argmax(softmax mean) */
int mm_eval_confusion(const float* logits2d, int ld2, const float* logits3d, int ld3, const int64_t* labels, int64_t N,
                      int C, int64_t ignore_index, int64_t* cm, mm_stream_t stream);
/* torch.optim.AdamW update over flat fp32 arenas (EXP/train.py:627-636); step counts from 1; g is multiplied by grad_scale.
 * skip_dev / nskip (0 .. 16, may be NULL / 0): device words - the update is a no-op when any of them is nonzero (the
 * data-parallel reducer's collective "this step's gradients are invalid" flags: decided on the device, no read-back) */
int mm_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int64_t step, double grad_scale, const int* skip_dev, int nskip,
                  mm_stream_t stream);
/* Loss-scaled steps (the fp16 kind of the 16-bit activation mode; torch.cuda.amp.GradScaler semantics as driven by the
 * reference's ``precision: 16`` trainer, train.yaml:11) WITHOUT a read-back: scale, non-finite flag, clean-step tracker and
 * step counter live on the device.  mm_grad_nonfinite: found_dev[0] = 1 if any gradient is inf / nan (the caller zeroes it);
 * mm_amp_prepare: the coefficients of one parameter group's update (mm_amp_coef_bytes bytes) incl. 1 / scale and "skip" =
 * any of found_dev[0 .. nfound) set - the flags of EVERY optimiser of the step (the reference's HybridOptim is one optimiser to
 * the GradScaler, EXP/train.py:627-636: an overflow in either network skips both updates) plus the caller's extra skip words;
 * mm_adamw_step_dev: mm_adamw_step with those coefficients, a no-op when skip is set; mm_amp_update: GradScaler.update(). */
int mm_grad_nonfinite(const float* g, int64_t n, int* found_dev, mm_stream_t stream);
int mm_amp_coef_bytes(void);
int mm_amp_prepare(const float* scale_dev, const int* found_dev, int nfound, int64_t* step_dev, int advance, double lr, double beta1,
                   double beta2, double eps, double weight_decay, double grad_scale, void* coef_dev, mm_stream_t stream);
int mm_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const void* coef_dev, mm_stream_t stream);
int mm_amp_update(float* scale_dev, int* tracker_dev, const int* found_dev, int nfound, double growth, double backoff, int interval,
                  mm_stream_t stream);

/* ---------------------------------------------------------------- dense 2D convolutions (csrc/conv2d.hip)
 * torch.nn.Conv2d / ConvTranspose2d of EXP/2d_net/backbones.py:43-65 and EXP/2d_net/model.py:64-82,104-123.
 * NHWC bf16 activations, fp32 accumulate.  mm_conv2d_gemm: out[m][n] = sum_{tap,k} A[src(m,tap)][k] * Wp[n][tap][k]
 * with m -> (b,gy,gx) on the base grid Hg x Wg, src = ((gy*sa+ty)/fr, (gx*sa+tx)/fr), out pixel = (gy*so+ooy(+z>>1),
 * gx*so+oox(+z&1)); covers Conv2d fwd / dgrad (stride 1, 2) and ConvTranspose2d(k2,s2) fwd / dgrad. */
int mm_conv2d_gemm(const void* A, int B, int Hi, int Wi, int Ca, int lda, void* O, int Ho, int Wo, int Cn, int ldo,
                   int out_f32, int Hg, int Wg, int so, int ooy, int oox, int sa, int fr, int ntaps, const int* ty,
                   const int* tx, const void* Wp, int nz, int64_t wz, int zpar, const float* bias, float* stats,
                   int64_t split_m, const void* addend, int ld_add, mm_stream_t stream);
/* ``addend`` (16-bit output only): a 16-bit map of the output's shape (pixel pitch ld_add) added to the result, element by element
 * as an add of the two maps would - the second gradient contribution of a map with two consumers (a BasicBlock input read by conv1
 * and by the 1x1 downsample; the decoder's concat slice) joins here instead of in an add kernel. */
/* The data gradient of a STRIDE-2 convolution (k = 1 or 3; EXP/2d_net/backbones.py layer2-4.0: conv1 and the 1x1 downsample) by output
 * parity: an input pixel of parity (py, px) receives only the taps with kh = py + pad, kw = px + pad (mod 2) - 1 + 2 + 2 + 4 of a 3x3
 * filter's nine, 1 + 0 + 0 + 0 of a 1x1's - so the four parities run as four tap windows of one launch instead of every tap for
 * every pixel with three quarters of them multiplying zeros (mm_conv2d_gemm with fr = 2).  dY [B,Ho,Wo,Cout] (pitch ldy), dX
 * [B,H,W,Cin] (pitch ldx; H, W even), Wd [Cin][k*k][Cout]; addend as in mm_conv2d_gemm.  Bit-identical with the generic form. */
int mm_conv2d_dgrad_s2(const void* dY, int B, int Ho, int Wo, int Cout, int ldy, void* dX, int H, int W, int Cin, int ldx, const void* Wd,
                       int k, int pad, const void* addend, int ld_add, mm_stream_t stream);
/* BatchNorm statistics in the epilogue (``stats`` non-NULL; 16-bit output only): the convolution also files, per 64-pixel
 * sub-block of its output and per statistics group, the per-channel sum and sum of squares of the ROUNDED outputs in
 *     stats[2 * sub + g][q][Cn] fp32   (q = 0: sum, 1: sum of squares; rows = mm_conv2d_gemm_stat_rows / _3x3s1_stat_rows)
 * every element of which is written by exactly one lane (no atomics, no zero fill needed).  Group 0 = GEMM rows [0, split_m)
 * (mm_conv2d_gemm, rows within one z slice) / batch entries [0, split_b) (mm_conv2d_3x3s1), group 1 = the others: the
 * [source | target] halves of the joint pass (xmuda.py:45-46 calls the network once per domain).  mm_bn2d_fwd_train_pre turns
 * the slab into batch statistics - the training BatchNorm2d behind a convolution then never reads the map for its sums. */
int64_t mm_conv2d_gemm_stat_rows(int64_t M, int nz);
int64_t mm_conv2d_3x3s1_stat_rows(int B, int H, int W);
/* 3x3 stride-1 pad-1 convolution (flip 0) or its data gradient (flip 1, Wp = [ci][tap][co]) from a halo tile staged once
 * for all 9 taps (EXP/2d_net/backbones.py ResNet34 BasicBlocks; EXP/2d_net/model.py:68-71 decoder convolutions).
 * flip | 2: a ragged last round of work items is NOT cut into half items (A/B measurements; results are the same sums).
 * flip bits 2-3 choose the kernel (A/B measurements and the identity tests; the results are the same sums):
 *   0  the round-6 kernels: k_conv3x3s (four multiplying + four loader waves, 16x16x32 MFMAs; 64 -> 64 layers with the nine weight
 *      tiles resident in LDS) - equal to the others up to the fp32 summation order inside one 32-deep product;
 *   4  the round-2 kernel k_conv3x3w (16 waves, 32x32x16 MFMAs; 64 -> 64: the round-3 weights-resident k_conv3x3r);
 *   8  k_conv3x3v (the 8-wave layout on 32x32x16 MFMAs): bit-identical with 4;
 *  12  k_conv3x3s in its streaming form also for 64 -> 64. */
int mm_conv2d_3x3s1(const void* A, int B, int H, int W, int Ca, int lda, void* O, int Cn, int ldo, const void* Wp,
                    const float* bias, int flip, float* stats, int split_b, mm_stream_t stream);
/* Two such convolutions (or data gradients) of ONE shape - the same layer of the RGB and of the depth encoder, EXP/2d_net/model.py:43-46 -
 * in one launch: one work-item list over both problems, so the persistent kernel's partly filled last round is shared (at the
 * bench's sizes a 256-channel layer alone leaves half the chip idle for a third of its time).  Same results as two calls. */
int mm_conv2d_3x3s1_pair(const void* A0, const void* A1, int B, int H, int W, int Ca, int lda, void* O0, void* O1, int Cn, int ldo,
                         const void* Wp0, const void* Wp1, int flip, float* stats0, float* stats1, int split_b, mm_stream_t stream);
size_t mm_conv2d_wgrad_ws_bytes(int64_t M, int Cn, int Ck, int ntaps);
/* dW[n*sn + t*st + k*sk] (+)= sum_m dY[m][n] * X[src(m,t)][k], base grid = dY pixels */
int mm_conv2d_wgrad(const void* X, int B, int Hi, int Wi, int Ck, int ldx, const void* dY, int Hg, int Wg, int Cn,
                    int ldy, int sa, int ntaps, const int* ty, const int* tx, float* dW, int64_t sn, int64_t st,
                    int64_t sk, int accumulate, void* ws, size_t ws_bytes, mm_stream_t stream);
/* The weight gradients of TWO 3x3 stride-1 pad-1 convolutions of one shape (the same layer of the two encoders) in the two launches
 * one of them takes: twice the pixels per workgroup, half the partial slabs per problem.  ws: mm_conv2d_wgrad_ws_bytes(B*H*W, Cn, Ck, 9). */
int mm_conv2d_wgrad3x3_pair(const void* X0, const void* X1, int B, int H, int W, int Ck, int ldx, const void* dY0, const void* dY1, int Cn,
                            int ldy, float* dW0, float* dW1, int64_t sn, int64_t st, int64_t sk, int accumulate, void* ws, size_t ws_bytes,
                            mm_stream_t stream);
/* The same two weight-gradient forms WITHOUT their slab sums (round 5): ``slabs`` (mm_conv2d_wgrad_ws_bytes bytes; the caller keeps
 * it until the sum has run) receives the fp32 partial slabs [*nsplit][Cn][ntaps][Ck] (pair: [2][*nsplit][Cn][9][Ck]), and ONE
 * mm_conv2d_wgrad_reduce_batch launch later sums the slabs of every layer of a backward pass into their gradients - the per-layer
 * sums were ~50 launches of ~14 us per training step.  Same sums in the same order: bit-identical with mm_conv2d_wgrad.
 * descs_dev: n descriptors of mm_conv2d_wgrad_reduce_desc_bytes() bytes on the device: {const float* slabs; float* dW; float* dW1
 * (the pair's second gradient, else NULL); int64 sn, st, sk; int32 nsplit, Cn, ntaps, Ck, accumulate, blk_first}, blk_first = the
 * running sum of mm_conv2d_wgrad_reduce_blocks(Cn, Ck, pair) over the preceding descriptors, total_blocks = the sum over all. */
int mm_conv2d_wgrad_slabs(const void* X, int B, int Hi, int Wi, int Ck, int ldx, const void* dY, int Hg, int Wg, int Cn, int ldy,
                          int sa, int ntaps, const int* ty, const int* tx, void* slabs, size_t slab_bytes, int* nsplit,
                          mm_stream_t stream);
int mm_conv2d_wgrad3x3_pair_slabs(const void* X0, const void* X1, int B, int H, int W, int Ck, int ldx, const void* dY0, const void* dY1,
                                  int Cn, int ldy, void* slabs, size_t slab_bytes, int* nsplit, mm_stream_t stream);
int mm_conv2d_wgrad_reduce_desc_bytes(void);
int64_t mm_conv2d_wgrad_reduce_blocks(int Cn, int Ck, int pair);
int mm_conv2d_wgrad_reduce_batch(const void* descs_dev, int n, int64_t total_blocks, mm_stream_t stream);
/* The 7x7 stride-1 stems (EXP/2d_net/backbones.py:23-25) on the staged image of mm_stem_prep: xb [B][Hb][Wb][8] with R = 8 / C image
 * rows stacked per buffer pixel, T = ceil(7 / R) taps of 8 pixels x 8 slots, Wp [64][T][64]; output O [B][H][W][64] (pitch ldo).
 * One persistent kernel with the weights resident in LDS and the RAW strip of a 16 x 16 tile staged once - the generic implicit GEMM
 * fetched 128 bytes per output pixel and tap, 16 times the strip's bytes.  stats / split_b: BatchNorm statistics slab as for
 * mm_conv2d_3x3s1, mm_conv2d_stem7_stat_rows(B, H, W) rows. */
int64_t mm_conv2d_stem7_stat_rows(int B, int H, int W);
int mm_conv2d_stem7(const void* xb, int B, int Hb, int Wb, int H, int W, int R, int T, void* O, int ldo, const void* Wp, float* stats,
                    int split_b, mm_stream_t stream);
/* stem input (EXP/2d_net/backbones.py:23-25, 7x7 stride-1 conv on 3 / 1 channels): NCHW fp32 -> zero-bordered NHWC8 bf16 */
int mm_stem_prep(const float* in, int B, int C, int H, int W, int pad, int Hb, int Wb, int R, void* out, mm_stream_t stream);
/* out[((z*N+n)*T+t)*K+k] = bf16(in[z*sz + n*sn + t*st + k*sk]) : fp32 master weights -> kernel layouts */
int mm_pack_weights_bf16(const float* in, void* out, int Z, int N, int T, int K, int64_t sz, int64_t sn, int64_t st,
                         int64_t sk, mm_stream_t stream);
/* One launch for a table of weights (every conv layer of a model after an optimiser step).  desc (device memory):
 * ndesc rows of 11 int64 {in, out, Z, N, T, K, sz, sn, st, sk, first_block}; first_block = running sum of
 * ceil(Z*N*T*K / 4096) over the preceding rows (Z*N*T*K < 2^31 per row), total_blocks = the sum over all rows. */
int mm_pack_weights_bf16_batch(const int64_t* desc, int ndesc, int64_t total_blocks, mm_stream_t stream);

/* ---------------------------------------------------------------- BatchNorm2d (+residual) (+ReLU), NHWC bf16 (csrc/bn2d.hip) */
size_t mm_bn2d_ws_bytes(int C);
/* Maps that fit on chip (every map of the headline step but the largest ones) take single-launch training kernels:
 * one workgroup per CU keeps its rows in registers / LDS across two grid barriers, so x (and dy) are read once - when the handle
 * allows it: mm_set_option(h, MM_OPT_BN2D_FUSED, mask), bit 0 = mm_bn2d_fwd_train, bit 1 = mm_bn2d_bwd, 0 = always the
 * reduce / finalize / apply kernels (see mm_set_option for when). */
/* Ns: rows [0,Ns) and [Ns,N) are normalised with their OWN batch statistics (the source / target halves of a jointly
 * batched step; train.py:186-292 calls each net once per domain) and the running buffers are updated group 0 first,
 * then group 1, as two consecutive calls would.  Ns = N (or 0): ordinary single batch.  save_mean/save_invstd: [G][C]. */
int mm_bn2d_fwd_train(mm_handle_t h, const void* x, int ld_x, const void* res, int ld_r, int64_t N, int64_t Ns, int C, const float* weight,
                      const float* bias, float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps,
                      float momentum, int relu, void* y, int ld_y, float* save_mean, float* save_invstd, void* ws,
                      size_t ws_bytes, mm_stream_t stream);
/* The same with the batch statistics taken from the slab the producing convolution filled (mm_conv2d_gemm / mm_conv2d_3x3s1
 * ``stats``, slab_rows = its *_stat_rows): a finalize launch (fp64 sums in a fixed order) + one streaming apply pass.  No pass
 * over x for the sums, no grid barrier, hence no handle and no residency rule. */
/* 1 when mm_bn2d_fwd_train (backward = 0) / mm_bn2d_bwd (1) would run as ONE launch for N rows of C channels under the handle's
 * present options (shape rule only).  The Python layer asks before a convolution: maps too large for one launch get their
 * statistics from the convolution's epilogue (mm_bn2d_fwd_train_pre), the others are read once by the single-launch kernel anyway. */
int mm_bn2d_single_launch(mm_handle_t h, int64_t N, int64_t Ns, int C, int backward);
int mm_bn2d_fwd_train_pre(const void* x, int ld_x, const void* res, int ld_r, int64_t N, int64_t Ns, int C, const float* weight,
                          const float* bias, float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps,
                          float momentum, int relu, void* y, int ld_y, float* save_mean, float* save_invstd, const float* slab,
                          int64_t slab_rows, void* ws, size_t ws_bytes, mm_stream_t stream);
/* Two BatchNorm2d problems of ONE shape (N, Ns, C and the scalars shared) - the same layer of the RGB and of the depth encoder,
 * EXP/2d_net/model.py:43-46 - as ONE single-launch kernel where the maps allow it: half the CUs per problem, one pair of grid barriers
 * and one statistics exchange for both (the small maps of the encoders' layers 2-4 are bound by those fixed costs, not by bytes);
 * otherwise the two problems run one after the other exactly as mm_bn2d_fwd_train / mm_bn2d_bwd run them.  Same results either way. */
typedef struct mm_bn2d_fwd_args {
  const void* x; int ld_x; const void* res; int ld_r; const float* weight; const float* bias; float* running_mean; float* running_var;
  int64_t* num_batches_tracked; void* y; int ld_y; float* save_mean; float* save_invstd;
} mm_bn2d_fwd_args;
typedef struct mm_bn2d_bwd_args {
  const void* x; int ld_x; const void* dy; int ld_dy; const void* dy2; int ld_dy2; const void* yout; int ld_y;
  const float* weight; const float* bias; const float* save_mean; const float* save_invstd; void* dx; int ld_dx; void* dres; int ld_dr;
  float* dweight; float* dbias;
} mm_bn2d_bwd_args;
int mm_bn2d_fwd_train_pair(mm_handle_t h, const mm_bn2d_fwd_args* a, const mm_bn2d_fwd_args* b, int64_t N, int64_t Ns, int C, float eps,
                           float momentum, int relu, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn2d_bwd_pair(mm_handle_t h, const mm_bn2d_bwd_args* a, const mm_bn2d_bwd_args* b, int relu, int64_t N, int64_t Ns, int C, int accumulate,
                     void* ws, size_t ws_bytes, mm_stream_t stream);
/* The stems' BatchNorm2d + ReLU + MaxPool2d(3, 2, 1) (EXP/2d_net/backbones.py:43-47) as ONE pass over the map (after the slab reduce +
 * finalize of mm_bn2d_fwd_train_pre): y = the normalised map [B][H][W][C] (pitch ld_y), ypool [B][H/2][W/2][C] its pooled version, idx
 * the winning taps (the outputs of mm_maxpool3x3s2_fwd); H, W even; images [0, Bs) are statistics group 0.  Backward
 * (mm_bn2d_bwd_pool): the gradient arrives as the POOLED map's gradient dyp + idx (gathered per pixel inside the reduce / apply passes,
 * so no full-resolution gradient map is written or read) and optionally a second full-resolution contribution dy2; three launches,
 * no grid barrier, same results as mm_maxpool3x3s2_bwd followed by mm_bn2d_bwd. */
int mm_bn2d_fwd_train_pre_pool(const void* x, int ld_x, int B, int H, int W, int Bs, int C, const float* weight, const float* bias,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps, float momentum, void* y,
                               int ld_y, void* ypool, void* idx, float* save_mean, float* save_invstd, const float* slab,
                               int64_t slab_rows, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn2d_bwd_pool(const void* x, int ld_x, const void* dyp, int ld_dyp, const void* idx, int B, int H, int W, int Bs, const void* dy2,
                     int ld_dy2, int C, const float* weight, const float* bias, const float* save_mean, const float* save_invstd, void* dx,
                     int ld_dx, float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn2d_fwd_eval(const void* x, int ld_x, const void* res, int ld_r, int64_t N, int C, const float* weight,
                     const float* bias, const float* running_mean, const float* running_var, float eps, int relu, void* y,
                     int ld_y, mm_stream_t stream);
/* yout == NULL with relu != 0 (forward without residual): the ReLU mask is recomputed from x and the saved statistics.
 * dy2 != NULL: the incoming gradient is dy + dy2 (a map with two consumers: residual / concat), summed in the kernel. */
int mm_bn2d_bwd(mm_handle_t h, const void* x, int ld_x, const void* dy, int ld_dy, const void* dy2, int ld_dy2, const void* yout, int ld_y,
                int relu, int64_t N, int64_t Ns, int C, const float* weight, const float* bias, const float* save_mean, const float* save_invstd, void* dx,
                int ld_dx, void* dres, int ld_dr, float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                mm_stream_t stream);

/* out[c] (+)= sum_rows x[row][c]: the conv bias gradient, torch's dy.sum((0,2,3)) (2d_net/model.py:68-81 convs with bias).
 * ws >= mm_bn2d_ws_bytes(C). */
int mm_colsum_bf16(const void* x, int ld_x, int64_t N, int C, float* out, int accumulate, void* ws, size_t ws_bytes,
                   mm_stream_t stream);

/* ---------------------------------------------------------------- concat, max-pool, fused heads (csrc/misc2d.hip) */
int mm_copy_rows_bf16(const void* src, int64_t ld_s, void* dst, int64_t ld_d, int64_t N, int C, mm_stream_t stream);
/* torch.cat(parts, dim=1) of NHWC bf16 maps in one launch (decoder concat [depth, up, rgb], 2d_net/model.py:104-123);
 * split != 0 is its backward: parts[i] = channel slice i of wide.  Channel counts multiples of 8, 1..4 parts. */
int mm_concat_bf16(void* const* parts, const int* channels, int nparts, void* wide, int64_t N, int split, mm_stream_t stream);
int mm_maxpool3x3s2_fwd(const void* x, int ldx, int B, int H, int W, int C, void* y, void* idx, mm_stream_t stream);
/* dy2 (optional): a second gradient of the pooled map (it had two consumers), summed with dy in fp32; ld_dy / ld_dy2 = pixel pitches */
int mm_maxpool3x3s2_bwd(const void* dy, int ld_dy, const void* dy2, int ld_dy2, const void* idx, int B, int H, int W, int C, void* dx,
                        mm_stream_t stream);
size_t mm_head_ws_bytes(int B, int h, int w, int Hp, int Wp, int C, int NJ);
/* AvgPool2d(5,1,2) + Conv2d 1x1 of both heads (EXP/2d_net/model.py:59-60,129-130,158,163-164): out NHWC fp32 [B,h,w,NJ] */
int mm_head_fwd(const void* x, int B, int Hp, int Wp, int ld, int h, int w, int C, const float* Wj, const float* bias,
                int NJ, float* out, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_head_bwd(const void* x, int B, int Hp, int Wp, int ld, int h, int w, int C, const float* Wj, int NJ,
                const float* dout, void* dx, float* dWj, void* ws, size_t ws_bytes, mm_stream_t stream);

/* ---------------------------------------------------------------- exact-fp32 2D convolutions (csrc/conv2d_f32.hip)
 * The `precision: 32` mode (config/run/test.yaml:8): plain fp32 FMAs in an LDS-tiled implicit GEMM; one kernel pair
 * expresses Conv2d / ConvTranspose2d forward, data and weight gradients through index maps (file header). */
int mm_conv2d_f32(const float* A, int B, int Hi, int Wi, int Ca, int ldA, float* O, int Ho, int Wo, int Cn, int ldO, int KH, int KW,
                  int so, int sgn, int off, int up, const float* W, int64_t w_sn, int64_t w_sc, int64_t w_sy, int64_t w_sx,
                  const float* bias, mm_stream_t stream);
size_t mm_conv2d_f32_wgrad_ws_bytes(int64_t n_pixels, int Cg, int Ca, int KH, int KW);
int mm_conv2d_f32_wgrad(const float* G, int B, int Hg, int Wg, int Cg, int ldG, const float* A, int Hi, int Wi, int Ca, int ldA,
                        int KH, int KW, int so, int sgn, int off, int up, float* dW, int64_t w_sn, int64_t w_sc, int64_t w_sy,
                        int64_t w_sx, int accumulate, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_colsum_f32(const float* x, int ld, int64_t N, int C, float* out, int accumulate, mm_stream_t stream);


/* ---------------------------------------------------------------- the dense 2D kernels over IEEE fp16 maps
 * The reference's 2D branch runs under ``precision: 16`` = fp16 autocast + GradScaler (config/run/train.yaml:11).  The kernels of
 * csrc/conv2d.hip, bn2d.hip and misc2d.hip are built a second time with IEEE fp16 as the 16-bit storage format
 * (v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16; csrc/h16.h) and exported under the suffix _f16: same arguments, same
 * semantics, "bf16" in the descriptions above reads "fp16".  One handle serves both builds (same option, same fault word).  mm_copy_rows_bf16 / mm_concat_bf16 move 2-byte elements and serve both.
 * Gradient maps in fp16 need the loss scale of mm2d3d_amd/amp.py (mm_grad_nonfinite ... mm_amp_update below). */
int mm_conv2d_gemm_f16(const void* A, int B, int Hi, int Wi, int Ca, int lda, void* O, int Ho, int Wo, int Cn, int ldo,
                   int out_f32, int Hg, int Wg, int so, int ooy, int oox, int sa, int fr, int ntaps, const int* ty,
                   const int* tx, const void* Wp, int nz, int64_t wz, int zpar, const float* bias, float* stats,
                   int64_t split_m, const void* addend, int ld_add, mm_stream_t stream);
int mm_conv2d_dgrad_s2_f16(const void* dY, int B, int Ho, int Wo, int Cout, int ldy, void* dX, int H, int W, int Cin, int ldx, const void* Wd,
                       int k, int pad, const void* addend, int ld_add, mm_stream_t stream);
int64_t mm_conv2d_gemm_stat_rows_f16(int64_t M, int nz);
int64_t mm_conv2d_3x3s1_stat_rows_f16(int B, int H, int W);
int mm_conv2d_3x3s1_f16(const void* A, int B, int H, int W, int Ca, int lda, void* O, int Cn, int ldo, const void* Wp,
                    const float* bias, int flip, float* stats, int split_b, mm_stream_t stream);
int mm_conv2d_3x3s1_pair_f16(const void* A0, const void* A1, int B, int H, int W, int Ca, int lda, void* O0, void* O1, int Cn, int ldo,
                         const void* Wp0, const void* Wp1, int flip, float* stats0, float* stats1, int split_b, mm_stream_t stream);
int mm_conv2d_wgrad3x3_pair_f16(const void* X0, const void* X1, int B, int H, int W, int Ck, int ldx, const void* dY0, const void* dY1, int Cn,
                            int ldy, float* dW0, float* dW1, int64_t sn, int64_t st, int64_t sk, int accumulate, void* ws, size_t ws_bytes,
                            mm_stream_t stream);
size_t mm_conv2d_wgrad_ws_bytes_f16(int64_t M, int Cn, int Ck, int ntaps);
int mm_conv2d_wgrad_f16(const void* X, int B, int Hi, int Wi, int Ck, int ldx, const void* dY, int Hg, int Wg, int Cn,
                    int ldy, int sa, int ntaps, const int* ty, const int* tx, float* dW, int64_t sn, int64_t st,
                    int64_t sk, int accumulate, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_conv2d_wgrad_slabs_f16(const void* X, int B, int Hi, int Wi, int Ck, int ldx, const void* dY, int Hg, int Wg, int Cn, int ldy,
                          int sa, int ntaps, const int* ty, const int* tx, void* slabs, size_t slab_bytes, int* nsplit,
                          mm_stream_t stream);
int mm_conv2d_wgrad3x3_pair_slabs_f16(const void* X0, const void* X1, int B, int H, int W, int Ck, int ldx, const void* dY0, const void* dY1,
                                  int Cn, int ldy, void* slabs, size_t slab_bytes, int* nsplit, mm_stream_t stream);
int mm_conv2d_wgrad_reduce_desc_bytes_f16(void);
int64_t mm_conv2d_wgrad_reduce_blocks_f16(int Cn, int Ck, int pair);
int mm_conv2d_wgrad_reduce_batch_f16(const void* descs_dev, int n, int64_t total_blocks, mm_stream_t stream);
int64_t mm_conv2d_stem7_stat_rows_f16(int B, int H, int W);
int mm_conv2d_stem7_f16(const void* xb, int B, int Hb, int Wb, int H, int W, int R, int T, void* O, int ldo, const void* Wp, float* stats,
                    int split_b, mm_stream_t stream);
int mm_stem_prep_f16(const float* in, int B, int C, int H, int W, int pad, int Hb, int Wb, int R, void* out, mm_stream_t stream);
int mm_pack_weights_f16(const float* in, void* out, int Z, int N, int T, int K, int64_t sz, int64_t sn, int64_t st,
                         int64_t sk, mm_stream_t stream);
int mm_pack_weights_f16_batch(const int64_t* desc, int ndesc, int64_t total_blocks, mm_stream_t stream);
size_t mm_bn2d_ws_bytes_f16(int C);
int mm_bn2d_fwd_train_f16(mm_handle_t h, const void* x, int ld_x, const void* res, int ld_r, int64_t N, int64_t Ns, int C, const float* weight,
                      const float* bias, float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps,
                      float momentum, int relu, void* y, int ld_y, float* save_mean, float* save_invstd, void* ws,
                      size_t ws_bytes, mm_stream_t stream);
int mm_bn2d_single_launch_f16(mm_handle_t h, int64_t N, int64_t Ns, int C, int backward);
int mm_bn2d_fwd_train_pre_f16(const void* x, int ld_x, const void* res, int ld_r, int64_t N, int64_t Ns, int C, const float* weight,
                          const float* bias, float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps,
                          float momentum, int relu, void* y, int ld_y, float* save_mean, float* save_invstd, const float* slab,
                          int64_t slab_rows, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn2d_fwd_train_pair_f16(mm_handle_t h, const mm_bn2d_fwd_args* a, const mm_bn2d_fwd_args* b, int64_t N, int64_t Ns, int C, float eps,
                           float momentum, int relu, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn2d_bwd_pair_f16(mm_handle_t h, const mm_bn2d_bwd_args* a, const mm_bn2d_bwd_args* b, int relu, int64_t N, int64_t Ns, int C, int accumulate,
                     void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn2d_fwd_train_pre_pool_f16(const void* x, int ld_x, int B, int H, int W, int Bs, int C, const float* weight, const float* bias,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float eps, float momentum, void* y,
                               int ld_y, void* ypool, void* idx, float* save_mean, float* save_invstd, const float* slab,
                               int64_t slab_rows, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn2d_bwd_pool_f16(const void* x, int ld_x, const void* dyp, int ld_dyp, const void* idx, int B, int H, int W, int Bs, const void* dy2,
                     int ld_dy2, int C, const float* weight, const float* bias, const float* save_mean, const float* save_invstd, void* dx,
                     int ld_dx, float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_bn2d_fwd_eval_f16(const void* x, int ld_x, const void* res, int ld_r, int64_t N, int C, const float* weight,
                     const float* bias, const float* running_mean, const float* running_var, float eps, int relu, void* y,
                     int ld_y, mm_stream_t stream);
int mm_bn2d_bwd_f16(mm_handle_t h, const void* x, int ld_x, const void* dy, int ld_dy, const void* dy2, int ld_dy2, const void* yout, int ld_y,
                int relu, int64_t N, int64_t Ns, int C, const float* weight, const float* bias, const float* save_mean, const float* save_invstd, void* dx,
                int ld_dx, void* dres, int ld_dr, float* dweight, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                mm_stream_t stream);
int mm_colsum_f16(const void* x, int ld_x, int64_t N, int C, float* out, int accumulate, void* ws, size_t ws_bytes,
                   mm_stream_t stream);
int mm_maxpool3x3s2_fwd_f16(const void* x, int ldx, int B, int H, int W, int C, void* y, void* idx, mm_stream_t stream);
int mm_maxpool3x3s2_bwd_f16(const void* dy, int ld_dy, const void* dy2, int ld_dy2, const void* idx, int B, int H, int W, int C, void* dx,
                        mm_stream_t stream);
size_t mm_head_ws_bytes_f16(int B, int h, int w, int Hp, int Wp, int C, int NJ);
int mm_head_fwd_f16(const void* x, int B, int Hp, int Wp, int ld, int h, int w, int C, const float* Wj, const float* bias,
                int NJ, float* out, void* ws, size_t ws_bytes, mm_stream_t stream);
int mm_head_bwd_f16(const void* x, int B, int Hp, int Wp, int ld, int h, int w, int C, const float* Wj, int NJ,
                const float* dout, void* dx, float* dWj, void* ws, size_t ws_bytes, mm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MM2D3D_H */
