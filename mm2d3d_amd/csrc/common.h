// Shared helpers for the gfx950 kernels of libmm2d3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define MM_OK 0
#define MM_ERR_ARG -1
#define MM_ERR_HIP -2
#define MM_ERR_WORKSPACE -3
#define MM_ERR_UNSUPPORTED -4

void mm_set_error(const char* fmt, ...);

#define MM_CHECK_ARG(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      mm_set_error(__VA_ARGS__);     \
      return MM_ERR_ARG;             \
    }                                \
  } while (0)

#define MM_HIP(call)                                                          \
  do {                                                                        \
    hipError_t e__ = (call);                                                  \
    if (e__ != hipSuccess) {                                                  \
      mm_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
      return MM_ERR_HIP;                                                      \
    }                                                                         \
  } while (0)

#define MM_LAUNCH_CHECK() MM_HIP(hipGetLastError())

static inline int64_t mm_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t mm_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Raising a kernel's dynamic-LDS limit (hipFuncSetAttribute) is a per-DEVICE attribute and idempotent: a call site remembers
// the devices it has done it on in one word (mm_attr_todo asks, mm_attr_done records).  These words are the library's only process-wide memory - caches of an idempotent
// driver call, no mode, no switch.
static inline bool mm_attr_todo(const unsigned* done) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 32) return true;
  return !(__atomic_load_n(done, __ATOMIC_RELAXED) & (1u << d));
}
// ... and marks the device only AFTER every hipFuncSetAttribute of the call site has succeeded (a failed or a racing
// first call must not leave the bit set with the limit not raised; two threads that both see "todo" both make the idempotent call)
static inline void mm_attr_done(unsigned* done) {
  int d = 0;
  if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 32) __atomic_fetch_or(done, 1u << d, __ATOMIC_RELAXED);
}

// bump allocator over a caller-provided workspace
struct MMArena {
  char* base;
  size_t cap, off;
  MMArena(void* p, size_t bytes) : base((char*)p), cap(bytes), off(0) {}
  template <typename T>
  T* take(size_t n) {
    size_t b = mm_align(n * sizeof(T));
    if (off + b > cap) return nullptr;
    T* r = (T*)(base + off);
    off += b;
    return r;
  }
};

// exclusive scan of int32 -> int32 (n up to 2^31), total written to *total_out (device); scan.hip.
// `in` and `out` must hold n + 1 elements when total_out is given (the single-pass path scans n + 1 and leaves the total in out[n]).
// no_spin != 0: only kernels whose workgroups never wait for each other (three plain launches instead of the decoupled look-back
// scan): what a build on a side stream beside grid-barrier kernels needs (fused_bn.h)
int mm_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total_out, void* ws, size_t ws_bytes,
                          hipStream_t s, int no_spin = 0);
int mm_exclusive_scan_nonneg_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes, hipStream_t s, int no_spin = 0);
size_t mm_scan_ws_bytes(int64_t n);

// ---------------------------------------------------------------------------------------------------------- per-device handle
// include/mm2d3d.h: mm_create / mm_destroy.  Everything that outlives a call lives HERE, in memory the caller supplied: the
// grid-barrier words and the fault word of the single-launch batch norms, their per-stream slot map, the mode switches.  The
// library keeps no process-wide mutable state (what is left are idempotent hipFuncSetAttribute calls that raise a kernel's
// dynamic-LDS limit, remembered per handle) and allocates no device memory.  One host thread per handle at a time.
enum {
  MM_OPT_BN2D_FUSED = 0,  // single-launch BatchNorm2d kernels: bit 0 forward, bit 1 backward (default 3)
  MM_OPT_BN3D_FUSED = 1,  // the same for the fp32 / 16-bit sparse rows (default 3)
  MM_OPT_OS_SORT = 2,     // tile-table sort: 0 rocPRIM Onesweep (default), 1 merge sort (no inter-workgroup spin waits)
  MM_OPT_SPCONV_TERMS = 3,  // bf16 terms per fp32 operand of the split-product sparse engines: 3 (fp32-faithful, default), 2, 0 = plain fp32
  MM_OPT_DW_WIDE = 4,     // sparse dW: wide channel tiles on rule lists >= 200k (default 1)
  MM_OPT_COUNT = 8
};
constexpr int MM_SYNC_SLOTS = 64;           // one barrier slot per stream: top arrival counter | release word | 8 group counters,
constexpr size_t MM_SYNC_SLOT_BYTES = 2048;  // each on a 128-byte line of its own (fused_bn.h fused_barrier)
constexpr size_t MM_SYNC_BYTES = (size_t)MM_SYNC_SLOTS * MM_SYNC_SLOT_BYTES;
constexpr size_t MM_FAULT_BYTES = 64;

struct MMHandle {
  unsigned magic;
  int device, cus, lds_max;
  unsigned* sync;        // device memory, MM_SYNC_BYTES, zero-filled by the caller before mm_create
  unsigned* fault_host;  // pinned + mapped host memory, MM_FAULT_BYTES: written by a kernel whose grid barrier timed out
  unsigned* fault_dev;   // its device address
  int opt[MM_OPT_COUNT];
  hipStream_t streams[MM_SYNC_SLOTS];
  int nstream;
  unsigned attr_done;    // bit per translation unit: the dynamic-LDS limit of its single-launch kernels is raised on this device
  int fused_ok;          // the device can run the single-launch kernels (CU count, LDS size); probed by mm_create
};
constexpr unsigned MM_HANDLE_MAGIC = 0x4D4D3244u;  // "MM2D"
static inline MMHandle* mm_handle(void* h) {
  MMHandle* p = (MMHandle*)h;
  return (p && p->magic == MM_HANDLE_MAGIC) ? p : nullptr;
}
#define MM_CHECK_HANDLE(h)                                                     \
  MMHandle* H = mm_handle(h);                                                  \
  do {                                                                         \
    if (!H) {                                                                  \
      mm_set_error("invalid handle (mm_create first)");                        \
      return MM_ERR_ARG;                                                       \
    }                                                                          \
  } while (0)
