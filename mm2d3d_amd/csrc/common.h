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
int mm_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* total_out, void* ws, size_t ws_bytes,
                          hipStream_t s);
size_t mm_scan_ws_bytes(int64_t n);
