// Per-device handle of the C ABI (include/mm2d3d.h: mm_create / mm_destroy / mm_set_option / mm_get_option / mm_fault_poll).
// "no allocation, no global state except per-device handle created by mm_create(device_id); re-entrant per
// handle; caller guarantees one host thread per handle".  The handle is a small host object; the device memory it points at (the
// grid-barrier words) and the pinned fault word are the CALLER's (sizes: mm_handle_sync_bytes / mm_handle_fault_bytes).
#include <new>

#include "common.h"
#include "fused_bn.h"  // FUSED_LDS: what a single-launch batch-norm workgroup needs

extern "C" {

size_t mm_handle_sync_bytes(void) { return MM_SYNC_BYTES; }
size_t mm_handle_fault_bytes(void) { return MM_FAULT_BYTES; }

// sync_dev: device memory of mm_handle_sync_bytes(), zero-filled; fault_host: pinned, device-mapped host memory of
// mm_handle_fault_bytes() (hipHostMalloc / torch pinned memory), zero-filled.  Both stay the caller's and must outlive the handle.
int mm_create(int device_id, void* sync_dev, size_t sync_bytes, void* fault_host, size_t fault_bytes, void** out) {
  MM_CHECK_ARG(out != nullptr, "mm_create: out is NULL");
  *out = nullptr;
  MM_CHECK_ARG(device_id >= 0, "mm_create: bad device id");
  MM_CHECK_ARG(sync_dev && sync_bytes >= MM_SYNC_BYTES, "mm_create: sync_dev must hold mm_handle_sync_bytes() bytes");
  MM_CHECK_ARG(fault_host && fault_bytes >= MM_FAULT_BYTES, "mm_create: fault_host must hold mm_handle_fault_bytes() bytes");
  MMHandle* H = new (std::nothrow) MMHandle();
  MM_CHECK_ARG(H != nullptr, "mm_create: out of host memory");
  H->magic = MM_HANDLE_MAGIC;
  H->device = device_id;
  H->sync = (unsigned*)sync_dev;
  H->fault_host = (unsigned*)fault_host;
  H->fault_dev = nullptr;
  H->nstream = 0;
  H->attr_done = 0;
  for (int i = 0; i < MM_OPT_COUNT; i++) H->opt[i] = 0;
  H->opt[MM_OPT_BN2D_FUSED] = 3;
  H->opt[MM_OPT_BN3D_FUSED] = 3;
  H->opt[MM_OPT_OS_SORT] = 0;
  H->opt[MM_OPT_SPCONV_TERMS] = 3;
  H->opt[MM_OPT_DW_WIDE] = 1;
  // one probe per handle; any failure marks it "three-kernel batch norms only" instead of failing every call
  int cus = 0, lds_max = 0;
  void* fdev = nullptr;
  H->fused_ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess &&
                hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess &&
                (size_t)lds_max >= FUSED_LDS && cus >= 2 && hipHostGetDevicePointer(&fdev, fault_host, 0) == hipSuccess;
  if (!H->fused_ok) (void)hipGetLastError();
  // fused_wave_sums combines at most 4 x 64 workgroups per statistics group: never plan a wider grid than that
  H->cus = cus < 256 ? cus : 256;
  H->lds_max = lds_max;
  H->fault_dev = (unsigned*)fdev;
  *out = H;
  return MM_OK;
}

int mm_destroy(void* h) {
  MM_CHECK_HANDLE(h);
  H->magic = 0;
  delete H;
  return MM_OK;
}

// returns the previous value (>= 0), or MM_ERR_ARG
int mm_set_option(void* h, int option, int value) {
  MM_CHECK_HANDLE(h);
  MM_CHECK_ARG(option >= 0 && option < MM_OPT_COUNT, "mm_set_option: unknown option");
  const int prev = H->opt[option];
  if (option == MM_OPT_BN2D_FUSED || option == MM_OPT_BN3D_FUSED) value &= 3;
  if (option == MM_OPT_SPCONV_TERMS) MM_CHECK_ARG(value == 0 || value == 2 || value == 3, "mm_set_option: spconv terms are 0, 2 or 3");
  H->opt[option] = value;
  return prev;
}

int mm_get_option(void* h, int option) {
  MM_CHECK_HANDLE(h);
  MM_CHECK_ARG(option >= 0 && option < MM_OPT_COUNT, "mm_get_option: unknown option");
  return H->opt[option];
}

// 1 if a single-launch batch-norm kernel launched through this handle gave up at its grid barrier since the last call (that
// launch's outputs, and everything computed from them, are invalid).  The single-launch kernels are then switched off on this
// handle (three-kernel path) and the barrier words re-armed (the one place the library waits for the device).  Costs one read of
// host memory when nothing happened.
int mm_fault_poll(void* h) {
  MM_CHECK_HANDLE(h);
  if (!__atomic_load_n(H->fault_host, __ATOMIC_RELAXED)) return 0;
  H->opt[MM_OPT_BN2D_FUSED] = H->opt[MM_OPT_BN3D_FUSED] = 0;
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(H->device);
  (void)hipDeviceSynchronize();
  (void)hipMemset(H->sync, 0, MM_SYNC_BYTES);
  (void)hipSetDevice(cur);
  __atomic_store_n(H->fault_host, 0u, __ATOMIC_RELAXED);
  return 1;
}

}  // extern "C"
