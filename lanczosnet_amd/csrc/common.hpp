// Shared helpers for the gfx950 LanczosNet kernels.  CDNA4 only: wave = 64 lanes,
// v_mfma_f32_32x32x2_f32 fragments as described in include/lanczosnet_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/lanczosnet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace lnz {

void set_error(const char* fmt, ...);
// the kernel (with its template arguments) a lnz_lanczosnet_* launcher selected, for lnz_last_kernel()
void note_kernel(const char* fmt, ...);

#define LNZ_REQUIRE(cond, code, ...)      \
  do {                                    \
    if (!(cond)) {                        \
      lnz::set_error(__VA_ARGS__);        \
      return (code);                      \
    }                                     \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return LNZ_ELAUNCH;
  }
  return LNZ_OK;
}

// Dynamic LDS beyond the 64 KiB default has to be granted per function AND per device (a process may
// drive several): asked for at every launch.  A part that cannot grant it (less than 160 KiB of LDS per
// workgroup) answers LNZ_ENOTSUP with a message — the callers' cue to take another path — instead of
// an opaque launch failure.
inline int set_dynamic_lds(const void* fn, size_t bytes, const char* what) {
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: %zu bytes of dynamic LDS per workgroup are not available on this device (%s)", what,
              bytes, hipGetErrorString(e));
    return LNZ_ENOTSUP;
  }
  return LNZ_OK;
}
#define LNZ_DYNAMIC_LDS(fn, bytes, what)                                       \
  do {                                                                         \
    const int rc_lds_ = lnz::set_dynamic_lds((const void*)(fn), (bytes), what); \
    if (rc_lds_ != LNZ_OK) return rc_lds_;                                     \
  } while (0)

// C/D row of accumulator register r for the lane half hh (= lane >> 5):
//   row = (r & 3) + 8 * (r >> 2) + 4 * hh          (v_mfma_f32_32x32x2_f32, col = lane & 31)
__host__ __device__ inline int cd_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ inline f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Wait until ALL of this wave's vector-memory operations have completed — including
// global_load_lds copies, whose LDS writes the compiler does not connect to later ds_reads of a
// different type (it then leaves vmcnt out of the s_waitcnt in front of a barrier: a wave could
// pass the barrier with its copies still in flight).  gfx9 encoding: vmcnt(0) expcnt(7) lgkmcnt(15).
__device__ inline void wait_vmcnt0() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// ... until at most N of them are still in flight (they complete in order): vmcnt(N)
template <int N>
__device__ inline void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

__device__ inline f32x16 splat16(float v) {
  f32x16 x;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = v;
  return x;
}

}  // namespace lnz
