// Error plumbing of the C ABI (include/nof_hip.h).
#include "nof_common.h"

static thread_local char g_nof_err[512] = "";

int nof_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_nof_err, sizeof(g_nof_err), fmt, ap);
  va_end(ap);
  return code;
}

int nof_cu_count(void) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n = prop.multiProcessorCount;
    (void)hipGetLastError();
    cus = n;
  }
  return cus;
}

extern "C" const char* nof_last_error(void) { return g_nof_err; }
extern "C" int nof_version(void) { return 100; }

// ---- test hook: fp32 atomic-add throughput for different address patterns (informs the hash scatter design) -------------
//   0: two instructions per entry (x then y), random entries      1: adjacent-lane pairs (lane 2m -> x, 2m+1 -> y), random
//   2: two instructions per entry, sequential entries             3: x only, random entries
//   4: like 0 but each wave's 64 entries are sorted (neighbouring lanes -> neighbouring entries)
__global__ __launch_bounds__(256) void k_atomic_probe(int variant, const uint32_t* __restrict__ idx, float* __restrict__ table,
                                                       int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (variant == 1) {
    const int64_t e = i >> 1;
    if (e < n) atomicAdd(&table[2 * (size_t)idx[e] + (i & 1)], 1.0f);
    return;
  }
  if (variant == 6 || variant == 7 || variant == 8) {
    // 16 lines per instruction, 4 lanes each: 6 = same ADDRESS from lanes l, l+16, l+32, l+48; 7 = same line, different words,
    // lanes 16 apart; 8 = same line, different words, ADJACENT lanes
    if (i >= n) return;
    const int l = threadIdx.x & 63;
    const int grp = variant == 8 ? (l >> 2) : (l & 15), sub = variant == 8 ? (l & 3) : (l >> 4);
    const size_t base = (size_t)(idx[(i >> 6) * 16 + grp] & ~3u);
    atomicAdd(&table[2 * (base + (variant == 6 ? 0 : sub))], 1.0f);
    return;
  }
  if (variant == 9) {                       // plain 4-byte stores, random entries (what an atomic-free scatter would cost)
    if (i < n) table[2 * (size_t)idx[i]] = 1.0f;
    return;
  }
  if (i >= n) return;
  const size_t e = (variant == 2) ? (size_t)(i & 0x7FFFF) : (size_t)idx[i];
  atomicAdd(&table[2 * e], 1.0f);
  if (variant != 3) atomicAdd(&table[2 * e + 1], 1.0f);
}

extern "C" int nof_atomic_probe(int32_t variant, const uint32_t* idx, float* table, int64_t n, void* stream) {
  NOF_ARG(idx && table && n > 0 && variant >= 0 && variant <= 9);
  const int64_t threads = variant == 1 ? 2 * n : n;
  hipLaunchKernelGGL(k_atomic_probe, dim3((unsigned)nof_div_up(threads, 256)), dim3(256), 0, (hipStream_t)stream, variant, idx,
                     table, n);
  NOF_LAUNCH_OK();
  return 0;
}
