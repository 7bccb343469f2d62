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
  if (variant >= 30) {
    // gather rate of 8-byte table rows (what the hash lookup does): 30 = every lane its own random row, 31 = groups of 8
    // adjacent lanes read the SAME row (a run of samples in one cell), 32 = one lane of every 8 reads, the others idle.
    // Each lane does 8 dependent-free gathers; the sum is stored so that nothing is optimised away.
    if (i >= n) return;
    const float2* t2 = reinterpret_cast<const float2*>(table);
    const int64_t base = variant == 30 ? i : (i & ~(int64_t)7);
    if (variant == 32 && (i & 7) != 0) return;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t e = idx[(base + (int64_t)k * 8191) % n];
      const float2 v = t2[e];
      acc += v.x + v.y;
    }
    table[2 * (size_t)(1 << 19) + i] = acc;                            // scratch area behind the 2^19 rows (the caller allocates it)
    return;
  }
  if (variant >= 22) {                      // other atomic types on random entries (one 8-byte slot per entry): 22 u32, 23 u64, 24 f64, 25 pk f16
    if (i >= n) return;
    void* a = &table[2 * (size_t)idx[i]];
    if (variant == 22) atomicAdd((unsigned int*)a, 1u);
    if (variant == 23) atomicAdd((unsigned long long*)a, 1ull);
    if (variant == 24) atomicAdd((double*)a, 1.0);
    if (variant == 25) { const uint32_t one2 = 0x3C003C00u; asm volatile("global_atomic_pk_add_f16 %0, %1, off" ::"v"(a), "v"(one2) : "memory"); }
    return;
  }
  if (variant >= 10) {
    // 10..13: x only, random entries; 14..17: the same with every entry folded into the eighth of the table that belongs to the
    // XCD the wave runs on (hardware XCC_ID), i.e. no line is touched by two XCDs; 18..21: the same with blockIdx % 8 instead
    // of XCC_ID.  Instruction flags: +0 none, +1 nt, +2 sc1, +3 sc0 (returns the old value)
    if (i >= n) return;
    const int flag = (variant - 10) & 3, part = (variant - 10) >> 2;
    uint32_t e = idx[i];
    if (part) {
      int xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      const uint32_t own = part == 1 ? (uint32_t)(xcc & 7) : (blockIdx.x & 7u);
      e = (e & 0xFFFFu) | (own << 16);
    }
    float* a = &table[2 * (size_t)e];
    const float one = 1.0f;
    if (flag == 0) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(a), "v"(one) : "memory");
    if (flag == 1) asm volatile("global_atomic_add_f32 %0, %1, off nt" ::"v"(a), "v"(one) : "memory");
    if (flag == 2) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(a), "v"(one) : "memory");
    if (flag == 3) { float r; asm volatile("global_atomic_add_f32 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(a), "v"(one) : "memory"); if (r == -1.0f) table[0] = r; }
    return;
  }
  if (i >= n) return;
  const size_t e = (variant == 2) ? (size_t)(i & 0x7FFFF) : (size_t)idx[i];
  atomicAdd(&table[2 * e], 1.0f);
  if (variant != 3) atomicAdd(&table[2 * e + 1], 1.0f);
}

extern "C" int nof_atomic_probe(int32_t variant, const uint32_t* idx, float* table, int64_t n, void* stream) {
  NOF_ARG(idx && table && n > 0 && variant >= 0 && variant <= 32);
  const int64_t threads = variant == 1 ? 2 * n : n;
  hipLaunchKernelGGL(k_atomic_probe, dim3((unsigned)nof_div_up(threads, 256)), dim3(256), 0, (hipStream_t)stream, variant, idx,
                     table, n);
  NOF_LAUNCH_OK();
  return 0;
}
