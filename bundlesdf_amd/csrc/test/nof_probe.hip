// TEST-ONLY library (libnof_probe.so, bundlesdf_amd/build.py:build_probe): the two hardware probes that used to be compiled into the
// product library.  Not declared in include/nof_hip.h, not linked into libnof_hip.so; tests/test_gpu_ops.py (the MFMA operand layout)
// and tools/atomic_probe.py (the atomic rate the table scatter is priced against) load it directly.
#include "../nof_mfma_dev.h"

// error plumbing of this library (the product's lives in nof_capi.hip)
static thread_local char g_probe_err[512] = "";
int nof_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_probe_err, sizeof(g_probe_err), fmt, ap);
  va_end(ap);
  return code;
}
extern "C" const char* nof_probe_last_error(void) { return g_probe_err; }

// ---- test hook: one 32x32 output tile with the operand layouts of nof_mlp.hip (nof_mfma_dev.h) -----------------------------
template <class P>
__global__ void k_mfma_probe(const float* __restrict__ Am, const float* __restrict__ Bm, float* __restrict__ D, int K) {
  constexpr int KR = P::KR;
  const int lane = threadIdx.x, hi = lane >> 5, i = lane & 31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int k0 = 0; k0 < K; k0 += 2 * KR) {
    float a[KR], b[KR];
#pragma unroll
    for (int t = 0; t < KR; ++t) {
      const int k = k0 + KR * hi + t;                                  // A[i][k] (32xK row-major), B[k][j] (Kx32 row-major)
      a[t] = Am[i * K + k];
      b[t] = Bm[k * 32 + i];
    }
    acc = P::mma(P::pack(a), P::pack(b), acc);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) D[nloc(hi, r) * 32 + i] = acc[r];
}

extern "C" int nof_mfma_probe(int32_t precision, const float* A, const float* Bm, float* D, int32_t K, void* stream) {
  NOF_ARG(A && Bm && D && K > 0 && K % 16 == 0 && precision >= 0 && precision <= 2);
  if (precision == 0) hipLaunchKernelGGL(k_mfma_probe<PrecF32>, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, D, K);
  else if (precision == 1) hipLaunchKernelGGL(k_mfma_probe<PrecBF16>, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, D, K);
  else hipLaunchKernelGGL(k_mfma_probe<PrecF16>, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, D, K);
  NOF_LAUNCH_OK();
  return 0;
}

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

// ---- test hook: what the queue leaves between two dependent launches (tools/gap_probe.py, round 6) -------------------------------
// kernel A streams `n` float4 into `dst` -- mode 0: plain stores, 1: __builtin_nontemporal_store, 2: stores with sc0 sc1 (write-through
// past the XCD's L2), 3: no stores (loads only) --, kernel B is one workgroup that touches one word.  Under rocprofv3 --kernel-trace the
// distance from A's end to B's start shows whether the end-of-kernel write-back of A's dirty L2 lines is what the gaps of the
// training step's timeline are made of.
__global__ __launch_bounds__(256) void k_gap_store(float4* __restrict__ dst, const float4* __restrict__ src, int64_t n, int mode) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float4 v = src[i];
    if (mode == 0) dst[i] = v;
    else if (mode == 1) {
      __builtin_nontemporal_store(v.x, &dst[i].x); __builtin_nontemporal_store(v.y, &dst[i].y);
      __builtin_nontemporal_store(v.z, &dst[i].z); __builtin_nontemporal_store(v.w, &dst[i].w);
    } else if (mode == 2) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      const f4v vv = {v.x, v.y, v.z, v.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&dst[i]), "v"(vv) : "memory");
    } else {
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (mode == 3 && acc.x == 12345.678f) dst[0] = acc;
}
__global__ void k_gap_touch(float* __restrict__ p) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f;
}
extern "C" int nof_gap_probe(float* dst, const float* src, int64_t n_float4, int32_t mode, int32_t blocks, float* word, void* stream) {
  NOF_ARG(dst && src && word && n_float4 > 0 && mode >= 0 && mode <= 3 && blocks > 0);
  hipLaunchKernelGGL(k_gap_store, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)dst, (const float4*)src, n_float4, (int)mode);
  NOF_LAUNCH_OK();
  hipLaunchKernelGGL(k_gap_touch, dim3(1), dim3(64), 0, (hipStream_t)stream, word);
  NOF_LAUNCH_OK();
  return 0;
}
