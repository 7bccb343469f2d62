// Shared helpers for the gfx950 Neural Object Field kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <math.h>
#include "../../include/nof_hip.h"

int nof_set_error(int code, const char* fmt, ...);

#define NOF_ARG(cond)                                                                     \
  do { if (!(cond)) return nof_set_error(-1, "%s: argument check failed: %s", __func__, #cond); } while (0)

#define NOF_LAUNCH_OK()                                                                   \
  do { hipError_t e_ = hipGetLastError();                                                 \
       if (e_ != hipSuccess) return nof_set_error((int)e_, "%s: %s", __func__, hipGetErrorString(e_)); } while (0)

#define NOF_HIP(call)                                                                     \
  do { hipError_t e_ = (call);                                                            \
       if (e_ != hipSuccess) return nof_set_error((int)e_, "%s: %s", __func__, hipGetErrorString(e_)); } while (0)

static inline int64_t nof_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

#define NOF_INF __builtin_huge_valf()

int nof_cu_count(void);                     // compute units of the current device (cached; 256 on MI355X)

// ---- NofTileList (include/nof_hip.h): uint32 head[4] = {count, n_tiles, 0, 0} | uint32 tiles[n_tiles + pad] | uint8 flags[n_tiles] ----
static inline uint32_t nof_tile_count(int64_t B) { return (uint32_t)((B + 31) / 32); }
static inline size_t nof_tile_list_words(uint32_t n_tiles) { return 4 + (((size_t)n_tiles + 1 + 3) & ~(size_t)3); }   // header + list (+1 pad), 16-byte multiple
static inline const uint8_t* nof_tile_flags(const void* tile_list, int64_t B) {
  return tile_list ? (const uint8_t*)((const uint32_t*)tile_list + nof_tile_list_words(nof_tile_count(B))) : nullptr;
}
