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
