// TEST INFRASTRUCTURE (oracle/).  Host stand-ins for the handful of CUDA / ATen names the reference's two native
// files use, so that mycuda/torch_ngp_grid_encoder/gridencoder.cu and mycuda/common.cu compile AS HOST C++ with g++
// from where they lie under /root/reference (recipe: oracle/ref_build.py).  No reference code is in here: this file
// only defines the execution model (a kernel launch = nested loops over blockIdx/threadIdx on one host thread,
// atomicAdd = "+=" in launch order) and minimal tensor/accessor/half types.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>
#include <algorithm>
#include <initializer_list>

using std::abs;   // device code calls abs(float): must not resolve to abs(int)

#define __global__
#define __device__
#define __host__
#define __restrict__
#define __forceinline__ inline

struct uint3 { unsigned int x, y, z; };
struct dim3 {
  unsigned int x, y, z;
  dim3(long long vx = 1, long long vy = 1, long long vz = 1) : x((unsigned)vx), y((unsigned)vy), z((unsigned)vz) {}
};
namespace cuda_host {
inline uint3 &block_idx() { static thread_local uint3 v{0, 0, 0}; return v; }
inline uint3 &thread_idx() { static thread_local uint3 v{0, 0, 0}; return v; }
inline dim3 &block_dim() { static thread_local dim3 v; return v; }
inline dim3 &grid_dim() { static thread_local dim3 v; return v; }
// one launch: every (block, thread) of the grid runs to completion, in order, on the calling thread
template <class Body> inline void run(dim3 grid, dim3 block, Body body) {
  grid_dim() = grid; block_dim() = block;
  for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
    block_idx() = uint3{bx, by, bz};
    for (unsigned tz = 0; tz < block.z; tz++) for (unsigned ty = 0; ty < block.y; ty++) for (unsigned tx = 0; tx < block.x; tx++) {
      thread_idx() = uint3{tx, ty, tz};
      body();
    }
  }
}
struct spin : std::runtime_error { spin() : std::runtime_error("device code entered its error spin loop (while(1){})") {} };
}  // namespace cuda_host
#define blockIdx (cuda_host::block_idx())
#define threadIdx (cuda_host::thread_idx())
#define blockDim (cuda_host::block_dim())
#define gridDim (cuda_host::grid_dim())

// ---- IEEE binary16 with round-to-nearest-even conversions (at::Half / __half semantics: arithmetic in float)
namespace cuda_host {
inline uint16_t f2h(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0));
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                  // rounds to inf
  if (x < 0x33000001u) return (uint16_t)sign;                                // rounds to 0
  int e = (int)(x >> 23) - 127; uint32_t m = (x & 0x7fffffu) | 0x800000u;
  int shift = (e < -14) ? (13 + (-14 - e)) : 13;
  uint32_t h = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (h & 1))) h++;
  if (e < -14) return (uint16_t)(sign | h);                                  // subnormal (h may carry into normal)
  return (uint16_t)(sign | (((uint32_t)(e + 15) << 10) + (h - 0x400u)));
}
inline float h2f(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else { int k = 0; while (!(m & 0x400u)) { m <<= 1; k++; } x = sign | ((uint32_t)(113 - k) << 23) | ((m & 0x3ffu) << 13); }
  } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
  else x = sign | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &x, 4); return f;
}
}  // namespace cuda_host
namespace at {
struct Half {
  uint16_t bits;
  Half() : bits(0) {}
  Half(float f) : bits(cuda_host::f2h(f)) {}
  operator float() const { return cuda_host::h2f(bits); }
  Half &operator+=(float f) { *this = Half((float)*this + f); return *this; }
};
}  // namespace at
typedef at::Half __half;
struct __half2 { __half x, y; };
inline float atomicAdd(float *p, float v) { float o = *p; *p = o + v; return o; }
inline double atomicAdd(double *p, double v) { double o = *p; *p = o + v; return o; }
inline __half2 atomicAdd(__half2 *p, __half2 v) {          // two independent half adds, each rounded to nearest even
  __half2 o = *p; p->x = __half((float)o.x + (float)v.x); p->y = __half((float)o.y + (float)v.y); return o;
}

// ---- the sliver of ATen the two files touch
namespace at {
enum class ScalarType { Float, Half, Double, Int, Long };
template <class T> struct st_of;
template <> struct st_of<float> { static constexpr ScalarType v = ScalarType::Float; };
template <> struct st_of<double> { static constexpr ScalarType v = ScalarType::Double; };
template <> struct st_of<Half> { static constexpr ScalarType v = ScalarType::Half; };
template <> struct st_of<int> { static constexpr ScalarType v = ScalarType::Int; };
template <> struct st_of<long> { static constexpr ScalarType v = ScalarType::Long; };
inline size_t st_size(ScalarType t) { return t == ScalarType::Half ? 2 : (t == ScalarType::Float || t == ScalarType::Int) ? 4 : 8; }
struct Device { bool cuda; bool is_cuda() const { return cuda; } };
}  // namespace at
namespace torch {
struct RestrictPtrTraits {};
template <class T, int N, class Traits = RestrictPtrTraits> struct PackedTensorAccessor32 {
  T *p; const int64_t *sizes_, *strides_;
  int64_t size(int i) const { return sizes_[i]; }
  PackedTensorAccessor32<T, N - 1, Traits> operator[](int64_t i) const { return {p + i * strides_[0], sizes_ + 1, strides_ + 1}; }
};
template <class T, class Traits> struct PackedTensorAccessor32<T, 1, Traits> {
  T *p; const int64_t *sizes_, *strides_;
  int64_t size(int i) const { return sizes_[i]; }
  T &operator[](int64_t i) const { return p[i * strides_[0]]; }
};
}  // namespace torch
namespace at {
struct Tensor {
  void *data = nullptr; ScalarType st = ScalarType::Float;
  std::vector<int64_t> shape, stride; bool contiguous = true, on_cuda = true;
  std::shared_ptr<std::vector<char>> owned;
  Tensor() {}
  Tensor(void *d, ScalarType t, std::vector<int64_t> s) : data(d), st(t), shape(std::move(s)) { set_strides(); }
  void set_strides() { stride.assign(shape.size(), 1); for (int i = (int)shape.size() - 2; i >= 0; i--) stride[i] = stride[i + 1] * shape[i + 1]; }
  Device device() const { return {on_cuda}; }
  bool is_cuda() const { return on_cuda; }
  bool is_contiguous() const { return contiguous; }
  ScalarType scalar_type() const { return st; }
  ScalarType type() const { return st; }
  const std::vector<int64_t> &sizes() const { return shape; }
  template <class T> T *data_ptr() const {
    if (st_of<T>::v != st) throw std::runtime_error("data_ptr: dtype mismatch");
    return (T *)data;
  }
  template <class T, int N, class Traits> torch::PackedTensorAccessor32<typename std::remove_const<T>::type, N, Traits> packed_accessor32() const {
    if ((int)shape.size() != N) throw std::runtime_error("packed_accessor32: rank mismatch");
    if (st_of<typename std::remove_const<T>::type>::v != st) throw std::runtime_error("packed_accessor32: dtype mismatch");
    return {(typename std::remove_const<T>::type *)data, shape.data(), stride.data()};
  }
};
}  // namespace at
namespace torch {
using at::Tensor;
enum DTypeTag { kFloat32 };
enum DeviceTag { kCUDA, kCPU };
struct TensorOptions {
  TensorOptions dtype(DTypeTag) const { return *this; }
  TensorOptions device(DeviceTag, int = 0) const { return *this; }
  TensorOptions requires_grad(bool) const { return *this; }
};
}  // namespace torch
namespace at {
inline Tensor zeros(std::initializer_list<int64_t> s, torch::TensorOptions = {}) {
  Tensor t; t.st = ScalarType::Float; t.shape.assign(s.begin(), s.end()); t.set_strides();
  size_t n = 4; for (auto v : t.shape) n *= (size_t)v;
  t.owned = std::make_shared<std::vector<char>>(n, 0); t.data = t.owned->data();
  return t;
}
}  // namespace at

#define NOF_STR2(x) #x
#define NOF_STR(x) NOF_STR2(x)
#define TORCH_CHECK(cond, ...) do { if (!(cond)) throw std::runtime_error("TORCH_CHECK failed: " #cond " at line " NOF_STR(__LINE__)); } while (0)
#define AT_ASSERTM(cond, ...) do { if (!(cond)) throw std::runtime_error("AT_ASSERTM failed: " #cond); } while (0)
#define NOF_DISPATCH_CASE(ST, T, ...) case at::ScalarType::ST: { using scalar_t = T; __VA_ARGS__(); break; }
#define AT_DISPATCH_FLOATING_TYPES_AND_HALF(TYPE, NAME, ...) \
  switch (TYPE) { NOF_DISPATCH_CASE(Float, float, __VA_ARGS__) NOF_DISPATCH_CASE(Double, double, __VA_ARGS__) \
                  NOF_DISPATCH_CASE(Half, at::Half, __VA_ARGS__) default: throw std::runtime_error(NAME ": not a floating type"); }
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) \
  switch (TYPE) { NOF_DISPATCH_CASE(Float, float, __VA_ARGS__) NOF_DISPATCH_CASE(Double, double, __VA_ARGS__) \
                  default: throw std::runtime_error(NAME ": not a floating type"); }
