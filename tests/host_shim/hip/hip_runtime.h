// Host stand-in for <hip/hip_runtime.h>, used ONLY by the CPU unit test of the device-side clip bookkeeping
// (tests/test_clip_tri_host.py compiles xugrid_amd/csrc/xr_clip_tri.h with g++ against this file): just enough for
// the per-lane device functions to compile as plain C++.  Never part of the product build.
#pragma once
#include <cmath>
#include <cstdint>
#define __device__
#define __host__
#define __forceinline__ inline
struct double2 { double x, y; };
struct double4 { double x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
struct HostThreadIdx { int x; };
static thread_local HostThreadIdx threadIdx{0};
