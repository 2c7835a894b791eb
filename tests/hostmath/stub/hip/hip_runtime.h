// Host stand-in for <hip/hip_runtime.h>, just enough to compile mpmavatar_amd/csrc/mpm_math.hpp with g++
// (tests/test_hip_math_on_host.py).  Test infrastructure only.
#pragma once
#include <cmath>
#define __device__
#define __host__
#define __forceinline__ inline
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }          // v_rcp_f32 (1 ulp on the device)
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }   // v_rsq_f32
static inline bool __any(bool b) { return b; }                                    // one "lane"
