// Host stub of <hip/hip_runtime.h>: lets g++ compile lowthrustopt_amd/csrc/dynamics.hpp so the CPU test-suite can
// check the PRODUCT's device formulas (RHS, variational coefficients, fused column path) against the oracle
// without a GPU.  The reduced-precision seeds mimic v_rsq_f64 / v_rcp_f64 (~2^-23) so the refinement steps are
// exercised too.
#pragma once
#include <cmath>
#define __device__
#define __host__
#define __forceinline__ inline
#define __constant__ static const
using std::fabs; using std::fmax; using std::fmin; using std::sqrt; using std::exp; using std::pow; using std::cbrt;
static inline double __builtin_amdgcn_rsq(double x) { return (double)(float)(1.0 / std::sqrt(x)); }
static inline double __builtin_amdgcn_rcp(double x) { return (double)(float)(1.0 / x); }
#include <cstring>
static inline long long __double_as_longlong(double x) { long long r; std::memcpy(&r, &x, 8); return r; }
static inline double __longlong_as_double(long long x) { double r; std::memcpy(&r, &x, 8); return r; }
static inline int __double2hiint(double x) { return (int)(__double_as_longlong(x) >> 32); }
static inline int __double2loint(double x) { return (int)(__double_as_longlong(x) & 0xffffffffll); }
static inline double __hiloint2double(int hi, int lo) { return __longlong_as_double(((long long)hi << 32) | (unsigned)lo); }
