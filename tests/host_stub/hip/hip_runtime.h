// Host stub of <hip/hip_runtime.h>: lets g++ compile lowthrustopt_amd/csrc/dynamics.hpp so the CPU test-suite can
// check the PRODUCT's device formulas (RHS, variational coefficients, fused column path) against the oracle
// without a GPU.  The reduced-precision seeds mimic v_rsq_f64 / v_rcp_f64 (~2^-23) so the refinement steps are
// exercised too.
#pragma once
#include <cmath>
#define __device__
#define __host__
#define __forceinline__ inline
using std::fabs; using std::fmax; using std::fmin; using std::sqrt; using std::exp; using std::pow; using std::cbrt;
static inline double __builtin_amdgcn_rsq(double x) { return (double)(float)(1.0 / std::sqrt(x)); }
static inline double __builtin_amdgcn_rcp(double x) { return (double)(float)(1.0 / x); }
