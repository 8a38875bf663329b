// TEST INFRASTRUCTURE: minimal stand-in for <hip/hip_runtime.h> so that the per-link device math
// headers (csrc/su3_math.hpp, csrc/su3_train_math.hpp) can be compiled for the host with g++ and
// checked on the CPU-only build container (tests/test_native_host.py).
#pragma once
#include <cmath>
#define __device__
#define __forceinline__ inline
struct double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return {x, y}; }
using std::fma; using std::sqrt; using std::cos; using std::sin; using std::atan2; using std::fabs; using std::acos; using std::frexp; using std::ldexp;
