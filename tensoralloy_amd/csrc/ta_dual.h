// First-order dual numbers for the gradients with respect to the constants of the analytic EAM
// functions: the function templates of ta_eam.hip are instantiated once with double (the inference
// path) and once with Dual, whose `d` carries the derivative with respect to ONE seeded constant.
#pragma once
#include <hip/hip_runtime.h>

#include "ta_math.h"

namespace ta {

struct Dual {
  double v, d;
};

__host__ __device__ __forceinline__ Dual make_dual(double v, double d = 0.0) { return Dual{v, d}; }

__device__ __forceinline__ Dual operator-(Dual a) { return {-a.v, -a.d}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator+(Dual a, double b) { return {a.v + b, a.d}; }
__device__ __forceinline__ Dual operator+(double a, Dual b) { return {a + b.v, b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return {a.v - b, a.d}; }
__device__ __forceinline__ Dual operator-(double a, Dual b) { return {a - b.v, -b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, fma(a.v, b.d, a.d * b.v)}; }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return {a.v * b, a.d * b}; }
__device__ __forceinline__ Dual operator*(double a, Dual b) { return {a * b.v, a * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const double q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual operator/(Dual a, double b) { return {a.v / b, a.d / b}; }
__device__ __forceinline__ Dual operator/(double a, Dual b) {
  const double q = a / b.v;
  return {q, -q * b.d / b.v};
}
__device__ __forceinline__ Dual &operator+=(Dual &a, Dual b) {
  a.v += b.v;
  a.d += b.d;
  return a;
}

// the scalar functions the potentials use, for both instantiations
__device__ __forceinline__ double t_val(double x) { return x; }
__device__ __forceinline__ double t_val(Dual x) { return x.v; }
__device__ __forceinline__ double t_exp(double x) { return ta_exp(x); }
__device__ __forceinline__ Dual t_exp(Dual x) {
  const double e = ta_exp(x.v);
  return {e, e * x.d};
}
__device__ __forceinline__ double t_exp_libm(double x) { return exp(x); }
__device__ __forceinline__ Dual t_exp_libm(Dual x) {
  const double e = exp(x.v);
  return {e, e * x.d};
}
__device__ __forceinline__ double t_log(double x) { return log(x); }
__device__ __forceinline__ Dual t_log(Dual x) { return {log(x.v), x.d / x.v}; }
__device__ __forceinline__ double t_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ Dual t_sqrt(Dual x) {
  const double s = sqrt(x.v);
  return {s, 0.5 * x.d / s};
}
__device__ __forceinline__ double t_pow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ Dual t_pow(Dual x, Dual y) {
  const double v = pow(x.v, y.v);
  // d(x^y) = x^y (y' ln x + y x' / x); the terms with a zero seed are skipped (x may be 0 there)
  double d = 0.0;
  if (y.d != 0.0) d += y.d * log(x.v);
  if (x.d != 0.0) d += y.v * x.d / x.v;
  return {v, v * d};
}
__device__ __forceinline__ Dual t_pow(Dual x, double y) { return t_pow(x, make_dual(y)); }
__device__ __forceinline__ Dual t_pow(double x, Dual y) { return t_pow(make_dual(x), y); }
__device__ __forceinline__ double t_floor_at(double x, double lo) { return fmax(x, lo); }
__device__ __forceinline__ Dual t_floor_at(Dual x, double lo) { return x.v > lo ? x : make_dual(lo); }

}  // namespace ta
