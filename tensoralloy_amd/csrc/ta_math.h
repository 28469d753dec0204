// Device math helpers (fp64) for the descriptor kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace ta {

// exp(x) for x <= ~50 (the kernels only call it with x <= 0): round-to-nearest
// range reduction with a two-part ln2 and a degree-13 Taylor polynomial on
// |r| <= ln2/2 (truncation 4e-18 relative), scaled by v_ldexp_f64.
__device__ __forceinline__ double ta_exp(double x) {
  const double kLog2e = 1.4426950408889634074;
  const double kLn2Hi = 6.93147180369123816490e-01;
  const double kLn2Lo = 1.90821492927058770002e-10;
  double kf = rint(x * kLog2e);
  double r = fma(-kf, kLn2Hi, x);
  r = fma(-kf, kLn2Lo, r);
  double p = 1.0 / 6227020800.0;
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  kf = fmax(kf, -1100.0);
  return ldexp(p, (int)kf);
}

// Coefficients of Y(u) = cos(pi sqrt(u) / 2) = sum_k (-pi^2 u / 4)^k / (2k)!.
// The cosine cutoff 0.5 (cos(pi r / rc) + 1) (reference nn/cutoff.py:43-48)
// equals Y(u)^2 with u = (r / rc)^2, so no square root or trigonometric
// instruction is needed for a distance known only through its square.
constexpr double kQ = -2.4674011002723396547;  // -(pi^2) / 4
constexpr double cos_coef(int k) {
  double v = 1.0;
  for (int j = 1; j <= k; ++j) v *= kQ / double((2 * j - 1) * (2 * j));
  return v;
}

// fc(u) and d fc / d u for u = (r / rc)^2 in [0, 1). Callers mask u >= 1.
__device__ __forceinline__ void cutoff_u(int kind, double u, double &f, double &dfdu) {
  if (kind == TA_CUTOFF_COSINE) {
    constexpr double c0 = cos_coef(0), c1 = cos_coef(1), c2 = cos_coef(2), c3 = cos_coef(3),
                     c4 = cos_coef(4), c5 = cos_coef(5), c6 = cos_coef(6), c7 = cos_coef(7),
                     c8 = cos_coef(8), c9 = cos_coef(9), c10 = cos_coef(10), c11 = cos_coef(11);
    double y = c11;
    y = fma(y, u, c10);
    y = fma(y, u, c9);
    y = fma(y, u, c8);
    y = fma(y, u, c7);
    y = fma(y, u, c6);
    y = fma(y, u, c5);
    y = fma(y, u, c4);
    y = fma(y, u, c3);
    y = fma(y, u, c2);
    y = fma(y, u, c1);
    y = fma(y, u, c0);
    double d = 11.0 * c11;
    d = fma(d, u, 10.0 * c10);
    d = fma(d, u, 9.0 * c9);
    d = fma(d, u, 8.0 * c8);
    d = fma(d, u, 7.0 * c7);
    d = fma(d, u, 6.0 * c6);
    d = fma(d, u, 5.0 * c5);
    d = fma(d, u, 4.0 * c4);
    d = fma(d, u, 3.0 * c3);
    d = fma(d, u, 2.0 * c2);
    d = fma(d, u, c1);
    f = y * y;
    dfdu = 2.0 * y * d;
  } else {
    // polynomial cutoff, gamma = 5 (reference nn/cutoff.py:78-85):
    // f = 1 + 5 x^6 - 6 x^5, x = sqrt(u);  df/du = 15 x^3 (x - 1)
    double x = sqrt(u);
    double x2 = u, x3 = x2 * x, x5 = x3 * x2;
    f = 1.0 + 5.0 * x5 * x - 6.0 * x5;
    dfdu = 15.0 * x3 * (x - 1.0);
  }
}

// fc(u), d fc / d u and d^2 fc / d u^2 (analytic Hessian-vector products, ta_hvp.hip)
__device__ __forceinline__ void cutoff_u2(int kind, double u, double &f, double &dfdu, double &d2fdu2) {
  if (kind == TA_CUTOFF_COSINE) {
    double y = 0.0, d = 0.0, dd = 0.0;
#pragma unroll
    for (int k = 11; k >= 0; --k) {  // Horner with first and second derivative
      dd = fma(dd, u, 2.0 * d);
      d = fma(d, u, y);
      y = fma(y, u, cos_coef(k));
    }
    f = y * y;
    dfdu = 2.0 * y * d;
    d2fdu2 = 2.0 * (d * d + y * dd);
  } else {
    const double x = sqrt(u);
    const double x3 = u * x, x5 = x3 * u;
    f = 1.0 + 5.0 * x5 * x - 6.0 * x5;
    dfdu = 15.0 * x3 * (x - 1.0);
    d2fdu2 = 30.0 * u - 22.5 * x;
  }
}

// value only
__device__ __forceinline__ double cutoff_u_value(int kind, double u) {
  if (kind == TA_CUTOFF_COSINE) {
    constexpr double c0 = cos_coef(0), c1 = cos_coef(1), c2 = cos_coef(2), c3 = cos_coef(3),
                     c4 = cos_coef(4), c5 = cos_coef(5), c6 = cos_coef(6), c7 = cos_coef(7),
                     c8 = cos_coef(8), c9 = cos_coef(9), c10 = cos_coef(10), c11 = cos_coef(11);
    double y = c11;
    y = fma(y, u, c10);
    y = fma(y, u, c9);
    y = fma(y, u, c8);
    y = fma(y, u, c7);
    y = fma(y, u, c6);
    y = fma(y, u, c5);
    y = fma(y, u, c4);
    y = fma(y, u, c3);
    y = fma(y, u, c2);
    y = fma(y, u, c1);
    y = fma(y, u, c0);
    return y * y;
  } else {
    double x = sqrt(u);
    double x5 = u * u * x;
    return 1.0 + 5.0 * x5 * x - 6.0 * x5;
  }
}

// base^(zi - 1) for integer zi >= 1 (wave-uniform exponent).
__device__ __forceinline__ double pow_int_m1(double base, int zi) {
  double r = 1.0, b = base;
  int e = zi - 1;
  while (e) {
    if (e & 1) r *= b;
    e >>= 1;
    if (e) b *= b;
  }
  return r;
}

// Cross-lane sums. Inside a row of 16 lanes the data moves with DPP row
// rotations on the VALU (no LDS traffic, unlike ds_bpermute-based shuffles):
// after rotations by 8, 4, 2, 1 every lane of the row holds the row's total.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_move<0x128>(v);  // row_ror:8
  v += dpp_move<0x124>(v);  // row_ror:4
  v += dpp_move<0x122>(v);  // row_ror:2
  v += dpp_move<0x121>(v);  // row_ror:1
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// 1 / d for d in [1, 3]: hardware seed + two Newton steps (no scaling / fix-up needed in that range)
__device__ __forceinline__ double rcp_1_3(double d) {
  double x = __builtin_amdgcn_rcp(d);
  double e = fma(-d, x, 1.0);
  x = fma(x, e, x);
  e = fma(-d, x, 1.0);
  return fma(x, e, x);
}

// softplus(x) = max(x, 0) + log1p(e) and its derivative 1 / (1 + exp(-x)), e = exp(-|x|), to 2-3 ulp
// in about a quarter of the instructions of libm's exp + log1p + two IEEE divisions (the per-pair
// nn functions of nn-EAM spend nearly all their time here):
//   e: |x| = n ln2 + r, |r| <= ln2 / 2, degree-13 Taylor of exp(-r), ldexp;
//   log1p(e) = [ln2 +] 2 atanh(s), s = f / (2 + f) with f = e, or (e - 1) / 2 when e > sqrt2 - 1
//   (|s| <= 0.172: ten odd terms); both quotients by rcp_1_3.
__device__ __forceinline__ void softplus_fn(double x, double &h, double &dh) {
  const double a = fmin(fabs(x), 708.0);
  const double n = __builtin_rint(a * 1.4426950408889634);
  double y = fma(n, -6.93147180369123816490e-01, a);
  y = -fma(n, -1.90821492927058770002e-10, y);
  double p = 1.0 / 6227020800.0;
  p = fma(p, y, 1.0 / 479001600.0);
  p = fma(p, y, 1.0 / 39916800.0);
  p = fma(p, y, 1.0 / 3628800.0);
  p = fma(p, y, 1.0 / 362880.0);
  p = fma(p, y, 1.0 / 40320.0);
  p = fma(p, y, 1.0 / 5040.0);
  p = fma(p, y, 1.0 / 720.0);
  p = fma(p, y, 1.0 / 120.0);
  p = fma(p, y, 1.0 / 24.0);
  p = fma(p, y, 1.0 / 6.0);
  p = fma(p, y, 0.5);
  p = fma(p, y, 1.0);
  p = fma(p, y, 1.0);
  const double e = __builtin_amdgcn_ldexp(p, -(int)n);
  const double inv = rcp_1_3(1.0 + e);
  dh = (x >= 0.0) ? inv : e * inv;
  const bool big = e > 0.41421356237309503;
  const double f = big ? 0.5 * (e - 1.0) : e;
  const double s = f * rcp_1_3(2.0 + f);
  const double z = s * s;
  double q = 1.0 / 21.0;
  q = fma(q, z, 1.0 / 19.0);
  q = fma(q, z, 1.0 / 17.0);
  q = fma(q, z, 1.0 / 15.0);
  q = fma(q, z, 1.0 / 13.0);
  q = fma(q, z, 1.0 / 11.0);
  q = fma(q, z, 1.0 / 9.0);
  q = fma(q, z, 1.0 / 7.0);
  q = fma(q, z, 1.0 / 5.0);
  q = fma(q, z, 1.0 / 3.0);
  q *= z;
  double l = fma(2.0 * s, q, 2.0 * s);
  if (big) l += 0.69314718055994531;
  h = fmax(x, 0.0) + l;
}

// activation value and derivative (reference nn/utils.py:39-74)
__device__ __forceinline__ void activation_fn(int act, double x, double &h, double &dh) {
  switch (act) {
    case TA_ACT_SOFTPLUS:
      softplus_fn(x, h, dh);
      break;
    case TA_ACT_RELU:
      h = fmax(x, 0.0);
      dh = x > 0.0 ? 1.0 : 0.0;
      break;
    case TA_ACT_LEAKY_RELU:
      h = x > 0.0 ? x : 0.2 * x;
      dh = x > 0.0 ? 1.0 : 0.2;
      break;
    case TA_ACT_TANH: {
      double t = tanh(x);
      h = t;
      dh = 1.0 - t * t;
      break;
    }
    case TA_ACT_SIGMOID: {
      double e = exp(-fabs(x));
      double s = (x >= 0.0) ? 1.0 / (1.0 + e) : e / (1.0 + e);
      h = s;
      dh = s * (1.0 - s);
      break;
    }
    case TA_ACT_SOFTSIGN: {
      double d = 1.0 + fabs(x);
      h = x / d;
      dh = 1.0 / (d * d);
      break;
    }
    case TA_ACT_ELU:
      h = x > 0.0 ? x : expm1(x);
      dh = x > 0.0 ? 1.0 : exp(x);
      break;
    case TA_ACT_SQUAREPLUS: {
      double s = sqrt(x * x + 4.0);
      h = 0.5 * (x + s);
      dh = 0.5 * (1.0 + x / s);
      break;
    }
    default:
      h = x;
      dh = 1.0;
  }
}

// activation value, first and second derivative (the weight gradient of a force / stress loss
// differentiates the MLP's input gradient once more, nn/losses.py:285-437)
__device__ __forceinline__ void activation_fn2(int act, double x, double &h, double &dh, double &d2h) {
  activation_fn(act, x, h, dh);
  switch (act) {
    case TA_ACT_SOFTPLUS:
    case TA_ACT_SIGMOID:
      // softplus' = sigmoid = s: s' = s (1 - s); sigmoid'' = s (1 - s)(1 - 2 s) with dh = s (1 - s)
      d2h = (act == TA_ACT_SOFTPLUS) ? dh * (1.0 - dh) : dh * (1.0 - 2.0 * h);
      break;
    case TA_ACT_TANH:
      d2h = -2.0 * h * dh;
      break;
    case TA_ACT_SOFTSIGN: {
      const double d = 1.0 + fabs(x);
      d2h = (x >= 0.0 ? -2.0 : 2.0) / (d * d * d);
      break;
    }
    case TA_ACT_ELU:
      d2h = x > 0.0 ? 0.0 : exp(x);
      break;
    case TA_ACT_SQUAREPLUS: {
      const double s = sqrt(x * x + 4.0);
      d2h = 2.0 / (s * s * s);
      break;
    }
    default:  // relu, leaky relu, linear
      d2h = 0.0;
  }
}

// `safe_pow` of reference extension/grad_ops.py:16-74. Without TENSORALLOY_USE_CUSTOM_POW it is plain
// `tf.pow`, whose gradient y x^(y-1) is Inf at x = 0 for y < 1 (safe = 0: IEEE pow, the same). The
// custom variant (safe = 1) zeroes an infinite value (:25-26) and an infinite or NaN gradient
// factor (:46-49); a NaN value (negative base, non-integer exponent) stays NaN in both.
__device__ __forceinline__ double safe_pow_value(int safe, double x, double y) {
  const double z = pow(x, y);
  return (safe && isinf(z)) ? 0.0 : z;
}
// x^(y - 1) as it enters d(x^y)/dx = y x^(y-1)
__device__ __forceinline__ double safe_pow_grad(int safe, double x, double ym1) {
  const double z = pow(x, ym1);
  return (safe && !isfinite(z)) ? 0.0 : z;
}

}  // namespace ta
