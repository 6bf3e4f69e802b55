// Device math shared by the fused kernels: cheap, branch-light replacements for the libm calls that
// dominated the VALU budget of the first version (rocprof r01: k_march was VALU-bound, ~2400 static VALU
// instructions per sample, 131 per sincosf, 174 per powf, 12 per IEEE division).
//
// Accuracy contracts (checked in numpy before adoption):
//   ug_sincos   max abs error 1.2e-7 on |x| <= 4 (1 ulp at 1.0), 1.7e-7 on |x| <= 16
//   ug_div_*    Markstein division: q = RN(x/d) except for rare double-rounding ties (<= 1 ulp)
//   ug_alpha    reproduces 1 - RN(pow(RN(1+e), -interval)) (a correctly rounded powf) exactly for
//               small alpha and to <= 1 ulp of pow elsewhere -- closer to glibc's powf than ocml powf is
#pragma once
#include <hip/hip_runtime.h>

// ---- division ---------------------------------------------------------------------------------
// reciprocal refined to (almost always) correct rounding: one Newton step on v_rcp_f32 (1 ulp)
__device__ __forceinline__ float ug_rcp_refined(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = fmaf(-d, r0, 1.0f);
  return fmaf(e, r0, r0);
}
// x / d given r = RN(1/d): quotient estimate, exact remainder by FMA, correction (Markstein)
__device__ __forceinline__ float ug_div_r(float x, float d, float r) {
  const float q0 = x * r;
  const float rem = fmaf(-q0, d, x);
  return fmaf(rem, r, q0);
}

// ---- sin / cos --------------------------------------------------------------------------------
// Reduction by pi (not pi/2): x = k*pi + r, |r| <= pi/2, so sin x = (-1)^k sin r and cos x = (-1)^k cos r --
// the quadrant fix-up is ONE shared sign bit (cvt, shift, two xors) instead of the swap-and-negate selects
// of a pi/2 reduction, which cost as much as the polynomials.  Two-FMA Cody-Waite reduction (k <= 20 for
// |x| <= 64 keeps k*PI_HI exact inside the FMA); least-squares polynomials on |r| <= pi/2, sin as
// r + r^3 P(r^2) so the leading term is exact.  Max abs error (numpy check, fp64 truth): 1.2e-7 on |x| <= 4,
// 1.7e-7 on |x| <= 16 -- one fp32 ulp at 1.0; the largest |sin| returned is 1 + 1 ulp.
__device__ __forceinline__ void ug_sincos(float x, float *s, float *c) {
  const float k = rintf(x * 0.3183098861837907f);
  float r = fmaf(-k, 3.1415927410125732f, x);
  r = fmaf(-k, -8.742278000372485e-08f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(2.605012241474469e-06f, z, -0.00019808979413937777f), z, 0.008333049714565277f), z,
                        -0.16666658222675323f);
  const float sp = fmaf(r * z, ps, r);
  const float cp = fmaf(fmaf(fmaf(fmaf(fmaf(-2.60495482962142e-07f, z, 2.4760051019256935e-05f), z,
                                            -0.0013888359535485506f), z, 0.04166663438081741f), z, -0.5f), z, 1.0f);
  const unsigned sign = (unsigned)((int)k) << 31;
  *s = __uint_as_float(__float_as_uint(sp) ^ sign);
  *c = __uint_as_float(__float_as_uint(cp) ^ sign);
}

// The same polynomials without the reduction, for |x| <= pi/2: there rint(x/pi) = 0 and both reduction FMAs return x
// itself, so the result is BIT-IDENTICAL to ug_sincos at 11 instead of 19 instructions.  Used for the first Fourier
// frequency, whose argument is the normalised grid coordinate itself (|u| <= 1 after contraction).
__device__ __forceinline__ void ug_sincos_small(float r, float *s, float *c) {
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(2.605012241474469e-06f, z, -0.00019808979413937777f), z, 0.008333049714565277f), z,
                        -0.16666658222675323f);
  *s = fmaf(r * z, ps, r);
  *c = fmaf(fmaf(fmaf(fmaf(fmaf(-2.60495482962142e-07f, z, 2.4760051019256935e-05f), z,
                                -0.0013888359535485506f), z, 0.04166663438081741f), z, -0.5f), z, 1.0f);
}

// ---- raw density -> alpha ----------------------------------------------------------------------
// alpha = 1 - (1 + exp(d + shift))^(-interval)   (render_utils_kernel.cu:439-441)
// evaluated as  t = RN(1+e);  L = log(t);  pw = RN(exp(-interval*L));  alpha = 1 - pw  with log1p / expm1
// style kernels where t is close to 1, so pw carries the same final rounding as a correctly rounded powf.
__device__ __forceinline__ float ug_alpha(float dens_plus_shift, float interval) {
  const float e = expf(dens_plus_shift);  // may be +inf
  const float t = 1.0f + e;
  const float eq = t - 1.0f;              // exact: the quantised e that pow() actually sees
  float L;
  if (eq < 0.0625f) {
    // log1p(x) = 2 atanh(x / (2 + x)),  s < 0.0303:  2s (1 + s^2/3 + s^4/5)
    const float den = 2.0f + eq;
    const float s = ug_div_r(eq, den, ug_rcp_refined(den));
    const float w = s * s;
    L = (2.0f * s) * fmaf(fmaf(0.2f, w, 0.333333343f), w, 1.0f);
  } else {
    L = logf(t);
  }
  const float y = -interval * L;
  float pw;
  if (y > -0.09f) {
    // expm1(y) = y + y^2 (1/2 + y/6 + y^2/24 + y^3/120)
    const float p = fmaf(fmaf(fmaf(fmaf(8.33333377e-3f, y, 4.16666679e-2f), y, 1.66666672e-1f), y, 0.5f), y * y, y);
    pw = 1.0f + p;
  } else {
    pw = expf(y);
  }
  return 1.0f - pw;
}

// ---- in-range trilinear cell set-up --------------------------------------------------------------
// grid_sample(align_corners=True) along one axis for a coordinate already inside [-1, 1] (contracted
// points and sin/cos level coordinates always are): ix = ((c+1)/2)*(n-1) in [0, n-1].
struct ug_axis_fast { int cell; float cellf, wlo, whi; };

__device__ __forceinline__ ug_axis_fast ug_axis_inrange(float c, int n) {
  // ((c+1)/2)*(n-1): fma(c, .5, .5) == RN(c+1)*.5 exactly (scaling by 2 commutes with rounding)
  const float ix = fmaf(c, 0.5f, 0.5f) * (float)(n - 1);
  // cell = floor(ix) clamped to [0, n-2].  For ix < n-1 this is torch's floor and the weights below are
  // torch's (ix - x0) and (x1 - ix) bit for bit: ix - cf is exact (Sterbenz), and (cf+1) - ix is exact for
  // cf >= 1, so it equals RN(1 - (ix - cf)); for cf == 0 the two expressions are literally the same.  At
  // ix == n-1 exactly (c == 1) the clamp moves to cell n-2 and the weights become (0, 1): the live corner n-1
  // with weight 1, as torch computes it from cell n-1.
  // NaN -> med3 returns 0 -> cell 0 (never an out-of-bounds address), weights NaN (sample is dropped).
  const float cf = __builtin_amdgcn_fmed3f(floorf(ix), 0.0f, (float)(n - 2));
  ug_axis_fast a;
  a.cell = (int)cf;
  a.cellf = cf;
  a.whi = ix - cf;
  a.wlo = 1.0f - a.whi;
  return a;
}
