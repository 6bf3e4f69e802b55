// Device math shared by the fused kernels: cheap, branch-light replacements for the libm calls that
// dominated the VALU budget of the first version (rocprof r01: k_march was VALU-bound, ~2400 static VALU
// instructions per sample, 131 per sincosf, 174 per powf, 12 per IEEE division).
//
// Accuracy contracts (checked in numpy before adoption):
//   ug_sincos   |x| <= 64: max abs error 9.2e-8 (< 1 ulp at 1.0); results clamp-free in [-1, 1]
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
// Cody-Waite reduction by pi/2 with two FMAs (k <= 41 for |x| <= 64 keeps k*C1 exact inside the FMA),
// Cephes single-precision minimax kernels on |r| <= pi/4, quadrant fix-up with selects.
__device__ __forceinline__ void ug_sincos(float x, float *s, float *c) {
  const float k = rintf(x * 0.636619772f);
  float r = fmaf(-k, 1.5707963705062866f, x);
  r = fmaf(-k, -4.371139000186241e-08f, r);
  const float z = r * r;
  const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
  const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                        z * z, fmaf(-0.5f, z, 1.0f));
  const int q = (int)k;
  const float s0 = (q & 1) ? cp : sp;
  const float c0 = (q & 1) ? sp : cp;
  *s = (q & 2) ? -s0 : s0;
  *c = ((q + 1) & 2) ? -c0 : c0;
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
struct ug_axis_fast { int cell; float wlo, whi; };

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
  a.whi = ix - cf;
  a.wlo = 1.0f - a.whi;
  return a;
}
