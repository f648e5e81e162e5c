// Small per-lane vector / quaternion / spatial algebra for the HIP physics kernels.
// Conventions: quaternions (w,x,y,z); 3x3 row-major; spatial vectors [rot(3); lin(3)].
#pragma once
#ifdef FB_EMULATE
#include "emu/hip_emu.hpp"
#else
#include <hip/hip_runtime.h>
#endif

#define FBD __device__ __forceinline__
// inlining policy of the heavy stages (tuned by measurement: isolating the solver loops keeps their register
// allocation independent of the rest of the stage machine)
#ifdef FB_EMULATE
#define FB_NOINLINE
#else
#define FB_NOINLINE __attribute__((noinline))
#endif
#ifndef FB_INL_A
#define FB_INL_A 0      // (1,1) trips an AMDGPU back-end assertion in ROCm 7.2 (private-base compare); (0,1) measured equal
#endif
#ifndef FB_INL_B
#define FB_INL_B 1
#endif
#if FB_INL_A
#define FB_STAGE_A __device__ FB_NOINLINE
#else
#define FB_STAGE_A __device__ __forceinline__
#endif
#ifndef FB_INL_FS
#define FB_INL_FS 1
#endif
#if FB_INL_FS
#define FB_STAGE_FS __device__ FB_NOINLINE
#else
#define FB_STAGE_FS __device__ __forceinline__
#endif
#if FB_INL_B
#define FB_STAGE_B __device__ FB_NOINLINE
#else
#define FB_STAGE_B __device__ __forceinline__
#endif
#define FB_MINV ((real)1e-15)

// division on solver hot paths: exact in FP64 (bit-faithful to the oracle), v_rcp_f32 (1 ulp) in FP32
// FP64: IEEE division costs ~25 dependent instructions (v_div_scale x2, v_rcp_f64, 5 FMAs, v_div_fmas, v_div_fixup) in the
// middle of the solver's serial chain.  Unless FB_EXACT_DIV64 is defined the quotient is v_rcp_f64 + two Newton steps + one residual
// correction (<= 1 ulp for the normal-range operands the solver produces, no denormal / overflow handling).
#if !defined(FB_EXACT_DIV64) && !defined(FB_EMULATE)
#ifndef FB_DIV_STEPS
#define FB_DIV_STEPS 2
#endif
FBD double fb_div(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
#pragma unroll
  for (int k = 0; k < FB_DIV_STEPS; k++) r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
  double q = a*r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
#else
FBD double fb_div(double a, double b) { return a / b; }
#endif
#ifdef FB_EMULATE
FBD float fb_div(float a, float b) { return a / b; }
#else
FBD float fb_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#endif
#if !defined(FB_EXACT_DIV64) && !defined(FB_EMULATE)
// (round 6: ONE Newton step behind v_rsq_f64 -- a few ulp instead of <= 1 -- measured +0.5 % env-steps/s with the 100-control-step divergence from the
//  oracle unchanged at 7e-15 / 4e-14 on qpos / qvel, profiles/r6/ab_reciprocals.txt; FB_RSQ_STEPS 2 restores the second)
#ifndef FB_RSQ_STEPS
#define FB_RSQ_STEPS 1
#endif
FBD double fb_rsqrt(double a) {
  double y = __builtin_amdgcn_rsq(a);
#pragma unroll
  for (int k = 0; k < FB_RSQ_STEPS; k++) y = y*__builtin_fma(-0.5*a*y, y, 1.5);
  return y;
}
#else
FBD double fb_rsqrt(double a) { return 1.0 / sqrt(a); }
#endif
#ifdef FB_EMULATE
FBD float fb_rsqrt(float a) { return 1.0f / sqrtf(a); }
#else
FBD float fb_rsqrt(float a) { return __builtin_amdgcn_rsqf(a); }
#endif
// sqrt / reciprocal on the same footing: v_rsq_f64 / v_rcp_f64 + Newton steps (<= 1 ulp) instead of the IEEE sequences (~30 dependent
// instructions each) -- used by the vector normalisations and the iterative collision query, which sit on serial chains.
// The host emulation and FB_EXACT_DIV64 builds keep the exact operations.
#if !defined(FB_EXACT_DIV64) && !defined(FB_EMULATE)
FBD double fb_sqrt(double a) { return a > 0 ? a*fb_rsqrt(a) : 0.0; }
FBD float fb_sqrt(float a) { return a > 0 ? a*fb_rsqrt(a) : 0.0f; }
#else
FBD double fb_sqrt(double a) { return sqrt(a); }
FBD float fb_sqrt(float a) { return sqrtf(a); }
#endif
// 1 / a: v_rcp_f64 + Newton steps.  fb_div(1, a) is rcp + THREE steps (its residual correction of q = 1 * r is a third one): the second already
// leaves the rounding of the last fma as the only error, and a reciprocal sits on the serial chain of every pivot (Gauss-Jordan on the
// Newton tiles) and of every level of the factorisations.  FB_INV_STEPS 3 restores the old sequence.
#ifndef FB_INV_STEPS
#define FB_INV_STEPS 2
#endif
#if !defined(FB_EXACT_DIV64) && !defined(FB_EMULATE) && FB_INV_STEPS < 3
FBD double fb_inv(double a) {
  double r = __builtin_amdgcn_rcp(a);
#pragma unroll
  for (int k = 0; k < FB_INV_STEPS; k++) r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
  return r;
}
#else
FBD double fb_inv(double a) { return fb_div(1.0, a); }
#endif
FBD float fb_inv(float a) { return fb_div(1.0f, a); }

template <typename real> FBD real dot3(const real* a, const real* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
template <typename real> FBD void cross3(real* r, const real* a, const real* b) {
  real x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename real> FBD void copy3(real* r, const real* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
template <typename real> FBD void sub3(real* r, const real* a, const real* b) { r[0] = a[0]-b[0]; r[1] = a[1]-b[1]; r[2] = a[2]-b[2]; }
template <typename real> FBD void add3(real* r, const real* a, const real* b) { r[0] = a[0]+b[0]; r[1] = a[1]+b[1]; r[2] = a[2]+b[2]; }
template <typename real> FBD void scl3(real* r, const real* a, real s) { r[0] = a[0]*s; r[1] = a[1]*s; r[2] = a[2]*s; }
template <typename real> FBD void addscl3(real* r, const real* a, real s) { r[0] += a[0]*s; r[1] += a[1]*s; r[2] += a[2]*s; }
template <typename real> FBD real norm3(const real* a) { return fb_sqrt(dot3(a, a)); }
#ifndef FB_RSQ_NORM
#define FB_RSQ_NORM 1
#endif
// norm and its reciprocal from ONE reciprocal square root (device builds: fb_sqrt is a * rsqrt(a) there anyway; the division that
// followed it was 8 more instructions on the chain of every normalisation and support-function call)
#if FB_RSQ_NORM && !defined(FB_EXACT_DIV64) && !defined(FB_EMULATE)
template <typename real> FBD real norm_rnorm(real n2, real& rn) { rn = n2 > 0 ? fb_rsqrt(n2) : (real)0; return n2*rn; }
#else
template <typename real> FBD real norm_rnorm(real n2, real& rn) { const real n = fb_sqrt(n2); rn = fb_inv(n > 0 ? n : (real)1); return n; }
#endif
template <typename real> FBD real normalize3(real* a) {
  real inv;
  const real n = norm_rnorm(dot3(a, a), inv);
  if (n < FB_MINV) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
  a[0] *= inv; a[1] *= inv; a[2] *= inv;
  return n;
}
template <typename real> FBD real dot6(const real* a, const real* b) {
  return a[0]*b[0] + a[1]*b[1] + a[2]*b[2] + a[3]*b[3] + a[4]*b[4] + a[5]*b[5];
}
template <typename real> FBD void mulmat3(real* r, const real* m, const real* v) {
  real x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  real y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  real z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename real> FBD void mulmatT3(real* r, const real* m, const real* v) {
  real x = m[0]*v[0] + m[3]*v[1] + m[6]*v[2];
  real y = m[1]*v[0] + m[4]*v[1] + m[7]*v[2];
  real z = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
template <typename real> FBD void mulquat(real* r, const real* a, const real* b) {
  real w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  real x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  real y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  real z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
template <typename real> FBD void normquat(real* q) {
  real n = fb_sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < FB_MINV) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { real inv = fb_inv(n); q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv; }
}
template <typename real> FBD void quat2mat(real* m, const real* q) {
  real q00 = q[0]*q[0], q01 = q[0]*q[1], q02 = q[0]*q[2], q03 = q[0]*q[3];
  real q11 = q[1]*q[1], q12 = q[1]*q[2], q13 = q[1]*q[3];
  real q22 = q[2]*q[2], q23 = q[2]*q[3], q33 = q[3]*q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2*(q12 - q03); m[2] = 2*(q13 + q02);
  m[3] = 2*(q12 + q03); m[5] = 2*(q23 - q01);
  m[6] = 2*(q13 - q02); m[7] = 2*(q23 + q01);
}
template <typename real> FBD void rotvecquat(real* r, const real* v, const real* q) {
  real m[9]; quat2mat(m, q); mulmat3(r, m, v);
}
template <typename real> FBD void axisangle2quat(real* q, const real* axis, real ang) {
  real s = sin((real)0.5*ang);
  q[0] = cos((real)0.5*ang); q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
template <typename real> FBD void mulinertvec(real* r, const real* i, const real* v) {
  r[0] = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  r[1] = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  r[2] = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  r[3] = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  r[4] = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  r[5] = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
}
template <typename real> FBD void crossmotion(real* r, const real* vel, const real* v) {
  r[0] = -vel[2]*v[1] + vel[1]*v[2];
  r[1] =  vel[2]*v[0] - vel[0]*v[2];
  r[2] = -vel[1]*v[0] + vel[0]*v[1];
  r[3] = -vel[2]*v[4] + vel[1]*v[5] - vel[5]*v[1] + vel[4]*v[2];
  r[4] =  vel[2]*v[3] - vel[0]*v[5] + vel[5]*v[0] - vel[3]*v[2];
  r[5] = -vel[1]*v[3] + vel[0]*v[4] - vel[4]*v[0] + vel[3]*v[1];
}
template <typename real> FBD void crossforce(real* r, const real* vel, const real* f) {
  r[0] = -vel[2]*f[1] + vel[1]*f[2] - vel[5]*f[4] + vel[4]*f[5];
  r[1] =  vel[2]*f[0] - vel[0]*f[2] + vel[5]*f[3] - vel[3]*f[5];
  r[2] = -vel[1]*f[0] + vel[0]*f[1] - vel[4]*f[3] + vel[3]*f[4];
  r[3] = -vel[2]*f[4] + vel[1]*f[5];
  r[4] =  vel[2]*f[3] - vel[0]*f[5];
  r[5] = -vel[1]*f[3] + vel[0]*f[4];
}
template <typename real> FBD void makeframe(real* f) {
  real* x = f; real* y = f + 3; real* z = f + 6;
  normalize3(x);
  if (x[1] > (real)0.5 || x[1] < (real)-0.5) { y[0] = 0; y[1] = 0; y[2] = 1; }
  else { y[0] = 0; y[1] = 1; y[2] = 0; }
  real d = dot3(x, y);
  addscl3(y, x, -d);
  normalize3(y);
  cross3(z, x, y);
}
template <typename real> FBD real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

// hide a register value from loop-invariant code motion: values derived from it are recomputed where they are
// used instead of being hoisted out of a loop into (spilled) registers
#ifdef FB_EMULATE
#define FB_OPAQUE(x) do {} while (0)
#define FB_SETPRIO(p) do {} while (0)
#define FB_LDS_AS
#else
#define FB_OPAQUE(x) asm volatile("" : "+v"(x))
#define FB_SETPRIO(p) do { int p_ = (p); if (p_ <= 0) __builtin_amdgcn_s_setprio(0); else if (p_ == 1) __builtin_amdgcn_s_setprio(1); else if (p_ == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); } while (0)
#define FB_LDS_AS __attribute__((address_space(3)))
#endif

// issue priority of the latency-bound stage class (fb_step.hpp: d_run; 0 = no stage classes)
#ifndef FB_LAT_PRIO
#define FB_LAT_PRIO 2
#endif

// Arguments of a non-inlined device function arrive in VGPRs even when they are wave-uniform.  Moving a uniform
// pointer to SGPRs (v_readfirstlane) frees two VGPRs per pointer and lets loads through it use scalar addressing.
#ifdef FB_EMULATE
template <typename T> FBD T* uniform_ptr(T* p) { return p; }
FBD int uniform_int(int v) { return v; }
#else
template <typename T> FBD T* uniform_ptr(T* p) {
  unsigned long long v = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
template <typename T> FBD FB_LDS_AS T* uniform_ptr(FB_LDS_AS T* p) {
  return (FB_LDS_AS T*)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)p);
}
FBD int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif
// any pointer flavour (generic / global / constant / LDS) to SGPRs
#ifdef FB_EMULATE
template <typename P> FBD P uniform_p(P p) { return p; }
#else
template <typename P> FBD P uniform_p(P p) {
  unsigned long long v = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (P)(((unsigned long long)hi << 32) | lo);
}
#endif
// A struct that sits in read-only device memory and is addressed uniformly: tell the compiler (constant address space),
// its fields then come in through the scalar cache and may be re-read or kept across fences freely.
#ifdef FB_EMULATE
template <typename T> FBD const T& as_constant(const T& r) { return r; }
#else
template <typename T> FBD const T& as_constant(const T& r) {
  const __attribute__((address_space(4))) T* p4 = (const __attribute__((address_space(4))) T*)uniform_ptr(&r);
  // opaque to the optimiser: otherwise the generic -> constant -> generic round trip is folded away in places and the
  // loads behind it fall back to flat_load
  asm volatile("" : "+s"(p4));
  return *(const T*)p4;
}
#endif

// ---- wavefront (64-lane) collectives -------------------------------------------------
#ifdef FB_EMULATE
template <typename T> FBD T rdlane(T v, int src) { return __shfl(v, src, 64); }
FBD double shfl_xor_r(double v, int m) { return __shfl_xor(v, m, 64); }
FBD float shfl_xor_r(float v, int m) { return __shfl_xor(v, m, 64); }
template <typename real> FBD real wave_sum(real v) {
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_r(v, m);
  return v;
}
#else
// broadcast from a wave-uniform source lane: v_readlane, no LDS crossbar round trip
FBD int rdlane(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
FBD float rdlane(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
FBD double rdlane(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
template <int CTRL> FBD float dpp_mov(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
template <int CTRL> FBD double dpp_mov(double v) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// full-wave sum in every lane: DPP butterflies inside each row of 16 lanes, then 4 readlanes.
// Must be called with all 64 lanes active.
template <typename real> FBD real wave_sum(real v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return (rdlane(v, 0) + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
}
#endif
// wave_sum of a value that is ZERO outside the first 16 lanes when `row0` (wave-uniform) says so: the three other row sums are exact
// zeros, so their six v_readlane and three additions are skipped -- same bits as wave_sum.  All 64 lanes must call.
#ifdef FB_EMULATE
template <typename real> FBD real wave_sum_lo(real v, bool) { return wave_sum(v); }
#else
template <typename real> FBD real wave_sum_lo(real v, bool row0) {
  v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
  real r = rdlane(v, 0);
  if (!row0) r = (r + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
  return r;
}
#endif
// sum over the four lanes of a quad (lanes 4k .. 4k+3), result in all four; every lane must call
#ifdef FB_EMULATE
template <typename real> FBD real quad_sum(real v) { v += shfl_xor_r(v, 1); v += shfl_xor_r(v, 2); return v; }
#else
template <typename real> FBD real quad_sum(real v) { v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); return v; }
#endif
FBD double shfl_xor_any(double v, int m) { return __shfl_xor(v, m, 64); }
FBD float shfl_xor_any(float v, int m) { return __shfl_xor(v, m, 64); }
FBD int wave_sum_i(int v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
// exclusive prefix sum of a small per-lane integer
FBD int wave_excl_scan(int v, int lane) {
  int s = v;
  for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(s, d, 64); if (lane >= d) s += t; }
  return s - v;
}
