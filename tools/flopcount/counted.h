// Operation-counting stand-in for `double`, used ONLY by tools/flopcount (instrumented build of the CPU oracle):
// every arithmetic operator and math function on the type bumps a per-category counter.  System headers are included
// first, then `double` is redefined for the oracle's translation units.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>
#include <omp.h>
#include <type_traits>

extern long long fbo_cnt[6];      // add/sub, mul, div, sqrt, transcendental, compare
struct CD {
  double v;
  CD() = default;
  template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> CD(T x) : v((double)x) {}
  explicit operator int() const { return (int)v; }
  explicit operator unsigned() const { return (unsigned)v; }
  explicit operator long() const { return (long)v; }
  explicit operator long long() const { return (long long)v; }
  explicit operator float() const { return (float)v; }
  explicit operator bool() const { return v != 0; }
  double raw() const { return v; }
  CD& operator+=(CD b) { fbo_cnt[0]++; v += b.v; return *this; }
  CD& operator-=(CD b) { fbo_cnt[0]++; v -= b.v; return *this; }
  CD& operator*=(CD b) { fbo_cnt[1]++; v *= b.v; return *this; }
  CD& operator/=(CD b) { fbo_cnt[2]++; v /= b.v; return *this; }
  CD operator-() const { CD r; r.v = -v; return r; }
  CD operator+() const { return *this; }
};
#define FB_AR(T) template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
inline CD operator+(CD a, CD b) { fbo_cnt[0]++; CD r; r.v = a.v + b.v; return r; }
inline CD operator-(CD a, CD b) { fbo_cnt[0]++; CD r; r.v = a.v - b.v; return r; }
inline CD operator*(CD a, CD b) { fbo_cnt[1]++; CD r; r.v = a.v * b.v; return r; }
inline CD operator/(CD a, CD b) { fbo_cnt[2]++; CD r; r.v = a.v / b.v; return r; }
FB_AR(T) inline CD operator+(CD a, T b) { return a + CD(b); }
FB_AR(T) inline CD operator+(T a, CD b) { return CD(a) + b; }
FB_AR(T) inline CD operator-(CD a, T b) { return a - CD(b); }
FB_AR(T) inline CD operator-(T a, CD b) { return CD(a) - b; }
FB_AR(T) inline CD operator*(CD a, T b) { return a * CD(b); }
FB_AR(T) inline CD operator*(T a, CD b) { return CD(a) * b; }
FB_AR(T) inline CD operator/(CD a, T b) { return a / CD(b); }
FB_AR(T) inline CD operator/(T a, CD b) { return CD(a) / b; }
#define FB_CMP(op) inline bool operator op(CD a, CD b) { fbo_cnt[5]++; return a.v op b.v; } \
  FB_AR(T) inline bool operator op(CD a, T b) { fbo_cnt[5]++; return a.v op (double)b; } \
  FB_AR(T) inline bool operator op(T a, CD b) { fbo_cnt[5]++; return (double)a op b.v; }
FB_CMP(<) FB_CMP(>) FB_CMP(<=) FB_CMP(>=) FB_CMP(==) FB_CMP(!=)
inline CD sqrt(CD a) { fbo_cnt[3]++; return CD(::sqrt(a.v)); }
inline CD fabs(CD a) { return CD(::fabs(a.v)); }
inline CD fmax(CD a, CD b) { fbo_cnt[5]++; return CD(::fmax(a.v, b.v)); }
inline CD fmin(CD a, CD b) { fbo_cnt[5]++; return CD(::fmin(a.v, b.v)); }
FB_AR(T) inline CD fmax(CD a, T b) { return fmax(a, CD(b)); }
FB_AR(T) inline CD fmax(T a, CD b) { return fmax(CD(a), b); }
FB_AR(T) inline CD fmin(CD a, T b) { return fmin(a, CD(b)); }
FB_AR(T) inline CD fmin(T a, CD b) { return fmin(CD(a), b); }
#define FB_TR1(f) inline CD f(CD a) { fbo_cnt[4]++; return CD(::f(a.v)); }
FB_TR1(sin) FB_TR1(cos) FB_TR1(tan) FB_TR1(asin) FB_TR1(acos) FB_TR1(atan) FB_TR1(exp) FB_TR1(log) FB_TR1(floor) FB_TR1(ceil) FB_TR1(tanh)
inline CD atan2(CD a, CD b) { fbo_cnt[4]++; return CD(::atan2(a.v, b.v)); }
inline CD pow(CD a, CD b) { fbo_cnt[4]++; return CD(::pow(a.v, b.v)); }
FB_AR(T) inline CD pow(CD a, T b) { return pow(a, CD(b)); }
inline CD fmod(CD a, CD b) { fbo_cnt[4]++; return CD(::fmod(a.v, b.v)); }
FB_AR(T) inline CD fmod(CD a, T b) { return fmod(a, CD(b)); }
inline bool isfinite(CD a) { return std::isfinite(a.v); }
inline bool isnan(CD a) { return std::isnan(a.v); }
#define double CD
#define restrict __restrict
