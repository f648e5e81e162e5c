/* CPU ORACLE (test infrastructure only) -- small vector/quaternion/spatial helpers.
 *
 * Conventions follow MuJoCo (the third-party engine the reference steps through,
 * SURVEY.md section 3.3): quaternions are (w,x,y,z); 3x3 matrices are row-major;
 * 6-D spatial vectors are [rotational(3); translational(3)].
 */
#ifndef FBO_MATH_H
#define FBO_MATH_H
#include <math.h>
#include <string.h>

#define FBO_MINVAL 1e-15
#define FBO_PI 3.14159265358979323846

static inline double dot3(const double* a, const double* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static inline void cross3(double* r, const double* a, const double* b) {
  double x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void copy3(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void zero3(double* r) { r[0] = r[1] = r[2] = 0; }
static inline void add3(double* r, const double* a, const double* b) { r[0] = a[0]+b[0]; r[1] = a[1]+b[1]; r[2] = a[2]+b[2]; }
static inline void sub3(double* r, const double* a, const double* b) { r[0] = a[0]-b[0]; r[1] = a[1]-b[1]; r[2] = a[2]-b[2]; }
static inline void scl3(double* r, const double* a, double s) { r[0] = a[0]*s; r[1] = a[1]*s; r[2] = a[2]*s; }
static inline void addscl3(double* r, const double* a, double s) { r[0] += a[0]*s; r[1] += a[1]*s; r[2] += a[2]*s; }
static inline double norm3(const double* a) { return sqrt(dot3(a, a)); }
static inline double normalize3(double* a) {
  double n = norm3(a);
  if (n < FBO_MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
  a[0] /= n; a[1] /= n; a[2] /= n; return n;
}
static inline double dot6(const double* a, const double* b) {
  return a[0]*b[0] + a[1]*b[1] + a[2]*b[2] + a[3]*b[3] + a[4]*b[4] + a[5]*b[5];
}
/* r = M v */
static inline void mulmat3(double* r, const double* m, const double* v) {
  double x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  double y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  double z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
/* r = M^T v */
static inline void mulmatT3(double* r, const double* m, const double* v) {
  double x = m[0]*v[0] + m[3]*v[1] + m[6]*v[2];
  double y = m[1]*v[0] + m[4]*v[1] + m[7]*v[2];
  double z = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulquat(double* r, const double* a, const double* b) {
  double w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  double x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  double y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  double z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static inline void normquat(double* q) {
  double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < FBO_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static inline void quat2mat(double* m, const double* q) {
  double q00 = q[0]*q[0], q01 = q[0]*q[1], q02 = q[0]*q[2], q03 = q[0]*q[3];
  double q11 = q[1]*q[1], q12 = q[1]*q[2], q13 = q[1]*q[3];
  double q22 = q[2]*q[2], q23 = q[2]*q[3], q33 = q[3]*q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2*(q12 - q03); m[2] = 2*(q13 + q02);
  m[3] = 2*(q12 + q03); m[5] = 2*(q23 - q01);
  m[6] = 2*(q13 - q02); m[7] = 2*(q23 + q01);
}
/* rotate vector by quaternion */
static inline void rotvecquat(double* r, const double* v, const double* q) {
  double m[9]; quat2mat(m, q); mulmat3(r, m, v);
}
static inline void axisangle2quat(double* q, const double* axis, double ang) {
  double s = sin(0.5*ang);
  q[0] = cos(0.5*ang); q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
/* integrate quaternion with body-frame angular velocity over dt */
static inline void quatintegrate(double* q, const double* w, double dt) {
  double ax[3] = {w[0], w[1], w[2]};
  double n = normalize3(ax);
  double ang = n*dt;
  double qr[4], res[4];
  axisangle2quat(qr, ax, ang);
  normquat(q);
  mulquat(res, q, qr);
  q[0] = res[0]; q[1] = res[1]; q[2] = res[2]; q[3] = res[3];
  normquat(q);
}
/* 10-number spatial inertia times 6-D motion vector */
static inline void mulinertvec(double* r, const double* i, const double* v) {
  r[0] = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  r[1] = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  r[2] = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  r[3] = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  r[4] = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  r[5] = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
}
/* spatial cross products */
static inline void crossmotion(double* r, const double* vel, const double* v) {
  r[0] = -vel[2]*v[1] + vel[1]*v[2];
  r[1] =  vel[2]*v[0] - vel[0]*v[2];
  r[2] = -vel[1]*v[0] + vel[0]*v[1];
  r[3] = -vel[2]*v[4] + vel[1]*v[5] - vel[5]*v[1] + vel[4]*v[2];
  r[4] =  vel[2]*v[3] - vel[0]*v[5] + vel[5]*v[0] - vel[3]*v[2];
  r[5] = -vel[1]*v[3] + vel[0]*v[4] - vel[4]*v[0] + vel[3]*v[1];
}
static inline void crossforce(double* r, const double* vel, const double* f) {
  r[0] = -vel[2]*f[1] + vel[1]*f[2] - vel[5]*f[4] + vel[4]*f[5];
  r[1] =  vel[2]*f[0] - vel[0]*f[2] + vel[5]*f[3] - vel[3]*f[5];
  r[2] = -vel[1]*f[0] + vel[0]*f[1] - vel[4]*f[3] + vel[3]*f[4];
  r[3] = -vel[2]*f[4] + vel[1]*f[5];
  r[4] =  vel[2]*f[3] - vel[0]*f[5];
  r[5] = -vel[1]*f[3] + vel[0]*f[4];
}
/* complete an orthonormal frame given its (unit) first row; rows of `f` are x,y,z */
static inline void makeframe(double* f) {
  double* x = f; double* y = f + 3; double* z = f + 6;
  normalize3(x);
  if (x[1] > 0.5 || x[1] < -0.5) { y[0] = 0; y[1] = 0; y[2] = 1; }
  else { y[0] = 0; y[1] = 1; y[2] = 0; }
  double d = dot3(x, y);
  addscl3(y, x, -d);
  normalize3(y);
  cross3(z, x, y);
}
#endif
