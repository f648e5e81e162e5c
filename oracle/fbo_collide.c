/* CPU ORACLE (test infrastructure only): collision detection.
 *
 * Restates the part of MuJoCo `mj_collision` the fruit-fly model exercises
 * (SURVEY.md 3.3 / 8a row H2): bounding-sphere mid-phase over the precomputed pair list,
 * analytic plane-{sphere,capsule,ellipsoid,cylinder}, sphere/capsule pairs, and a
 * Minkowski-Portal-Refinement (MPR) penetration query on support functions for every pair
 * involving an ellipsoid or a cylinder (the algorithm MuJoCo's libccd back-end uses; it is
 * restated here from the published XenoCollide/libccd description, not from MuJoCo sources,
 * which are absent from /root/reference).
 */
#include "fbo.h"
#include "fbo_math.h"

#define MPR_TOL 1e-6
#define MPR_ITER 50
#define MPR_EPS 1e-14

typedef struct { const double *pos, *mat, *size; int type; double margin; } cgeom;

/* farthest point of a geom (inflated by margin/2) along world direction dir (unit) */
static void support(const cgeom* g, const double* dir, double* out) {
  double l[3], p[3];
  mulmatT3(l, g->mat, dir);
  switch (g->type) {
    case FBO_GEOM_SPHERE:
      scl3(p, l, g->size[0]); break;
    case FBO_GEOM_CAPSULE:
      scl3(p, l, g->size[0]);
      p[2] += (l[2] >= 0 ? g->size[1] : -g->size[1]); break;
    case FBO_GEOM_ELLIPSOID: {
      double s[3] = {g->size[0]*l[0], g->size[1]*l[1], g->size[2]*l[2]};
      double n = norm3(s);
      if (n < FBO_MINVAL) { p[0] = g->size[0]; p[1] = p[2] = 0; }
      else { p[0] = g->size[0]*s[0]/n; p[1] = g->size[1]*s[1]/n; p[2] = g->size[2]*s[2]/n; }
      break; }
    case FBO_GEOM_CYLINDER: {
      double n = sqrt(l[0]*l[0] + l[1]*l[1]);
      if (n < FBO_MINVAL) { p[0] = p[1] = 0; }
      else { p[0] = g->size[0]*l[0]/n; p[1] = g->size[0]*l[1]/n; }
      p[2] = (l[2] >= 0 ? g->size[1] : -g->size[1]);
      break; }
    default: zero3(p);
  }
  addscl3(p, l, 0.5*g->margin);
  mulmat3(out, g->mat, p);
  add3(out, out, g->pos);
}

typedef struct { double v[3], v1[3], v2[3]; } mpr_pt;

static void md_support(const cgeom* a, const cgeom* b, const double* dir, mpr_pt* s) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  support(a, dir, s->v1);
  support(b, nd, s->v2);
  sub3(s->v, s->v1, s->v2);
}

static void portal_dir(const mpr_pt* p, double* dir) {
  double a[3], b[3];
  sub3(a, p[2].v, p[1].v); sub3(b, p[3].v, p[1].v);
  cross3(dir, a, b); normalize3(dir);
}

static int reach_tol(const mpr_pt* p, const mpr_pt* v4, const double* dir) {
  double dv4 = dot3(v4->v, dir);
  double d1 = dv4 - dot3(p[1].v, dir), d2 = dv4 - dot3(p[2].v, dir), d3 = dv4 - dot3(p[3].v, dir);
  double dm = fmin(d1, fmin(d2, d3));
  return dm <= MPR_TOL;
}

static void expand_portal(mpr_pt* p, const mpr_pt* v4) {
  double v4v0[3];
  cross3(v4v0, v4->v, p[0].v);
  if (dot3(p[1].v, v4v0) > 0) {
    if (dot3(p[2].v, v4v0) > 0) p[1] = *v4; else p[3] = *v4;
  } else {
    if (dot3(p[3].v, v4v0) > 0) p[2] = *v4; else p[1] = *v4;
  }
}

/* squared distance from origin to triangle (a,b,c); witness returned */
static double origin_tri_dist2(const double* a, const double* b, const double* c, double* wit) {
  double ab[3], ac[3], ap[3] = {-a[0], -a[1], -a[2]};
  sub3(ab, b, a); sub3(ac, c, a);
  double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
  if (d1 <= 0 && d2 <= 0) { copy3(wit, a); return dot3(a, a); }
  double bp[3] = {-b[0], -b[1], -b[2]};
  double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
  if (d3 >= 0 && d4 <= d3) { copy3(wit, b); return dot3(b, b); }
  double vc = d1*d4 - d3*d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1/(d1 - d3); copy3(wit, a); addscl3(wit, ab, v); return dot3(wit, wit); }
  double cp[3] = {-c[0], -c[1], -c[2]};
  double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
  if (d6 >= 0 && d5 <= d6) { copy3(wit, c); return dot3(c, c); }
  double vb = d5*d2 - d1*d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { double w = d2/(d2 - d6); copy3(wit, a); addscl3(wit, ac, w); return dot3(wit, wit); }
  double va = d3*d6 - d5*d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    double w = (d4 - d3)/((d4 - d3) + (d5 - d6));
    double bc[3]; sub3(bc, c, b); copy3(wit, b); addscl3(wit, bc, w); return dot3(wit, wit);
  }
  double den = 1.0/(va + vb + vc);
  double v = vb*den, w = vc*den;
  copy3(wit, a); addscl3(wit, ab, v); addscl3(wit, ac, w);
  return dot3(wit, wit);
}

static void find_pos(const mpr_pt* p, double* pos) {
  double dir[3], t[3], b[4];
  portal_dir(p, dir);
  cross3(t, p[1].v, p[2].v); b[0] = dot3(t, p[3].v);
  cross3(t, p[3].v, p[2].v); b[1] = dot3(t, p[0].v);
  cross3(t, p[0].v, p[1].v); b[2] = dot3(t, p[3].v);
  cross3(t, p[2].v, p[1].v); b[3] = dot3(t, p[0].v);
  double sum = b[0] + b[1] + b[2] + b[3];
  if (sum <= MPR_EPS) {
    b[0] = 0;
    cross3(t, p[2].v, p[3].v); b[1] = dot3(t, dir);
    cross3(t, p[3].v, p[1].v); b[2] = dot3(t, dir);
    cross3(t, p[1].v, p[2].v); b[3] = dot3(t, dir);
    sum = b[1] + b[2] + b[3];
  }
  double inv = 1.0/sum, p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
  for (int i = 0; i < 4; i++) { addscl3(p1, p[i].v1, b[i]); addscl3(p2, p[i].v2, b[i]); }
  for (int k = 0; k < 3; k++) pos[k] = 0.5*inv*(p1[k] + p2[k]);
}

/* returns 1 on penetration (of the margin-inflated shapes); depth>=0, dir from a to b */
static int mpr_penetration(const cgeom* a, const cgeom* b, double* depth, double* dir, double* pos) {
  mpr_pt p[4], v4;
  double d[3], va[3], vb[3];
  /* ---- discover portal */
  copy3(p[0].v1, a->pos); copy3(p[0].v2, b->pos); sub3(p[0].v, a->pos, b->pos);
  if (dot3(p[0].v, p[0].v) < MPR_EPS*MPR_EPS) p[0].v[0] += 1e-9;
  scl3(d, p[0].v, -1); normalize3(d);
  md_support(a, b, d, &p[1]);
  if (dot3(p[1].v, d) < 0) return 0;
  cross3(d, p[0].v, p[1].v);
  if (dot3(d, d) < MPR_EPS*MPR_EPS) {
    /* origin on the v0-v1 ray */
    if (dot3(p[1].v, p[1].v) < MPR_EPS*MPR_EPS) { *depth = 0; zero3(dir); dir[0] = 1; }
    else { *depth = norm3(p[1].v); copy3(dir, p[1].v); normalize3(dir); }
    for (int k = 0; k < 3; k++) pos[k] = 0.5*(p[1].v1[k] + p[1].v2[k]);
    return 1;
  }
  normalize3(d);
  md_support(a, b, d, &p[2]);
  if (dot3(p[2].v, d) < 0) return 0;
  sub3(va, p[1].v, p[0].v); sub3(vb, p[2].v, p[0].v);
  cross3(d, va, vb); normalize3(d);
  if (dot3(d, p[0].v) > 0) { mpr_pt t = p[1]; p[1] = p[2]; p[2] = t; scl3(d, d, -1); }
  for (int it = 0;; it++) {
    if (it > 4*MPR_ITER) return 0;
    md_support(a, b, d, &p[3]);
    if (dot3(p[3].v, d) < 0) return 0;
    int cont = 0;
    cross3(va, p[1].v, p[3].v);
    if (dot3(va, p[0].v) < -MPR_EPS) { p[2] = p[3]; cont = 1; }
    if (!cont) {
      cross3(va, p[3].v, p[2].v);
      if (dot3(va, p[0].v) < -MPR_EPS) { p[1] = p[3]; cont = 1; }
    }
    if (!cont) break;
    sub3(va, p[1].v, p[0].v); sub3(vb, p[2].v, p[0].v);
    cross3(d, va, vb); normalize3(d);
  }
  /* ---- refine portal until the origin is enclosed */
  for (int it = 0;; it++) {
    portal_dir(p, d);
    if (dot3(d, p[1].v) >= 0) break;             /* origin inside the portal: intersection */
    md_support(a, b, d, &v4);
    if (dot3(v4.v, d) < 0 || reach_tol(p, &v4, d) || it > MPR_ITER) return 0;
    expand_portal(p, &v4);
  }
  /* ---- find penetration */
  for (int it = 0;; it++) {
    portal_dir(p, d);
    md_support(a, b, d, &v4);
    if (reach_tol(p, &v4, d) || it > MPR_ITER) {
      double wit[3];
      double d2 = origin_tri_dist2(p[1].v, p[2].v, p[3].v, wit);
      *depth = sqrt(d2);
      if (*depth < MPR_EPS) copy3(dir, d); else { copy3(dir, wit); normalize3(dir); }
      find_pos(p, pos);
      return 1;
    }
    expand_portal(p, &v4);
  }
}

/* ------------------------------------------------------------------ contact helpers */
static int add_contact(fbo_data* d, int pair, double dist, const double* pos, const double* normal) {
  const fbo_model* m = d->m;
  if (d->ncon >= FBO_MAXCON) return 0;
  fbo_contact* c = d->contact + d->ncon;
  c->dist = dist; copy3(c->pos, pos);
  copy3(c->frame, normal); zero3(c->frame + 3); zero3(c->frame + 6);
  makeframe(c->frame);
  c->geom1 = m->pair_geom1[pair]; c->geom2 = m->pair_geom2[pair];
  c->dim = m->pair_condim[pair];
  c->includemargin = m->pair_margin[pair] - m->pair_gap[pair];
  memcpy(c->friction, m->pair_friction + 5*pair, sizeof(double)*5);
  memcpy(c->solref, m->pair_solref + 2*pair, sizeof(double)*2);
  memcpy(c->solimp, m->pair_solimp + 5*pair, sizeof(double)*5);
  c->mu = 0; c->exclude = 0; c->efc_address = -1;
  d->ncon++;
  return 1;
}

static int sphere_sphere(fbo_data* d, int pair, const double* p1, double r1, const double* p2, double r2, double margin) {
  double n[3]; sub3(n, p2, p1);
  double len = norm3(n);
  double dist = len - r1 - r2;
  if (dist > margin) return 0;
  if (len < FBO_MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else scl3(n, n, 1.0/len);
  double pos[3]; copy3(pos, p1); addscl3(pos, n, r1 + 0.5*dist);
  return add_contact(d, pair, dist, pos, n);
}

static int plane_sphere(fbo_data* d, int pair, const double* ppos, const double* n, const double* spos, double r, double margin) {
  double dif[3]; sub3(dif, spos, ppos);
  double dist = dot3(dif, n) - r;
  if (dist > margin) return 0;
  double pos[3]; copy3(pos, spos); addscl3(pos, n, -(r + 0.5*dist));
  return add_contact(d, pair, dist, pos, n);
}

static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

static int capsule_capsule(fbo_data* d, int pair, const double* p1, const double* m1, const double* s1,
                           const double* p2, const double* m2, const double* s2, double margin) {
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
  double dif[3]; sub3(dif, p1, p2);
  double ma = 1.0, mb = -dot3(a1, a2), mc = 1.0;
  double u = -dot3(a1, dif), v = dot3(a2, dif);
  double det = ma*mc - mb*mb;
  double l1 = s1[1], l2 = s2[1];
  if (fabs(det) >= 1e-12) {
    double x1 = (mc*u - mb*v)/det, x2 = (ma*v - mb*u)/det;
    if (x1 > l1) { x1 = l1; x2 = (v - mb*l1)/mc; }
    else if (x1 < -l1) { x1 = -l1; x2 = (v + mb*l1)/mc; }
    if (x2 > l2) { x2 = l2; x1 = clampd((u - mb*l2)/ma, -l1, l1); }
    else if (x2 < -l2) { x2 = -l2; x1 = clampd((u + mb*l2)/ma, -l1, l1); }
    double v1[3], v2[3];
    copy3(v1, p1); addscl3(v1, a1, x1);
    copy3(v2, p2); addscl3(v2, a2, x2);
    return sphere_sphere(d, pair, v1, s1[0], v2, s2[0], margin);
  }
  /* parallel axes: test both ends of capsule 1 against the clamped projection on capsule 2 */
  int n = 0;
  double last[3] = {1e30, 1e30, 1e30};
  for (int sgn = 1; sgn >= -1; sgn -= 2) {
    double x1 = sgn*l1;
    double x2 = clampd((v - mb*x1)/mc, -l2, l2);
    double v1[3], v2[3];
    copy3(v1, p1); addscl3(v1, a1, x1);
    copy3(v2, p2); addscl3(v2, a2, x2);
    double t[3]; sub3(t, v2, last);
    if (n && dot3(t, t) < 1e-20) {
      /* both ends project to the same point of capsule 2: use the closest point on capsule 1 instead */
      x1 = clampd((u - mb*x2)/ma, -l1, l1);
      copy3(v1, p1); addscl3(v1, a1, x1);
      d->ncon -= 1; n = 0;
    }
    copy3(last, v2);
    n += sphere_sphere(d, pair, v1, s1[0], v2, s2[0], margin);
  }
  return n;
}

static int sphere_capsule(fbo_data* d, int pair, const double* p1, double r1, const double* p2, const double* m2, const double* s2, double margin) {
  double ax[3] = {m2[2], m2[5], m2[8]}, dif[3];
  sub3(dif, p1, p2);
  double x = clampd(dot3(ax, dif), -s2[1], s2[1]);
  double v[3]; copy3(v, p2); addscl3(v, ax, x);
  return sphere_sphere(d, pair, p1, r1, v, s2[0], margin);
}

static int plane_capsule(fbo_data* d, int pair, const double* ppos, const double* n, const double* cpos, const double* cmat, const double* size, double margin) {
  double ax[3] = {cmat[2], cmat[5], cmat[8]};
  int cnt = 0;
  for (int sgn = 1; sgn >= -1; sgn -= 2) {
    double e[3]; copy3(e, cpos); addscl3(e, ax, sgn*size[1]);
    cnt += plane_sphere(d, pair, ppos, n, e, size[0], margin);
  }
  return cnt;
}

static int plane_ellipsoid(fbo_data* d, int pair, const double* ppos, const double* n, const double* epos, const double* emat, const double* size, double margin) {
  double nl[3]; mulmatT3(nl, emat, n);
  double s[3] = {size[0]*nl[0], size[1]*nl[1], size[2]*nl[2]};
  double len = norm3(s);
  double loc[3] = {-size[0]*s[0]/len, -size[1]*s[1]/len, -size[2]*s[2]/len};
  double pt[3]; mulmat3(pt, emat, loc); add3(pt, pt, epos);
  double dif[3]; sub3(dif, pt, ppos);
  double dist = dot3(dif, n);
  if (dist > margin) return 0;
  double pos[3]; copy3(pos, pt); addscl3(pos, n, -0.5*dist);
  return add_contact(d, pair, dist, pos, n);
}

static int plane_cylinder(fbo_data* d, int pair, const double* ppos, const double* n, const double* cpos, const double* cmat, const double* size, double margin) {
  double ax[3] = {cmat[2], cmat[5], cmat[8]};
  double dif[3]; sub3(dif, cpos, ppos);
  double dist0 = dot3(dif, n);
  double prjaxis = dot3(n, ax);
  if (prjaxis > 0) { scl3(ax, ax, -1); prjaxis = -prjaxis; }
  double vec[3] = {ax[0]*prjaxis - n[0], ax[1]*prjaxis - n[1], ax[2]*prjaxis - n[2]};
  double len = norm3(vec);
  if (len < 1e-12) { vec[0] = cmat[0]*size[0]; vec[1] = cmat[3]*size[0]; vec[2] = cmat[6]*size[0]; }
  else scl3(vec, vec, size[0]/len);
  double prjvec = dot3(vec, n);
  scl3(ax, ax, size[1]); prjaxis *= size[1];
  int cnt = 0;
  double dist = dist0 + prjaxis + prjvec;
  if (dist > margin) return 0;
  double pos[3];
  for (int k = 0; k < 3; k++) pos[k] = cpos[k] + vec[k] + ax[k] - n[k]*dist*0.5;
  cnt += add_contact(d, pair, dist, pos, n);
  dist = dist0 - prjaxis + prjvec;
  if (dist <= margin) {
    for (int k = 0; k < 3; k++) pos[k] = cpos[k] + vec[k] - ax[k] - n[k]*dist*0.5;
    cnt += add_contact(d, pair, dist, pos, n);
  }
  dist = dist0 + prjaxis - 0.5*prjvec;
  if (dist <= margin) {
    double v1[3]; cross3(v1, vec, ax); normalize3(v1); scl3(v1, v1, size[0]*sqrt(3.0)/2);
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
      for (int k = 0; k < 3; k++) pos[k] = cpos[k] + sgn*v1[k] + ax[k] - 0.5*vec[k] - n[k]*dist*0.5;
      cnt += add_contact(d, pair, dist, pos, n);
    }
  }
  return cnt;
}

static int convex_pair(fbo_data* d, int pair, const cgeom* a, const cgeom* b, double margin) {
  cgeom A = *a, B = *b;
  A.margin = margin; B.margin = margin;
  double depth, dir[3], pos[3];
  if (!mpr_penetration(&A, &B, &depth, dir, pos)) return 0;
  double dist = margin - depth;
  if (dist > margin) return 0;
  return add_contact(d, pair, dist, pos, dir);
}

/* ------------------------------------------------------------------ second broad-phase filter: oriented boxes
 * MuJoCo's broad phase is bounding spheres (mj_collideGeoms; rbound), which for the fly's long thin leg capsules and flat body
 * ellipsoids passes ~85 pairs per step of which 2-3 touch.  A pair whose oriented bounding boxes -- box of geom 1 inflated by the
 * margin -- are disjoint (exact 15-axis separating-axis test) is farther apart than the margin, so no narrow-phase routine can
 * return a contact for it: skipping it changes nothing but the work.  Half-extents: capsule (r, r, l + r), cylinder (r, r, l),
 * ellipsoid = its semi-axes, sphere (r, r, r). */
static void box_extents(int type, const double* size, double* e) {
  if (type == FBO_GEOM_CAPSULE) { e[0] = e[1] = size[0]; e[2] = size[1] + size[0]; }
  else if (type == FBO_GEOM_CYLINDER) { e[0] = e[1] = size[0]; e[2] = size[1]; }
  else if (type == FBO_GEOM_ELLIPSOID) { e[0] = size[0]; e[1] = size[1]; e[2] = size[2]; }
  else { e[0] = e[1] = e[2] = size[0]; }
}

/* 1: the boxes (A inflated by `margin`) may overlap; 0: separated */
static int boxes_may_touch(const double* pa, const double* ma, const double* ea0, const double* pb, const double* mb, const double* eb, double margin) {
  double ea[3] = {ea0[0] + margin, ea0[1] + margin, ea0[2] + margin};
  double R[3][3], AR[3][3], t[3], tw[3];
  sub3(tw, pb, pa);
  for (int i = 0; i < 3; i++) {
    t[i] = ma[i]*tw[0] + ma[3+i]*tw[1] + ma[6+i]*tw[2];                     /* centre of B in A's frame */
    for (int j = 0; j < 3; j++) {
      R[i][j] = ma[i]*mb[j] + ma[3+i]*mb[3+j] + ma[6+i]*mb[6+j];            /* axis j of B in A's frame */
      AR[i][j] = fabs(R[i][j]) + 1e-9;                                       /* (parallel edges: keep the cross-axis tests conservative) */
    }
  }
  for (int i = 0; i < 3; i++)
    if (fabs(t[i]) > ea[i] + (eb[0]*AR[i][0] + eb[1]*AR[i][1] + eb[2]*AR[i][2])) return 0;
  for (int j = 0; j < 3; j++)
    if (fabs(t[0]*R[0][j] + t[1]*R[1][j] + t[2]*R[2][j]) > (ea[0]*AR[0][j] + ea[1]*AR[1][j] + ea[2]*AR[2][j]) + eb[j]) return 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      int i1 = (i+1)%3, i2 = (i+2)%3, j1 = (j+1)%3, j2 = (j+2)%3;
      double ra = ea[i1]*AR[i2][j] + ea[i2]*AR[i1][j], rb = eb[j1]*AR[i][j2] + eb[j2]*AR[i][j1];
      if (fabs(t[i2]*R[i1][j] - t[i1]*R[i2][j]) > ra + rb) return 0;
    }
  return 1;
}

/* ------------------------------------------------------------------ driver */
void fbo_collision(fbo_data* d) {
  const fbo_model* m = d->m;
  d->ncon = 0;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    double margin = m->pair_margin[p];
    const double *p1 = d->geom_xpos + 3*g1, *p2 = d->geom_xpos + 3*g2;
    const double *m1 = d->geom_xmat + 9*g1, *m2 = d->geom_xmat + 9*g2;
    const double *s1 = m->geom_size + 3*g1, *s2 = m->geom_size + 3*g2;
    if (t1 == FBO_GEOM_PLANE) {
      double n[3] = {m1[2], m1[5], m1[8]}, dif[3];
      sub3(dif, p2, p1);
      if (dot3(dif, n) > m->geom_rbound[g2] + margin) continue;
      switch (t2) {
        case FBO_GEOM_SPHERE: plane_sphere(d, p, p1, n, p2, s2[0], margin); break;
        case FBO_GEOM_CAPSULE: plane_capsule(d, p, p1, n, p2, m2, s2, margin); break;
        case FBO_GEOM_ELLIPSOID: plane_ellipsoid(d, p, p1, n, p2, m2, s2, margin); break;
        case FBO_GEOM_CYLINDER: plane_cylinder(d, p, p1, n, p2, m2, s2, margin); break;
      }
      continue;
    }
    double dif[3]; sub3(dif, p2, p1);
    double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
    if (dot3(dif, dif) > bound*bound) continue;
    {
      double e1[3], e2[3];
      box_extents(t1, s1, e1); box_extents(t2, s2, e2);
      if (!boxes_may_touch(p1, m1, e1, p2, m2, e2, margin)) continue;
    }
    if (t1 == FBO_GEOM_SPHERE && t2 == FBO_GEOM_SPHERE) sphere_sphere(d, p, p1, s1[0], p2, s2[0], margin);
    else if (t1 == FBO_GEOM_SPHERE && t2 == FBO_GEOM_CAPSULE) sphere_capsule(d, p, p1, s1[0], p2, m2, s2, margin);
    else if (t1 == FBO_GEOM_CAPSULE && t2 == FBO_GEOM_CAPSULE) capsule_capsule(d, p, p1, m1, s1, p2, m2, s2, margin);
    else {
      cgeom A = {p1, m1, s1, t1, 0}, B = {p2, m2, s2, t2, 0};
      convex_pair(d, p, &A, &B, margin);
    }
  }
}
