/* CPU ORACLE (test infrastructure only): constraint construction and solvers.
 *
 * Restates MuJoCo mj_makeConstraint / mj_projectConstraint / mj_solPGS / mj_solNoSlip for the
 * constraint types the fruit-fly model produces: joint limits,
 * frictionless contacts (condim 1) and elliptic-cone frictional contacts (condim 3).
 * Reference option block: fruitfly.xml:4 (cone="elliptic", noslip_iterations="3").
 */
#include "fbo.h"
#include "fbo_math.h"
#include <stdlib.h>

#define MINIMP 0.0001
#define MAXIMP 0.9999

/* ------------------------------------------------------------------ rows */
static int add_row(fbo_data* d, int type, int id, double pos, double margin, double diagApprox) {
  if (d->nefc >= FBO_MAXEFC) return -1;
  int r = d->nefc++;
  memset(d->efc_J + (size_t)r*d->m->nv, 0, sizeof(double)*d->m->nv);
  d->efc_type[r] = type; d->efc_id[r] = id;
  d->efc_pos[r] = pos; d->efc_margin[r] = margin; d->efc_diagApprox[r] = diagApprox;
  return r;
}

/* row = dir^T (Jp(body2) - Jp(body1)) at `point` */
static void jac_dif_row(const fbo_data* d, double* row, const double* dir, const double* point, int b1, int b2) {
  const fbo_model* m = d->m;
  for (int side = 0; side < 2; side++) {
    int body = side ? b2 : b1;
    double sgn = side ? 1.0 : -1.0;
    if (body <= 0) continue;
    double off[3]; sub3(off, point, d->subtree_com + 3*m->body_rootid[body]);
    int b = body;
    while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parent[b];
    if (b <= 0) continue;
    for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parentid[i]) {
      const double* c = d->cdof + 6*i;
      double t[3]; cross3(t, c, off);
      double jp[3] = {c[3] + t[0], c[4] + t[1], c[5] + t[2]};
      row[i] += sgn*dot3(dir, jp);
    }
  }
}

void fbo_make_constraint(fbo_data* d) {
  const fbo_model* m = d->m;
  d->nefc = 0;
  /* joint limits */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] != FBO_JNT_HINGE) continue;
    double value = d->qpos[m->jnt_qposadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side*(m->jnt_range[2*j + (side+1)/2] - value);
      if (dist < m->jnt_margin[j]) {
        int r = add_row(d, FBO_CNSTR_LIMIT, j, dist, m->jnt_margin[j], m->dof_invweight0[m->jnt_dofadr[j]]);
        if (r >= 0) d->efc_J[(size_t)r*m->nv + m->jnt_dofadr[j]] = -side;
      }
    }
  }
  /* contacts */
  for (int c = 0; c < d->ncon; c++) {
    fbo_contact* con = d->contact + c;
    con->efc_address = -1;
    if (con->dist >= con->includemargin) { con->exclude = 1; continue; }   /* in gap */
    con->exclude = 0;
    int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
    double tran = m->body_invweight0[2*b1] + m->body_invweight0[2*b2];
    int dim = (con->dim == 1) ? 1 : 3;
    if (d->nefc + dim > FBO_MAXEFC) { con->exclude = 1; continue; }
    for (int k = 0; k < dim; k++) {
      int r = add_row(d, dim == 1 ? FBO_CNSTR_CONTACT_FRICTIONLESS : FBO_CNSTR_CONTACT_ELLIPTIC, c,
                      con->dist, con->includemargin, tran);
      if (k == 0) con->efc_address = r;
      jac_dif_row(d, d->efc_J + (size_t)r*m->nv, con->frame + 3*k, con->pos, b1, b2);
    }
  }
}

/* ------------------------------------------------------------------ impedance, reference */
static void get_impedance(const double* solimp_in, double pos, double margin, double* imp) {
  double s[5];
  s[0] = fmin(MAXIMP, fmax(MINIMP, solimp_in[0]));
  s[1] = fmin(MAXIMP, fmax(MINIMP, solimp_in[1]));
  s[2] = fmax(0.0, solimp_in[2]);
  s[3] = fmin(MAXIMP, fmax(MINIMP, solimp_in[3]));
  s[4] = fmax(1.0, solimp_in[4]);
  if (s[0] == s[1] || s[2] <= FBO_MINVAL) { *imp = 0.5*(s[0] + s[1]); return; }
  double x = fabs((pos - margin)/s[2]);
  if (x >= 1) { *imp = s[1]; return; }
  if (x <= 0) { *imp = s[0]; return; }
  double y;
  if (s[4] == 1) y = x;
  else if (s[4] == 2) y = (x <= s[3]) ? x*x/s[3] : 1 - (1 - x)*(1 - x)/(1 - s[3]);      /* default power: same value as pow(x, 2), spelled like the kernel */
  else if (x <= s[3]) y = pow(x, s[4]) / pow(s[3], s[4] - 1);
  else y = 1 - pow(1 - x, s[4]) / pow(1 - s[3], s[4] - 1);
  *imp = s[0] + y*(s[1] - s[0]);
}

static void make_impedance(fbo_data* d) {
  const fbo_model* m = d->m;
  for (int i = 0; i < d->nefc; i++) {
    const double *solref, *solimp;
    int friction_row = 0;
    if (d->efc_type[i] == FBO_CNSTR_LIMIT) { solref = m->jnt_solref + 2*d->efc_id[i]; solimp = m->jnt_solimp + 5*d->efc_id[i]; }
    else {
      const fbo_contact* con = d->contact + d->efc_id[i];
      solref = con->solref; solimp = con->solimp;
      friction_row = (d->efc_type[i] == FBO_CNSTR_CONTACT_ELLIPTIC && i != con->efc_address);
    }
    double imp;
    get_impedance(solimp, d->efc_pos[i], d->efc_margin[i], &imp);
    double dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1]));
    double K, B;
    if (solref[0] > 0) {
      double tc = fmax(solref[0], 2*m->timestep);          /* refsafe */
      K = 1.0 / fmax(FBO_MINVAL, dmax*dmax*tc*tc*solref[1]*solref[1]);
      B = 2.0 / fmax(FBO_MINVAL, dmax*tc);
    } else {
      K = -solref[0] / fmax(FBO_MINVAL, dmax*dmax);
      B = -solref[1] / fmax(FBO_MINVAL, dmax);
    }
    if (friction_row) K = 0;
    d->efc_KBIP[4*i] = K; d->efc_KBIP[4*i+1] = B; d->efc_KBIP[4*i+2] = imp; d->efc_KBIP[4*i+3] = 0;
    d->efc_R[i] = fmax(FBO_MINVAL, (1 - imp)*d->efc_diagApprox[i]/imp);
  }
  /* friction rows of elliptic contacts */
  for (int c = 0; c < d->ncon; c++) {
    fbo_contact* con = d->contact + c;
    if (con->efc_address < 0 || con->dim == 1) continue;
    int i = con->efc_address;
    d->efc_R[i+1] = d->efc_R[i] / fmax(FBO_MINVAL, m->impratio);
    con->mu = con->friction[0] * sqrt(d->efc_R[i+1] / d->efc_R[i]);
    d->efc_R[i+2] = d->efc_R[i+1]*con->friction[0]*con->friction[0]/(con->friction[1]*con->friction[1]);
  }
  for (int i = 0; i < d->nefc; i++) d->efc_D[i] = 1.0 / d->efc_R[i];
}

/* adhesion actuators pull along the mean normal of the contacts of their body
 * (transmission type "body"; fruitfly.xml:889-896) */
void fbo_transmission(fbo_data* d) {
  const fbo_model* m = d->m;
  int nv = m->nv;
  for (int i = 0; i < m->nu; i++) {
    double* mom = d->actuator_moment + (size_t)i*nv;
    memset(mom, 0, sizeof(double)*nv);
    int id = m->actuator_trnid[i];
    switch (m->actuator_trntype[i]) {
      case FBO_TRN_JOINT:
        d->actuator_length[i] = d->qpos[m->jnt_qposadr[id]];
        mom[m->jnt_dofadr[id]] = 1.0;
        break;
      case FBO_TRN_TENDON:
        d->actuator_length[i] = d->ten_length[id];
        for (int w = m->tendon_adr[id]; w < m->tendon_adr[id] + m->tendon_num[id]; w++) mom[m->wrap_dofid[w]] = m->wrap_coef[w];
        break;
      case FBO_TRN_BODY: {
        d->actuator_length[i] = 0;
        int counter = 0;
        for (int c = 0; c < d->ncon; c++) {
          const fbo_contact* con = d->contact + c;
          int b1 = m->geom_bodyid[con->geom1], b2 = m->geom_bodyid[con->geom2];
          if (b1 != id && b2 != id) continue;
          double* row = d->scratch;
          memset(row, 0, sizeof(double)*nv);
          jac_dif_row(d, row, con->frame, con->pos, b1, b2);
          for (int k = 0; k < nv; k++) mom[k] -= row[k];   /* attractive */
          counter++;
        }
        if (counter) for (int k = 0; k < nv; k++) mom[k] /= counter;
        break; }
    }
  }
}

/* AR = J M^-1 J^T + diag(R) through the half-solve  JM2 = J L^-1 D^-1/2 */
void fbo_project_constraint(fbo_data* d) {
  const fbo_model* m = d->m;
  int nv = m->nv, n = d->nefc;
  make_impedance(d);
  if (!n) return;
  double* Y = d->scratch;   /* n x nv */
  for (int r = 0; r < n; r++) {
    double* x = Y + (size_t)r*nv;
    memcpy(x, d->efc_J + (size_t)r*nv, sizeof(double)*nv);
    for (int i = nv - 1; i >= 0; i--) {
      if (x[i] == 0) continue;
      int adr = m->dof_Madr[i] + 1;
      for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[j] -= d->qLD[adr++]*x[i];
    }
    for (int i = 0; i < nv; i++) x[i] *= sqrt(d->qLDiagInv[i]);
  }
  for (int r = 0; r < n; r++)
    for (int c = 0; c <= r; c++) {
      const double *a = Y + (size_t)r*nv, *b = Y + (size_t)c*nv;
      double s = 0;
      for (int k = 0; k < nv; k++) s += a[k]*b[k];
      d->efc_AR[(size_t)r*n + c] = s; d->efc_AR[(size_t)c*n + r] = s;
    }
  for (int r = 0; r < n; r++) d->efc_AR[(size_t)r*n + r] += d->efc_R[r];
}

/* ------------------------------------------------------------------ small QCQP (2 friction dims) */
/* min 0.5 x'Ax + x'b  s.t.  sum (x_i/d_i)^2 <= r^2 ; returns 1 if the constraint is active */
static int qcqp2(double* res, const double* Ain, const double* bin, const double* dd, double r) {
  double A11 = Ain[0]*dd[0]*dd[0], A22 = Ain[3]*dd[1]*dd[1], A12 = Ain[1]*dd[0]*dd[1];
  double b1 = bin[0]*dd[0], b2 = bin[1]*dd[1];
  double la = 0, v1 = 0, v2 = 0;
  for (int it = 0; it < 20; it++) {
    double det = (A11 + la)*(A22 + la) - A12*A12;
    if (det < 1e-10) { res[0] = res[1] = 0; return 0; }
    double detinv = 1.0/det;
    double P11 = (A22 + la)*detinv, P22 = (A11 + la)*detinv, P12 = -A12*detinv;
    v1 = -P11*b1 - P12*b2; v2 = -P12*b1 - P22*b2;
    double val = v1*v1 + v2*v2 - r*r;
    if (val < 1e-10) break;
    double deriv = -2.0*(P11*v1*v1 + 2.0*P12*v1*v2 + P22*v2*v2);
    double delta = -val/deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1*dd[0]; res[1] = v2*dd[1];
  return la != 0;
}

/* ------------------------------------------------------------------ PGS */
static double dual_cost(const fbo_data* d, const double* f) {
  int n = d->nefc;
  double cost = 0;
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int k = 0; k < n; k++) s += d->efc_AR[(size_t)i*n + k]*f[k];
    cost += f[i]*(0.5*s + d->efc_b[i]);
  }
  return cost;
}

/* force implied by a candidate acceleration (primal mapping), used for warm start */
static void constraint_update(fbo_data* d, const double* jar) {
  for (int i = 0; i < d->nefc;) {
    if (d->efc_type[i] != FBO_CNSTR_CONTACT_ELLIPTIC) {
      d->efc_force[i] = jar[i] < 0 ? -d->efc_D[i]*jar[i] : 0;
      i++;
      continue;
    }
    const fbo_contact* con = d->contact + d->efc_id[i];
    double mu = con->mu;
    double U0 = jar[i]*mu, U1 = jar[i+1]*con->friction[0], U2 = jar[i+2]*con->friction[1];
    double N = U0, T = sqrt(U1*U1 + U2*U2);
    if (N >= mu*T || (T <= 0 && N >= 0)) { d->efc_force[i] = d->efc_force[i+1] = d->efc_force[i+2] = 0; }
    else if (mu*N + T <= 0 || (T <= 0 && N < 0)) { for (int k = 0; k < 3; k++) d->efc_force[i+k] = -d->efc_D[i+k]*jar[i+k]; }
    else {
      double Dm = d->efc_D[i] / fmax(FBO_MINVAL, mu*mu*(1 + mu*mu));
      double NT = N - mu*T;
      double f0 = -Dm*NT*mu;
      d->efc_force[i] = f0;
      d->efc_force[i+1] = -f0/T*U1*con->friction[0];
      d->efc_force[i+2] = -f0/T*U2*con->friction[1];
    }
    i += 3;
  }
}

static void solve_pgs(fbo_data* d) {
  const fbo_model* m = d->m;
  int n = d->nefc;
  double* f = d->efc_force;
  double scale = 1.0 / (m->meaninertia * (m->nv > 1 ? m->nv : 1));
  d->solver_niter = 0;
  for (int it = 0; it < m->iterations; it++) {
    double improvement = 0;
    for (int i = 0; i < n;) {
      int dim = (d->efc_type[i] == FBO_CNSTR_CONTACT_ELLIPTIC) ? 3 : 1;
      double res[3], old[3];
      for (int j = 0; j < dim; j++) {
        double s = d->efc_b[i+j];
        const double* row = d->efc_AR + (size_t)(i+j)*n;
        for (int k = 0; k < n; k++) s += row[k]*f[k];
        res[j] = s; old[j] = f[i+j];
      }
      if (dim == 1) {
        double a = d->efc_AR[(size_t)i*n + i];
        f[i] -= res[0]/a;
        if (f[i] < 0) f[i] = 0;
        double del = f[i] - old[0];
        improvement -= 0.5*del*del*a + del*res[0];
      } else {
        const fbo_contact* con = d->contact + d->efc_id[i];
        double A[9], bc[3];
        for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) A[3*j+k] = d->efc_AR[(size_t)(i+j)*n + i + k];
        for (int j = 0; j < 3; j++) bc[j] = res[j] - (A[3*j]*old[0] + A[3*j+1]*old[1] + A[3*j+2]*old[2]);
        /* (1) update along the ray through the current force (normal only when inactive) */
        double v[3];
        if (f[i] < FBO_MINVAL) { v[0] = 1; v[1] = v[2] = 0; } else { v[0] = old[0]; v[1] = old[1]; v[2] = old[2]; }
        double Av[3] = {A[0]*v[0] + A[1]*v[1] + A[2]*v[2], A[3]*v[0] + A[4]*v[1] + A[5]*v[2], A[6]*v[0] + A[7]*v[1] + A[8]*v[2]};
        double denom = v[0]*Av[0] + v[1]*Av[1] + v[2]*Av[2];
        if (denom >= FBO_MINVAL) {
          double x = -(v[0]*res[0] + v[1]*res[1] + v[2]*res[2])/denom;
          if (f[i] + x*v[0] < 0) x = -f[i]/v[0];
          for (int k = 0; k < 3; k++) f[i+k] += x*v[k];
        }
        /* (2) friction update with the normal force fixed */
        if (f[i] < FBO_MINVAL) { f[i] = 0; f[i+1] = f[i+2] = 0; }
        else {
          double Ac[4] = {A[4], A[5], A[7], A[8]};
          double bf[2] = {bc[1] + A[3]*f[i], bc[2] + A[6]*f[i]};
          double fr[2];
          int active = qcqp2(fr, Ac, bf, con->friction, f[i]);
          if (active) {
            double s = sqrt((fr[0]/con->friction[0])*(fr[0]/con->friction[0]) + (fr[1]/con->friction[1])*(fr[1]/con->friction[1]));
            if (s > FBO_MINVAL) { fr[0] *= f[i]/s; fr[1] *= f[i]/s; }
          }
          f[i+1] = fr[0]; f[i+2] = fr[1];
        }
        double del[3] = {f[i] - old[0], f[i+1] - old[1], f[i+2] - old[2]};
        double q = 0, l = 0;
        for (int j = 0; j < 3; j++) { l += del[j]*res[j]; for (int k = 0; k < 3; k++) q += del[j]*A[3*j+k]*del[k]; }
        improvement -= 0.5*q + l;
      }
      i += dim;
    }
    d->solver_niter = it + 1;
    if (improvement*scale < m->tolerance) break;
  }
}

static void solve_noslip(fbo_data* d) {
  const fbo_model* m = d->m;
  int n = d->nefc;
  double* f = d->efc_force;
  double scale = 1.0 / (m->meaninertia * (m->nv > 1 ? m->nv : 1));
  d->noslip_niter = 0;
  for (int it = 0; it < m->noslip_iterations; it++) {
    double improvement = 0;
    for (int c = 0; c < d->ncon; c++) {
      const fbo_contact* con = d->contact + c;
      if (con->efc_address < 0 || con->dim == 1) continue;
      int i = con->efc_address;
      double res[2], old[2] = {f[i+1], f[i+2]};
      for (int j = 0; j < 2; j++) {
        double s = d->efc_b[i+1+j];
        const double* row = d->efc_AR + (size_t)(i+1+j)*n;
        for (int k = 0; k < n; k++) s += row[k]*f[k];
        res[j] = s - d->efc_R[i+1+j]*f[i+1+j];          /* remove regularisation */
      }
      double Ac[4] = {d->efc_AR[(size_t)(i+1)*n + i+1] - d->efc_R[i+1], d->efc_AR[(size_t)(i+1)*n + i+2],
                      d->efc_AR[(size_t)(i+2)*n + i+1], d->efc_AR[(size_t)(i+2)*n + i+2] - d->efc_R[i+2]};
      double bc[2] = {res[0] - (Ac[0]*old[0] + Ac[1]*old[1]), res[1] - (Ac[2]*old[0] + Ac[3]*old[1])};
      if (f[i] < FBO_MINVAL) { f[i+1] = f[i+2] = 0; }
      else {
        double fr[2];
        int active = qcqp2(fr, Ac, bc, con->friction, f[i]);
        if (active) {
          double s = sqrt((fr[0]/con->friction[0])*(fr[0]/con->friction[0]) + (fr[1]/con->friction[1])*(fr[1]/con->friction[1]));
          if (s > FBO_MINVAL) { fr[0] *= f[i]/s; fr[1] *= f[i]/s; }
        }
        f[i+1] = fr[0]; f[i+2] = fr[1];
      }
      double del[2] = {f[i+1] - old[0], f[i+2] - old[1]};
      improvement -= 0.5*(del[0]*(Ac[0]*del[0] + Ac[1]*del[1]) + del[1]*(Ac[2]*del[0] + Ac[3]*del[1])) + del[0]*res[0] + del[1]*res[1];
    }
    d->noslip_niter = it + 1;
    if (improvement*scale < m->noslip_tolerance) break;
  }
}


/* ------------------------------------------------------------------ Newton (the reference model's solver: fruitfly.xml:4 sets no
 * `solver`, i.e. MuJoCo's default, Newton)
 *
 * MuJoCo's Newton solver minimises the PRIMAL cost  P(a) = 1/2 (a - a_s)' M (a - a_s) + sum_i s_i(J a - aref)  over the
 * acceleration a; s is the convex constraint cost whose negative gradient is the constraint force (constraint_update above: zero /
 * quadratic rows, and the three zones of an elliptic cone).  PGS solves the dual of the same problem, so both have one solution.
 * Restated here in constraint space: every Newton iterate has the form a = a_s + M^-1 J' lam, so the state is lam (nefc numbers,
 * equal to the force at the solution), jar = b + A lam with A = J M^-1 J' (the Delassus matrix WITHOUT the regulariser R), and
 *     P(lam) = 1/2 lam' A lam + s(jar),   gradient (wrt a) = J'(lam - f(jar)),   Hessian = M + J' H J,  H = d2s/djar2 (block diagonal).
 * The Newton direction in lam-coordinates solves (I + H A) dlam = f - lam.  With H = F F' (F: one column per active scalar row / three
 * per contact in the bottom zone / the two rank-one factors of the cone Hessian in the middle zone) the Woodbury identity gives
 *     dlam = r - F (I + F' A F)^-1 F' A r,   r = f - lam,
 * i.e. one Cholesky factorisation of the SPD matrix K = I + F'AF (pivots >= 1) per iteration; then an exact (to ls_tolerance) line
 * search on the 1-D convex function P(lam + alpha dlam) by safeguarded Newton steps on its derivative.  Terminates when the bound
 * 1/2 r'Ar on the attainable improvement, scaled like MuJoCo's `improvement` (1/(meaninertia nv)), drops below opt.tolerance. */
#define NEWTON_LS_TOL 0.01
#define NEWTON_LS_MAX 20
/* Newton at every system size (MuJoCo).  The HIP kernel keeps one row per lane of a wavefront and falls back to PGS beyond 64 rows
 * (flagged there: FB_WARN_SOLVER_FALLBACK); a model blob with opt_newton_maxrows = 64 makes this oracle do the same, for the tests that
 * check the kernel's fallback path itself. */
#define NEWTON_OK(m, n) ((m)->solver == FBO_SOLVER_NEWTON && ((m)->newton_maxrows <= 0 || (n) <= (m)->newton_maxrows))

typedef struct { double f, cost; int base; double frow[3], fcol[3]; } nrow;

/* force, cost and Hessian factor of every row at `jar`; rows of one contact get identical block data */
static double newton_update(const fbo_data* d, const double* jar, nrow* q) {
  int n = d->nefc; double cost = 0;
  for (int i = 0; i < n;) {
    if (d->efc_type[i] != FBO_CNSTR_CONTACT_ELLIPTIC) {
      nrow* r = q + i; r->base = i; r->f = 0; r->frow[0] = r->frow[1] = r->frow[2] = 0; r->fcol[0] = r->fcol[1] = r->fcol[2] = 0;
      if (jar[i] < 0) { r->f = -d->efc_D[i]*jar[i]; cost += 0.5*d->efc_D[i]*jar[i]*jar[i]; r->frow[0] = r->fcol[0] = sqrt(d->efc_D[i]); }
      i++; continue;
    }
    const fbo_contact* con = d->contact + d->efc_id[i];
    double mu = con->mu, s[3] = {mu, con->friction[0], con->friction[1]};
    double U0 = jar[i]*s[0], U1 = jar[i+1]*s[1], U2 = jar[i+2]*s[2];
    double N = U0, T = sqrt(U1*U1 + U2*U2);
    for (int k = 0; k < 3; k++) { nrow* r = q + i + k; r->base = i; r->f = 0; for (int c = 0; c < 3; c++) r->frow[c] = r->fcol[c] = 0; }
    if (N >= mu*T || (T <= 0 && N >= 0)) { /* top zone: no force */ }
    else if (mu*N + T <= 0 || (T <= 0 && N < 0)) {
      for (int k = 0; k < 3; k++) { double D = d->efc_D[i+k]; q[i+k].f = -D*jar[i+k]; cost += 0.5*D*jar[i+k]*jar[i+k]; q[i+k].frow[k] = q[i+k].fcol[k] = sqrt(D); }
    } else {
      double Dm = d->efc_D[i] / fmax(FBO_MINVAL, mu*mu*(1 + mu*mu));
      double NT = N - mu*T, t1 = U1/T, t2 = U2/T;
      cost += 0.5*Dm*NT*NT;
      double f0 = -Dm*NT*mu;
      q[i].f = f0; q[i+1].f = -f0*t1*s[1]; q[i+2].f = -f0*t2*s[2];
      /* cone Hessian in U-space: Dm (e_n - mu t)(e_n - mu t)' + Dm mu (mu - N/T) t_perp t_perp' ; jar-space: scale rows by s */
      double g1 = sqrt(Dm), g2 = sqrt(Dm*mu*(mu - N/T));
      double c1[3] = {g1*s[0], -g1*mu*t1*s[1], -g1*mu*t2*s[2]};       /* column 0 of the block factor */
      double c2[3] = {0, -g2*t2*s[1], g2*t1*s[2]};                      /* column 1; column 2 is zero */
      for (int k = 0; k < 3; k++) { q[i+k].frow[0] = c1[k]; q[i+k].frow[1] = c2[k]; q[i+k].frow[2] = 0; }
      for (int k = 0; k < 3; k++) { q[i].fcol[k] = c1[k]; q[i+1].fcol[k] = c2[k]; q[i+2].fcol[k] = 0; }
    }
    i += 3;
  }
  return cost;
}

static void solve_newton(fbo_data* d) {
  const fbo_model* m = d->m;
  const int n = d->nefc;
  const double* AR = d->efc_AR; const double* b = d->efc_b;
  double scale = 1.0 / (m->meaninertia * (m->nv > 1 ? m->nv : 1));
  double* W = (double*)malloc(sizeof(double)*((size_t)n*n + 12*(size_t)n));
  double *K = W, *lam = W + (size_t)n*n, *jar = lam + n, *r = jar + n, *qv = r + n, *p = qv + n, *z = p + n, *dl = z + n, *Adl = dl + n, *tj = Adl + n;
  nrow* rw = (nrow*)malloc(sizeof(nrow)*(size_t)n);
#define AMUL(out, x) for (int i_ = 0; i_ < n; i_++) { double s_ = 0; for (int k_ = 0; k_ < n; k_++) s_ += AR[(size_t)i_*n + k_]*(x)[k_]; (out)[i_] = s_ - d->efc_R[i_]*(x)[i_]; }
  /* warm start: the force implied by the previous acceleration, unless the zero force is cheaper */
  memcpy(lam, d->efc_force, sizeof(double)*n);
  AMUL(jar, lam);
  double lAl = 0; for (int i = 0; i < n; i++) { lAl += lam[i]*jar[i]; jar[i] += b[i]; }
  double c_ws = 0.5*lAl + newton_update(d, jar, rw), c_0 = newton_update(d, b, rw);
  if (c_ws > c_0) { memset(lam, 0, sizeof(double)*n); memcpy(jar, b, sizeof(double)*n); }
  d->solver_niter = 0;
  for (int it = 0; it < m->iterations; it++) {
    newton_update(d, jar, rw);
    for (int i = 0; i < n; i++) r[i] = rw[i].f - lam[i];
    AMUL(qv, r);
    double dec = 0; for (int i = 0; i < n; i++) dec += r[i]*qv[i];
    if (0.5*dec*scale < m->tolerance) break;
    /* K = I + F'AF (lower triangle), p = F'q */
    for (int j = 0; j < n; j++) {
      int bj = rw[j].base, nj = (d->efc_type[j] == FBO_CNSTR_CONTACT_ELLIPTIC) ? 3 : 1;
      double pj = 0; for (int a = 0; a < nj; a++) pj += rw[j].fcol[a]*qv[bj + a];
      p[j] = pj;
      for (int k = 0; k <= j; k++) {
        int bk = rw[k].base, nk = (d->efc_type[k] == FBO_CNSTR_CONTACT_ELLIPTIC) ? 3 : 1;
        double s = 0;
        for (int a = 0; a < nj; a++) {
          double wa = 0;
          for (int c = 0; c < nk; c++) { int ra = bj + a, rc = bk + c; wa += (AR[(size_t)ra*n + rc] - (ra == rc ? d->efc_R[ra] : 0.0))*rw[k].fcol[c]; }
          s += rw[j].fcol[a]*wa;
        }
        K[(size_t)j*n + k] = s + (j == k ? 1.0 : 0.0);
      }
    }
    /* Cholesky K = L L' (right-looking), forward substitution folded in; then back substitution */
    for (int j = 0; j < n; j++) {
      double inv = 1.0/sqrt(K[(size_t)j*n + j]);
      K[(size_t)j*n + j] = inv;                      /* the diagonal slot keeps 1/L_jj */
      double yj = p[j]*inv; p[j] = yj;
      for (int i = j + 1; i < n; i++) { double l = K[(size_t)i*n + j]*inv; K[(size_t)i*n + j] = l; p[i] -= l*yj; }
      for (int i = j + 1; i < n; i++) { double l = K[(size_t)i*n + j]; for (int k = j + 1; k <= i; k++) K[(size_t)i*n + k] -= l*K[(size_t)k*n + j]; }
    }
    for (int j = n - 1; j >= 0; j--) {
      double zj = p[j]*K[(size_t)j*n + j]; z[j] = zj;
      for (int i = 0; i < j; i++) p[i] -= K[(size_t)j*n + i]*zj;
    }
    for (int i = 0; i < n; i++) {
      int bi = rw[i].base, ni = (d->efc_type[i] == FBO_CNSTR_CONTACT_ELLIPTIC) ? 3 : 1;
      double s = 0; for (int c = 0; c < ni; c++) s += rw[i].frow[c]*z[bi + c];
      dl[i] = r[i] - s;
    }
    AMUL(Adl, dl);
    double lAd = 0, dAd = 0;
    for (int i = 0; i < n; i++) { lAd += (jar[i] - b[i])*dl[i]; dAd += dl[i]*Adl[i]; }
    /* line search on phi(alpha) = P(lam + alpha dl):  phi' = lAd + alpha dAd - f(jar + alpha Adl).Adl,  phi'' = dAd + |F' Adl|^2 */
    double alpha = 0, g0 = 0, lo = 0, hi = -1;
    for (int k = 0; k <= NEWTON_LS_MAX; k++) {
      for (int i = 0; i < n; i++) tj[i] = jar[i] + alpha*Adl[i];
      newton_update(d, tj, rw);
      double g = lAd + alpha*dAd, h = dAd;
      for (int i = 0; i < n; i++) {
        g -= rw[i].f*Adl[i];
        int bi = rw[i].base, ni = (d->efc_type[i] == FBO_CNSTR_CONTACT_ELLIPTIC) ? 3 : 1;
        double wv = 0; for (int a = 0; a < ni; a++) wv += rw[i].fcol[a]*Adl[bi + a];
        h += wv*wv;
      }
      if (k == 0) { g0 = g; if (!(g0 < 0) || !(h > FBO_MINVAL)) break; alpha = -g0/h; continue; }
      if (fabs(g) <= NEWTON_LS_TOL*fabs(g0) || k == NEWTON_LS_MAX) break;
      if (g < 0) lo = alpha; else hi = alpha;
      double an = (h > FBO_MINVAL) ? alpha - g/h : -1;
      if (!(an > lo) || (hi >= 0 && !(an < hi))) an = (hi < 0) ? 2*alpha : 0.5*(lo + hi);
      alpha = an;
    }
    if (!(alpha > 0)) break;
    for (int i = 0; i < n; i++) { lam[i] += alpha*dl[i]; jar[i] += alpha*Adl[i]; }
    d->solver_niter = it + 1;
    /* MuJoCo's own stopping test: the improvement of the iteration (scaled) below opt.tolerance.  phi is convex with phi'(0) = g0 < 0:
     * the cost fell by at most -g0 alpha.  Ends a solve that stagnates at the rounding floor of the decrement just above the tolerance. */
    if (-g0*alpha*scale < m->tolerance) break;
  }
  newton_update(d, jar, rw);
  for (int i = 0; i < n; i++) d->efc_force[i] = rw[i].f;
#undef AMUL
  free(W); free(rw);
}

void fbo_fwd_constraint(fbo_data* d) {
  const fbo_model* m = d->m;
  int nv = m->nv, n = d->nefc;
  if (!n) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double)*nv);
    memset(d->qfrc_constraint, 0, sizeof(double)*nv);
    memcpy(d->qacc_warmstart, d->qacc, sizeof(double)*nv);
    d->solver_niter = 0;
    return;
  }
  /* reference acceleration and b = J qacc_smooth - aref */
  for (int i = 0; i < n; i++) {
    const double* row = d->efc_J + (size_t)i*nv;
    double vel = 0, ja = 0;
    for (int k = 0; k < nv; k++) { vel += row[k]*d->qvel[k]; ja += row[k]*d->qacc_smooth[k]; }
    d->efc_vel[i] = vel;
    d->efc_aref[i] = -d->efc_KBIP[4*i+1]*vel - d->efc_KBIP[4*i]*d->efc_KBIP[4*i+2]*(d->efc_pos[i] - d->efc_margin[i]);
    d->efc_b[i] = ja - d->efc_aref[i];
  }
  /* warm start from the previous acceleration */
  {
    double* jar = d->scratch;
    for (int i = 0; i < n; i++) {
      const double* row = d->efc_J + (size_t)i*nv;
      double s = 0;
      for (int k = 0; k < nv; k++) s += row[k]*d->qacc_warmstart[k];
      jar[i] = s - d->efc_aref[i];
    }
    constraint_update(d, jar);
    if (!NEWTON_OK(m, n) && dual_cost(d, d->efc_force) > 0) memset(d->efc_force, 0, sizeof(double)*n);
  }
  if (NEWTON_OK(m, n)) solve_newton(d);
  else solve_pgs(d);
  if (m->noslip_iterations > 0) solve_noslip(d);
  /* qfrc_constraint = J^T f ; qacc = qacc_smooth + M^-1 qfrc_constraint */
  memset(d->qfrc_constraint, 0, sizeof(double)*nv);
  for (int i = 0; i < n; i++) {
    double fi = d->efc_force[i];
    if (fi == 0) continue;
    const double* row = d->efc_J + (size_t)i*nv;
    for (int k = 0; k < nv; k++) d->qfrc_constraint[k] += row[k]*fi;
  }
  memcpy(d->qacc, d->qfrc_constraint, sizeof(double)*nv);
  fbo_solve_m(d, d->qacc, d->qLD, d->qLDiagInv);
  for (int k = 0; k < nv; k++) d->qacc[k] += d->qacc_smooth[k];
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double)*nv);   /* saved for the next solve */
}
