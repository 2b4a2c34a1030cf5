/* Oracle C port -- TEST INFRASTRUCTURE / CPU BASELINE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the reference's js.model.step() for SoftContacts + semi-implicit
 * Euler, structured like the reference: dense 6x6 Pluecker transforms and inertias, body-frame
 * recursions, one environment at a time (the reference's vmap lane), OpenMP over environments.
 * It is validated against the NumPy oracle (tests/test_oracle_cport.py) and timed by
 * bench.py as `cpu_baseline` (kind "port": the reference's JAX-CPU path cannot run here).
 * Parity status: unpinned by execution, pinned analytically -- same as the NumPy oracle.
 *
 * Follows: api/model.py:2601-2681 (step), api/actuation_model.py:7-126, api/ode.py:16-131,
 * rbda/contacts/soft.py:195-444, rbda/collidable_points.py:9-65, api/contact.py:557-603,
 * rbda/aba.py:12-292, api/kin_dyn_parameters.py:396-451, rbda/forward_kinematics.py:12-113,
 * api/integrators.py:14-88, math/{adjoint,cross,inertia,quaternion,rotation}.py.
 *
 * Compiled twice: -DREAL=double (oracle_step_f64) and -DREAL=float (oracle_step_f32).
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

#include "../../include/jaxsim_amd.h"

#ifndef REAL
#define REAL double
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#ifndef SUFFIX
#define SUFFIX f64
#endif
#define FN(name) CAT(name##_, SUFFIX)

typedef REAL real;
#define MAXL 64

static real eps_of(void) { return sizeof(real) == 4 ? (real)1.1920928955078125e-07 : (real)2.220446049250313e-16; }

/* ---- 3x3 / 6x6 helpers (row-major) -------------------------------------------------------- */
static void m3mul(const real* a, const real* b, real* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
static void m3vec(const real* a, const real* x, real* o) {
  for (int i = 0; i < 3; ++i) o[i] = a[3 * i] * x[0] + a[3 * i + 1] * x[1] + a[3 * i + 2] * x[2];
}
static void m3t(const real* a, real* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o[3 * i + j] = a[3 * j + i];
}
static void wedge(const real* v, real* S) {
  S[0] = 0; S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2]; S[4] = 0; S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
static void cross3(const real* a, const real* b, real* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static void m6mul(const real* a, const real* b, real* o) {
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      real s = 0;
      for (int k = 0; k < 6; ++k) s += a[6 * i + k] * b[6 * k + j];
      o[6 * i + j] = s;
    }
}
static void m6vec(const real* a, const real* x, real* o) {
  for (int i = 0; i < 6; ++i) {
    real s = 0;
    for (int k = 0; k < 6; ++k) s += a[6 * i + k] * x[k];
    o[i] = s;
  }
}
static void m6tvec(const real* a, const real* x, real* o) { /* a^T x */
  for (int i = 0; i < 6; ++i) {
    real s = 0;
    for (int k = 0; k < 6; ++k) s += a[6 * k + i] * x[k];
    o[i] = s;
  }
}
static void m6t(const real* a, real* o) {
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) o[6 * i + j] = a[6 * j + i];
}
/* Adjoint.from_rotation_and_translation (math/adjoint.py:66-107) */
static void adjoint(const real* R, const real* p, int inverse, real* X) {
  real S[9], T[9], Rt[9];
  memset(X, 0, 36 * sizeof(real));
  wedge(p, S);
  if (!inverse) {
    m3mul(S, R, T);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        X[6 * i + j] = R[3 * i + j];
        X[6 * i + 3 + j] = T[3 * i + j];
        X[6 * (3 + i) + 3 + j] = R[3 * i + j];
      }
  } else {
    m3t(R, Rt);
    m3mul(Rt, S, T);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        X[6 * i + j] = Rt[3 * i + j];
        X[6 * i + 3 + j] = -T[3 * i + j];
        X[6 * (3 + i) + 3 + j] = Rt[3 * i + j];
      }
  }
}
/* Adjoint.inverse (math/adjoint.py:135-160) */
static void adjoint_inverse(const real* X, real* O) {
  real Rt[9], Tm[9], A[9], B[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Rt[3 * i + j] = X[6 * j + i];
      Tm[3 * i + j] = X[6 * i + 3 + j];
    }
  m3mul(Rt, Tm, A);
  m3mul(A, Rt, B);
  memset(O, 0, 36 * sizeof(real));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      O[6 * i + j] = Rt[3 * i + j];
      O[6 * i + 3 + j] = -B[3 * i + j];
      O[6 * (3 + i) + 3 + j] = Rt[3 * i + j];
    }
}
/* Cross.vx / vx_star (math/cross.py:14-58) */
static void vx(const real* v6, real* X) {
  real Sw[9], Sv[9];
  wedge(v6 + 3, Sw);
  wedge(v6, Sv);
  memset(X, 0, 36 * sizeof(real));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      X[6 * i + j] = Sw[3 * i + j];
      X[6 * i + 3 + j] = Sv[3 * i + j];
      X[6 * (3 + i) + 3 + j] = Sw[3 * i + j];
    }
}
static void vx_star(const real* v6, real* X) {
  real V[36];
  vx(v6, V);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) X[6 * i + j] = -V[6 * j + i];
}
/* jaxlie SO3.as_matrix (normalises implicitly) */
static void quat_to_R(const real* q, real* R) {
  real nsq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  real sc = (real)sqrt((double)(2.0 / nsq));
  real w = q[0] * sc, x = q[1] * sc, y = q[2] * sc, z = q[3] * sc;
  R[0] = 1 - y * y - z * z; R[1] = x * y - z * w; R[2] = x * z + y * w;
  R[3] = x * y + z * w; R[4] = 1 - x * x - z * z; R[5] = y * z - x * w;
  R[6] = x * z - y * w; R[7] = y * z + x * w; R[8] = 1 - x * x - y * y;
}
/* Rotation.from_axis_angle (math/rotation.py:58-84) */
static void axis_angle_R(const real* vec, real* R) {
  real th = (real)sqrt((double)(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2]));
  real s = (real)sin((double)th), c = (real)cos((double)th);
  real hs = (real)sin((double)th / 2.0);
  real c1 = 2 * hs * hs;
  real st = th == 0 ? 1 : th;
  real u[3] = {vec[0] / st, vec[1] / st, vec[2] / st}, S[9], M[9];
  wedge(u, S);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[3 * i + j] = c * (i == j) - s * S[3 * i + j] + c1 * u[i] * u[j];
  m3t(M, R);
}

typedef struct {
  const jxs_model_desc* d;
  int nL, n, n_cp, rows, row_s, row_vlin, row_vang, row_sd, row_m;
  real M[MAXL][36]; /* link spatial inertias, Inertia.to_sixd */
} ctx_t;

static void build_ctx(const jxs_model_desc* d, ctx_t* c) {
  c->d = d;
  c->nL = d->n_links;
  c->n = d->n_links - 1;
  c->n_cp = d->n_points;
  c->row_s = 7;
  c->row_vlin = 7 + c->n;
  c->row_vang = 10 + c->n;
  c->row_sd = 13 + c->n;
  c->row_m = 13 + 2 * c->n;
  c->rows = 13 + 2 * c->n + 3 * c->n_cp;
  for (int i = 0; i < c->nL; ++i) { /* math/inertia.py:14-41 */
    real m = (real)d->link_mass[i], com[3], S[9], St[9], SS[9];
    for (int k = 0; k < 3; ++k) com[k] = (real)d->link_com[3 * i + k];
    wedge(com, S);
    m3t(S, St);
    m3mul(S, St, SS);
    real* M = c->M[i];
    memset(M, 0, 36 * sizeof(real));
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) {
        M[6 * r + q] = m * (r == q);
        M[6 * r + 3 + q] = m * St[3 * r + q];
        M[6 * (3 + r) + q] = m * S[3 * r + q];
        M[6 * (3 + r) + 3 + q] = (real)d->link_inertia[9 * i + 3 * r + q] + m * SS[3 * r + q];
      }
  }
}

/* One environment: state column `e` of [rows][N] arrays. */
static void step_env(const ctx_t* c, const real* sin_, real* sout, const real* tau_ref, const real* Wf_ext, int N, int e) {
  const jxs_model_desc* d = c->d;
  const int nL = c->nL, n = c->n;
  const real eps = eps_of();
#define ST(row) sin_[(size_t)(row)*N + e]
#define OUT(row) sout[(size_t)(row)*N + e]
  real pB[3], q[4], vW[3], om[3], s[MAXL], sd[MAXL], tau[MAXL];
  for (int k = 0; k < 3; ++k) { pB[k] = ST(k); vW[k] = ST(c->row_vlin + k); om[k] = ST(c->row_vang + k); }
  for (int k = 0; k < 4; ++k) q[k] = ST(3 + k);
  for (int j = 0; j < n; ++j) { s[j] = ST(c->row_s + j); sd[j] = ST(c->row_sd + j); }

  /* actuation (api/actuation_model.py:7-126) */
  for (int j = 0; j < n; ++j) {
    const int i = j + 1;
    real smin = (real)d->position_limit_min[i], smax = (real)d->position_limit_max[i];
    real lower = s[j] - smin; if (lower > 0) lower = 0;
    real upper = s[j] - smax; if (upper < 0) upper = 0;
    real tpl = -((real)d->position_limit_spring[i] * (lower + upper));
    tpl = tpl - tpl * ((real)d->position_limit_damper[i] * sd[j]);
    real tfr = 0;
    if (d->enable_friction) {
      real sg = (sd[j] > 0) - (sd[j] < 0);
      tfr = -((real)d->friction_static[i] * sg + (real)d->friction_viscous[i] * sd[j]);
    }
    real tot = (tau_ref ? tau_ref[(size_t)j * N + e] : 0) + tfr + tpl;
    real av = (real)fabs((double)sd[j]), lim;
    if (av <= (real)d->omega_th) lim = (real)d->torque_max;
    else if (av <= (real)d->omega_max) lim = (real)d->torque_max * (1 - (av - (real)d->omega_th) / ((real)d->omega_max - (real)d->omega_th));
    else lim = 0;
    tau[j] = tot > lim ? lim : (tot < -lim ? -lim : tot);
  }

  /* base_orientation (api/data.py:267-286) */
  real qn[4];
  {
    real nr = (real)sqrt((double)(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
    real den = nr + eps * (nr == 0);
    for (int k = 0; k < 4; ++k) qn[k] = q[k] / den;
  }
  real R0[9];
  quat_to_R(qn, R0);
  real W_X_B[36], B_X_W[36];
  adjoint(R0, pB, 0, W_X_B);
  adjoint(R0, pB, 1, B_X_W);

  /* joint transforms i_X_lambda (api/kin_dyn_parameters.py:396-451) */
  static _Thread_local real X[MAXL][36], WXi[MAXL][36], iX0[MAXL][36];
  static _Thread_local real v[MAXL][6], cc[MAXL][6], pA[MAXL][6], MA[MAXL][36], U[MAXL][6], dd[MAXL], uu[MAXL], a[MAXL][6];
  static _Thread_local real WH_R[MAXL][9], WH_p[MAXL][3], Wv[MAXL][6], Wf[MAXL][6];
  for (int i = 0; i < nL; ++i) {
    real R[9], p[3];
    if (i == 0) { /* lambda_H_pre = I, pre_H_suc = W_H_B, suc_H_i[0] */
      real Rs[9], ps[3], t[3];
      for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) Rs[3 * r + k] = (real)d->suc_H_i[4 * r + k]; ps[r] = (real)d->suc_H_i[4 * r + 3]; }
      m3mul(R0, Rs, R);
      m3vec(R0, ps, t);
      for (int k = 0; k < 3; ++k) p[k] = pB[k] + t[k];
    } else {
      real Rp[9], pp[3], Rs[9], ps[3], Rj[9], pj[3] = {0, 0, 0}, T1[9], t1[3], t2[3];
      const double* Hp = d->lambda_H_pre + 16 * i;
      const double* Hs = d->suc_H_i + 16 * i;
      for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) { Rp[3 * r + k] = (real)Hp[4 * r + k]; Rs[3 * r + k] = (real)Hs[4 * r + k]; }
        pp[r] = (real)Hp[4 * r + 3];
        ps[r] = (real)Hs[4 * r + 3];
      }
      real ax[3] = {(real)d->joint_axis[3 * i], (real)d->joint_axis[3 * i + 1], (real)d->joint_axis[3 * i + 2]};
      if (d->joint_type[i] == 1) {
        real vec[3] = {s[i - 1] * ax[0], s[i - 1] * ax[1], s[i - 1] * ax[2]};
        axis_angle_R(vec, Rj);
      } else {
        for (int k = 0; k < 9; ++k) Rj[k] = (k % 4 == 0);
        for (int k = 0; k < 3; ++k) pj[k] = s[i - 1] * ax[k];
      }
      m3mul(Rj, Rs, T1);
      m3mul(Rp, T1, R);
      m3vec(Rj, ps, t1);
      for (int k = 0; k < 3; ++k) t1[k] += pj[k];
      m3vec(Rp, t1, t2);
      for (int k = 0; k < 3; ++k) p[k] = pp[k] + t2[k];
    }
    adjoint(R, p, 1, X[i]);
  }

  /* forward kinematics = the cached link transforms / velocities (rbda/forward_kinematics.py:12-113) */
  adjoint_inverse(X[0], WXi[0]);
  for (int k = 0; k < 3; ++k) { Wv[0][k] = vW[k]; Wv[0][3 + k] = om[k]; }
  for (int i = 1; i < nL; ++i) {
    real Xi[36], Sv[6], t[6];
    adjoint_inverse(X[i], Xi);
    m6mul(WXi[d->parent[i]], Xi, WXi[i]);
    for (int k = 0; k < 6; ++k) Sv[k] = 0;
    for (int k = 0; k < 3; ++k) Sv[(d->joint_type[i] == 1 ? 3 : 0) + k] = (real)d->joint_axis[3 * i + k] * sd[i - 1];
    m6vec(WXi[i], Sv, t);
    for (int k = 0; k < 6; ++k) Wv[i][k] = Wv[d->parent[i]][k] + t[k];
  }
  for (int i = 0; i < nL; ++i) { /* Adjoint.to_transform (math/adjoint.py:109-133) */
    real Rt[9], oxR[9], P[9];
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) { WH_R[i][3 * r + k] = WXi[i][6 * r + k]; oxR[3 * r + k] = WXi[i][6 * r + 3 + k]; }
    m3t(WH_R[i], Rt);
    m3mul(oxR, Rt, P);
    WH_p[i][0] = P[7]; WH_p[i][1] = P[2]; WH_p[i][2] = P[3];
  }

  /* external + contact link wrenches, inertial-fixed (api/ode.py:16-131) */
  for (int i = 0; i < nL; ++i)
    for (int k = 0; k < 6; ++k) Wf[i][k] = Wf_ext ? Wf_ext[(size_t)(6 * i + k) * N + e] : 0;
  const real K = (real)d->K, D = (real)d->D, mu = (real)d->mu;
  for (int cidx = 0; cidx < c->n_cp; ++cidx) {
    real m[3];
    for (int k = 0; k < 3; ++k) m[k] = ST(c->row_m + 3 * cidx + k);
    if (!d->point_enabled[cidx]) { /* disabled points keep m_dot = 0 (soft.py:427-442) */
      for (int k = 0; k < 3; ++k) OUT(c->row_m + 3 * cidx + k) = m[k];
      continue;
    }
    const int b = d->point_body[cidx];
    real Lp[3] = {(real)d->point_position[3 * cidx], (real)d->point_position[3 * cidx + 1], (real)d->point_position[3 * cidx + 2]};
    real pw[3], pd[3], t[3];
    m3vec(WH_R[b], Lp, pw);
    for (int k = 0; k < 3; ++k) pw[k] += WH_p[b][k];
    cross3(Wv[b] + 3, pw, t); /* [I, -S(p)] v = v_lin + w x p (collidable_points.py:50-53) */
    for (int k = 0; k < 3; ++k) pd[k] = Wv[b][k] + t[k];
    /* penetration (contacts/common.py:25-63), flat terrain */
    real nh[3] = {(real)d->terrain_normal[0], (real)d->terrain_normal[1], (real)d->terrain_normal[2]};
    /* PlaneTerrain.height (terrain/terrain.py:180-215); FlatTerrain is the normal (0,0,1) */
    real height = (real)(-(d->terrain_normal[0] * pw[0] + d->terrain_normal[1] * pw[1] - d->terrain_normal[2] * d->terrain_height) / d->terrain_normal[2]);
    real h = (height - pw[2]) * nh[2]; /* h . n with h = [0,0,height - p_z] */
    real delta = h > 0 ? h : 0;
    real ddot = delta > 0 ? -(pd[0] * nh[0] + pd[1] * nh[1] + pd[2] * nh[2]) : 0;
    real dp = (real)pow((double)(delta + eps), d->p), dq = (real)pow((double)(delta + eps), d->q);
    real fn = (K * dp) * delta + (D * dq) * ddot;
    if (fn < 0) fn = 0;
    real vdotn = pd[0] * nh[0] + pd[1] * nh[1] + pd[2] * nh[2], mdotn = m[0] * nh[0] + m[1] * nh[1] + m[2] * nh[2];
    real vt[3], mn[3], mt[3], ft[3];
    for (int k = 0; k < 3; ++k) { vt[k] = pd[k] - vdotn * nh[k]; mn[k] = mdotn * nh[k]; mt[k] = m[k] - mdotn * nh[k]; }
    for (int k = 0; k < 3; ++k) ft[k] = -((K * dp) * mt[k] + (D * dq) * vt[k]);
    real ft2 = ft[0] * ft[0] + ft[1] * ft[1] + ft[2] * ft[2];
    int no_contact = delta <= 0;
    int sticking = no_contact || (ft2 <= (mu * fn) * (mu * fn));
    real nrm = (real)sqrt((double)ft2);
    real den = nrm + eps * (nrm == 0);
    real mag = mu * fn < nrm ? mu * fn : nrm;
    if (!sticking) for (int k = 0; k < 3; ++k) ft[k] = mag * (ft[k] / den);
    if (no_contact) for (int k = 0; k < 3; ++k) ft[k] = 0;
    real md[3];
    for (int k = 0; k < 3; ++k) {
      real nc = -(K / D) * m[k], stv = vt[k] - (K / D) * mn[k], sl = -(ft[k] + (K * dp) * mt[k]) / (D * dq);
      md[k] = no_contact ? nc : (sticking ? stv : sl);
    }
    real f[3] = {fn * nh[0] + ft[0], fn * nh[1] + ft[1], fn * nh[2] + ft[2]}, mom[3];
    cross3(pw, f, mom); /* W_f = [f; p x f] (soft.py:377-388) */
    for (int k = 0; k < 3; ++k) { Wf[b][k] += f[k]; Wf[b][3 + k] += mom[k]; }
    for (int k = 0; k < 3; ++k) OUT(c->row_m + 3 * cidx + k) = m[k] + (real)d->time_step * md[k];
  }

  /* ABA (rbda/aba.py:12-292) */
  const int floating = d->floating_base;
  real Wg[6] = {0, 0, (real)d->gravity, 0, 0, 0};
  memset(v, 0, sizeof(real) * 6 * nL);
  memset(cc, 0, sizeof(real) * 6 * nL);
  memset(pA, 0, sizeof(real) * 6 * nL);
  memset(MA, 0, sizeof(real) * 36 * nL);
  memset(iX0, 0, sizeof(real) * 36 * nL);
  for (int k = 0; k < 6; ++k) iX0[0][7 * k] = 1;
  real Sm[MAXL][6];
  for (int i = 0; i < nL; ++i) {
    for (int k = 0; k < 6; ++k) Sm[i][k] = 0;
    if (i > 0)
      for (int k = 0; k < 3; ++k) Sm[i][(d->joint_type[i] == 1 ? 3 : 0) + k] = (real)d->joint_axis[3 * i + k];
  }
  if (floating) {
    real Wv0[6] = {vW[0], vW[1], vW[2], om[0], om[1], om[2]}, VS[36], T[36], t1[6], t2[6];
    m6vec(B_X_W, Wv0, v[0]);
    memcpy(MA[0], c->M[0], 36 * sizeof(real));
    vx_star(v[0], VS);
    m6mul(VS, MA[0], T);
    m6vec(T, v[0], t1);
    m6tvec(W_X_B, Wf[0], t2);
    for (int k = 0; k < 6; ++k) pA[0][k] = t1[k] - t2[k];
  }
  for (int i = 1; i < nL; ++i) { /* pass 1 */
    const int l = d->parent[i];
    real vJ[6], t[6], VX[36], VS[36], T[36], T2[36], Xf[36], t1[6], t2[6];
    for (int k = 0; k < 6; ++k) vJ[k] = Sm[i][k] * sd[i - 1];
    m6vec(X[i], v[l], t);
    for (int k = 0; k < 6; ++k) v[i][k] = t[k] + vJ[k];
    vx(v[i], VX);
    m6vec(VX, vJ, cc[i]);
    memcpy(MA[i], c->M[i], 36 * sizeof(real));
    m6mul(X[i], iX0[l], iX0[i]);
    m6mul(iX0[i], B_X_W, T);
    adjoint_inverse(T, T2);
    m6t(T2, Xf);
    vx_star(v[i], VS);
    m6mul(VS, c->M[i], T);
    m6vec(T, v[i], t1);
    m6vec(Xf, Wf[i], t2);
    for (int k = 0; k < 6; ++k) pA[i][k] = t1[k] - t2[k];
  }
  for (int i = nL - 1; i >= 1; --i) { /* pass 2 */
    const int l = d->parent[i];
    real Ma[36], pa[6], t[6];
    m6vec(MA[i], Sm[i], U[i]);
    dd[i] = 0; real sp = 0;
    for (int k = 0; k < 6; ++k) { dd[i] += Sm[i][k] * U[i][k]; sp += Sm[i][k] * pA[i][k]; }
    uu[i] = tau[i - 1] - sp;
    for (int r = 0; r < 6; ++r)
      for (int k = 0; k < 6; ++k) Ma[6 * r + k] = MA[i][6 * r + k] - (U[i][r] / dd[i]) * U[i][k];
    m6vec(Ma, cc[i], t);
    for (int k = 0; k < 6; ++k) pa[k] = pA[i][k] + t[k] + U[i][k] * (uu[i] / dd[i]);
    if (l != 0 || floating) {
      real Xt[36], T[36], T2[36];
      m6t(X[i], Xt);
      m6mul(Xt, Ma, T);
      m6mul(T, X[i], T2);
      for (int k = 0; k < 36; ++k) MA[l][k] += T2[k];
      m6tvec(X[i], pa, t);
      for (int k = 0; k < 6; ++k) pA[l][k] += t[k];
    }
  }
  if (floating) { /* a0 = solve(-MA[0], pA[0]) by Gaussian elimination with partial pivoting */
    real Aug[6][7];
    for (int r = 0; r < 6; ++r) { for (int k = 0; k < 6; ++k) Aug[r][k] = -MA[0][6 * r + k]; Aug[r][6] = pA[0][r]; }
    for (int col = 0; col < 6; ++col) {
      int piv = col;
      for (int r = col + 1; r < 6; ++r) if (fabs((double)Aug[r][col]) > fabs((double)Aug[piv][col])) piv = r;
      if (piv != col) for (int k = 0; k < 7; ++k) { real tmp = Aug[col][k]; Aug[col][k] = Aug[piv][k]; Aug[piv][k] = tmp; }
      for (int r = col + 1; r < 6; ++r) {
        real fct = Aug[r][col] / Aug[col][col];
        for (int k = col; k < 7; ++k) Aug[r][k] -= fct * Aug[col][k];
      }
    }
    for (int r = 5; r >= 0; --r) {
      real acc = Aug[r][6];
      for (int k = r + 1; k < 6; ++k) acc -= Aug[r][k] * a[0][k];
      a[0][r] = acc / Aug[r][r];
    }
  } else {
    real t[6];
    m6vec(B_X_W, Wg, t);
    for (int k = 0; k < 6; ++k) a[0][k] = -t[k];
  }
  real sdd[MAXL];
  for (int i = 1; i < nL; ++i) { /* pass 3 */
    real ai[6], ua = 0;
    m6vec(X[i], a[d->parent[i]], ai);
    for (int k = 0; k < 6; ++k) { ai[k] += cc[i][k]; ua += U[i][k] * ai[k]; }
    sdd[i - 1] = (uu[i] - ua) / dd[i];
    for (int k = 0; k < 6; ++k) a[i][k] = ai[k] + Sm[i][k] * sdd[i - 1];
  }
  real Wa[6] = {0, 0, 0, 0, 0, 0};
  if (floating) {
    m6vec(W_X_B, a[0], Wa);
    for (int k = 0; k < 6; ++k) Wa[k] += Wg[k];
  }

  /* semi-implicit Euler (api/integrators.py:14-88) */
  const real dt = (real)d->time_step;
  real vn[3], wn[3], pd[3], t[3];
  for (int k = 0; k < 3; ++k) { vn[k] = vW[k] + dt * Wa[k]; wn[k] = om[k] + dt * Wa[3 + k]; }
  cross3(wn, pB, t);
  for (int k = 0; k < 3; ++k) pd[k] = vn[k] + t[k];
  real nw = (real)sqrt((double)(wn[0] * wn[0] + wn[1] * wn[1] + wn[2] * wn[2]));
  real nq = (real)sqrt((double)(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]));
  real h0 = (real)0.1 * nw * (1 - nq);
  real Qd[4];
  Qd[0] = (real)0.5 * (qn[0] * h0 - qn[1] * wn[0] - qn[2] * wn[1] - qn[3] * wn[2]);
  Qd[1] = (real)0.5 * (qn[1] * h0 + qn[0] * wn[0] + qn[3] * wn[1] - qn[2] * wn[2]);
  Qd[2] = (real)0.5 * (qn[2] * h0 - qn[3] * wn[0] + qn[0] * wn[1] + qn[1] * wn[2]);
  Qd[3] = (real)0.5 * (qn[3] * h0 + qn[2] * wn[0] - qn[1] * wn[1] + qn[0] * wn[2]);
  real Qn[4], nn = 0;
  for (int k = 0; k < 4; ++k) { Qn[k] = qn[k] + dt * Qd[k]; nn += Qn[k] * Qn[k]; }
  nn = (real)sqrt((double)nn);
  if (nn == 0) nn = 1;
  for (int k = 0; k < 4; ++k) Qn[k] /= nn;
  /* data.replace re-normalises once more (api/data.py:434-440) */
  nn = (real)sqrt((double)(Qn[0] * Qn[0] + Qn[1] * Qn[1] + Qn[2] * Qn[2] + Qn[3] * Qn[3]));
  if (nn == 0) nn = 1;
  for (int k = 0; k < 4; ++k) OUT(3 + k) = Qn[k] / nn;
  for (int k = 0; k < 3; ++k) { OUT(k) = pB[k] + dt * pd[k]; OUT(c->row_vlin + k) = vn[k]; OUT(c->row_vang + k) = wn[k]; }
  for (int j = 0; j < n; ++j) {
    real sdn = sd[j] + dt * sdd[j];
    OUT(c->row_sd + j) = sdn;
    OUT(c->row_s + j) = s[j] + dt * sdn;
  }
#undef ST
#undef OUT
}

/* state_in/state_out: host [rows][N]; tau: [n][N] or NULL; link_forces: inertial-fixed
 * [nL*6][N] or NULL.  n_threads <= 0 -> OpenMP default.  Returns 0, or -1 for nL > 64. */
int FN(oracle_step)(const jxs_model_desc* d, const REAL* state_in, REAL* state_out, const REAL* tau,
                    const REAL* link_forces, int N, int n_steps, int n_threads) {
  if (d->n_links > MAXL) return -1;
  if (d->integrator != 0) return -2; /* the C port restates the semi-implicit Euler step only */
  ctx_t* c = (ctx_t*)malloc(sizeof(ctx_t));
  build_ctx(d, c);
  if (state_out != state_in) memcpy(state_out, state_in, sizeof(REAL) * (size_t)c->rows * N);
  int nt = n_threads;
#ifdef _OPENMP
  if (nt <= 0) nt = omp_get_max_threads();
#else
  nt = 1;
#endif
  /* Environments never interact, so each thread advances its environments through all the
   * steps: one parallel region, no per-step barrier. */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nt)
#endif
  for (int e = 0; e < N; ++e)
    for (int it = 0; it < n_steps; ++it) step_env(c, state_out, state_out, tau, link_forces, N, e);
  free(c);
  return 0;
}
