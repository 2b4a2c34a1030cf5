// EXPERIMENT: the soft-contact step with ONE ENVIRONMENT PER LANE
// and serial sweeps over the links -- no cross-lane traffic, every table value wave-uniform (SGPR operands),
// per-link intermediates in private (scratch) memory.  Same formulation as the product kernel (frame C: origin
// at the base position, world-aligned axes; jaxsim_amd/csrc/jxs_core.h), same arithmetic helpers, plain
// [rows][N] state layout.  Built and driven by tools/experiments/env_per_lane.py, which checks the result against
// the product kernel and times both.  Not part of the library.
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kMaxLinks = 32;
constexpr int kMaxPoints = 64;

struct ELink {
  int parent, jtype;  // jtype: 0 base, 1 revolute, 2 prismatic
  float axis[3], Rpre[9], ppre[3], mass, com[3], I[6];
  float kc, kv, smin, smax, klim, dlim;
  int p0, p1;  // points of this link: [p0, p1)
};
struct EModel {
  int nL, n, n_points, n_rows, floating;
  float dt, g, K, D, mu, eps, K_over_D, quat_K, tau_max, w_th, w_max, inv_w_range, terrain_h;
  ELink link[kMaxLinks];
  float ppos[kMaxPoints][3];
  int prow[kMaxPoints];
};

__device__ __forceinline__ float frcp(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float fsqrt(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float s = x * y;
  const float e = __builtin_fmaf(-s, s, x);
  const float r = __builtin_fmaf(0.5f * y, e, s);
  return x > 0.0f ? r : x;
}
__device__ __forceinline__ void cross3(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void mat3vec(const float* R, const float* x, float* o) {
  o[0] = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  o[1] = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  o[2] = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
}
__device__ __forceinline__ constexpr int sidx(int i, int j) {
  return (i <= j) ? (i * 6 - (i * (i - 1)) / 2 + (j - i)) : (j * 6 - (j * (j - 1)) / 2 + (i - j));
}

__global__ __launch_bounds__(64) void step_env_per_lane(const EModel* __restrict__ Mp, const float* sin, float* sout, int N) {
  const int env = blockIdx.x * 64 + threadIdx.x;
  if (env >= N) return;
  const EModel& M = *Mp;
  const int nL = M.nL, n = M.n;
  const int row_quat = 3, row_s = 7, row_vlin = 7 + n, row_vang = 10 + n, row_sd = 13 + n, row_m = 13 + 2 * n;
  auto ld = [&](int row) { return sin[(size_t)row * N + env]; };
  auto st = [&](int row, float v) { sout[(size_t)row * N + env] = v; };
  float pB[3], q[4], vW[3], om[3];
  for (int k = 0; k < 3; ++k) pB[k] = ld(k), vW[k] = ld(row_vlin + k), om[k] = ld(row_vang + k);
  for (int k = 0; k < 4; ++k) q[k] = ld(row_quat + k);
  {
    const float nrm = fsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float inv = frcp(nrm + (nrm == 0.0f ? M.eps : 0.0f));
    for (int k = 0; k < 4; ++k) q[k] *= inv;
  }
  // per-link intermediates (private memory)
  float R[kMaxLinks][9], r[kMaxLinks][3], vl[kMaxLinks][3], va[kMaxLinks][3];
  float S6[kMaxLinks][6], c6[kMaxLinks][6], MA[kMaxLinks][21], pA[kMaxLinks][6], U[kMaxLinks][6], dinv[kMaxLinks], uu[kMaxLinks], tauj[kMaxLinks];
  float vBc[3];
  cross3(om, pB, vBc);
  for (int k = 0; k < 3; ++k) vBc[k] += vW[k];
  // ---- pass 1: kinematics, contacts, inertia, bias (links in BFS order: parents first)
  for (int i = 0; i < nL; ++i) {
    const ELink& Lk = M.link[i];
    float s = 0.0f, sd = 0.0f;
    if (i == 0) {
      const float w = q[0], x = q[1], y = q[2], z = q[3];
      R[0][0] = 1 - 2 * (y * y + z * z), R[0][1] = 2 * (x * y - w * z), R[0][2] = 2 * (x * z + w * y);
      R[0][3] = 2 * (x * y + w * z), R[0][4] = 1 - 2 * (x * x + z * z), R[0][5] = 2 * (y * z - w * x);
      R[0][6] = 2 * (x * z - w * y), R[0][7] = 2 * (y * z + w * x), R[0][8] = 1 - 2 * (x * x + y * y);
      for (int k = 0; k < 3; ++k) r[0][k] = 0.0f, vl[0][k] = M.floating ? vBc[k] : 0.0f, va[0][k] = M.floating ? om[k] : 0.0f;
      for (int k = 0; k < 6; ++k) S6[0][k] = 0.0f, c6[0][k] = 0.0f;
      tauj[0] = 0.0f;
    } else {
      s = ld(row_s + i - 1), sd = ld(row_sd + i - 1);
      const int lam = Lk.parent;
      float sh, ch;
      __sincosf((Lk.jtype == 1 ? s : 0.0f) * 0.5f, &sh, &ch);
      const float sn = 2 * sh * ch, c1 = 2 * sh * sh, cs = 1 - c1;
      const float* ax = Lk.axis;
      float Rj[9] = {cs + c1 * ax[0] * ax[0], -sn * ax[2] + c1 * ax[0] * ax[1], sn * ax[1] + c1 * ax[0] * ax[2],
                     sn * ax[2] + c1 * ax[1] * ax[0], cs + c1 * ax[1] * ax[1], -sn * ax[0] + c1 * ax[1] * ax[2],
                     -sn * ax[1] + c1 * ax[2] * ax[0], sn * ax[0] + c1 * ax[2] * ax[1], cs + c1 * ax[2] * ax[2]};
      float pj[3] = {0, 0, 0};
      if (Lk.jtype == 2) for (int k = 0; k < 3; ++k) pj[k] = s * ax[k];
      float Rl[9], pl[3];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Rl[3 * a + b] = Lk.Rpre[3 * a] * Rj[b] + Lk.Rpre[3 * a + 1] * Rj[3 + b] + Lk.Rpre[3 * a + 2] * Rj[6 + b];
      mat3vec(Lk.Rpre, pj, pl);
      for (int k = 0; k < 3; ++k) pl[k] += Lk.ppre[k];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) R[i][3 * a + b] = R[lam][3 * a] * Rl[b] + R[lam][3 * a + 1] * Rl[3 + b] + R[lam][3 * a + 2] * Rl[6 + b];
      float t[3];
      mat3vec(R[lam], pl, t);
      for (int k = 0; k < 3; ++k) r[i][k] = t[k] + r[lam][k];
      float Ra[3], rxa[3];
      mat3vec(R[i], ax, Ra);
      cross3(r[i], Ra, rxa);
      for (int k = 0; k < 3; ++k) {
        S6[i][3 + k] = Lk.jtype == 1 ? Ra[k] : 0.0f;
        S6[i][k] = Lk.jtype == 1 ? rxa[k] : Ra[k];
      }
      float vJl[3], vJa[3];
      for (int k = 0; k < 3; ++k) vJl[k] = S6[i][k] * sd, vJa[k] = S6[i][3 + k] * sd, vl[i][k] = vl[lam][k] + vJl[k], va[i][k] = va[lam][k] + vJa[k];
      float t0[3], t1[3], ca[3];
      cross3(va[i], vJl, t0);
      cross3(vl[i], vJa, t1);
      cross3(va[i], vJa, ca);
      for (int k = 0; k < 3; ++k) c6[i][k] = t0[k] + t1[k], c6[i][3 + k] = ca[k];
      // actuation (api/actuation_model.py:7-126), tau_ref = 0
      const float lower = fminf(s - Lk.smin, 0.0f), upper = fmaxf(s - Lk.smax, 0.0f);
      float tau_pl = -(Lk.klim * (lower + upper));
      tau_pl = tau_pl - tau_pl * (Lk.dlim * sd);
      const float sgn = sd > 0 ? 1.0f : (sd < 0 ? -1.0f : 0.0f);
      const float tot = -(Lk.kc * sgn + Lk.kv * sd) + tau_pl;
      const float av = fabsf(sd);
      const float lim = av <= M.w_th ? M.tau_max : (av <= M.w_max ? M.tau_max * (1.0f - (av - M.w_th) * M.inv_w_range) : 0.0f);
      tauj[i] = fmaxf(fminf(tot, lim), -lim);
    }
    // soft contacts of this link's points
    float fl[3] = {0, 0, 0}, fa[3] = {0, 0, 0};
    for (int p = Lk.p0; p < Lk.p1; ++p) {
      float rc[3], pw[3], pd[3], t[3], m[3];
      mat3vec(R[i], M.ppos[p], rc);
      for (int k = 0; k < 3; ++k) rc[k] += r[i][k], pw[k] = rc[k] + pB[k], m[k] = ld(row_m + 3 * M.prow[p] + k);
      cross3(va[i], rc, t);
      for (int k = 0; k < 3; ++k) pd[k] = vl[i][k] + t[k];
      const float delta = fmaxf(0.0f, M.terrain_h - pw[2]);
      const bool in_contact = delta > 0.0f;
      const float ddelta = in_contact ? -pd[2] : 0.0f;
      const float dp = fsqrt(delta + M.eps);
      const float Kdp = M.K * dp, Ddq = M.D * dp;
      const float fn = fmaxf(0.0f, Kdp * delta + Ddq * ddelta);
      float vt[3] = {pd[0], pd[1], 0.0f}, mn[3] = {0.0f, 0.0f, m[2]}, mt[3] = {m[0], m[1], 0.0f}, ft[3];
      for (int k = 0; k < 3; ++k) ft[k] = -(Kdp * mt[k] + Ddq * vt[k]);
      const float ft2 = ft[0] * ft[0] + ft[1] * ft[1] + ft[2] * ft[2];
      const float mufn = M.mu * fn;
      const bool sticking = !in_contact || (ft2 <= mufn * mufn);
      const float nrm = fsqrt(ft2);
      const float scale = fminf(mufn, nrm) * frcp(nrm + (nrm == 0.0f ? M.eps : 0.0f));
      const float inv_Ddq = frcp(Ddq);
      float md[3];
      for (int k = 0; k < 3; ++k) {
        ft[k] = sticking ? ft[k] : scale * ft[k];
        ft[k] = in_contact ? ft[k] : 0.0f;
        const float md_nc = -(M.K_over_D * m[k]), md_st = vt[k] - M.K_over_D * mn[k], md_sl = -(ft[k] + Kdp * mt[k]) * inv_Ddq;
        md[k] = !in_contact ? md_nc : (sticking ? md_st : md_sl);
        st(row_m + 3 * M.prow[p] + k, m[k] + M.dt * md[k]);
      }
      const float f[3] = {ft[0], ft[1], fn + ft[2]};
      float mo[3];
      cross3(rc, f, mo);
      for (int k = 0; k < 3; ++k) fl[k] += f[k], fa[k] += mo[k];
    }
    // link inertia in C and bias force
    float cw[3], Ic[6];
    mat3vec(R[i], Lk.com, cw);
    for (int k = 0; k < 3; ++k) cw[k] += r[i][k];
    {
      const float I9[9] = {Lk.I[0], Lk.I[1], Lk.I[2], Lk.I[1], Lk.I[3], Lk.I[4], Lk.I[2], Lk.I[4], Lk.I[5]};
      float Tm[9];
      const float* Ri = R[i];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) Tm[3 * a + b] = Ri[3 * a] * I9[b] + Ri[3 * a + 1] * I9[3 + b] + Ri[3 * a + 2] * I9[6 + b];
      Ic[0] = Tm[0] * Ri[0] + Tm[1] * Ri[1] + Tm[2] * Ri[2];
      Ic[1] = Tm[0] * Ri[3] + Tm[1] * Ri[4] + Tm[2] * Ri[5];
      Ic[2] = Tm[0] * Ri[6] + Tm[1] * Ri[7] + Tm[2] * Ri[8];
      Ic[3] = Tm[3] * Ri[3] + Tm[4] * Ri[4] + Tm[5] * Ri[5];
      Ic[4] = Tm[3] * Ri[6] + Tm[4] * Ri[7] + Tm[5] * Ri[8];
      Ic[5] = Tm[6] * Ri[6] + Tm[7] * Ri[7] + Tm[8] * Ri[8];
    }
    const float mass = Lk.mass;
    float hl[3], ha[3], t[3], Iw[3], bl[3], ba[3], t0[3], t1[3];
    cross3(va[i], cw, t);
    for (int k = 0; k < 3; ++k) hl[k] = mass * (vl[i][k] + t[k]);
    Iw[0] = Ic[0] * va[i][0] + Ic[1] * va[i][1] + Ic[2] * va[i][2];
    Iw[1] = Ic[1] * va[i][0] + Ic[3] * va[i][1] + Ic[4] * va[i][2];
    Iw[2] = Ic[2] * va[i][0] + Ic[4] * va[i][1] + Ic[5] * va[i][2];
    cross3(cw, hl, t);
    for (int k = 0; k < 3; ++k) ha[k] = Iw[k] + t[k];
    cross3(va[i], hl, bl);
    cross3(vl[i], hl, t0);
    cross3(va[i], ha, t1);
    for (int k = 0; k < 3; ++k) ba[k] = t0[k] + t1[k], pA[i][k] = bl[k] - fl[k], pA[i][3 + k] = ba[k] - fa[k];
    float* Mi = MA[i];
    const float mcx = mass * cw[0], mcy = mass * cw[1], mcz = mass * cw[2], cc = cw[0] * cw[0] + cw[1] * cw[1] + cw[2] * cw[2];
    Mi[sidx(0, 0)] = mass, Mi[sidx(0, 1)] = 0, Mi[sidx(0, 2)] = 0, Mi[sidx(1, 1)] = mass, Mi[sidx(1, 2)] = 0, Mi[sidx(2, 2)] = mass;
    Mi[sidx(0, 3)] = 0, Mi[sidx(0, 4)] = mcz, Mi[sidx(0, 5)] = -mcy, Mi[sidx(1, 3)] = -mcz, Mi[sidx(1, 4)] = 0, Mi[sidx(1, 5)] = mcx;
    Mi[sidx(2, 3)] = mcy, Mi[sidx(2, 4)] = -mcx, Mi[sidx(2, 5)] = 0;
    Mi[sidx(3, 3)] = Ic[0] + mass * (cc - cw[0] * cw[0]), Mi[sidx(3, 4)] = Ic[1] - mcx * cw[1], Mi[sidx(3, 5)] = Ic[2] - mcx * cw[2];
    Mi[sidx(4, 4)] = Ic[3] + mass * (cc - cw[1] * cw[1]), Mi[sidx(4, 5)] = Ic[4] - mcy * cw[2], Mi[sidx(5, 5)] = Ic[5] + mass * (cc - cw[2] * cw[2]);
  }
  // ---- pass 2: leaves to base (rbda/aba.py:184-224); in frame C parents simply add
  for (int i = nL - 1; i >= 1; --i) {
    const int lam = M.link[i].parent;
    float d = 0.0f, sp = 0.0f;
    for (int a = 0; a < 6; ++a) {
      float acc = 0.0f;
      for (int b = 0; b < 6; ++b) acc += MA[i][sidx(a, b)] * S6[i][b];
      U[i][a] = acc;
    }
    for (int a = 0; a < 6; ++a) d += U[i][a] * S6[i][a], sp += pA[i][a] * S6[i][a];
    const float inv = frcp(d), u = tauj[i] - sp;
    dinv[i] = inv, uu[i] = u;
    if (lam != 0 || M.floating) {
      float Ud[6], Ma[21];
      for (int a = 0; a < 6; ++a) Ud[a] = U[i][a] * inv;
      for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) Ma[sidx(a, b)] = MA[i][sidx(a, b)] - Ud[a] * U[i][b];
      for (int a = 0; a < 6; ++a) {
        float acc = pA[i][a] + Ud[a] * u;
        for (int b = 0; b < 6; ++b) acc += Ma[sidx(a, b)] * c6[i][b];
        pA[lam][a] += acc;
      }
      for (int e = 0; e < 21; ++e) MA[lam][e] += Ma[e];
    }
  }
  // ---- pass 3
  float a[kMaxLinks][6], sdd[kMaxLinks];
  if (M.floating) {
    // LDL^T solve of MA_0 a0 = -pA_0
    float Lm[6][6], Dd[6], Di[6], y[6];
    for (int j = 0; j < 6; ++j) {
      float dj = MA[0][sidx(j, j)];
      for (int k = 0; k < j; ++k) dj -= Lm[j][k] * Lm[j][k] * Dd[k];
      Dd[j] = dj, Di[j] = frcp(dj);
      for (int i = j + 1; i < 6; ++i) {
        float lij = MA[0][sidx(i, j)];
        for (int k = 0; k < j; ++k) lij -= Lm[i][k] * Lm[j][k] * Dd[k];
        Lm[i][j] = lij * Di[j];
      }
    }
    for (int i = 0; i < 6; ++i) {
      float acc = -pA[0][i];
      for (int k = 0; k < i; ++k) acc -= Lm[i][k] * y[k];
      y[i] = acc;
    }
    for (int i = 5; i >= 0; --i) {
      float acc = y[i] * Di[i];
      for (int k = i + 1; k < 6; ++k) acc -= Lm[k][i] * a[0][k];
      a[0][i] = acc;
    }
  } else {
    for (int k = 0; k < 6; ++k) a[0][k] = 0.0f;
    a[0][2] = -M.g;
  }
  for (int i = 1; i < nL; ++i) {
    const int lam = M.link[i].parent;
    float ai[6], ua = 0.0f;
    for (int k = 0; k < 6; ++k) ai[k] = a[lam][k] + c6[i][k], ua += U[i][k] * ai[k];
    const float sddi = (uu[i] - ua) * dinv[i];
    sdd[i] = sddi;
    for (int k = 0; k < 6; ++k) a[i][k] = ai[k] + S6[i][k] * sddi;
  }
  // ---- semi-implicit Euler (api/integrators.py:14-88)
  const float dt = M.dt;
  for (int i = 1; i < nL; ++i) {
    const float sd = ld(row_sd + i - 1) + dt * sdd[i];
    st(row_sd + i - 1, sd);
    st(row_s + i - 1, ld(row_s + i - 1) + dt * sd);
  }
  float acl[3], aca[3], omn[3], pd[3], t[3];
  for (int k = 0; k < 3; ++k) acl[k] = M.floating ? a[0][k] : 0.0f, aca[k] = M.floating ? a[0][3 + k] : 0.0f;
  if (M.floating) acl[2] += M.g;
  cross3(aca, pB, t);
  for (int k = 0; k < 3; ++k) omn[k] = om[k] + dt * aca[k], pd[k] = vBc[k] + dt * acl[k], vW[k] += dt * (acl[k] - t[k]);
  {
    const float nw = fsqrt(omn[0] * omn[0] + omn[1] * omn[1] + omn[2] * omn[2]);
    const float nq = fsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float h0 = M.quat_K * nw * (1.0f - nq);
    float qd[4];
    qd[0] = 0.5f * (q[0] * h0 - q[1] * omn[0] - q[2] * omn[1] - q[3] * omn[2]);
    qd[1] = 0.5f * (q[1] * h0 + q[0] * omn[0] + q[3] * omn[1] - q[2] * omn[2]);
    qd[2] = 0.5f * (q[2] * h0 - q[3] * omn[0] + q[0] * omn[1] + q[1] * omn[2]);
    qd[3] = 0.5f * (q[3] * h0 + q[2] * omn[0] - q[1] * omn[1] + q[0] * omn[2]);
    float qn[4];
    for (int k = 0; k < 4; ++k) qn[k] = q[k] + dt * qd[k];
    const float nn = fsqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    const float invn = frcp(nn == 0.0f ? 1.0f : nn);
    for (int k = 0; k < 4; ++k) st(row_quat + k, qn[k] * invn);
  }
  for (int k = 0; k < 3; ++k) st(k, pB[k] + dt * pd[k]), st(row_vlin + k, vW[k]), st(row_vang + k, omn[k]);
}

extern "C" {
int epl_upload_model(const EModel* host, void** dev) {
  if (hipMalloc(dev, sizeof(EModel)) != hipSuccess) return -1;
  return hipMemcpy(*dev, host, sizeof(EModel), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
int epl_model_bytes() { return (int)sizeof(EModel); }
// `steps` in-place launches on the default stream; returns microseconds per step (HIP events)
int epl_run(const void* dmodel, void* state, int N, int steps, float* us_per_step) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int blocks = (N + 63) / 64;
  hipEventRecord(e0, 0);
  for (int i = 0; i < steps; ++i)
    hipLaunchKernelGGL(step_env_per_lane, dim3(blocks), dim3(64), 0, 0, (const EModel*)dmodel, (const float*)state, (float*)state, N);
  hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return -1;
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  *us_per_step = ms * 1e3f / (float)steps;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
}
